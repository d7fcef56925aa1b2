timeout 900 python tools/r06_probes/teacher_copy_debug.py fcos 2>&1 | grep -a "largest\|student AP\|loss-scale\|Error\|error" | cut -c1-600
