timeout 600 python tools/r06_probes/gnb_debug.py 2>&1 | tail -30
