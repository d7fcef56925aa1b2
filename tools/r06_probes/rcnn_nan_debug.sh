for s in -1 1 2 3; do timeout 900 python tools/r06_probes/rcnn_nan_debug.py $s 2>&1 | grep -v amdgpu.ids | tail -12; done
