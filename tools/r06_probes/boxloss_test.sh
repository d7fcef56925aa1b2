timeout 900 python -m pytest tests/test_rcnn_kernels_gpu.py tests/test_rcnn_step_gpu.py -x -q 2>&1 | tail -12
