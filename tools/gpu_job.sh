#!/bin/bash
python -m pytest tests/test_rcnn_step_gpu.py tests/test_rcnn_kernels_gpu.py tests/test_dp_gpu.py -x -q -m gpu > gpurun_out/t8.log 2>&1; tail -5 gpurun_out/t8.log
