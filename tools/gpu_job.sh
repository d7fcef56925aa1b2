#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_rcnn_kernels_gpu.py tests/test_rcnn_step_gpu.py -x -q -m gpu > gpurun_out/t17.log 2>&1
tail -5 gpurun_out/t17.log
