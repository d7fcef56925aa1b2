#!/bin/bash
# scratch: the GPU job of the moment
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_fcos_step_gpu.py tests/test_dp_gpu.py tests/test_rcnn_step_gpu.py -x -q -m gpu > gpurun_out/t16.log 2>&1
tail -6 gpurun_out/t16.log
