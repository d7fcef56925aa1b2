#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fcos_step_gpu.py tests/test_rcnn_step_gpu.py -x -q -m gpu -k "trainable_stem" > gpurun_out/t19.log 2>&1
tail -30 gpurun_out/t19.log
