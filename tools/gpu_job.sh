#!/bin/bash
# scratch: the GPU job of the moment
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_data_pipeline_gpu.py -x -q -m gpu > gpurun_out/t11.log 2>&1
tail -5 gpurun_out/t11.log
