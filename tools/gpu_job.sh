#!/bin/bash
# scratch: the GPU job of the moment
cd /root/repo
timeout 600 python -m pytest tests/test_conv_bf16_gpu.py -x -q -m gpu -k "tile or big or w8" > gpurun_out/t13.log 2>&1
tail -3 gpurun_out/t13.log
bash tools/measure_record.sh r03
