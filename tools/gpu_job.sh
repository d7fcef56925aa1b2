#!/bin/bash
# scratch: the GPU job of the moment
cd /root/repo
bash tools/measure_record.sh r03 > gpurun_out/measure.log 2>&1
tail -c 100 gpurun_out/r03_bench_f16.json
