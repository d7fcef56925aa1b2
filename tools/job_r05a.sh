#!/bin/bash
# first GPU job of round 5 (re-entry): power probe, GPU suite, default bench line
mkdir -p gpurun_out
timeout 120 python tools/power_probe.py 4 > gpurun_out/r05_power_probe.txt 2>&1
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r05_gputests.log 2>&1
timeout 900 python bench.py > gpurun_out/r05_bench_f16.json 2> gpurun_out/r05_bench_f16.err
tail -5 gpurun_out/r05_gputests.log; cat gpurun_out/r05_power_probe.txt; head -c 1500 gpurun_out/r05_bench_f16.json
