#!/bin/bash
# scratch: the GPU job of the moment
cd /root/repo
mkdir -p gpurun_out
UTV2_W8=0 timeout 300 python tools/check_w8.py save /tmp/ref.pt > gpurun_out/pp3_check.txt 2>&1
UTV2_PP=3 timeout 300 python tools/check_w8.py cmp /tmp/ref.pt >> gpurun_out/pp3_check.txt 2>&1
tail -13 gpurun_out/pp3_check.txt
UTV2_PP=2 timeout 300 python tools/bench_pp_overhead.py > gpurun_out/pp_overhead_2.txt 2>&1
UTV2_PP=3 timeout 300 python tools/bench_pp_overhead.py > gpurun_out/pp_overhead_3.txt 2>&1
tail -8 gpurun_out/pp_overhead_2.txt; echo ---; tail -8 gpurun_out/pp_overhead_3.txt
