#!/bin/bash
mkdir -p gpurun_out
for v in full a3 a9 full a3 a9 nodma; do timeout 60 tools/probe/pp_power_$v 3; done > gpurun_out/r05_pp_power_span.txt 2>&1
cat gpurun_out/r05_pp_power_span.txt
