#!/bin/bash
mkdir -p gpurun_out
for v in full nomfma nodma noread mfmaonly dmaonly readonly skeleton full; do timeout 60 tools/probe/pp_power_$v 3; done > gpurun_out/r05_pp_power.txt 2>&1
cat gpurun_out/r05_pp_power.txt
