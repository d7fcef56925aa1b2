#!/bin/bash
mkdir -p gpurun_out
timeout 120 python tools/power_probe.py 4 > gpurun_out/r05_power_probe.txt 2>&1
(timeout 100 tools/probe/mfma_peak 2.5 256 2; timeout 100 tools/probe/mfma_peak 2.5 256 1; timeout 100 tools/probe/mfma_peak 2.5 128 2) > gpurun_out/r05_mfma_peak.txt 2>&1
cat gpurun_out/r05_power_probe.txt | cut -c1-700; cat gpurun_out/r05_mfma_peak.txt
