#!/bin/bash
mkdir -p gpurun_out
(echo "## 256 workgroups x 512 threads (two waves per SIMD, one workgroup)"; timeout 100 tools/probe/mfma_peak 2 256 2 2
 echo "## 512 workgroups x 256 threads (two waves per SIMD, two workgroups per CU)"; timeout 100 tools/probe/mfma_peak 2 512 3 2
 echo "## 256 workgroups x 256 threads (one wave per SIMD)"; timeout 100 tools/probe/mfma_peak 2 256 1 2) > gpurun_out/r05_mfma_peak2.txt 2>&1
cat gpurun_out/r05_mfma_peak2.txt
