#!/bin/bash
mkdir -p gpurun_out
( UTV2_WGRAD_PP=0 timeout 200 python tools/check_wgrad_pp.py save /tmp/wref.pt | tail -1 | cut -c1-300
  echo "--- UTV2_WGRAD_PP=1"; timeout 200 python tools/check_wgrad_pp.py cmp /tmp/wref.pt | grep -v amdgpu
  for rep in 1 2; do
  echo "--- bench_tower pp-wgrad"; TOWER_N=12 timeout 120 python tools/bench_tower.py relu | grep wgrad
  echo "--- bench_tower w8-wgrad"; UTV2_WGRAD_PP=0 TOWER_N=12 timeout 120 python tools/bench_tower.py relu | grep wgrad
  done ) > gpurun_out/wg1.txt 2>&1
cat gpurun_out/wg1.txt
