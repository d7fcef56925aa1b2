#!/bin/bash
mkdir -p gpurun_out
( UTV2_W8=0 timeout 300 python tools/check_w8.py save /tmp/ref.pt | tail -1
  echo "--- default (row span)"; timeout 300 python tools/check_w8.py cmp /tmp/ref.pt
  echo "--- UTV2_PP=1 (row span, one tile per workgroup)"; UTV2_PP=1 timeout 300 python tools/check_w8.py cmp /tmp/ref.pt
  echo "--- UTV2_PP_RS=0"; UTV2_PP_RS=0 timeout 300 python tools/check_w8.py cmp /tmp/ref.pt
  for i in 1 2; do
  echo "--- bench_tower rs"; TOWER_N=12 timeout 120 python tools/bench_tower.py relu
  echo "--- bench_tower pp"; UTV2_PP_RS=0 TOWER_N=12 timeout 120 python tools/bench_tower.py relu
  done ) > gpurun_out/rs1.txt 2>&1
cat gpurun_out/rs1.txt
