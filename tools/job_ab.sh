#!/bin/bash
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
( for rep in 1 2; do for w in 256 224 192; do
  echo "WGRAD_W8_WGS=$w: $(UTV2_WGRAD_W8_WGS=$w timeout 300 python bench.py --no-cpu-baseline --no-rcnn --no-f32 --timed-only --steps 40 --warmup 8 2>/dev/null | python -c "$P")"
done; done
for rep in 1 2; do for w in 256 240; do
  echo "PP_WGS=$w: $(UTV2_PP_WGS=$w timeout 300 python bench.py --no-cpu-baseline --no-rcnn --no-f32 --timed-only --steps 40 --warmup 8 2>/dev/null | python -c "$P")"
done; done ) > gpurun_out/r05_knobs_ab.txt 2>&1
cat gpurun_out/r05_knobs_ab.txt
