#!/bin/bash
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline_wgrad"]["achieved"])'
( (time timeout 1500 python -m pytest tests -m gpu -x -q) 2>&1 | grep -E "passed|failed|real"
for rep in 1 2; do for w in default 256; do
  if [ $w = default ]; then unset UTV2_WGRAD_W8_WGS; else export UTV2_WGRAD_W8_WGS=$w; fi
  echo "fcos f16 WGRAD_W8_WGS=$w: $(timeout 300 python bench.py --no-cpu-baseline --no-rcnn --no-f32 --timed-only --steps 40 --warmup 8 2>/dev/null | python -c "$P")"
done; done ) > gpurun_out/r05_knobs_ab3.txt 2>&1
cat gpurun_out/r05_knobs_ab3.txt
