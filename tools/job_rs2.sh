#!/bin/bash
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"].get("sustained_clock_ghz"), d["roofline_wgrad"]["achieved"])'
( for rep in 1 2 3; do for rs in 0 1; do
  echo "fcos f16 rs=$rs: $(UTV2_PP_RS=$rs timeout 300 python bench.py --no-cpu-baseline --no-rcnn --no-f32 --timed-only --steps 40 --warmup 8 2>/dev/null | python -c "$P")"
done; done
for rs in 0 1; do echo "rcnn bf16 rs=$rs: $(UTV2_PP_RS=$rs timeout 300 python bench.py --model rcnn --no-cpu-baseline --no-f32 --timed-only --steps 40 --warmup 8 2>/dev/null | python -c "$P")"; done
echo "--- power probe rs"; timeout 100 python tools/power_probe.py 3 2>/dev/null | grep tower | cut -c1-250
echo "--- power probe pp"; UTV2_PP_RS=0 timeout 100 python tools/power_probe.py 3 2>/dev/null | grep tower | cut -c1-250 ) > gpurun_out/rs2.txt 2>&1
cat gpurun_out/rs2.txt
