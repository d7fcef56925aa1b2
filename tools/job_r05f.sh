#!/bin/bash
# per-GPU batch 2+2 (configs[2] / [4]: 16+16 over 8 GPUs) and 1+1: is the step host-bound there?
mkdir -p gpurun_out
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["host"]["enqueue_ms_per_step"], d["host"].get("cabi_calls_per_step"))'
for m in rcnn fcos; do for b in 4 2 1; do
  echo "$m ${b}+${b} bf16: $(timeout 300 python bench.py --model $m --dtype bf16 --label $b --unlabel $b --no-cpu-baseline --no-rcnn --no-f32 --timed-only --steps 40 --warmup 8 2>/dev/null | python -c "$P")"
done; done > gpurun_out/r05_small_batch.txt 2>&1
cat gpurun_out/r05_small_batch.txt
