#!/bin/bash
# scratch: the GPU job of the moment
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_conv_bf16_gpu.py tests/test_fcos_step_gpu.py -x -q -m gpu > gpurun_out/t12.log 2>&1
tail -8 gpurun_out/t12.log
B="--steps 23 --warmup 5 --no-cpu-baseline --no-rcnn --no-f32 --timed-only"
for r in 1 2 3; do
  for f in 0 1; do
    UTV2_GRAD_HANDOFF=$f timeout 600 python bench.py $B > gpurun_out/ab_G${f}_${r}.json 2> gpurun_out/ab_err.txt
  done
done
for r in 1 2; do
  for f in 0 1; do
    UTV2_GRAD_HANDOFF=$f timeout 600 python bench.py $B --model rcnn > gpurun_out/ab_RG${f}_${r}.json 2> gpurun_out/ab_err.txt
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_G*.json") + glob.glob("gpurun_out/ab_RG*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["dtype"], round(d["value"], 2), round(d["ms_per_step"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
