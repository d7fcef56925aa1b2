#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fcos_step_gpu.py tests/test_fcos_kernels_gpu.py tests/test_conv_bf16_gpu.py -x -q -m gpu -k "not trainable_stem" > gpurun_out/t21.log 2>&1
tail -4 gpurun_out/t21.log
B="--steps 23 --warmup 5 --no-cpu-baseline --no-rcnn --no-f32 --timed-only"
for r in 1 2 3; do
  UTV2_FLIP_AHEAD=0 UTV2_SCALE_ML=0 timeout 600 python bench.py $B > gpurun_out/ab_F0_${r}.json 2> gpurun_out/ab_err.txt
  timeout 600 python bench.py $B > gpurun_out/ab_F1_${r}.json 2> gpurun_out/ab_err.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_F[01]_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["dtype"], round(d["value"], 2), round(d["ms_per_step"], 3), d["losses"]["loss_fcos_loc"])
    except Exception as e:
        print(f, "ERR", e)
PY
