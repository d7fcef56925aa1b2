#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
B="--steps 23 --warmup 5 --no-cpu-baseline --no-rcnn --no-f32 --timed-only --model rcnn"
for r in 1 2 3; do
  for f in 0 1; do
    UTV2_AUX_STREAM=$f timeout 600 python bench.py $B > gpurun_out/ab_X${f}_${r}.json 2> gpurun_out/ab_err_$f.txt
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_X[01]_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["dtype"], round(d["value"], 2), round(d["ms_per_step"], 3), d["losses"]["loss_rpn_cls_pseudo"], d["losses"]["loss_cls"])
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/ab_err_1.txt
