#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
B="--steps 23 --warmup 5 --no-cpu-baseline --no-rcnn --no-f32 --timed-only"
for r in 1 2; do
  for f in 256 240 224; do
    UTV2_PP_WGS=$f timeout 600 python bench.py $B > gpurun_out/ab_G${f}_${r}.json 2> gpurun_out/ab_err.txt
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_G2[0-9][0-9]_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["dtype"], round(d["value"], 2), round(d["ms_per_step"], 3), round(d["roofline"]["frac"], 4))
    except Exception as e:
        print(f, "ERR", e)
PY
