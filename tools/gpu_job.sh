#!/bin/bash
# scratch GPU job of the round (edited per run; outputs under gpurun_out/)
python -m pytest tests/test_conv_bf16_gpu.py tests/test_conv_ml_gpu.py tests/test_conv_gpu.py tests/test_fcos_step_gpu.py tests/test_rcnn_step_gpu.py -x -q -m gpu > gpurun_out/t7.log 2>&1; tail -5 gpurun_out/t7.log
python tools/bench_pp_overhead.py 2>&1 | tail -3
for rep in 1 2 3; do
  python bench.py --dtype bf16 --no-cpu-baseline --no-rcnn --no-f32 --timed-only > gpurun_out/ab_F_$rep.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_F_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["dtype"], round(d["value"],2), round(d["ms_per_step"],3), round(d["roofline"]["frac"],4))
PY
