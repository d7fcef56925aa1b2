#!/bin/bash
# scratch GPU job of the round (edited per run; outputs under gpurun_out/)
python -m pytest tests/test_rcnn_step_gpu.py tests/test_rcnn_kernels_gpu.py -x -q -m gpu > gpurun_out/t8.log 2>&1; tail -5 gpurun_out/t8.log
for rep in 1 2 3; do
  UTV2_FUSE_LOSS_TAIL=1 python bench.py --model rcnn --no-cpu-baseline --timed-only > gpurun_out/ab_H1_$rep.json 2>/dev/null
  UTV2_FUSE_LOSS_TAIL=0 python bench.py --model rcnn --no-cpu-baseline --timed-only > gpurun_out/ab_H0_$rep.json 2>/dev/null
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_H*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["dtype"], round(d["value"],2), round(d["ms_per_step"],3), d["host"]["cabi_calls_per_step"], round(d["host"]["enqueue_ms_per_step"],2), {k: round(v,5) for k,v in d["losses"].items()})
PY
