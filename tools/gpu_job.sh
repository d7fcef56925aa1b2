#!/bin/bash
# scratch GPU job of the round (edited per run; outputs under gpurun_out/)
R=$PWD
python -m pytest tests/test_fcos_kernels_gpu.py tests/test_conv_ml_gpu.py tests/test_fcos_step_gpu.py tests/test_conv_bf16_gpu.py -x -q -m gpu > gpurun_out/t6.log 2>&1; tail -5 gpurun_out/t6.log
for rep in 1 2 3; do
  python bench.py --no-cpu-baseline --no-rcnn --no-f32 --timed-only > gpurun_out/ab_E_$rep.json 2>/dev/null
done
python bench.py --no-cpu-baseline --no-rcnn > gpurun_out/bench_r3f.json 2> gpurun_out/bench_r3f.err; tail -c 300 gpurun_out/bench_r3f.err
python - <<'PY'
import json, glob
for f in ["gpurun_out/bench_r3f.json"] + sorted(glob.glob("gpurun_out/ab_E_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, round(d["value"],2), round(d["ms_per_step"],3))
            if "f32" in d: print(json.dumps({k: d["f32"].get(k) for k in ("f16_vs_f32_first_step_rel_dev", "bf16_vs_f32_first_step_rel_dev", "f16")}))
PY
