#!/bin/bash
# scratch: the GPU job of the moment
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fcos_kernels_gpu.py tests/test_rcnn_kernels_gpu.py -x -q -m gpu > gpurun_out/t20.log 2>&1
tail -4 gpurun_out/t20.log
B="--steps 23 --warmup 5 --no-cpu-baseline --no-rcnn --no-f32 --timed-only"
P=$PWD/unbiased-teacher-v2_amd/lib_prev
for r in 1 2 3; do
  UTV2_LIB_DIR=$P timeout 600 python bench.py $B --model rcnn > gpurun_out/ab_RT0_${r}.json 2> gpurun_out/ab_err.txt
  timeout 600 python bench.py $B --model rcnn > gpurun_out/ab_RT1_${r}.json 2> gpurun_out/ab_err.txt
done
for r in 1 2; do
  UTV2_LIB_DIR=$P timeout 600 python bench.py $B > gpurun_out/ab_T0_${r}.json 2> gpurun_out/ab_err.txt
  timeout 600 python bench.py $B > gpurun_out/ab_T1_${r}.json 2> gpurun_out/ab_err.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_RT*.json") + glob.glob("gpurun_out/ab_T[01]*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["dtype"], round(d["value"], 2), round(d["ms_per_step"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
