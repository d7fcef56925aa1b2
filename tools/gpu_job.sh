#!/bin/bash
# scratch GPU job of the round (edited per run; outputs under gpurun_out/)
R=$PWD
python -m pytest tests/test_fcos_kernels_gpu.py tests/test_fcos_step_gpu.py -x -q -m gpu > gpurun_out/t5.log 2>&1; tail -5 gpurun_out/t5.log
python bench.py --no-cpu-baseline --no-rcnn > gpurun_out/bench_r3e.json 2> gpurun_out/bench_r3e.err; tail -c 300 gpurun_out/bench_r3e.err
for rep in 1 2; do
  for cfg in "A UTV2_GN_REVERSE=1" "B UTV2_GN_REVERSE=0" "C UTV2_WGRAD_W8_WGS=216" "D UTV2_GN_STATS_FUSED=0"; do
    set -- $cfg
    env $2 python bench.py --no-cpu-baseline --no-rcnn --no-f32 --timed-only > gpurun_out/ab_$1_$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in ["gpurun_out/bench_r3e.json"] + sorted(glob.glob("gpurun_out/ab_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, round(d["value"],2), round(d["ms_per_step"],3))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/_kt -o run -- python $R/bench.py --no-cpu-baseline --no-f32 --no-rcnn --timed-only --steps 8 --warmup 2 > $R/gpurun_out/kt.log 2>&1 < /dev/null
cd $R
db=$(find gpurun_out/_kt -name '*.db' | head -1)
python tools/rocpd_stats.py "$db" > gpurun_out/r3e_kernel_stats.txt 2>&1
python tools/rocpd_timeline.py "$db" steps 10 6 > gpurun_out/r3e_timeline.txt 2>&1
rm -rf gpurun_out/_kt
head -40 gpurun_out/r3e_kernel_stats.txt | cut -c1-200
