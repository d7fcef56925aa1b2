#!/bin/bash
# scratch GPU job of the round (edited per run; outputs under gpurun_out/)
R=$PWD
(time python bench.py) > gpurun_out/bench_r3g.json 2> gpurun_out/bench_r3g.err; tail -c 400 gpurun_out/bench_r3g.err
python bench.py --dtype bf16 --no-cpu-baseline --no-rcnn --no-f32 --timed-only > gpurun_out/bench_r3g_bf16.json 2>/dev/null
python bench.py --model rcnn --no-cpu-baseline --timed-only > gpurun_out/bench_r3g_rcnn.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_r3g","bench_r3g_bf16","bench_r3g_rcnn"):
    for l in open("gpurun_out/%s.json" % f):
        if l.startswith("{"):
            d=json.loads(l); print(f, d["dtype"], round(d["value"],2), round(d["ms_per_step"],3), d.get("loss_scale_state"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/_kt -o run -- python $R/bench.py --no-cpu-baseline --no-f32 --no-rcnn --timed-only --steps 8 --warmup 2 > $R/gpurun_out/kt.log 2>&1 < /dev/null
cd $R
db=$(find gpurun_out/_kt -name '*.db' | head -1)
python tools/rocpd_gaps.py "$db" 10 6 30 > gpurun_out/r3g_gaps.txt 2>&1
python tools/rocpd_gaps.py "$db" 10 6 30 1 > gpurun_out/r3g_gaps_q1.txt 2>&1
rm -rf gpurun_out/_kt
head -40 gpurun_out/r3g_gaps.txt | cut -c1-200
