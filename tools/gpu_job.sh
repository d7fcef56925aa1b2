#!/bin/bash
# scratch GPU job of the round (edited per run; outputs under gpurun_out/)
R=$PWD
python tools/aten_census.py rcnn > gpurun_out/aten_rcnn.txt 2>&1; head -70 gpurun_out/aten_rcnn.txt | cut -c1-170
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/_rkt -o run -- python $R/bench.py --model rcnn --no-cpu-baseline --no-f32 --timed-only --steps 8 --warmup 2 > $R/gpurun_out/rkt.log 2>&1 < /dev/null
cd $R
db=$(find gpurun_out/_rkt -name '*.db' | head -1)
python tools/rocpd_gaps.py "$db" 10 6 45 > gpurun_out/r3_rcnn_gaps.txt 2>&1
rm -rf gpurun_out/_rkt
head -50 gpurun_out/r3_rcnn_gaps.txt | cut -c1-190
