#!/bin/bash
# scratch: the GPU job of the moment (run through gpurun from the repo root)
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/_kt -o run -- python $R/bench.py --no-cpu-baseline --no-f32 --timed-only --steps 8 --warmup 2 > $R/gpurun_out/r04_kt.log 2>&1 < /dev/null
cd $R
DB=$(find gpurun_out/_kt -name '*.db' | head -1)
python tools/rocpd_timeline.py "$DB" steps 10 6 > gpurun_out/r04_base_timeline.txt 2>&1
python tools/rocpd_gaps.py "$DB" 10 6 30 > gpurun_out/r04_base_gaps.txt 2>&1
python tools/rocpd_stats.py "$DB" > gpurun_out/r04_base_kernel_stats.txt 2>&1
python tools/rocpd_solo.py "$DB" 10 6 40 > gpurun_out/r04_base_solo.txt 2>&1
cp "$DB" gpurun_out/r04_base_kt.db
rm -rf gpurun_out/_kt
head -40 gpurun_out/r04_base_timeline.txt
