#!/bin/bash
mkdir -p gpurun_out
V=$PWD/unbiased-teacher-v2_amd/lib_v
for x in p0 p1 p3 p1l; do
  UTV2_LIB_DIR=$V/$x timeout 300 python tools/profile_shapes.py --dtype bf16 > gpurun_out/r05_shapes_$x.txt 2> gpurun_out/r05_shapes_$x.err
done
for rep in 1 2 3; do for x in p0 p1 p3 p1l; do
  echo "$x $(UTV2_LIB_DIR=$V/$x timeout 300 python bench.py --dtype bf16 --no-cpu-baseline --no-rcnn --no-f32 --timed-only --steps 40 --warmup 8 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline_wgrad"]["achieved"])')"
done; done > gpurun_out/r05_prio_ab.txt 2>&1
cat gpurun_out/r05_prio_ab.txt; head -3 gpurun_out/r05_shapes_p0.txt; head -3 gpurun_out/r05_shapes_p1.txt | tail -2;  head -3 gpurun_out/r05_shapes_p3.txt | tail -2; head -3 gpurun_out/r05_shapes_p1l.txt | tail -2
