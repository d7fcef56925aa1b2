#!/bin/bash
# kernel traces of the FCOS f16 4+4 step with UTV2_GN_BWD_FUSE=0 / 1: which GroupNorm / tower kernels run and for how long
R=$PWD; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  UTV2_GN_BWD_FUSE=$f timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/_gnb$f -o run -- python $R/bench.py --no-cpu-baseline --no-rcnn --no-graph --no-small --no-f32 --timed-only --steps 8 --warmup 2 --dtype f16 > $R/gpurun_out/gnb_trace_$f.log 2>&1 < /dev/null
done
cd $R
for f in 0 1; do
  timeout 120 python tools/rocpd_stats.py "$(find gpurun_out/_gnb$f -name '*.db' | head -1)" > gpurun_out/gnb_kernel_stats_fuse$f.txt 2>&1
  timeout 120 python tools/rocpd_timeline.py "$(find gpurun_out/_gnb$f -name '*.db' | head -1)" steps 10 6 > gpurun_out/gnb_timeline_fuse$f.txt 2>&1
  rm -rf gpurun_out/_gnb$f
done
grep -h "gn_\|_rs<\|igemm_bf16_v2<128, true, 64" gpurun_out/gnb_kernel_stats_fuse0.txt | head -20
echo ----
grep -h "gn_\|_rs<\|igemm_bf16_v2<128, true, 64" gpurun_out/gnb_kernel_stats_fuse1.txt | head -20
