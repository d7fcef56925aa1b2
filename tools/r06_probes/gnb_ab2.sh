#!/bin/bash
# longer A/B of UTV2_GN_BWD_FUSE at 4+4 (FCOS f16): 6 interleaved pairs of 100 timed steps on one box
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-rcnn --no-graph --no-small --no-f32 --timed-only --steps 100 --warmup 10 --dtype f16"
for rep in 1 2 3 4 5 6; do
  for f in 0 1; do
    UTV2_GN_BWD_FUSE=$f timeout 600 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$f rep=$rep  %.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" >> gpurun_out/gnb_ab2.txt
  done
done
cat gpurun_out/gnb_ab2.txt
