#!/bin/bash
# A/B of UTV2_GN_BWD_FUSE (GroupNorm backward's first reduction in the producing dgrad's epilogue): FCOS f16 4+4 / 2+2, interleaved runs on one box
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-rcnn --no-graph --no-small --no-f32 --timed-only --steps 50 --warmup 10 --dtype f16"
for rep in 1 2 3; do
  for f in 0 1; do
    for lb in 4 2; do
      echo "fuse=$f batch=$lb+$lb rep=$rep" >> gpurun_out/gnb_ab.txt
      UTV2_GN_BWD_FUSE=$f timeout 600 $B --label $lb --unlabel $lb 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  value %.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" >> gpurun_out/gnb_ab.txt
    done
  done
done
cat gpurun_out/gnb_ab.txt
