#!/bin/bash
# Round measurement of record (run on the GPU box through gpurun): bench lines, kernel traces, PMC passes for the FCOS (headline) and the
# Faster-RCNN step.  usage: tools/measure_record.sh <tag>     outputs -> gpurun_out/<tag>_*
set -u
TAG=${1:-r06}
R=$PWD
mkdir -p gpurun_out
# the default FCOS run is the fp16 AMP mode (the reference's own autocast type) since round 3; file names keep the "4p4_bf16" stem of the
# earlier rounds (same kernels, the 16-bit element type of the library build differs)
timeout 1200 python bench.py > gpurun_out/${TAG}_bench_f16.json 2> gpurun_out/${TAG}_bench_f16.err < /dev/null
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-rcnn > gpurun_out/${TAG}_bench_bf16.json 2> /dev/null < /dev/null
timeout 600 python bench.py --model rcnn > gpurun_out/${TAG}_bench_rcnn_bf16.json 2> /dev/null < /dev/null
timeout 600 python bench.py --model rcnn --dtype f32 --steps 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_rcnn_f32.json 2> /dev/null < /dev/null
# per-shape replay of every conv launch of the step (tools/profile_shapes.py): the evidence behind the kernel target list
timeout 600 python tools/profile_shapes.py > gpurun_out/${TAG}_conv_shapes.txt 2> /dev/null < /dev/null
timeout 600 python tools/profile_shapes.py --model rcnn > gpurun_out/${TAG}_rcnn_conv_shapes.txt 2> /dev/null < /dev/null
cd /tmp && export TMPDIR=/tmp
prof() {  # prof <dir> <extra rocprof args> -- <bench args>
  local d=$1; shift
  local pmc=()
  while [ "$1" != "--" ]; do pmc+=("$1"); shift; done
  shift
  timeout 600 rocprofv3 --kernel-trace "${pmc[@]}" -d $R/gpurun_out/$d -o run -- python $R/bench.py --no-cpu-baseline --no-f32 --timed-only "$@" > $R/gpurun_out/${TAG}_$d.log 2>&1 < /dev/null
}
prof _kt -- --steps 8 --warmup 2
prof _pf --pmc FETCH_SIZE -- --steps 3 --warmup 1
prof _pw --pmc WRITE_SIZE -- --steps 3 --warmup 1
prof _rkt -- --model rcnn --steps 8 --warmup 2
prof _rpf --pmc FETCH_SIZE -- --model rcnn --steps 3 --warmup 1
prof _rpw --pmc WRITE_SIZE -- --model rcnn --steps 3 --warmup 1
cd $R
db() { find gpurun_out/$1 -name '*.db' | head -1; }
timeout 120 python tools/rocpd_stats.py "$(db _kt)" > gpurun_out/${TAG}_fcos_4p4_f16_kernel_stats.txt 2>&1 < /dev/null
timeout 120 python tools/rocpd_pmc.py "$(db _pf)" 40 > gpurun_out/${TAG}_fcos_4p4_f16_pmc_fetch.txt 2>&1 < /dev/null
timeout 120 python tools/rocpd_pmc.py "$(db _pw)" 40 > gpurun_out/${TAG}_fcos_4p4_f16_pmc_write.txt 2>&1 < /dev/null
timeout 60 python tools/make_traffic.py gpurun_out/${TAG}_fcos_4p4_f16_pmc_fetch.txt gpurun_out/${TAG}_fcos_4p4_f16_pmc_write.txt \
  gpurun_out/${TAG}_fcos_4p4_f16_kernel_stats.txt gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.log 2>&1 < /dev/null
timeout 120 python tools/rocpd_timeline.py "$(db _kt)" steps 10 6 > gpurun_out/${TAG}_fcos_4p4_f16_timeline.txt 2>&1 < /dev/null
timeout 120 python tools/rocpd_gaps.py "$(db _kt)" 10 6 30 > gpurun_out/${TAG}_fcos_4p4_f16_gaps.txt 2>&1 < /dev/null
timeout 120 python tools/rocpd_solo.py "$(db _kt)" 10 6 40 > gpurun_out/${TAG}_fcos_4p4_f16_solo.txt 2>&1 < /dev/null
timeout 120 python tools/rocpd_timeline.py "$(db _rkt)" steps 10 6 > gpurun_out/${TAG}_rcnn_4p4_bf16_timeline.txt 2>&1 < /dev/null
timeout 120 python tools/rocpd_stats.py "$(db _rkt)" > gpurun_out/${TAG}_rcnn_4p4_bf16_kernel_stats.txt 2>&1 < /dev/null
timeout 120 python tools/rocpd_pmc.py "$(db _rpf)" 60 > gpurun_out/${TAG}_rcnn_4p4_bf16_pmc_fetch.txt 2>&1 < /dev/null
timeout 120 python tools/rocpd_pmc.py "$(db _rpw)" 60 > gpurun_out/${TAG}_rcnn_4p4_bf16_pmc_write.txt 2>&1 < /dev/null
timeout 60 python tools/make_traffic.py gpurun_out/${TAG}_rcnn_4p4_bf16_pmc_fetch.txt gpurun_out/${TAG}_rcnn_4p4_bf16_pmc_write.txt \
  gpurun_out/${TAG}_rcnn_4p4_bf16_kernel_stats.txt gpurun_out/${TAG}_rcnn_traffic.json > gpurun_out/${TAG}_rcnn_traffic.log 2>&1 < /dev/null
rm -rf gpurun_out/_kt gpurun_out/_pf gpurun_out/_pw gpurun_out/_rkt gpurun_out/_rpf gpurun_out/_rpw
tail -c 600 gpurun_out/${TAG}_bench_f16.json; echo; tail -c 400 gpurun_out/${TAG}_bench_rcnn_bf16.json; echo; tail -30 gpurun_out/${TAG}_rcnn_traffic.log
