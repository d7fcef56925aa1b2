#!/bin/bash
# Round measurement of record (run on the GPU box through gpurun): bench lines, kernel trace, PMC passes.
# usage: tools/measure_record.sh <tag>     outputs -> gpurun_out/<tag>_*
set -u
TAG=${1:-r01}
R=$PWD
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err < /dev/null
timeout 600 python bench.py --dtype f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_f32.json 2> /dev/null < /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/_kt -o run -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_kt.log 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/_pf -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/_pw -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 < /dev/null
cd $R
for d in _kt _pf _pw; do f=$(find gpurun_out/$d -name "*.db" | head -1); echo "$d $f"; done
timeout 120 python tools/rocpd_stats.py "$(find gpurun_out/_kt -name '*.db' | head -1)" > gpurun_out/${TAG}_kernel_stats.txt 2>&1 < /dev/null
timeout 120 python tools/rocpd_pmc.py "$(find gpurun_out/_pf -name '*.db' | head -1)" 30 > gpurun_out/${TAG}_pmc_fetch.txt 2>&1 < /dev/null
timeout 120 python tools/rocpd_pmc.py "$(find gpurun_out/_pw -name '*.db' | head -1)" 30 > gpurun_out/${TAG}_pmc_write.txt 2>&1 < /dev/null
timeout 60 python tools/make_traffic.py gpurun_out/${TAG}_pmc_fetch.txt gpurun_out/${TAG}_pmc_write.txt gpurun_out/${TAG}_traffic.json > /dev/null 2>&1 < /dev/null
rm -rf gpurun_out/_kt gpurun_out/_pf gpurun_out/_pw
