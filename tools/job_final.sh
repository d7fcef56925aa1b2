#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/final_build.log 2>&1
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/final_gputests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -3 gpurun_out/final_build.log; grep -E "passed|failed" gpurun_out/final_gputests.log | tail -2; tail -2 gpurun_out/final_smoke.log; head -c 400 gpurun_out/final_bench.json
