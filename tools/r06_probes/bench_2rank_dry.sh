#!/bin/bash
# bench.py's N > 1 path on a ONE-GPU box: two ranks share the device (UTV2_BENCH_SINGLE_DEVICE=1), gloo carries the collectives
mkdir -p gpurun_out
UTV2_BENCH_SINGLE_DEVICE=1 UTV2_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 6 --warmup 3 --label 2 --unlabel 2 > gpurun_out/bench_2rank_dry.json 2> gpurun_out/bench_2rank_dry.err
echo rc $?
tail -c 1500 gpurun_out/bench_2rank_dry.json; tail -5 gpurun_out/bench_2rank_dry.err | cut -c1-300
