mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 3000 gpurun_out/bench_default.json
