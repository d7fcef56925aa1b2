mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke.txt 2>&1; tail -2 gpurun_out/r06_smoke.txt
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r06_gpu_suite_d.txt 2>&1; tail -3 gpurun_out/r06_gpu_suite_d.txt
