mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/gpu_suite.txt 2>&1
tail -20 gpurun_out/gpu_suite.txt
