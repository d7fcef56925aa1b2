mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_bf16_gpu.py tests/test_conv_ml_gpu.py tests/test_fcos_step_gpu.py tests/test_rcnn_step_gpu.py tests/test_fullsize_gpu.py tests/test_dp_gpu.py -q -m gpu -x > gpurun_out/r06_probe7_tests.txt 2>&1
tail -4 gpurun_out/r06_probe7_tests.txt
timeout 600 python tools/ragged_probe.py fcos f16 8 3 > gpurun_out/r06_ragged_probe.txt 2>&1
timeout 600 python tools/ragged_probe.py rcnn bf16 8 3 >> gpurun_out/r06_ragged_probe.txt 2>&1
grep -v amdgpu gpurun_out/r06_ragged_probe.txt
