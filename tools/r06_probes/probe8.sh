mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_conv_bf16_gpu.py tests/test_conv_ml_gpu.py tests/test_fcos_step_gpu.py tests/test_rcnn_step_gpu.py tests/test_fullsize_gpu.py tests/test_dp_gpu.py -q -m gpu -x > gpurun_out/r06_probe8_tests.txt 2>&1
tail -4 gpurun_out/r06_probe8_tests.txt
timeout 600 python tools/ragged_probe.py fcos f16 8 3 > gpurun_out/r06_ragged_probe.txt 2>&1
timeout 600 python tools/ragged_probe.py rcnn bf16 8 3 >> gpurun_out/r06_ragged_probe.txt 2>&1
grep -v amdgpu gpurun_out/r06_ragged_probe.txt
O=gpurun_out/r06_fold_ab.txt; : > $O
for rep in 1 2; do
for f in 1 0; do
  for m in "fcos f16 2" "fcos f16 4" "rcnn bf16 2" "rcnn bf16 4"; do
    echo "UTV2_WGRAD_FOLD=$f" >> $O
    UTV2_WGRAD_FOLD=$f PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py $m 60 >> $O 2>/dev/null
  done
done
done
grep -v "^$" $O | paste - -
