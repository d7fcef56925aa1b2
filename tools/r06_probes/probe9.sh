mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_rcnn_kernels_gpu.py tests/test_rcnn_step_gpu.py tests/test_fcos_kernels_gpu.py -q -m gpu -x > gpurun_out/r06_probe9_tests.txt 2>&1
tail -4 gpurun_out/r06_probe9_tests.txt
O=gpurun_out/r06_fold_cap_ab.txt; : > $O
for rep in 1 2; do
for cap in 0 2 4 8; do
  for m in "fcos f16 2" "rcnn bf16 2" "rcnn bf16 4"; do
    if [ $cap = 0 ]; then echo "UTV2_WGRAD_FOLD=0" >> $O; UTV2_WGRAD_FOLD=0 PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py $m 60 >> $O 2>/dev/null
    else echo "UTV2_WGRAD_FOLD_CAP=$cap" >> $O; UTV2_WGRAD_FOLD_CAP=$cap PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py $m 60 >> $O 2>/dev/null; fi
  done
done
done
echo "UTV2_ROI_TOPK=0" >> $O; UTV2_WGRAD_FOLD=0 UTV2_ROI_TOPK=0 PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py rcnn bf16 2 60 >> $O 2>/dev/null
echo "UTV2_ROI_TOPK=0" >> $O; UTV2_WGRAD_FOLD=0 UTV2_ROI_TOPK=0 PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py rcnn bf16 2 60 >> $O 2>/dev/null
grep -v "^$" $O | paste - -
