mkdir -p gpurun_out
O=gpurun_out/r06_roi_bwd_ab.txt; : > $O
OLD=$PWD/unbiased-teacher-v2_amd/lib_v/rcnn_old
for rep in 1 2; do for n in 12 6; do
  echo "before (one dependent load per bin)" >> $O; UTV2_LIB_DIR=$OLD timeout 120 python tools/bench_roi_bwd.py $n 2>/dev/null >> $O
  echo "after (a bin row of loads in flight)" >> $O; timeout 120 python tools/bench_roi_bwd.py $n 2>/dev/null >> $O
done; done
for rep in 1 2; do
  echo "before" >> $O; UTV2_LIB_DIR=$OLD PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py rcnn bf16 4 40 >> $O 2>/dev/null
  echo "after" >> $O; PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py rcnn bf16 4 40 >> $O 2>/dev/null
  echo "before" >> $O; UTV2_LIB_DIR=$OLD PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py rcnn bf16 2 40 >> $O 2>/dev/null
  echo "after" >> $O; PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py rcnn bf16 2 40 >> $O 2>/dev/null
done
grep -v "^$" $O | paste - -
timeout 600 python -m pytest tests/test_rcnn_kernels_gpu.py tests/test_rcnn_step_gpu.py -q -m gpu -x 2>&1 | tail -2
