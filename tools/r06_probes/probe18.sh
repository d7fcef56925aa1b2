mkdir -p gpurun_out
O=gpurun_out/r06_roi_bwd_ab.txt; : > $O
OLD=$PWD/unbiased-teacher-v2_amd/lib_v/rcnn_old
for rep in 1 2; do for n in 12 6; do
  echo "one-stage sum (round 2)" >> $O; UTV2_LIB_DIR=$OLD timeout 120 python tools/bench_roi_bwd.py $n 2>/dev/null >> $O
  echo "two-stage separable sum" >> $O; timeout 120 python tools/bench_roi_bwd.py $n 2>/dev/null >> $O
done; done
grep -v "^$" $O | paste - -
timeout 900 python -m pytest tests/test_rcnn_kernels_gpu.py tests/test_rcnn_step_gpu.py tests/test_dp_gpu.py tests/test_fullsize_gpu.py -q -m gpu 2>&1 | tail -4
