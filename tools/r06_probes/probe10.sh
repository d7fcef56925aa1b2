mkdir -p gpurun_out
O=gpurun_out/r06_mfma16_timing.txt; : > $O
for rep in 1 2 3; do
  echo "shipped kernel (32x32x16)" >> $O; TOWER_N=12 timeout 120 python tools/bench_tower.py relu 2>/dev/null | grep fwd >> $O
  echo "timing build (2 x 16x16x32 per 32x32x16, garbage results)" >> $O; TOWER_N=12 UTV2_LIB_DIR=$PWD/unbiased-teacher-v2_amd/lib_v/mfma16 timeout 120 python tools/bench_tower.py relu 2>/dev/null | grep fwd >> $O
done
cat $O
