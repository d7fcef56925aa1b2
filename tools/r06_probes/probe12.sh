mkdir -p gpurun_out
O=gpurun_out/r06_rs_imm_offsets.txt; : > $O
BASE=$PWD/unbiased-teacher-v2_amd/lib_v/base
UTV2_LIB_DIR=$BASE timeout 300 python tools/check_w8.py save /tmp/refb.pt > /dev/null 2>&1
echo "## new build, 32x32x16 form (UTV2_RS_MFMA16=0) against the committed build" >> $O
UTV2_RS_MFMA16=0 timeout 300 python tools/check_w8.py cmp /tmp/refb.pt 2>/dev/null >> $O
echo "## new build, 16x16x32 form against the committed build" >> $O
UTV2_RS_MFMA16=1 timeout 300 python tools/check_w8.py cmp /tmp/refb.pt 2>/dev/null >> $O
echo "## single tower launch, 12 images, post-ReLU data (tools/bench_tower.py relu): committed build | new 32x32x16 | new 16x16x32" >> $O
for rep in 1 2 3 4; do
  echo "committed" >> $O; TOWER_N=12 UTV2_LIB_DIR=$BASE timeout 120 python tools/bench_tower.py relu 2>/dev/null | grep fwd >> $O
  echo "new, UTV2_RS_MFMA16=0" >> $O; TOWER_N=12 UTV2_RS_MFMA16=0 timeout 120 python tools/bench_tower.py relu 2>/dev/null | grep fwd >> $O
  echo "new, UTV2_RS_MFMA16=1" >> $O; TOWER_N=12 UTV2_RS_MFMA16=1 timeout 120 python tools/bench_tower.py relu 2>/dev/null | grep fwd >> $O
done
cat $O | grep -v "bit-identical"
grep -c "bit-identical" $O
