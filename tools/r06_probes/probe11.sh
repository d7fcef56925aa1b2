mkdir -p gpurun_out
O=gpurun_out/r06_rs_mfma16.txt; : > $O
UTV2_RS_MFMA16=0 timeout 300 python tools/check_w8.py save /tmp/ref16.pt > /dev/null 2>&1
echo "## rs on 16x16x32 against rs on 32x32x16 (check_w8 cases)" >> $O
UTV2_RS_MFMA16=1 timeout 300 python tools/check_w8.py cmpclose /tmp/ref16.pt 2>/dev/null >> $O
echo "## determinism of the 16x16x32 form (two processes)" >> $O
UTV2_RS_MFMA16=1 timeout 300 python tools/check_w8.py save /tmp/ref16b.pt > /dev/null 2>&1
UTV2_RS_MFMA16=1 timeout 300 python tools/check_w8.py cmp /tmp/ref16b.pt 2>/dev/null >> $O
echo "## single tower launch, 12 images, post-ReLU data (tools/bench_tower.py relu)" >> $O
for rep in 1 2 3; do
  for m in 0 1; do echo "UTV2_RS_MFMA16=$m" >> $O; TOWER_N=12 UTV2_RS_MFMA16=$m timeout 120 python tools/bench_tower.py relu 2>/dev/null | grep fwd >> $O; done
done
cat $O
