mkdir -p gpurun_out
O=gpurun_out/r06_small_grid_n64_ab.txt; : > $O
for rep in 1 2; do
for t in 0 256 512 1024; do
  for m in "fcos f16 2" "fcos f16 4" "rcnn bf16 2"; do
    echo "UTV2_SMALL_GRID_N64=$t" >> $O
    UTV2_SMALL_GRID_N64=$t PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py $m 60 >> $O 2>/dev/null
  done
done
done
grep -v "^$" $O | paste - - | sed 's/ovl=1 wgs=1: eager//'
