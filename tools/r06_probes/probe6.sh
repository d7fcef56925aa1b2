mkdir -p gpurun_out
timeout 300 python tools/aten_census.py rcnn 2 > gpurun_out/r06_aten_census_rcnn.txt 2>&1
timeout 300 python tools/aten_census.py fcos 2 > gpurun_out/r06_aten_census_fcos.txt 2>&1
O=gpurun_out/r06_wgs_pct_ab.txt; : > $O
for rep in 1 2; do
for pct in 100 75 50; do
  for m in "fcos f16 2" "fcos f16 4" "rcnn bf16 2"; do
    echo "UTV2_WGRAD_SMALL_WGS_PCT=$pct" >> $O
    UTV2_WGRAD_SMALL_WGS_PCT=$pct PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py $m 60 >> $O 2>/dev/null
  done
done
done
for lanes in 1 3 4; do
  echo "UTV2_WGRAD_LANES=$lanes" >> $O
  UTV2_WGRAD_LANES=$lanes PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py fcos f16 2 60 >> $O 2>/dev/null
done
cat $O
head -50 gpurun_out/r06_aten_census_rcnn.txt
