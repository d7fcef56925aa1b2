mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/mall_reuse tools/probe/mall_reuse.hip 2>/dev/null && timeout 300 /tmp/mall_reuse > gpurun_out/r06_mall_reuse.txt 2>&1
cat gpurun_out/r06_mall_reuse.txt
O=gpurun_out/r06_step_gc_ab.txt; : > $O
for rep in 1 2; do
for gc in 1 0; do
  for m in "fcos f16" "rcnn bf16"; do
    echo "UTV2_STEP_GC=$gc" >> $O
    UTV2_STEP_GC=$gc PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py $m 2 100 >> $O 2>/dev/null
  done
done
done
cat $O
timeout 300 python tools/bench_gn.py 12 > gpurun_out/r06_bench_gn.txt 2>&1; timeout 300 python tools/bench_gn.py 6 >> gpurun_out/r06_bench_gn.txt 2>&1; timeout 300 python tools/bench_gn.py 3 >> gpurun_out/r06_bench_gn.txt 2>&1; cat gpurun_out/r06_bench_gn.txt
