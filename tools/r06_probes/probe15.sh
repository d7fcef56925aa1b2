mkdir -p gpurun_out
O=gpurun_out/r06_host_opt_ab.txt; : > $O
H=unbiased-teacher-v2_amd/ubteacher/hip.py
cp $H /tmp/hip_new.py
for rep in 1 2; do
  for v in new old; do
    if [ $v = old ]; then cp tools/probe/hip_before.py.txt $H; else cp /tmp/hip_new.py $H; fi
    for m in "fcos f16" "rcnn bf16"; do
      echo "hip.py $v" >> $O
      PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py $m 2 60 small >> $O 2>/dev/null
      echo "hip.py $v" >> $O
      PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py $m 2 60 >> $O 2>/dev/null
    done
  done
done
cp /tmp/hip_new.py $H
grep -v "^$" $O | paste - -
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r06_gpu_suite_c.txt 2>&1
tail -3 gpurun_out/r06_gpu_suite_c.txt
