mkdir -p gpurun_out
O=gpurun_out/r06_small_batch_probe.txt
: > $O
for m in "fcos f16" "rcnn bf16"; do
  timeout 300 python tools/small_batch_probe.py $m 2 30 >> $O 2>gpurun_out/probe_err.txt
  UTV2_OVERLAP_TEACHER=0 UTV2_WGRAD_STREAM=0 timeout 300 python tools/small_batch_probe.py $m 2 30 >> $O 2>>gpurun_out/probe_err.txt
  timeout 300 python tools/small_batch_probe.py $m 2 30 small >> $O 2>>gpurun_out/probe_err.txt
  UTV2_OVERLAP_TEACHER=0 UTV2_WGRAD_STREAM=0 timeout 300 python tools/small_batch_probe.py $m 2 30 small >> $O 2>>gpurun_out/probe_err.txt
done
timeout 300 python tools/host_profile.py fcos 10 2 small > gpurun_out/r06_host_profile_fcos.txt 2>&1
timeout 300 python tools/host_profile.py rcnn 10 2 small > gpurun_out/r06_host_profile_rcnn.txt 2>&1
cat $O
