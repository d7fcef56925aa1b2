# kernel traces of the 2+2 step replayed as a hipGraph (no host in the way): single stream and the step's own four streams
R=$PWD; mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
for m in "fcos f16" "rcnn bf16"; do
  set -- $m
  for ovl in 0 1; do
    d=_kt_$1_$ovl
    UTV2_OVERLAP_TEACHER=$ovl UTV2_WGRAD_STREAM=$ovl timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/$d -o run -- python $R/tools/small_batch_probe.py $1 $2 2 40 > $R/gpurun_out/r06_trace_$1_$ovl.log 2>&1 < /dev/null
    db=$(find $R/gpurun_out/$d -name '*.db' | head -1)
    timeout 120 python $R/tools/rocpd_steps.py "$db" 30 90 > $R/gpurun_out/r06_$1_2p2_graph_streams${ovl}_steps.txt 2>&1
    rm -rf $R/gpurun_out/$d
  done
done
cd $R; grep -h "enqueue" gpurun_out/r06_trace_*.log; head -12 gpurun_out/r06_fcos_2p2_graph_streams0_steps.txt
