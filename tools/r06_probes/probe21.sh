mkdir -p gpurun_out
O=gpurun_out/r06_teacher_graph_ab.txt; : > $O
for rep in 1 2; do
for tg in 0 1; do
  for m in "fcos f16 2" "rcnn bf16 2" "fcos f16 4" "rcnn bf16 4"; do
    echo "UTV2_TEACHER_GRAPH=$tg" >> $O
    UTV2_TEACHER_GRAPH=$tg PROBE_NO_GRAPH=1 timeout 300 python tools/small_batch_probe.py $m 60 >> $O 2>gpurun_out/r06_tg_err.txt || tail -3 gpurun_out/r06_tg_err.txt >> $O
  done
done
done
grep -v "^$" $O | paste - - | sed 's/ovl=1 wgs=1: eager//'
