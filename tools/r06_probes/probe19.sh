mkdir -p gpurun_out
timeout 600 python bench.py --subrecord-only --ragged --model fcos --dtype f16 --label 4 --unlabel 4 --steps 24 --warmup 16 > gpurun_out/r06_ragged_sub.json 2>gpurun_out/r06_ragged_sub.err
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06_ragged_sub.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["enqueue_ms_per_step"], r["pseudo_boxes_per_batch_of_the_cycle"], {k: round(v, 3) for k, v in r["losses"].items()})
PY
timeout 600 python bench.py --subrecord-only --ragged --model rcnn --dtype bf16 --label 4 --unlabel 4 --steps 24 --warmup 16 > gpurun_out/r06_ragged_sub_rcnn.json 2>gpurun_out/r06_ragged_sub.err
python - <<'PY'
import json
r = json.loads([l for l in open("gpurun_out/r06_ragged_sub_rcnn.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["enqueue_ms_per_step"], r["pseudo_boxes_per_batch_of_the_cycle"], {k: round(v, 3) for k, v in r["losses"].items()})
PY
