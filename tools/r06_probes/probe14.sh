mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err
python - <<'PY'
import json
b = json.loads([l for l in open("gpurun_out/r06_bench_b.json") if l.startswith("{")][-1])
print("headline", b["value"], b["ms_per_step"])
for k in ("fcos", "rcnn"):
    r = b["small_batch"][k]; print("small", k, r.get("value"), r.get("ms_per_step"), r.get("enqueue_ms_per_step"), r.get("error"))
r = b["ragged_canvases"]; print("ragged", r.get("value"), r.get("ms_per_step"), r.get("enqueue_ms_per_step"), r.get("error"))
print("rcnn", b["rcnn"]["value"], b["rcnn"]["parity_fullsize"]["within_tolerance"], b["parity_fullsize"]["within_tolerance"])
PY
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r06_gpu_suite_b.txt 2>&1
tail -4 gpurun_out/r06_gpu_suite_b.txt
