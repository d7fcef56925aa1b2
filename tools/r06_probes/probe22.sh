mkdir -p gpurun_out
timeout 900 python - > gpurun_out/r06_rcnn_first_step_dev.json 2> gpurun_out/r06_rcnn_first_step_dev.err <<'PY'
import sys, json, argparse, torch
sys.path.insert(0, "."); sys.path.insert(0, "unbiased-teacher-v2_amd")
import bench
from ubteacher import hip
hip.load()
a = argparse.Namespace(label=4, unlabel=4)
print(json.dumps(bench.rcnn_first_step_deviation(a, 0)))
PY
tail -c 1500 gpurun_out/r06_rcnn_first_step_dev.json; tail -3 gpurun_out/r06_rcnn_first_step_dev.err
