mkdir -p gpurun_out
timeout 900 python - > gpurun_out/r06_fullsize_rcnn_parity.json 2> gpurun_out/r06_fullsize_rcnn_parity.err <<'PY'
import sys, json, torch, tempfile, os
sys.path.insert(0, "."); sys.path.insert(0, "unbiased-teacher-v2_amd")
import bench
from tests.test_fullsize_gpu import rebuild
d = rebuild("rcnn")
p = os.path.join(tempfile.gettempdir(), "fs.pt"); torch.save(d, p)
print(json.dumps(bench.parity_fullsize(p, 0)))
PY
timeout 1500 python -m pytest tests -q -m gpu -x --deselect "tests/test_fullsize_gpu.py::test_fullsize_step_parity_vs_oracle_fixture[rcnn]" > gpurun_out/r06_gpu_suite_a.txt 2>&1
tail -5 gpurun_out/r06_gpu_suite_a.txt
timeout 1200 python bench.py > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err
tail -c 1500 gpurun_out/r06_bench_a.json
