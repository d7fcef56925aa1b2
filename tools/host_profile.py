"""Where does the HOST spend its time enqueuing one UTv2 step?  cProfile over K steps at benchmark size with the GPU left to run behind
(no sync inside the window), sorted by own time.   python tools/host_profile.py [fcos|rcnn] [K] [images per list, default 4] [small]
`small`: 96 x 128 images - the GPU work is negligible, the step time IS the Python / launch cost (bench.py host.ms_per_step_on_96x128_images)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
import bench
from ubteacher.engine import UBTeacherTrainer, UBRCNNTeacherTrainer
from ubteacher.presets import get_config

model = sys.argv[1] if len(sys.argv) > 1 else "fcos"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
small = len(sys.argv) > 4 and sys.argv[4] == "small"
cfg = get_config(model, 1, ["SOLVER.IMG_PER_BATCH_LABEL", B, "SOLVER.IMG_PER_BATCH_UNLABEL", B, "SEMISUPNET.BURN_UP_STEP", 0,
                            "SOLVER.AMP.ENABLED", True, "MODEL.DEVICE", "cuda"])
torch.manual_seed(0)
loader = None
if small:
    from ubteacher.data.synthetic import SyntheticTwoCropLoader
    loader = SyntheticTwoCropLoader(cfg, height=96, width=128)
tr = (UBRCNNTeacherTrainer if model == "rcnn" else UBTeacherTrainer)(cfg, data_loader=loader)
(bench.tune_rcnn_for_pseudo_labels if model == "rcnn" else bench.tune_for_pseudo_labels)(tr, tr._data_loader.batches[0])
tr.iter = 1; tr.log_period = 10 ** 9
for g in tr.optimizer.param_groups:
    g["lr"] = 1e-12
for _ in range(5):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    tr.run_step_full_semisup(); tr.iter += 1
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s: enqueue %.2f ms/step, wall %.2f ms/step (no profiler)" % (model, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    tr.run_step_full_semisup(); tr.iter += 1
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
tot = sum(v[2] for v in st.stats.values())
print("profiled own time %.2f ms/step" % (tot / K * 1e3))
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:60]
for (fn, line, name), (cc, nc, tt, ct, _) in rows:
    print("%8.3f ms/step own %8.3f cum  %7.1f calls/step  %s:%d %s" % (tt / K * 1e3, ct / K * 1e3, nc / K, fn.replace(ROOT + "/", "")[-60:], line, name))
