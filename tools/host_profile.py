import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
import bench
from ubteacher.engine import UBTeacherTrainer, UBRCNNTeacherTrainer
from ubteacher.presets import get_config
model = sys.argv[1] if len(sys.argv) > 1 else "fcos"
cfg = get_config(model, 1, ["SOLVER.IMG_PER_BATCH_LABEL", 4, "SOLVER.IMG_PER_BATCH_UNLABEL", 4, "SEMISUPNET.BURN_UP_STEP", 0, "SOLVER.AMP.ENABLED", True, "MODEL.DEVICE", "cuda"])
torch.manual_seed(0)
tr = (UBRCNNTeacherTrainer if model == "rcnn" else UBTeacherTrainer)(cfg)
(bench.tune_rcnn_for_pseudo_labels if model == "rcnn" else bench.tune_for_pseudo_labels)(tr, tr._data_loader.batches[0])
tr.iter = 1; tr.log_period = 10 ** 9
for _ in range(5):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.synchronize()
# where inside a step does the host spend its time?  cProfile over 10 steps
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(10):
    tr.run_step_full_semisup(); tr.iter += 1
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step, then %.2f ms until the GPU is done" % ((t1 - t0) * 100, (t2 - t1) * 1e3))
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(30)
