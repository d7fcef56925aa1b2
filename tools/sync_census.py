import os, sys, warnings, traceback, collections
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
for _ in range(3):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.synchronize()
seen = collections.Counter()
def showwarning(message, category, filename, lineno, file=None, line=None):
    st = [f for f in traceback.extract_stack() if "unbiased-teacher-v2_amd" in f.filename]
    where = "%s:%d %s" % (st[-1].filename.split("unbiased-teacher-v2_amd/")[-1], st[-1].lineno, st[-1].name) if st else "?"
    seen[(str(message)[:60], where)] += 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
for _ in range(2):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
for (m, w), n in seen.most_common(40):
    print("%5.1f/step  %-62s %s" % (n / 2, m, w))
