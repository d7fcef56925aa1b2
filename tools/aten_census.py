"""Which ATen ops (and from which line of the package) does one UTv2 FCOS step launch?  torch.profiler over 2 steps, grouped by
(op, innermost frame inside unbiased-teacher-v2_amd/)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
import bench
from ubteacher.engine import UBTeacherTrainer, UBRCNNTeacherTrainer
from ubteacher.presets import get_config

model = sys.argv[1] if len(sys.argv) > 1 else "fcos"
cfg = get_config(model, 1, ["SOLVER.IMG_PER_BATCH_LABEL", 4, "SOLVER.IMG_PER_BATCH_UNLABEL", 4, "SEMISUPNET.BURN_UP_STEP", 0,
                            "SOLVER.AMP.ENABLED", True, "MODEL.DEVICE", "cuda"])
torch.manual_seed(0)
tr = (UBRCNNTeacherTrainer if model == "rcnn" else UBTeacherTrainer)(cfg)
(bench.tune_rcnn_for_pseudo_labels if model == "rcnn" else bench.tune_for_pseudo_labels)(tr, tr._data_loader.batches[0])
tr.iter = 1; tr.log_period = 10 ** 9
for _ in range(3):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
K = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(K):
        tr.run_step_full_semisup(); tr.iter += 1
    torch.cuda.synchronize()
agg = collections.Counter()
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA if hasattr(torch.autograd, "DeviceType") else False:
        continue
for ev in prof.events():
    name = ev.name
    if not name.startswith("aten::"):
        continue
    # only leaf-ish ops that launch kernels
    if not ev.kernels:
        continue
    where = "?"
    for fr in (ev.stack or []):
        if "unbiased-teacher-v2_amd" in fr or "bench.py" in fr:
            where = fr.split("unbiased-teacher-v2_amd/")[-1]
            break
    agg[(name, where)] += len(ev.kernels)
tot = sum(agg.values())
print("ATen kernel launches per step: %.1f" % (tot / K))
for (name, where), n in agg.most_common(70):
    print("%6.1f  %-28s %s" % (n / K, name, where))
