"""Who launches the fill / add / copy kernels of one UTv2 step?  torch.profiler with Python stacks over one step (all threads, so the
autograd thread's ops are seen too), grouped by (op, innermost package frames).  usage: fill_census.py [fcos|rcnn] [op substrings...]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
import bench
from ubteacher.engine import UBTeacherTrainer, UBRCNNTeacherTrainer
from ubteacher.presets import get_config

model = sys.argv[1] if len(sys.argv) > 1 else "fcos"
wanted = sys.argv[2:] or ["fill_", "zero_", "zeros", "ones", "full"]
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.run_step_full_semisup(); tr.iter += 1
    torch.cuda.synchronize()
agg = collections.Counter()
allops = collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith("aten::"):
        continue
    if not ev.kernels:
        continue
    allops[ev.name] += len(ev.kernels)
    if not any(w in ev.name for w in wanted):
        continue
    frames = [s for s in (ev.stack or []) if "unbiased-teacher-v2_amd" in s or "bench.py" in s]
    where = " <- ".join(f.split("unbiased-teacher-v2_amd/")[-1].strip() for f in frames[:3]) or "(autograd engine / no package frame)"
    agg[(ev.name, where)] += len(ev.kernels)
print("kernels launched by ATen ops in one step: %d" % sum(allops.values()))
for k, n in allops.most_common(40):
    print("%5d  %s" % (n, k))
print("--- %s ---" % wanted)
for (name, where), n in agg.most_common(60):
    print("%5d  %-18s %s" % (n, name, where))
