"""Per-step wall time of the ragged-canvas step (every image at its own ResizeShortestEdge size, the reference recipes' INPUT.MIN_SIZE_TRAIN
range): NB different batches cycled - the first cycle sees every shape for the first time (geometry tables, workspaces, allocator), later
cycles hit the caches.  usage: python tools/ragged_probe.py [fcos|rcnn] [f16|bf16] [NB=8] [cycles=3] [images per list=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
import bench
from ubteacher.engine import UBTeacherTrainer, UBRCNNTeacherTrainer
from ubteacher.presets import get_config
from ubteacher.data.synthetic import SyntheticTwoCropLoader

model = sys.argv[1] if len(sys.argv) > 1 else "fcos"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cycles = int(sys.argv[4]) if len(sys.argv) > 4 else 3
B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
bench.set_amp_type(dtype)
cfg = get_config(model, 1, ["SOLVER.IMG_PER_BATCH_LABEL", B, "SOLVER.IMG_PER_BATCH_UNLABEL", B, "SEMISUPNET.BURN_UP_STEP", 0,
                            "SOLVER.AMP.ENABLED", True, "MODEL.DEVICE", "cuda"])
torch.manual_seed(0)
loader = SyntheticTwoCropLoader(cfg, num_batches=NB, ragged=(400, 1200, 1333))
tr = (UBRCNNTeacherTrainer if model == "rcnn" else UBTeacherTrainer)(cfg, data_loader=loader)
tr.iter = 1; tr.log_period = 10 ** 9
tr.optimizer.param_groups[0]["lr"] = 1e-12
(bench.tune_rcnn_for_pseudo_labels if model == "rcnn" else bench.tune_for_pseudo_labels)(tr, loader.batches[0])
loader._i = 0
for c in range(cycles):
    ts = []
    for _ in range(NB):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.run_step_full_semisup(); tr.iter += 1
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("%s %s cycle %d: per-step ms (synchronised each step) %s  mean %.1f" % (model, dtype, c, " ".join("%.0f" % t for t in ts), sum(ts) / len(ts)), flush=True)
print("memory: allocated %.1f GB reserved %.1f GB" % (torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30))
