"""Can one whole UTv2 step be captured as a hipGraph (torch.cuda.graph) and replayed?  Eager vs replay time on the bench workload.
usage: python tools/graph_probe.py [fcos|rcnn] [steps]      (UTV2_FLIP_AHEAD=0 is forced: the dgrad weight images are refreshed inside
the step, a refresh that starts after the optimizer and is joined by the NEXT step's backward cannot live inside a one-step capture)"""
import os, sys, time
os.environ.setdefault("UTV2_FLIP_AHEAD", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
import bench
from ubteacher.engine import UBTeacherTrainer, UBRCNNTeacherTrainer
from ubteacher.presets import get_config

model = sys.argv[1] if len(sys.argv) > 1 else "fcos"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if model == "fcos":
    os.environ["UTV2_PRECISION"] = "fp16"
cfg = get_config(model, 1, ["SOLVER.IMG_PER_BATCH_LABEL", 4, "SOLVER.IMG_PER_BATCH_UNLABEL", 4, "SEMISUPNET.BURN_UP_STEP", 0,
                            "SOLVER.AMP.ENABLED", True, "MODEL.DEVICE", "cuda"])
torch.manual_seed(0)
tr = (UBRCNNTeacherTrainer if model == "rcnn" else UBTeacherTrainer)(cfg)
tr.iter = 1; tr.log_period = 10 ** 9
tr.optimizer.param_groups[0]["lr"] = 1e-12
(bench.tune_rcnn_for_pseudo_labels if model == "rcnn" else bench.tune_for_pseudo_labels)(tr, tr._data_loader.batches[0])
for _ in range(6):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / K * 1e3
rec_e = dict(tr.flush_metrics())
print("eager  %.3f ms/step" % eager, flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        tr.run_step_full_semisup()
except Exception as e:  # noqa: BLE001
    import traceback
    traceback.print_exc()
    print("CAPTURE FAILED:", repr(e)[:400])
    sys.exit(0)
torch.cuda.synchronize()
pend = tr._pending_metrics
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    g.replay()
torch.cuda.synchronize()
rep = (time.perf_counter() - t0) / K * 1e3
tr._pending_metrics = pend
rec_g = dict(tr.flush_metrics())
print("replay %.3f ms/step   (eager %.3f)" % (rep, eager))
print("losses eager ", {k: round(v, 5) for k, v in rec_e.items() if k.startswith("loss")})
print("losses replay", {k: round(v, 5) for k, v in rec_g.items() if k.startswith("loss")})
