"""How far ahead of the GPU does the host run?  Enqueue time per UTv2 step (no synchronisation) vs wall time per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
import bench
from ubteacher.engine import UBTeacherTrainer
from ubteacher.presets import get_config

cfg = get_config("fcos", 1, ["SOLVER.IMG_PER_BATCH_LABEL", 4, "SOLVER.IMG_PER_BATCH_UNLABEL", 4, "SEMISUPNET.BURN_UP_STEP", 0,
                             "SOLVER.AMP.ENABLED", True, "MODEL.DEVICE", "cuda"])
torch.manual_seed(0)
small = len(sys.argv) > 1 and sys.argv[1] == "small"   # tiny images: the GPU is fast, the step time is the HOST cost per step
if small:
    from ubteacher.data.synthetic import SyntheticTwoCropLoader
    tr = UBTeacherTrainer(cfg, data_loader=SyntheticTwoCropLoader(cfg, height=96, width=128))
else:
    tr = UBTeacherTrainer(cfg)
bench.tune_for_pseudo_labels(tr, tr._data_loader.batches[0])
tr.iter = 1; tr.log_period = 10 ** 9
for _ in range(3):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    tr.run_step_full_semisup(); tr.iter += 1
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue %.2f ms/step, wall %.2f ms/step (host ahead by %.1f ms at the end of %d steps)" % (1e3 * (t1 - t0) / K, 1e3 * (t2 - t0) / K, 1e3 * (t2 - t1), K))

# GPU-side lag of the host at the step boundary: an event recorded right after a step's last launch and one right before the next step's
# first launch are neighbours in the stream; the time between them is how long the GPU sat at the boundary waiting for the host
ends, starts = [], []
for _ in range(K):
    e0 = torch.cuda.Event(enable_timing=True); e0.record(); starts.append(e0)
    tr.run_step_full_semisup(); tr.iter += 1
    e1 = torch.cuda.Event(enable_timing=True); e1.record(); ends.append(e1)
torch.cuda.synchronize()
lag = [ends[i].elapsed_time(starts[i + 1]) * 1e3 for i in range(K - 1)]
step = [starts[i].elapsed_time(ends[i]) for i in range(K)]
print("GPU time per step %.3f ms; boundary lag (GPU idle until the host's next step arrives): mean %.1f us, max %.1f us" % (sum(step) / K, sum(lag) / len(lag), max(lag)))
