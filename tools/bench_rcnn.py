"""Step time of the Faster-RCNN UTv2 trainer on the benchmark shapes: 4 labeled + 4 unlabeled 1333x800 images,
random-init weights (no teacher tuning: pseudo boxes may be absent, the compute is the same).  Not the headline metric."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher.engine import UBRCNNTeacherTrainer
from ubteacher.presets import get_config

amp = len(sys.argv) > 1 and sys.argv[1] == "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = get_config("rcnn", 1, ["SOLVER.IMG_PER_BATCH_LABEL", 4, "SOLVER.IMG_PER_BATCH_UNLABEL", 4, "SEMISUPNET.BURN_UP_STEP", 0,
                             "SOLVER.AMP.ENABLED", amp, "MODEL.DEVICE", "cuda"])
torch.manual_seed(0)
tr = UBRCNNTeacherTrainer(cfg)
tr.iter = 1
tr.log_period = 10 ** 9
for _ in range(3):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({"model": "faster_rcnn_R_50_FPN_ut2", "dtype": "bf16" if amp else "f32", "ms_per_step": 1e3 * dt, "images_per_sec": 8 / dt}))
