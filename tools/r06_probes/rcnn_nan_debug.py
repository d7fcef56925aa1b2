"""first non-finite box-regression loss of the Faster-RCNN trainer started from random weights with the shipped config (soak run finding)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd")); sys.path.insert(0, ROOT)
import torch
from ubteacher import hip
from ubteacher.engine import UBRCNNTeacherTrainer
from ubteacher.presets import get_config
cfg = get_config("rcnn", 1, ["SOLVER.IMG_PER_BATCH_LABEL", 2, "SOLVER.IMG_PER_BATCH_UNLABEL", 2, "SEMISUPNET.BURN_UP_STEP", 40, "MODEL.DEVICE", "cuda"])
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if seed >= 0:
    torch.manual_seed(seed)
tr = UBRCNNTeacherTrainer(cfg)
tr.log_period = 1
orig = hip.roi_box_loss
state = {"n": 0, "hit": False}
def wrapped(deltas, std, cls, prop, gtb, gstd, *a):
    out = orig(deltas, std, cls, prop, gtb, gstd, *a)
    state["n"] += 1
    if not state["hit"] and not torch.isfinite(out[0]).all():
        state["hit"] = True
        fg = (cls >= 0) & (cls < a[0])
        print("call", state["n"], "iter", tr.iter, "mode", a[1], "R", deltas.shape[0], "fg", int(fg.sum()), "empty", int((cls < 0).sum()))
        for nm, t in (("deltas", deltas), ("std", std), ("prop", prop), ("gtb", gtb)):
            t = t.float()
            print(" ", nm, "finite rows all/fg/bg/empty:", bool(torch.isfinite(t).all()), bool(torch.isfinite(t[fg]).all()),
                  bool(torch.isfinite(t[(cls == a[0])]).all()), bool(torch.isfinite(t[cls < 0]).all()),
                  "absmax fg %.3g" % float(t[fg].abs().max()) if fg.any() else "")
        s = std.float()
        print("  std logits min/max fg", float(s[fg].min()), float(s[fg].max()), "all", float(s.min()), float(s.max()))
        w = (prop[:, 2] - prop[:, 0]); h = (prop[:, 3] - prop[:, 1])
        print("  fg prop w min %.3g h min %.3g; gt w min %.3g" % (float(w[fg].min()), float(h[fg].min()), float((gtb[:, 2] - gtb[:, 0])[fg].min())))
        print("  grads finite:", bool(torch.isfinite(out[1]).all()), bool(torch.isfinite(out[2]).all()))
    return out
hip.roi_box_loss = wrapped
from ubteacher.d2.events import EventStorage
with EventStorage(0) as tr.storage:
    for it in range(60):
        tr.iter = it
        tr.run_step_full_semisup()
        tr.scheduler.step()
        tr.storage.step()
        m = tr.flush_metrics()
        if state["hit"] or any(v != v for v in m.values()):
            break
print("seed", seed, "calls", state["n"], "stopped at iter", tr.iter, {k: round(v, 4) for k, v in m.items() if "loss" in k})
