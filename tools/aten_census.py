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
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
os.environ.setdefault("UTV2_PRECISION", "fp16" if model == "fcos" else "bf16")
cfg = get_config(model, 1, ["SOLVER.IMG_PER_BATCH_LABEL", B, "SOLVER.IMG_PER_BATCH_UNLABEL", B, "SEMISUPNET.BURN_UP_STEP", 0,
                            "SOLVER.AMP.ENABLED", True, "MODEL.DEVICE", "cuda"])
torch.manual_seed(0)
tr = (UBRCNNTeacherTrainer if model == "rcnn" else UBTeacherTrainer)(cfg)
(bench.tune_rcnn_for_pseudo_labels if model == "rcnn" else bench.tune_for_pseudo_labels)(tr, tr._data_loader.batches[0])
tr.iter = 1; tr.log_period = 10 ** 9
for _ in range(3):
    tr.run_step_full_semisup(); tr.iter += 1
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
K = 2
agg = collections.Counter()


class Census(TorchDispatchMode):
    """every ATen op that reaches the dispatcher with a CUDA tensor, keyed by the innermost frame inside the package"""

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        flat = [a for a in list(args) + list((kwargs or {}).values()) + ([out] if isinstance(out, torch.Tensor) else list(out) if isinstance(out, (tuple, list)) else [])
                if isinstance(a, torch.Tensor)]
        if any(t.is_cuda for t in flat):
            name = str(func).replace("aten.", "")
            if not any(name.startswith(v) for v in ("view", "_unsafe_view", "reshape", "permute", "transpose", "slice", "select", "expand", "as_strided",
                                                    "unsqueeze", "squeeze", "detach", "alias", "t.", "empty", "unbind", "split", "_local_scalar", "narrow", "new_empty")):
                where = "?"
                for fr in reversed(traceback.extract_stack()):
                    if "unbiased-teacher-v2_amd" in fr.filename and "torch" not in fr.filename.split("unbiased-teacher-v2_amd")[-1]:
                        where = "%s:%d %s" % (fr.filename.split("unbiased-teacher-v2_amd/")[-1], fr.lineno, fr.name)
                        break
                agg[(name, where)] += 1
        return out


with Census():
    for _ in range(K):
        tr.run_step_full_semisup(); tr.iter += 1
    torch.cuda.synchronize()
tot = sum(agg.values())
print("ATen ops on CUDA tensors per step (views excluded): %.1f" % (tot / K))
for (name, where), n in agg.most_common(90):
    print("%6.1f  %-28s %s" % (n / K, name, where))
