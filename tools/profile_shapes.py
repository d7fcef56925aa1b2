"""Per-shape timing of every conv launch (fwd / dgrad / wgrad) inside real FCOS UTv2 steps."""
import os, sys, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
import bench

recs = []
orig_call = hip.call
enabled = [False]

def call(name, *args):
    if enabled[0] and name in ("utv2_conv2d_nhwc_fwd", "utv2_conv2d_nhwc_wgrad"):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); orig_call(name, *args); e1.record()
        if name == "utv2_conv2d_nhwc_fwd":
            N, H, W, C, K, KH, KW, stride, pad, in_dil, OH, OW = args[6:18]
            kind = "dgrad" if (args[3].value is None and args[4].value is None and in_dil >= 1 and args[19] == 0 and args[18] == 0 and pad == KH - 1 - (KH // 2) and False) else "fwd"
            key = ("igemm", N, H, W, C, K, KH, stride, in_dil, OH, OW)
            fl = 2.0 * N * OH * OW * K * KH * KW * C / (in_dil * in_dil if in_dil > 1 else 1)
        else:
            N, H, W, C, K, KH, KW, stride, pad, OH, OW = args[4:15]
            key = ("wgrad", N, H, W, C, K, KH, stride, 1, OH, OW)
            fl = 2.0 * N * OH * OW * K * KH * KW * C
        recs.append((key, e0, e1, fl))
    else:
        orig_call(name, *args)

hip.call = call
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
# run bench main but toggle recording around the timed region by patching ConvTimer.enabled setter
class T(bench.ConvTimer):
    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)
        if k == "enabled": enabled[0] = bool(v)
bench.ConvTimer = T
bench.main()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1, fl in recs:
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print("total conv ms (2 steps): %.1f" % tot)
print("%-8s %3s %4s %4s %5s %5s %2s %2s %2s %4s %4s | %5s %9s %8s %6s" % ("kind","N","H","W","C","K","k","s","d","OH","OW","calls","ms","TF","pct"))
for key, (n, ms, fl) in rows[:60]:
    print("%-8s %3d %4d %4d %5d %5d %2d %2d %2d %4d %4d | %5d %9.3f %8.1f %5.1f%%" % (key + (n, ms, fl / ms / 1e9, 100 * ms / tot)))
