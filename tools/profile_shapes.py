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
    names = ("utv2_conv2d_nhwc_fwd", "utv2_conv2d_nhwc_wgrad", "utv2_conv2d_nhwc_fwd_bf16", "utv2_conv2d_ml_fwd_bf16",
             "utv2_conv2d_ml_fwd", "utv2_conv2d_wgrad_bf16", "utv2_conv2d_ml_wgrad")
    if enabled[0] and name in names:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); orig_call(name, *args); e1.record()
        if name in ("utv2_conv2d_nhwc_fwd", "utv2_conv2d_nhwc_fwd_bf16"):
            N, H, W, C, K, KH, KW, stride, pad, in_dil, OH, OW = args[6:18]
            key = ("ig16" if "bf16" in name else "ig32", N, H, W, C, K, KH, stride, in_dil, OH, OW)
            fl = 2.0 * N * OH * OW * K * KH * KW * C / (in_dil * in_dil if in_dil > 1 else 1)
        elif name in ("utv2_conv2d_ml_fwd_bf16", "utv2_conv2d_ml_fwd"):
            nlev, _, _, N, C, K, KH, KW = args[6:14]
            key = ("ml16" if "bf16" in name else "ml32", N, 0, 0, C, K, KH, 1, 1, 0, 0)
            fl = 2.0 * N * 29841 * K * KH * KW * C
        elif name == "utv2_conv2d_wgrad_bf16":
            M, C, K, KH, KW = args[5:10]
            key = ("wg16", M, 0, 0, C, K, KH, 1, 1, 0, 0)
            fl = 2.0 * M * K * KH * KW * C
        elif name == "utv2_conv2d_ml_wgrad":
            key = ("mlwg32", 0, 0, 0, 0, 0, 0, 1, 1, 0, 0); fl = 0
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
for key, (n, ms, fl) in rows[:45]:
    print("%-8s %6d %4d %4d %5d %5d %2d %2d %2d %4d %4d | %5d %9.3f %8.1f %5.1f%%" % (key + (n, ms, fl / ms / 1e9, 100 * ms / tot)))
