"""Per-shape timing of every conv launch (fwd / dgrad / wgrad) inside real FCOS UTv2 steps."""
import os, sys, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
import bench

recs = []
replay = {}
orig_call = hip.call
enabled = [False]

def call(name, *args):
    names = ("utv2_conv2d_nhwc_fwd", "utv2_conv2d_nhwc_wgrad", "utv2_conv2d_nhwc_fwd_bf16", "utv2_conv2d_ml_fwd_bf16",
             "utv2_conv2d_ml_fwd", "utv2_conv2d_wgrad_bf16", "utv2_conv2d_ml_wgrad")
    if enabled[0] and name in names:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); orig_call(name, *args); e1.record()
        if name in ("utv2_conv2d_nhwc_fwd", "utv2_conv2d_nhwc_fwd_bf16"):
            b16 = "bf16" in name
            o = 10 if b16 else 6
            N, H, W, C, K, KH, KW, stride, pad, in_dil, OH, OW = args[o:o + 12]
            xb = (2 if args[1] else 4) if b16 else 4
            yb = (2 if args[4] else 4) if b16 else 4
            res, acc = (args[7], args[23]) if b16 else (args[5], args[19])
            key = ("ig16" if "bf16" in name else "ig32", N, H, W, C, K, KH, stride, in_dil, OH, OW)
            fl = 2.0 * N * OH * OW * K * KH * KW * C / (in_dil * in_dil if in_dil > 1 else 1)
            by = xb * N * H * W * C + yb * N * OH * OW * K * (1 + bool(res.value if hasattr(res, "value") else res) + bool(acc)) + 2.0 * K * KH * KW * C
        elif name in ("utv2_conv2d_ml_fwd_bf16", "utv2_conv2d_ml_fwd"):
            b16 = "bf16" in name
            o = 8 if b16 else 6
            nlev, _, _, N, C, K, KH, KW = args[o:o + 8]
            key = ("ml16" if b16 else "ml32", N, 0, 0, C, K, KH, 1, 1, 0, 0)
            fl = 2.0 * N * 22400 * K * KH * KW * C
            xb = (2 if args[1] else 4) if b16 else 4
            yb = (2 if args[4] else 4) if b16 else 4
            by = N * 22400 * (xb * C + yb * K)
        elif name == "utv2_conv2d_wgrad_bf16":
            M, C, K, KH, KW = args[9:14]
            key = ("wg16", M, 0, 0, C, K, KH, 1, 1, 0, 0)
            fl = 2.0 * M * K * KH * KW * C
            by = M * ((2 if args[1] else 4) * C + (2 if args[3] else 4) * K)
        elif name == "utv2_conv2d_ml_wgrad":
            key = ("mlwg32", 0, 0, 0, 0, 0, 0, 1, 1, 0, 0); fl = 0; by = 0
        else:
            N, H, W, C, K, KH, KW, stride, pad, OH, OW = args[4:15]
            key = ("wgrad", N, H, W, C, K, KH, stride, 1, OH, OW)
            fl = 2.0 * N * OH * OW * K * KH * KW * C
            by = 4.0 * N * (H * W * C + OH * OW * K)
        recs.append((key, e0, e1, fl, by))
        if key not in replay:
            replay[key] = (name, args, fl, by)
    else:
        orig_call(name, *args)

hip.call = call
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-f32", "--no-rcnn", "--timed-only"]
# run bench main but toggle recording around the timed region by patching ConvTimer.enabled setter
class T(bench.ConvTimer):
    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)
        if k == "enabled": enabled[0] = bool(v)
bench.ConvTimer = T
bench.main()
torch.cuda.synchronize()
# replay every distinct launch back to back (the in-step event pairs above include the host gaps of short launches)
REP = 10
iso = {}
for key, (name, args, fl, by) in replay.items():
    orig_call(name, *args)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        orig_call(name, *args)
    e1.record()
    torch.cuda.synchronize()
    iso[key] = e0.elapsed_time(e1) / REP
agg = collections.OrderedDict()
for key, e0, e1, fl, by in recs:
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += iso[key]; a[2] += fl; a[3] += by
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values())
print("total conv ms (2 steps): %.1f" % tot)
print("%-8s %3s %4s %4s %5s %5s %2s %2s %2s %4s %4s | %5s %9s %8s %8s %6s" % ("kind","N","H","W","C","K","k","s","d","OH","OW","calls","ms","TF","GB/s","pct"))
for key, (n, ms, fl, by) in rows[:70]:
    print("%-8s %6d %4d %4d %5d %5d %2d %2d %2d %4d %4d | %5d %9.3f %8.1f %8.0f %5.1f%%" % (key + (n, ms, fl / ms / 1e9, by / ms / 1e6, 100 * ms / tot)))
