"""What would a split-bf16 conv mode (x = hi + lo, w = hi + lo in bf16; y = hi*hi + hi*lo + lo*hi on the bf16 MFMA kernels, fp32 accumulate:
products exact to ~2^-16 relative, no TF32 on this chip) buy the Faster-RCNN trainer at the precision of its own YAML (fp32: 50 img/s,
0.62-0.74 of the fp32-MFMA ceiling)?  Kernel-level evaluation on representative layers: time of the exact-f32 conv (v_mfma_f32_32x32x2_f32)
against three accumulating launches of the 16-bit kernel with fp32 output (+ the split passes a fused kernel would do in one), and the
error of each against an fp64 reference.  usage: python tools/eval_split_bf16.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
import torch.nn.functional as F
from ubteacher import hip, ops
ops.set_precision("bf16")
BF = torch.bfloat16
torch.manual_seed(0)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def split(t):
    hi = t.to(BF)
    lo = (t - hi.float()).to(BF)
    return hi, lo


print("%-44s %9s %9s %9s %9s | %9s %9s %9s" % ("layer (12 images)", "f32 us", "3x16 us", "+split us", "speedup", "err f32", "err bf16", "err split"))
for name, (N, H, W, C, K, k) in (("res3 conv2 3x3 128->128 @100x168", (12, 100, 168, 128, 128, 3)), ("res4 conv2 3x3 256->256 @50x84", (12, 50, 84, 256, 256, 3)),
                                 ("FPN output 3x3 256->256 @200x336", (12, 200, 336, 256, 256, 3)), ("res3 conv3 1x1 128->512 @100x168", (12, 100, 168, 128, 512, 1)),
                                 ("res4 conv1 1x1 1024->256 @50x84", (12, 50, 84, 1024, 256, 1))):
    x = torch.relu(torch.randn(N, H, W, C, device="cuda"))
    w = torch.randn(K, k * k * C, device="cuda") * (1.0 / (k * k * C) ** 0.5)
    pad = k // 2
    y32 = torch.empty(N, H, W, K, device="cuda")
    t32 = timeit(lambda: hip.conv2d_fwd(x, w, pad=pad, kh=k, kw=k, out=y32))
    xh, xl = split(x)
    wh, wl = split(w)
    ys = torch.empty(N, H, W, K, device="cuda")

    def three():
        hip.conv2d_fwd_bf16(xh, wh, pad=pad, kh=k, kw=k, out=ys, out_dtype=torch.float32)
        hip.conv2d_fwd_bf16(xh, wl, pad=pad, kh=k, kw=k, out=ys, out_dtype=torch.float32, accumulate=True)
        hip.conv2d_fwd_bf16(xl, wh, pad=pad, kh=k, kw=k, out=ys, out_dtype=torch.float32, accumulate=True)
    t3 = timeit(three)
    tsp = timeit(lambda: split(x))       # three torch passes; one fused pass would read 4 and write 4 bytes per element
    # errors against fp64 on the first image
    x1 = x[:1].double().permute(0, 3, 1, 2)
    wd = w.double().view(K, k, k, C).permute(0, 3, 1, 2)
    ref = F.conv2d(x1, wd, padding=pad).permute(0, 2, 3, 1)
    sc = float(ref.abs().max())
    hip.conv2d_fwd(x, w, pad=pad, kh=k, kw=k, out=y32)
    three()
    yb = hip.conv2d_fwd_bf16(xh, wh, pad=pad, kh=k, kw=k, out_dtype=torch.float32)
    e32 = float((y32[:1].double() - ref).abs().max()) / sc
    eb = float((yb[:1].double() - ref).abs().max()) / sc
    es = float((ys[:1].double() - ref).abs().max()) / sc
    print("%-44s %9.1f %9.1f %9.1f %8.2fx | %9.2e %9.2e %9.2e" % (name, t32, t3, t3 + tsp, t32 / (t3 + tsp), e32, eb, es))
