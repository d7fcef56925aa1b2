"""Micro-benchmark of the wgrad / fwd kernels on the FCOS tower (multi-level) and backbone shapes."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

N = 8
level_hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
P = N * sum(h * w for h, w in level_hw)
C = K = 256
x = torch.randn(P, C, device="cuda"); dy = torch.randn(P, K, device="cuda")
w = torch.randn(K, 9 * C, device="cuda") * 0.05
w16 = w.to(torch.bfloat16)
dw = torch.zeros_like(w)
ri = hip.rowinfo_ml(N, level_hw, 1, 3, "cuda")
fl = 2.0 * P * K * 9 * C
t = timeit(lambda: hip.conv2d_wgrad_bf16(x, dy, dw, ri, C, 3, 3, accumulate=True)); print("tower wgrad bf16 %.3f ms %.1f TF" % (t, fl / t / 1e9))
t = timeit(lambda: hip.conv2d_ml_wgrad(x, dy, dw, level_hw, N, 3, 1, accumulate=True)); print("tower wgrad f32  %.3f ms %.1f TF" % (t, fl / t / 1e9))
y = torch.empty(P, K, device="cuda")
t = timeit(lambda: hip.conv2d_ml_fwd_bf16(x, w16, level_hw, N, k=3, pad=1, out=y)); print("tower fwd   bf16 %.3f ms %.1f TF" % (t, fl / t / 1e9))
t = timeit(lambda: hip.conv2d_ml_fwd(x, w, level_hw, N, k=3, pad=1, out=y)); print("tower fwd   f32  %.3f ms %.1f TF" % (t, fl / t / 1e9))
for (n, h, ww, c, k, ks, s, p) in [(8, 50, 84, 1024, 256, 1, 1, 0), (8, 50, 84, 256, 1024, 1, 1, 0), (8, 50, 84, 256, 256, 3, 1, 1), (8, 100, 168, 128, 128, 3, 1, 1), (8, 25, 42, 512, 512, 3, 1, 1), (8, 100, 168, 512, 128, 1, 1, 0)]:
    xx = torch.randn(n, h, ww, c, device="cuda"); wt = torch.randn(k, ks * ks * c, device="cuda") * 0.05
    yy = hip.conv2d_fwd(xx, wt, stride=s, pad=p, kh=ks, kw=ks)
    dyy = torch.randn_like(yy); dww = torch.zeros_like(wt)
    rr = hip.rowinfo_nhwc(n, h, ww, yy.shape[1], yy.shape[2], s, p, ks, ks, "cuda")
    f = 2.0 * yy.shape[0] * yy.shape[1] * yy.shape[2] * k * ks * ks * c
    t1 = timeit(lambda: hip.conv2d_wgrad_bf16(xx, dyy.view(-1, k), dww, rr, c, ks, ks, accumulate=True))
    t2 = timeit(lambda: hip.conv2d_wgrad(xx, dyy, dww, s, p, ks, ks, accumulate=True))
    t3 = timeit(lambda: hip.conv2d_fwd_bf16(xx, wt.to(torch.bfloat16), stride=s, pad=p, kh=ks, kw=ks, out=yy))
    print("%s wgrad16 %.3f ms %.0f TF | wgrad32 %.3f ms %.0f TF | fwd16 %.3f ms %.0f TF" % ((n, h, ww, c, k, ks), t1, f / t1 / 1e9, t2, f / t2 / 1e9, t3, f / t3 / 1e9))
