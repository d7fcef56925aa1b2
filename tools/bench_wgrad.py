"""Micro-benchmark of the bf16 wgrad / fwd kernels (bf16 activations) on the FCOS tower (multi-level) and backbone shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip

BF = torch.bfloat16


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
x = torch.randn(P, C, device="cuda").to(BF); dy = torch.randn(P, K, device="cuda").to(BF)
w = torch.randn(K, 9 * C, device="cuda") * 0.05
w16 = w.to(BF)
dw = torch.zeros_like(w)
ri = hip.rowinfo_ml(N, level_hw, 1, 3, "cuda")
fl = 2.0 * P * K * 9 * C
t = timeit(lambda: hip.conv2d_wgrad_bf16(x, dy, dw, ri, C, 3, 3, accumulate=True)); print("tower wgrad %.3f ms %.1f TF" % (t, fl / t / 1e9))
y = torch.empty(P, K, device="cuda", dtype=BF)
t = timeit(lambda: hip.conv2d_ml_fwd_bf16(x, w16, level_hw, N, k=3, pad=1, out=y)); print("tower fwd   %.3f ms %.1f TF" % (t, fl / t / 1e9))
tot = [0.0, 0.0]
for (n, h, ww, c, k, ks, s, p) in [(8, 50, 84, 1024, 256, 1, 1, 0), (8, 50, 84, 256, 1024, 1, 1, 0), (8, 50, 84, 256, 256, 3, 1, 1),
                                   (8, 100, 168, 128, 128, 3, 1, 1), (8, 25, 42, 512, 512, 3, 1, 1), (8, 100, 168, 512, 128, 1, 1, 0),
                                   (8, 100, 168, 128, 512, 1, 1, 0), (8, 25, 42, 2048, 512, 1, 1, 0), (8, 25, 42, 512, 2048, 1, 1, 0),
                                   (8, 200, 336, 64, 256, 1, 1, 0), (8, 200, 336, 256, 512, 1, 2, 0), (8, 50, 84, 1024, 2048, 1, 2, 0)]:
    xx = torch.randn(n, h, ww, c, device="cuda").to(BF); wt = (torch.randn(k, ks * ks * c, device="cuda") * 0.05)
    yy = hip.conv2d_fwd_bf16(xx, wt.to(BF), stride=s, pad=p, kh=ks, kw=ks)
    dyy = torch.randn_like(yy); dww = torch.zeros_like(wt)
    rr = hip.rowinfo_nhwc(n, h, ww, yy.shape[1], yy.shape[2], s, p, ks, ks, "cuda")
    f = 2.0 * yy.shape[0] * yy.shape[1] * yy.shape[2] * k * ks * ks * c
    by = 2.0 * (xx.numel() + yy.numel())
    t1 = timeit(lambda: hip.conv2d_wgrad_bf16(xx, dyy.view(-1, k), dww, rr, c, ks, ks, accumulate=True))
    res = torch.randn_like(yy)
    t3 = timeit(lambda: hip.conv2d_fwd_bf16(xx, wt.to(BF), residual=res, relu=True, stride=s, pad=p, kh=ks, kw=ks, out=yy))
    tot[0] += t1; tot[1] += t3
    print("%-32s wgrad %.3f ms %4.0f TF %5.0f GB/s | fwd(+res) %.3f ms %4.0f TF %5.0f GB/s" % (
        (n, h, ww, c, k, ks, s), t1, f / t1 / 1e9, by / t1 / 1e6, t3, f / t3 / 1e9, (by + 2.0 * yy.numel()) / t3 / 1e6))
print("sum wgrad %.3f ms, fwd %.3f ms" % tuple(tot))
