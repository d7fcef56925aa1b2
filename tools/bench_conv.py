"""Micro-benchmark of the implicit-GEMM conv kernels on the dominant UTv2 shapes."""
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

SHAPES = [
    ("tower3x3_p3", 8, 100, 168, 256, 256, 3, 1, 1),
    ("tower3x3_p4", 8, 50, 84, 256, 256, 3, 1, 1),
    ("res2_3x3", 8, 200, 336, 64, 64, 3, 1, 1),
    ("res2_1x1b", 8, 200, 336, 64, 256, 1, 1, 0),
    ("res3_3x3", 8, 100, 168, 128, 128, 3, 1, 1),
    ("res4_3x3", 8, 50, 84, 256, 256, 3, 1, 1),
    ("res5_3x3", 8, 25, 42, 512, 512, 3, 1, 1),
    ("res4_1x1", 8, 50, 84, 1024, 256, 1, 1, 0),
    ("res5_1x1s2", 8, 50, 84, 1024, 512, 1, 2, 0),
    ("cls_logits", 8, 100, 168, 256, 80, 3, 1, 1),
]
out = []
for name, N, H, W, C, K, k, s, p in SHAPES:
    x = torch.randn(N, H, W, C, device="cuda")
    w = torch.randn(K, k * k * C, device="cuda") * 0.05
    y = hip.conv2d_fwd(x, w, stride=s, pad=p, kh=k, kw=k)
    OH, OW = y.shape[1:3]
    flops = 2.0 * N * OH * OW * K * k * k * C
    t_f = timeit(lambda: hip.conv2d_fwd(x, w, stride=s, pad=p, kh=k, kw=k, out=y))
    dy = torch.randn_like(y)
    wt = hip.weight_flip_transpose(w, K, k, k, C)
    dx = torch.empty_like(x)
    t_d = timeit(lambda: hip.conv2d_dgrad(dy, wt, tuple(x.shape), s, p, k, k, out=dx))
    dw = torch.zeros_like(w)
    t_w = timeit(lambda: hip.conv2d_wgrad(x, dy, dw, s, p, k, k, accumulate=True))
    r = dict(name=name, fwd_ms=t_f, fwd_tf=flops / t_f / 1e9, dgrad_ms=t_d, dgrad_tf=flops / t_d / 1e9,
             wgrad_ms=t_w, wgrad_tf=flops / t_w / 1e9)
    print(json.dumps(r)); out.append(r)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_conv.json"), "w"), indent=1)
