"""fused frozen identity bottleneck (csrc/bottleneck.hip) against the three conv launches it replaces, at the benchmark's res2 shapes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
if len(sys.argv) > 1:
    hip.set_h16(sys.argv[1])
h16 = hip.h16_dtype()
C, MID = 256, 64
g = torch.Generator().manual_seed(0)
w1 = (torch.randn(MID, C, generator=g) * 0.06).to(h16).cuda(); w2 = (torch.randn(MID, 9 * MID, generator=g) * 0.05).to(h16).cuda()
w3 = (torch.randn(C, MID, generator=g) * 0.1).to(h16).cuda()
s1, s2 = torch.ones(MID).cuda(), torch.ones(MID).cuda(); b1, b2 = torch.zeros(MID).cuda(), torch.zeros(MID).cuda()
s3, b3 = torch.ones(C).cuda(), torch.zeros(C).cuda()


def timeit(fn, rep=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


wsc = (torch.randn(C, 64, generator=g) * 0.1).to(h16).cuda(); w1s = (torch.randn(MID, 64, generator=g) * 0.1).to(h16).cuda()
for N in (12, 4):
    for shortcut in (False, True):
        cin = 64 if shortcut else C
        x = (torch.randn(N, 200, 336, cin, generator=g).clamp(min=0) * 0.7).to(h16).cuda()
        y = torch.empty((N, 200, 336, C), dtype=h16, device="cuda")
        wa = w1s if shortcut else w1
        kw = dict(wsc=wsc, ssc=s3, bsc=b3) if shortcut else {}
        def fused():
            hip.bottleneck_fwd_bf16(x, wa, w2, w3, s1, b1, s2, b2, s3, b3, out=y, **kw)
        def chain():
            c1 = hip.conv2d_fwd_bf16(x, wa, scale=s1, bias=b1, relu=True)
            c2 = hip.conv2d_fwd_bf16(c1, w2, scale=s2, bias=b2, relu=True, kh=3, kw=3, pad=1)
            r = hip.conv2d_fwd_bf16(x, wsc, scale=s3, bias=b3) if shortcut else x
            hip.conv2d_fwd_bf16(c2, w3, scale=s3, bias=b3, relu=True, residual=r, out=y)
        tf, tc = timeit(fused), timeit(chain)
        px = N * 200 * 336
        fl = 2.0 * px * (cin * MID + 9 * MID * MID + MID * C + (cin * C if shortcut else 0))
        by = 2.0 * px * (cin + C)
        print("N=%d shortcut=%d  fused %.1f us (%.0f TF/s, %.2f TB/s on x + y)   conv chain %.1f us   ratio %.2f" % (
            N, shortcut, tf, fl / tf / 1e6, by / tf / 1e6, tc, tc / tf))
