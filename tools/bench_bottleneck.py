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


for N in (12, 4):
    x = (torch.randn(N, 200, 336, C, generator=g).clamp(min=0) * 0.7).to(h16).cuda()
    y = torch.empty_like(x)
    def fused():
        hip.bottleneck_identity_fwd_bf16(x, w1, w2, w3, s1, b1, s2, b2, s3, b3, out=y)
    def chain():
        c1 = hip.conv2d_fwd_bf16(x, w1, scale=s1, bias=b1, relu=True)
        c2 = hip.conv2d_fwd_bf16(c1, w2, scale=s2, bias=b2, relu=True, kh=3, kw=3, pad=1)
        hip.conv2d_fwd_bf16(c2, w3, scale=s3, bias=b3, relu=True, residual=x, out=y)
    tf, tc = timeit(fused), timeit(chain)
    px = N * 200 * 336
    fl = 2.0 * px * (C * MID + 9 * MID * MID + MID * C)
    by = 2.0 * px * C * 2
    print("N=%d  fused %.1f us (%.0f TF/s, %.2f TB/s on x + y)   three convs %.1f us   ratio %.2f" % (N, tf, fl / tf / 1e6, by / tf / 1e6, tc, tc / tf))
