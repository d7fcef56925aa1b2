"""Tower-shaped bf16 wgrad: time + checksum of the result (A/B of kernel variants selected by environment variables)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
BF = torch.bfloat16
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
level_hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
P = N * sum(h * w for h, w in level_hw)
C = K = 256
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.relu(torch.randn(P, C, device="cuda", generator=g)).to(BF)
dy = torch.randn(P, K, device="cuda", generator=g).to(BF)
dw = torch.zeros(K, 9 * C, device="cuda")
db = torch.zeros(K, device="cuda")
ri = hip.rowinfo_ml(N, level_hw, 1, 3, "cuda")
fl = 2.0 * P * K * 9 * C
for name, kw in (("wgrad", {}), ("wgrad+bias", {"db": db})):
    fn = lambda: hip.conv2d_wgrad_bf16(x, dy, dw, ri, C, 3, 3, accumulate=False, **kw)
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print("N=%d %s %.3f ms %.1f TF  checksum %.6e %.6e" % (N, name, t, fl / t / 1e9, float(dw.double().sum()), float(dw.double().abs().sum())))
