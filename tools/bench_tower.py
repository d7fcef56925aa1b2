"""Tower-shaped (multi-level 3x3, C=K=256) bf16 fwd and wgrad launches only - target of the PMC passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
BF = torch.bfloat16
N = int(os.environ.get("TOWER_N", "8"))
level_hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
P = N * sum(h * w for h, w in level_hw)
C = K = 256
sparse = len(sys.argv) > 1 and sys.argv[1] == "relu"
x = torch.randn(P, C, device="cuda"); dy = torch.randn(P, K, device="cuda").to(BF)
if sparse:
    x = torch.relu(x)
x = x.to(BF)
w16 = (torch.randn(K, 9 * C, device="cuda") * 0.05).to(BF)
dw = torch.zeros(K, 9 * C, device="cuda")
ri = hip.rowinfo_ml(N, level_hw, 1, 3, "cuda")
y = torch.empty(P, K, device="cuda", dtype=BF)
fl = 2.0 * P * K * 9 * C
for name, fn in (("fwd", lambda: hip.conv2d_ml_fwd_bf16(x, w16, level_hw, N, k=3, pad=1, out=y)),
                 ("wgrad", lambda: hip.conv2d_wgrad_bf16(x, dy, dw, ri, C, 3, 3, accumulate=True))):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print("%s %.3f ms %.1f TF" % (name, t, fl / t / 1e9))
