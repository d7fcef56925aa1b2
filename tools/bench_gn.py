"""Tower-sized segmented GroupNorm+ReLU forward / backward launches (bf16, 12 images x 5 FPN levels x 256 channels): time per call and
the HBM rate against the algorithmic bytes (fwd: read x twice, write y; bwd: read dy, x twice (y not needed when beta is given), write dx)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
level_hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
seg_rows = [h * w for h, w in level_hw for _ in range(N)]
P, C = sum(seg_rows), 256
BF = torch.bfloat16
x = torch.randn(P, C, device="cuda").to(BF)
dy = torch.randn(P, C, device="cuda").to(BF)
gamma = torch.rand(C, device="cuda") + 0.5; beta = torch.randn(C, device="cuda") * 0.1
dgamma = torch.zeros(C, device="cuda"); dbeta = torch.zeros(C, device="cuda")
y, mean, rstd = hip.groupnorm_relu_seg_fwd(x, seg_rows, gamma, beta)
tb = P * C * 2


def timeit(name, fn, nbytes, reps=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    print("%-10s %.3f ms  %.0f GB/s (algorithmic %d MB)" % (name, t, nbytes / t / 1e6, nbytes / 1e6))


timeit("gn fwd", lambda: hip.groupnorm_relu_seg_fwd(x, seg_rows, gamma, beta), 3 * tb)
timeit("gn bwd", lambda: hip.groupnorm_relu_seg_bwd(dy, y, x, seg_rows, mean, rstd, gamma, dgamma, dbeta, beta=beta), 5 * tb)
