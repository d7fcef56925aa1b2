"""How much of a 256 x 256 tile's time on conv_igemm_bf16_pp is outside its K loop?  The same whole-rounds-only launch (1024 row tiles x 1 or
2 column tiles) with K loops of 9, 18, 36 and 72 chunks (C = 64 .. 512 input channels): time is a + b * chunks per round; a = prologue
(geometry decode, first DMA pieces) + epilogue (LDS bounce, stores) + launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
BF = torch.bfloat16
level_hw = [(128, 256)]
N = 8                       # P = 262144 rows = 1024 row tiles of 256
P = N * 128 * 256
res = []
for K in (256, 512):
    for C in (128, 256, 512):   # Kred >= 1024 is the 256-tile kernel's launch rule
        x = torch.randn(P, C, device="cuda").to(BF)
        w16 = (torch.randn(K, 9 * C, device="cuda") * 0.05).to(BF)
        y = torch.empty(P, K, device="cuda", dtype=BF)
        fn = lambda: hip.conv2d_ml_fwd_bf16(x, w16, level_hw, N, k=3, pad=1, out=y)
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e3
        rounds = (P // 256) * (K // 256) / 256
        chunks = 9 * C // 64
        fl = 2.0 * P * K * 9 * C
        print("K %d C %d: %8.1f us  %6.1f TF  rounds %d  chunks/tile %d  -> %.2f us per round, %.3f us per chunk" % (K, C, t, fl / t / 1e6, rounds, chunks, t / rounds, t / rounds / chunks))
        res.append((K, C, t / rounds, chunks))
for K in (256, 512):
    r = [x for x in res if x[0] == K]
    (_, _, t1, c1), (_, _, t2, c2) = r[0], r[-1]
    b = (t2 - t1) / (c2 - c1)
    a = t1 - b * c1
    print("K %d: per round %.2f us fixed + %.3f us per chunk (fixed share at 36 chunks: %.1f %%)" % (K, a, b, 100 * a / (a + 36 * b)))
