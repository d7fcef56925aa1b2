"""Would a narrow-K (256 x 128) tile on the ping-pong skeleton beat the 128 x 128 kernel on the res3 / res2 3x3 layers?  Measured bound
without writing it: the same layer with its output channels zero-padded to 256 runs on conv_igemm_bf16_pp today; a 256 x 128 tile keeps
that kernel's LOAD slots (A fragments, barriers, DMA issue) and halves its MFMAs, so it cannot be faster than the padded run's time
minus half of its MFMA time - printed next to what the layer takes on conv_igemm_bf16_v2<128, false, 64> now."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
BF = torch.bfloat16


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (N, H, W, C, K) in ((12, 100, 168, 128, 128), (12, 200, 336, 64, 64), (4, 100, 168, 128, 128)):
    x = torch.randn(N, H, W, C, device="cuda").to(BF)
    w = (torch.randn(K, 9 * C, device="cuda") * 0.05).to(BF)
    wpad = torch.zeros(256, 9 * C, device="cuda", dtype=BF); wpad[:K] = w
    fl = 2.0 * N * H * W * K * 9 * C
    t_now = timeit(lambda: hip.conv2d_fwd_bf16(x, w, pad=1, kh=3, kw=3, relu=True))
    if 9 * C >= 1024:
        t_pad = timeit(lambda: hip.conv2d_fwd_bf16(x, wpad, pad=1, kh=3, kw=3, relu=True))
        chunks = 9 * C // 64
        tiles = (N * H * W + 255) // 256
        rounds = -(-tiles // 256)
        mfma_us = chunks * 64 * 32.4 / 1.9e3     # 64 MFMAs of a SIMD per chunk at 32.4 cycles each, ~1.9 GHz: the padded run's MFMA time per tile
        bound = t_pad - rounds * mfma_us / 2
        print("N %d %dx%d C %d K %d: now %.1f us (%.0f TF); padded to K 256 on the 256-tile kernel %.1f us; a 256 x 128 tile >= %.1f us (%.0f TF)"
              % (N, H, W, C, K, t_now, fl / t_now / 1e6, t_pad, bound, fl / bound / 1e6))
    else:
        print("N %d %dx%d C %d K %d: now %.1f us (%.0f TF); Kred %d < 1024: the 256-tile kernel's K loop would be %d chunks against ~12 us of fixed cost per tile"
              % (N, H, W, C, K, t_now, fl / t_now / 1e6, 9 * C, 9 * C // 64))
