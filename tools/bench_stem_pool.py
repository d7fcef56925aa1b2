"""fused frozen stem + max pool (csrc/stem_pool.hip) against the two launches it replaces, at the benchmark size"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
if len(sys.argv) > 1:
    hip.set_h16(sys.argv[1])
h16 = hip.h16_dtype()
g = torch.Generator().manual_seed(0)
w208 = torch.zeros(64, 208); w208[:, :196] = torch.randn(64, 196, generator=g) * 0.05
w16s = hip.stem_weight_image(w208.cuda())
sc, sh = torch.ones(64).cuda(), torch.zeros(64).cuda()


def timeit(fn, rep=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


for N in (12, 4):
    imgs = [(torch.rand(3, 800, 1333, generator=g) * 255).cuda() for _ in range(N)]
    x4, _ = hip.preprocess_images(imgs, [103.53, 116.28, 123.675], [57.0, 57.0, 58.0], 32, bf16_stem=True)
    tf = timeit(lambda: hip.stem_pool_fwd_bf16(x4, w16s, sc, sh))
    tc = timeit(lambda: hip.maxpool3x3s2(hip.conv2d_stem_fwd_bf16(x4, w16s, sc, sh, True, h16)))
    fl = 2.0 * N * 400 * 672 * 64 * 147
    print("N=%d  fused %.1f us (%.0f TF/s)   stem conv + max pool %.1f us   ratio %.2f" % (N, tf, fl / tf / 1e6, tc, tc / tf))
