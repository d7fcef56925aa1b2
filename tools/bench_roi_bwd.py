"""RoIAlign backward (utv2_roi_align_bwd_tiled) at the Faster-RCNN step's size: N images x 512 sampled ROIs, 256 channels, 7 x 7 bins, p2-p5 of
an 800 x 1344 canvas, 16-bit gradients.  Prints the time per launch and a checksum of the maps (two builds / two runs must agree exactly).
usage: python tools/bench_roi_bwd.py [N=12]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
BF = torch.bfloat16
torch.manual_seed(0)
P, C = 512, 256
shapes = [(N, 200, 336, C), (N, 100, 168, C), (N, 50, 84, C), (N, 25, 42, C)]
scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
# proposal-like boxes: log-uniform sizes 16..600 px, anywhere in the 1333 x 800 image
g = torch.Generator().manual_seed(1)
cx = torch.rand(N * P, generator=g) * 1333; cy = torch.rand(N * P, generator=g) * 800
w = torch.exp(torch.rand(N * P, generator=g) * (6.4 - 2.8) + 2.8); h = torch.exp(torch.rand(N * P, generator=g) * (6.4 - 2.8) + 2.8)
rois = torch.stack(((cx - w / 2).clamp(0, 1333), (cy - h / 2).clamp(0, 800), (cx + w / 2).clamp(0, 1333), (cy + h / 2).clamp(0, 800)), 1).cuda().contiguous()
valid = torch.ones(N * P, dtype=torch.uint8, device="cuda")
dy = torch.randn(N * P, 7, 7, C, device="cuda").to(BF)
outs = hip.roi_align_bwd_tiled(shapes, BF, scales, 2, rois, valid, dy, P)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    hip.roi_align_bwd_tiled(shapes, BF, scales, 2, rois, valid, dy, P, outs=outs)
e1.record(); torch.cuda.synchronize()
hsh = hashlib.sha1(b"".join(o.view(torch.int16).cpu().numpy().tobytes() for o in outs)).hexdigest()[:16]
print("roi_align_bwd_tiled N=%d: %.1f us per launch, maps sha1 %s" % (N, e0.elapsed_time(e1) / 10 * 1e3, hsh))
