"""Weight gradients of the 256-tile kernel (conv_wgrad_bf16_pp) saved by one process and compared bit for bit by another: two builds /
two switch settings that keep the accumulation order (round 5: the ping-pong schedule against the lock-step one; round 6: determinism
across processes) must agree exactly.  usage: check_wgrad_pp.py save|cmp FILE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
BF = torch.bfloat16
torch.manual_seed(0)
outs = {}
level_hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
for name, N, C, K, G in (("tower", 3, 256, 256, 1), ("paired", 5, 256, 512, 2), ("wide", 2, 256, 512, 1)):
    P = N * sum(h * w for h, w in level_hw)
    x = torch.relu(torch.randn(P, C * G, device="cuda")).to(BF)
    dy = torch.randn(P, K, device="cuda").to(BF)
    dw = torch.zeros(K, 9 * C, device="cuda")
    ri = hip.rowinfo_ml(N, level_hw, 1, 3, "cuda")
    hip.conv2d_wgrad_bf16(x, dy, dw, ri, C, 3, 3, accumulate=False, groups=G, x_pitch=C * G)
    outs[name] = dw.clone()
    hip.conv2d_wgrad_bf16(x, dy, dw, ri, C, 3, 3, accumulate=True, groups=G, x_pitch=C * G)
    outs[name + "_acc"] = dw.clone()
# a plain NHWC 3x3 (res4 conv2-like) with a ragged pixel count, and a deep 1x1
Nn, H, W, C, K = 6, 50, 83, 256, 256
xn = torch.relu(torch.randn(Nn, H, W, C, device="cuda")).to(BF)
dyn = torch.randn(Nn, H, W, K, device="cuda").to(BF)
dwn = torch.zeros(K, 9 * C, device="cuda")
rin = hip.rowinfo_nhwc(Nn, H, W, H, W, 1, 1, 3, 3, "cuda")
hip.conv2d_wgrad_bf16(xn, dyn.reshape(-1, K), dwn, rin, C, 3, 3, accumulate=False)
outs["nhwc3"] = dwn.clone()
torch.cuda.synchronize()
if sys.argv[1] == "save":
    torch.save({k: v.cpu() for k, v in outs.items()}, sys.argv[2])
    print("saved", {k: (tuple(v.shape), float(v.abs().max())) for k, v in outs.items()})
else:
    ref = torch.load(sys.argv[2])
    for k, v in outs.items():
        same = torch.equal(v.cpu(), ref[k])
        d = (v.cpu().float() - ref[k].float()).abs().max().item()
        print(k, "bit-identical" if same else "DIFF max %.4g (ref max %.4g)" % (d, ref[k].abs().max().item()), "nan" if torch.isnan(v).any() else "")
