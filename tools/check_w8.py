"""A/B of the 256x256 8-wave conv kernel (UTV2_W8=1) against the 128x128 kernel (UTV2_W8=0): same fp32 accumulation order, so the
outputs must be BIT-identical.  usage: check_w8.py save|cmp FILE"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch
from ubteacher import hip
BF = torch.bfloat16
torch.manual_seed(0)
outs = {}
# multi-level tower conv (3 images), with bias+relu epilogue
N = 3
level_hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
P = N * sum(h * w for h, w in level_hw)
x = torch.randn(P, 256, device="cuda").to(BF)
w16 = (torch.randn(256, 9 * 256, device="cuda") * 0.05).to(BF)
y = torch.empty(P, 256, device="cuda", dtype=BF)
hip.conv2d_ml_fwd_bf16(x, w16, level_hw, N, k=3, pad=1, out=y)
outs["ml"] = y.clone()
# plain conv, K = 512, C = 512 3x3 (res5-like), stride 1, mask + residual epilogue, fp32 output too
xn = torch.randn(8, 100, 100, 512, device="cuda").to(BF)
wn = (torch.randn(512, 9 * 512, device="cuda") * 0.03).to(BF)
sc = torch.rand(512, device="cuda") + 0.5; bi = torch.randn(512, device="cuda")
res = torch.randn(8, 100, 100, 512, device="cuda").to(BF)
msk = torch.randn(8, 100, 100, 512, device="cuda").to(BF)
outs["n1"] = hip.conv2d_fwd_bf16(xn, wn, scale=sc, bias=bi, residual=res, stride=1, pad=1, relu=True, kh=3, kw=3, mask=msk)
outs["n2"] = hip.conv2d_fwd_bf16(xn, wn, stride=1, pad=1, kh=3, kw=3, out_dtype=torch.float32)
# 1x1 wide (C = 1024 -> K = 256, Kred = 1024)
x1 = torch.randn(4, 128, 160, 1024, device="cuda").to(BF)
w1 = (torch.randn(256, 1024, device="cuda") * 0.03).to(BF)
outs["p1"] = hip.conv2d_fwd_bf16(x1, w1)
# the epilogue's plain path with everything it carries: bias, GroupNorm partial sums (level-first tower conv) / scale + bias + ReLU and the
# ReLU bit plane (NHWC conv)
bias = torch.randn(256, device="cuda")
part = hip.gn_part_buffer(P, 256, x.device)
outs["ml_gn"] = hip.conv2d_ml_fwd_bf16(x, w16, level_hw, N, bias=bias, k=3, pad=1, gn_part=part)
outs["ml_gn_part"] = part.clone()
# the paired FCOS towers: grouped (2 x 256 -> 2 x 256) level-first conv with GroupNorm partials, 12 images (a persistent-grid launch)
N2 = 12
P2 = N2 * sum(h * w for h, w in level_hw)
x2 = torch.relu(torch.randn(P2, 512, device="cuda")).to(BF)
w2 = (torch.randn(512, 9 * 256, device="cuda") * 0.05).to(BF)
part2 = hip.gn_part_buffer(P2, 512, x2.device)
outs["ml_g2"] = hip.conv2d_ml_fwd_bf16(x2, w2, level_hw, N2, bias=torch.randn(512, device="cuda"), k=3, pad=1, groups=2, gn_part=part2)
outs["ml_g2_part"] = part2.clone()
bits = hip.relu_bits_buffer((8, 100, 100, 512), xn.device)
outs["nb"] = hip.conv2d_fwd_bf16(xn, wn, scale=sc, bias=bi, stride=1, pad=1, relu=True, kh=3, kw=3, relu_bits=bits)
outs["nb_bits"] = bits.clone()
# the lean epilogue forms with operands: residual + ReLU (+ bit plane) - a bottleneck's conv3; mask plane - its conv3 / conv2 dgrads;
# residual + post-mask plane - its conv1 dgrad (3x3 here so that the 256-tile kernels take it too)
def pack_bits(t):
    b = (t.float() > 0).reshape(t.shape[:-1] + (t.shape[-1] // 8, 8)).to(torch.int32)
    return (b << torch.arange(8, device=t.device, dtype=torch.int32)).sum(-1).to(torch.uint8)
bits1 = hip.relu_bits_buffer((8, 100, 100, 512), xn.device)
outs["m1"] = hip.conv2d_fwd_bf16(xn, wn, scale=sc, bias=bi, residual=res, stride=1, pad=1, relu=True, kh=3, kw=3, relu_bits=bits1)
outs["m1_bits"] = bits1.clone()
mb, pb = pack_bits(msk), pack_bits(res)
outs["m2"] = hip.conv2d_dgrad_bf16(xn, wn, (8, 100, 100, 512), 1, 1, 3, 3, mask_bits=mb)
outs["m5"] = hip.conv2d_dgrad_bf16(xn, wn, (8, 100, 100, 512), 1, 1, 3, 3, residual=res, post_mask_bits=pb)
torch.cuda.synchronize()
if sys.argv[1] == "save":
    torch.save({k: v.cpu() for k, v in outs.items()}, sys.argv[2])
    print("saved", {k: tuple(v.shape) for k, v in outs.items()})
elif sys.argv[1] == "cmpclose":
    # two kernels that sum the same products in ANOTHER order (the row-span kernel on v_mfma_f32_16x16x32 against the 32x32x16 kernels): the
    # fp32 sums differ in their last bits, so a 16-bit output may land on the neighbouring value - never further - on a small fraction of
    # the elements; fp32 outputs agree to 1e-5 of the tensor's range; bit planes / GroupNorm partial sums follow their tensors
    ref = torch.load(sys.argv[2])
    for k, v in outs.items():
        a, b = v.cpu(), ref[k]
        if torch.equal(a, b):
            print(k, "bit-identical")
            continue
        if a.dtype == torch.uint8:       # ReLU bit planes: a sign can only flip where the value rounds to +-0 neighbours
            frac = float((a != b).float().mean())
            print(k, "close" if frac < 1e-3 else "DIFF", "bytes that differ %.2e" % frac)
            continue
        af, bf = a.float(), b.float()
        d = (af - bf).abs()
        if a.dtype == torch.float32:
            lim = 1e-5 * float(bf.abs().max()) if "part" not in k else 2e-3 * float(bf.abs().max())
            ok = float(d.max()) <= lim
            print(k, "close" if ok else "DIFF", "max abs dev %.3g (limit %.3g)" % (float(d.max()), lim))
        else:
            ulp = torch.maximum(af.abs(), bf.abs()) * (2.0 ** -7 if a.dtype == torch.bfloat16 else 2.0 ** -10) + 1e-30
            worst = float((d / ulp).max())
            frac = float((d > 0).float().mean())
            ok = worst <= 1.01 and frac < 0.05 and not torch.isnan(af).any()
            print(k, "close" if ok else "DIFF", "elements that differ %.2e, worst %.2f ulp16" % (frac, worst))
else:
    ref = torch.load(sys.argv[2])
    for k, v in outs.items():
        same = torch.equal(v.cpu(), ref[k])
        d = (v.cpu().float() - ref[k].float()).abs().max().item()
        print(k, "bit-identical" if same else "DIFF max %.4g" % d, "nan" if torch.isnan(v).any() else "")
