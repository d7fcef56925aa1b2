"""HIP implicit-GEMM conv (fwd / dgrad / wgrad) vs torch fp32 CPU conv2d on seeded inputs.
Tolerance: fp32 accumulation-order differences only -> rtol 2e-4 on the max-norm."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


CASES = [
    # N, H, W, C, K, k, stride, pad
    (2, 25, 42, 256, 256, 3, 1, 1),
    (2, 13, 21, 256, 256, 3, 2, 1),
    (1, 50, 84, 64, 64, 1, 1, 0),
    (2, 50, 84, 256, 512, 1, 2, 0),
    (2, 20, 20, 256, 80, 3, 1, 1),
    (1, 17, 23, 32, 73, 3, 1, 1),
    (1, 9, 11, 20, 12, 3, 1, 1),      # generic path (C % 16 != 0)
    (1, 12, 12, 12, 7, 1, 1, 0),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_bwd(case):
    from ubteacher import hip
    N, H, W, C, K, k, s, p = case
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, k, k, generator=g) * 0.05
    b = torch.randn(K, generator=g)
    sc = torch.rand(K, generator=g) + 0.5
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    yref = F.conv2d(xr, wr, None, s, p) * sc.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    dy = torch.randn(yref.shape, generator=g)
    yref.backward(dy)

    dev = "cuda"
    xh = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wh = w.permute(0, 2, 3, 1).contiguous().reshape(K, -1).to(dev)
    y = hip.conv2d_fwd(xh, wh, scale=sc.to(dev), bias=b.to(dev), stride=s, pad=p, kh=k, kw=k)
    torch.cuda.synchronize()
    assert _relerr(y.cpu().permute(0, 3, 1, 2), yref.detach()) < 2e-4

    # relu + residual epilogue
    res = torch.randn(yref.shape, generator=g)
    y2 = hip.conv2d_fwd(xh, wh, scale=sc.to(dev), bias=b.to(dev), residual=res.permute(0, 2, 3, 1).contiguous().to(dev),
                        stride=s, pad=p, kh=k, kw=k, relu=True)
    ref2 = torch.relu(yref.detach() + res)
    assert _relerr(y2.cpu().permute(0, 3, 1, 2), ref2) < 2e-4

    # dgrad / wgrad of the scaled conv: upstream grad is dy*scale
    gs = (dy * sc.view(1, -1, 1, 1)).permute(0, 2, 3, 1).contiguous().to(dev)
    wt = hip.weight_flip_transpose(wh, K, k, k, C)
    dx = hip.conv2d_dgrad(gs, wt, (N, H, W, C), s, p, k, k)
    assert _relerr(dx.cpu().permute(0, 3, 1, 2), xr.grad) < 2e-4
    dw = torch.zeros(K, k * k * C, device=dev)
    hip.conv2d_wgrad(xh, gs, dw, s, p, k, k, accumulate=True)
    hip.conv2d_wgrad(xh, gs, dw, s, p, k, k, accumulate=True)   # accumulates
    dwref = wr.grad.permute(0, 2, 3, 1).reshape(K, -1)
    assert _relerr(dw.cpu() / 2, dwref) < 2e-4
    db = torch.zeros(K, device=dev)
    hip.colsum(gs.reshape(-1, K), db, accumulate=False)
    assert _relerr(db.cpu(), (dy * sc.view(1, -1, 1, 1)).sum((0, 2, 3))) < 2e-4


def test_conv_stem_c4():
    from ubteacher import hip
    g = torch.Generator().manual_seed(1)
    N, H, W = 2, 64, 96
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    yref = torch.relu(F.conv2d(x, w, None, 2, 3))
    x4 = torch.zeros(N, H, W, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    w4 = torch.zeros(64, 7, 7, 4)
    w4[..., :3] = w.permute(0, 2, 3, 1)
    wp = torch.zeros(64, 208)
    wp[:, :196] = w4.reshape(64, 196)
    y = hip.conv2d_fwd(x4.cuda(), wp.cuda(), stride=2, pad=3, kh=7, kw=7, relu=True)
    assert _relerr(y.cpu().permute(0, 3, 1, 2), yref) < 2e-4
