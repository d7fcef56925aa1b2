"""Multi-level (level-first) 'same' conv: one launch for all FPN levels == per-level convs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


@pytest.mark.parametrize("k,C,K", [(3, 64, 80), (3, 32, 32), (1, 48, 16)])
def test_ml_conv_fwd_dgrad_wgrad(k, C, K):
    from ubteacher import hip
    from ubteacher.ops import LevelMeta
    g = torch.Generator().manual_seed(0)
    N = 3
    level_hw = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    meta = LevelMeta(N, level_hw)
    xs = [torch.randn(N, C, h, w, generator=g) for h, w in level_hw]
    wt = torch.randn(K, C, k, k, generator=g) * 0.1
    b = torch.randn(K, generator=g)
    xr = [x.clone().requires_grad_(True) for x in xs]
    wr = wt.clone().requires_grad_(True)
    ys = [F.conv2d(x, wr, b, 1, (k - 1) // 2) for x in xr]
    dys = [torch.randn(y.shape, generator=g) for y in ys]
    torch.autograd.backward(ys, dys)
    big = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, C) for x in xs]).cuda()
    w2 = wt.permute(0, 2, 3, 1).reshape(K, -1).contiguous().cuda()
    if k == 1:
        pytest.skip("1x1 on a level-first matrix is a plain GEMM (covered by test_conv_gpu)")
    y = hip.conv2d_ml_fwd(big, w2, level_hw, N, bias=b.cuda(), k=k, pad=(k - 1) // 2)
    for l, yr in enumerate(ys):
        assert relerr(meta.level_view(y, l).permute(0, 3, 1, 2).cpu(), yr.detach()) < 2e-4
    dy = torch.cat([d.permute(0, 2, 3, 1).reshape(-1, K) for d in dys]).cuda()
    wtt = hip.weight_flip_transpose(w2, K, k, k, C)
    dx = hip.conv2d_ml_dgrad(dy, wtt, level_hw, N, k, (k - 1) // 2)
    for l, x in enumerate(xr):
        assert relerr(meta.level_view(dx, l).permute(0, 3, 1, 2).cpu(), x.grad) < 2e-4
    dw = torch.zeros_like(w2)
    hip.conv2d_ml_wgrad(big, dy, dw, level_hw, N, k, (k - 1) // 2, accumulate=True)
    assert relerr(dw.cpu(), wr.grad.permute(0, 2, 3, 1).reshape(K, -1)) < 2e-4
