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


def test_padded_dgrad_of_an_80_channel_head_conv():
    """A multi-level 3x3 conv with 80 output channels and an fp32 output (the FCOS prediction convs, fcos.py:283-307) under AMP: its
    backward pads the gradient to 96 bf16 channels (utv2_pad_cols_bf16) and runs the LDS-DMA kernel on a zero-padded weight image
    (Conv.dgrad_cout) - the same dx, to one bf16 rounding of the accumulation-order difference, as the generic kernel on the fp32
    gradient; the image from the batched flip launch equals the per-layer one padded; weight / bias gradients are untouched."""
    from ubteacher import hip, ops
    from ubteacher.params import ParamStore
    g = torch.Generator().manual_seed(3)
    N, C, K = 2, 256, 80
    level_hw = [(20, 24), (10, 12), (5, 6)]
    meta = ops.LevelMeta(N, level_hw)
    try:
        ops.set_precision("bf16")
        store = ParamStore()
        w = store.new((K, 9 * C), "decay", lambda t: t.normal_(0.0, 0.05))
        b = store.new((K,), "decay", lambda t: t.normal_(0.0, 0.1))
        conv = ops.Conv(w, C, K, 3, 1, 1, bias=b, out_fp32=True)
        store.finalize("cuda")
        assert conv.dgrad_cout() == 96
        x = (torch.randn(meta.P, C, generator=g) * 0.5).to(torch.bfloat16).cuda().requires_grad_(True)
        dy = torch.randn(meta.P, K, generator=g).cuda()
        dxs, imgs = [], []
        for _ in range(2):      # first backward: per-layer image (registers the layer); second: the batched launch
            x.grad = None
            store.grad.zero_()
            y = conv(x, meta=meta)
            assert y.dtype == torch.float32
            y.backward(dy)
            torch.cuda.synchronize()
            dxs.append(x.grad.clone())
            imgs.append(conv.wt16(None).clone())
            ops.bump_version()
        assert torch.equal(dxs[0], dxs[1]) and torch.equal(imgs[0], imgs[1])
        img = imgs[0].view(C, 9, 96)
        plain = hip.weight_flip_transpose_bf16(w.t, K, 3, 3, C).view(C, 9, K)
        assert torch.equal(img[:, :, :K], plain) and float(img[:, :, K:].abs().max()) == 0.0
        ref = hip.conv2d_ml_fwd_bf16(dy, plain.reshape(C, -1).contiguous(), level_hw, N, k=3, pad=1, out_dtype=torch.bfloat16)
        err = (dxs[0].float() - ref.float()).abs().max().item()
        assert err <= 2 ** -7 * ref.float().abs().max().item(), err
        assert float(w.g.abs().sum()) > 0 and float(b.g.abs().sum()) > 0
        # the bias gradient is summed next to the wgrad's dY staging, from the operands as the MFMA sees them (bf16-rounded)
        assert relerr(b.g.cpu(), dy.to(torch.bfloat16).float().sum(0).cpu()) < 1e-4
    finally:
        ops.set_precision("fp32")


@pytest.mark.parametrize("level_hw,N", [([(20, 24), (10, 12), (5, 6)], 2), ([(128, 160), (64, 80)], 2)])
def test_grouped_and_pitched_ml_conv_bf16(level_hw, N):
    """The paired FCOS towers (cls | bbox: two independent 256 -> 256 chains, fcos/fcos.py:252-304) run as ONE grouped launch per depth:
    utv2_conv2d_ml_fwd_bf16_g with groups = 2 on a [P, 512] matrix == the two 256 -> 256 convs on its column halves, BIT-identical
    (same tiles, same accumulation order) - on the 128-tile kernel (small case) and on whole rounds of the 256-tile ping-pong kernel
    + its 128-tile remainder (large case); a conv reading a column SLICE (row pitch 512) and one writing into a column slice equal the
    convs on contiguous copies; the grouped weight gradient equals the two separate ones (fp32 accumulation, different split counts)."""
    from ubteacher import hip
    from ubteacher.ops import LevelMeta
    g = torch.Generator().manual_seed(7)
    C = 256
    meta = LevelMeta(N, level_hw)
    P = meta.P
    x = (torch.randn(P, 2 * C, generator=g) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(2 * C, 9 * C, generator=g) * 0.02).to(torch.bfloat16).cuda()
    b = torch.randn(2 * C, generator=g).cuda()
    y = hip.conv2d_ml_fwd_bf16(x, w, level_hw, N, bias=b, k=3, pad=1, groups=2)
    xa, xb = x[:, :C].contiguous(), x[:, C:].contiguous()
    ya = hip.conv2d_ml_fwd_bf16(xa, w[:C].contiguous(), level_hw, N, bias=b[:C].contiguous(), k=3, pad=1)
    yb = hip.conv2d_ml_fwd_bf16(xb, w[C:].contiguous(), level_hw, N, bias=b[C:].contiguous(), k=3, pad=1)
    assert y.shape == (P, 2 * C) and torch.equal(y[:, :C], ya) and torch.equal(y[:, C:], yb)
    # against torch on the same bf16-rounded operands (one level, group 1)
    h0, w0 = level_hw[0]
    xr = meta.level_view(xb, 0).float().permute(0, 3, 1, 2).cpu()
    ref = F.conv2d(xr, w[C:].float().view(C, 3, 3, C).permute(0, 3, 1, 2).cpu(), b[C:].cpu(), 1, 1)
    got = meta.level_view(yb, 0).float().permute(0, 3, 1, 2).cpu()
    assert relerr(got, ref) < 2e-2
    # column slices: input with row pitch 512, output written into the right half of a [P, 512] buffer
    ys = hip.conv2d_ml_fwd_bf16(x[:, C:], w[C:].contiguous(), level_hw, N, bias=b[C:].contiguous(), k=3, pad=1)
    assert torch.equal(ys, yb)
    out = torch.zeros(P, 2 * C, dtype=torch.bfloat16, device="cuda")
    hip.conv2d_ml_fwd_bf16(xa, w[:C].contiguous(), level_hw, N, bias=b[:C].contiguous(), k=3, pad=1, out=out[:, C:])
    assert torch.equal(out[:, C:], ya) and float(out[:, :C].float().abs().max()) == 0.0
    # a narrow (80-channel, fp32-output) conv reading a slice: the prediction convs behind the paired towers
    w80 = (torch.randn(80, 9 * C, generator=g) * 0.02).to(torch.bfloat16).cuda()
    p_slice = hip.conv2d_ml_fwd_bf16(x[:, :C], w80, level_hw, N, k=3, pad=1, out_dtype=torch.float32)
    p_copy = hip.conv2d_ml_fwd_bf16(xa, w80, level_hw, N, k=3, pad=1, out_dtype=torch.float32)
    assert torch.equal(p_slice, p_copy)
    # weight gradients: grouped == per group; sliced x == contiguous x
    dy = (torch.randn(P, 2 * C, generator=g) * 0.1).to(torch.bfloat16).cuda()
    ri = hip.rowinfo_ml(N, level_hw, 1, 3, "cuda")
    dwg = torch.zeros(2 * C, 9 * C, device="cuda")
    hip.conv2d_wgrad_bf16(x, dy, dwg, ri, C, 3, 3, accumulate=True, groups=2)
    dwa, dwb = torch.zeros(C, 9 * C, device="cuda"), torch.zeros(C, 9 * C, device="cuda")
    hip.conv2d_wgrad_bf16(xa, dy[:, :C].contiguous(), dwa, ri, C, 3, 3, accumulate=True)
    hip.conv2d_wgrad_bf16(xb, dy[:, C:].contiguous(), dwb, ri, C, 3, 3, accumulate=True)
    assert relerr(dwg[:C].cpu(), dwa.cpu()) < 1e-5 and relerr(dwg[C:].cpu(), dwb.cpu()) < 1e-5
    dy80 = torch.randn(P, 80, generator=g).cuda()
    d1, d2 = torch.zeros(80, 9 * C, device="cuda"), torch.zeros(80, 9 * C, device="cuda")
    hip.conv2d_wgrad_bf16(x[:, C:], dy80, d1, ri, C, 3, 3, accumulate=True, x_pitch=2 * C)
    hip.conv2d_wgrad_bf16(xb, dy80, d2, ri, C, 3, 3, accumulate=True)
    assert torch.equal(d1, d2)


@pytest.mark.parametrize("groups", [1, 2])
def test_groupnorm_statistics_from_the_conv_epilogue(groups):
    """A tower conv under AMP leaves per-(32-row block, 8-channel group) sum / sum of squares of its stored bf16 output
    (utv2_conv2d_ml_fwd_bf16_g gn_part); the GroupNorm that follows (fcos/fcos.py:263-264) takes mean / rstd from them plus the few
    segment-edge rows (utv2_groupnorm_relu_seg_fwd_p32) instead of a statistics pass over the tensor: same mean / rstd to fp32
    summation order, same normalised output to one bf16 rounding - with segments that start off the 32-row grid, on the 128-tile
    kernel and on the 256-tile kernel + its remainder."""
    from ubteacher import hip
    from ubteacher.ops import LevelMeta
    g = torch.Generator().manual_seed(11)
    C = 256
    for level_hw, N in (([(20, 24), (10, 12), (5, 6), (3, 3), (2, 2)], 3), ([(128, 160), (64, 80), (7, 11)], 2)):
        meta = LevelMeta(N, level_hw)
        P, K = meta.P, 2 * C
        x = (torch.randn(P, groups * C, generator=g) * 0.5).to(torch.bfloat16).cuda()
        w = (torch.randn(K, 9 * C, generator=g) * 0.02).to(torch.bfloat16).cuda()
        b = torch.randn(K, generator=g).cuda()
        ga, be = (torch.rand(K, generator=g) + 0.5).cuda(), torch.randn(K, generator=g).cuda()
        part = hip.gn_part_buffer(P, K, "cuda")
        part.fill_(float("nan"))
        y = hip.conv2d_ml_fwd_bf16(x, w, level_hw, N, bias=b, k=3, pad=1, groups=groups, gn_part=part)
        y0 = hip.conv2d_ml_fwd_bf16(x, w, level_hw, N, bias=b, k=3, pad=1, groups=groups)
        assert torch.equal(y, y0)
        # every block that holds a row was written, and equals the sums of the stored tensor
        yf = y.float()
        nb = (P + 31) // 32
        pad = torch.zeros(nb * 32 - P, K, device="cuda")
        blk = torch.cat((yf, pad)).view(nb, 32, K // 8, 8)
        ref_s, ref_q = blk.sum(dim=(1, 3)), (blk * blk).sum(dim=(1, 3))
        assert torch.isfinite(part).all()
        assert torch.allclose(part[:, :, 0], ref_s, rtol=1e-5, atol=1e-3) and torch.allclose(part[:, :, 1], ref_q, rtol=1e-5, atol=1e-3)
        z1, m1, r1 = hip.groupnorm_relu_seg_fwd_p32(y, meta.seg_rows, ga, be, part, K // 8)
        z0, m0, r0 = hip.groupnorm_relu_seg_fwd(y, meta.seg_rows, ga, be, K // 8)
        assert torch.allclose(m1, m0, rtol=1e-5, atol=1e-6) and torch.allclose(r1, r0, rtol=1e-5)
        d = (z1.float() - z0.float()).abs()
        assert float(d.max()) <= 2 ** -7 * float(z0.float().abs().max()) and float((d > 0).float().mean()) < 1e-3


@pytest.mark.parametrize("K", [80, 72, 96])
@pytest.mark.parametrize("out_dtype", ["f32", "h16"])
def test_narrow_prediction_convs_on_the_96_wide_tile(K, out_dtype):
    """The 256 -> 80 prediction convs of the FCOS head (fcos/fcos.py:306-376: cls_logits; bbox_pred | bbox_pred_std | ctrness fused)
    run on the 256 x 96 tile of conv_igemm_bf16_v2<96> (round 4) when the level-first matrix is large enough: the same values as the
    leading K columns of a 128-channel conv with the weight zero-padded (that one runs on the 128-wide tile), with a bias, reading a column slice (row pitch 512), fp32 and 16-bit outputs; and against
    torch on the same rounded operands.  A ragged last row tile and levels that start off the tile grid are part of the geometry."""
    from ubteacher import hip
    from ubteacher.ops import LevelMeta
    g = torch.Generator().manual_seed(13)
    C, N = 256, 2
    level_hw = [(100, 84), (50, 42), (25, 21), (13, 11), (7, 6)]
    meta = LevelMeta(N, level_hw)
    P = meta.P
    assert P >= 8192 and P % 256 != 0
    h16 = hip.h16_dtype()
    od = torch.float32 if out_dtype == "f32" else h16
    x = (torch.randn(P, 2 * C, generator=g) * 0.5).to(h16).cuda()
    w = (torch.randn(K, 9 * C, generator=g) * 0.02).to(h16).cuda()
    b = torch.randn(K, generator=g).cuda()
    y = hip.conv2d_ml_fwd_bf16(x[:, C:], w, level_hw, N, bias=b, k=3, pad=1, out_dtype=od)
    assert tuple(y.shape) == (P, K) and y.dtype == od
    w128 = torch.zeros(128, 9 * C, dtype=h16, device="cuda")
    w128[:K] = w
    b128 = torch.zeros(128, device="cuda")
    b128[:K] = b
    y128 = hip.conv2d_ml_fwd_bf16(x[:, C:].contiguous(), w128, level_hw, N, bias=b128, k=3, pad=1, out_dtype=od)
    # (the 128-wide tile walks the K loop in 64-channel chunks, this one in 32-channel chunks: the same products summed in another
    # order in fp32 - equal to accumulation-order rounding, not bit for bit)
    # a 16-bit output may flip by one unit in the last place: 2^-7 of the top binade in bf16 (8 significant bits), 2^-10 in fp16 -
    # and only a small fraction of the elements may differ at all
    tol = 2e-5 if out_dtype == "f32" else (2 ** -7 if h16 == torch.bfloat16 else 2 ** -10)
    dy = (y.float() - y128[:, :K].float()).abs()
    assert float(dy.max()) <= tol * float(y128.float().abs().max())
    assert out_dtype == "f32" or float((dy > 0).float().mean()) < 0.02
    xr = meta.level_view(x[:, C:].contiguous(), 1).float().permute(0, 3, 1, 2).cpu()
    ref = F.conv2d(xr, w.float().view(K, 3, 3, C).permute(0, 3, 1, 2).cpu(), b.cpu(), 1, 1)
    got = meta.level_view(y, 1).float().permute(0, 3, 1, 2).cpu()
    assert relerr(got, ref) < (2e-4 if out_dtype == "f32" else 1e-2)
    # relu + accumulate forms of the epilogue on both column blocks
    y2 = hip.conv2d_ml_fwd_bf16(x[:, :C], w, level_hw, N, bias=b, k=3, pad=1, out_dtype=od, relu=True)
    y2r = hip.conv2d_ml_fwd_bf16(x[:, :C].contiguous(), w128, level_hw, N, bias=b128, k=3, pad=1, out_dtype=od, relu=True)
    assert float((y2.float() - y2r[:, :K].float()).abs().max()) <= tol * float(y2r.float().abs().max()) and float(y2.float().min()) >= 0.0
    # bit-deterministic
    y3 = hip.conv2d_ml_fwd_bf16(x[:, :C], w, level_hw, N, bias=b, k=3, pad=1, out_dtype=od, relu=True)
    assert torch.equal(y2, y3)
