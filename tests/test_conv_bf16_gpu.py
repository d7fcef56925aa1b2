"""bf16-MFMA (AMP) conv kernels: identical to an fp32 conv on bf16-rounded operands up to accumulation
order (rtol 2e-4), i.e. the ONLY precision change is the declared operand rounding."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def relerr(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)


def r16(t):
    return t.to(torch.bfloat16).to(torch.float32)


BF = torch.bfloat16


def close16(y, ref):
    """y (bf16 tensor) == RNE(ref') for some ref' within accumulation-order noise of ref"""
    y = y.float()
    tol = 2.0 ** -8 * ref.abs() + 2e-4 * ref.abs().max()
    return bool(((y - ref).abs() <= tol).all())


@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (BF, BF), (torch.float32, BF), (BF, torch.float32)])
@pytest.mark.parametrize("case", [(2, 25, 42, 256, 256, 3, 1, 1), (2, 13, 21, 256, 256, 3, 2, 1), (2, 50, 84, 64, 256, 1, 1, 0),
                                  (2, 50, 84, 256, 512, 1, 2, 0), (1, 20, 20, 128, 80, 3, 1, 1), (1, 9, 9, 32, 40, 3, 1, 1), (2, 12, 10, 80, 256, 3, 1, 1), (1, 7, 9, 24, 64, 1, 1, 0)])
def test_conv_bf16_fwd_dgrad(case, xdt, ydt):
    from ubteacher import hip
    N, H, W, C, K, k, s, p = case
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, k, k, generator=g) * 0.05
    b = torch.randn(K, generator=g)
    yref = F.conv2d(r16(x), r16(w), b, s, p)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda().to(xdt)
    w2 = w.permute(0, 2, 3, 1).reshape(K, -1).contiguous().cuda()
    w16 = torch.empty_like(w2, dtype=torch.bfloat16)
    hip.f32_to_bf16(w2, w16)
    assert torch.equal(w16.cpu(), w2.cpu().to(torch.bfloat16))       # RNE conversion == torch's
    y = hip.conv2d_fwd_bf16(xh, w16, bias=b.cuda(), stride=s, pad=p, kh=k, kw=k, out_dtype=ydt)
    assert y.dtype == ydt
    yc = y.cpu().permute(0, 3, 1, 2)
    assert relerr(yc, yref) < 2e-4 if ydt == torch.float32 else close16(yc, yref)
    # fused epilogue: residual (y's type) + ReLU, one rounding at the store
    res = torch.randn(yref.shape, generator=g)
    resh = res.permute(0, 2, 3, 1).contiguous().cuda().to(ydt)
    y2 = hip.conv2d_fwd_bf16(xh, w16, bias=b.cuda(), residual=resh, stride=s, pad=p, kh=k, kw=k, relu=True, out_dtype=ydt)
    ref2 = torch.relu(yref + resh.float().cpu().permute(0, 3, 1, 2))
    y2c = y2.cpu().permute(0, 3, 1, 2)
    assert relerr(y2c, ref2) < 2e-4 if ydt == torch.float32 else close16(y2c, ref2)
    if K % 8 == 0:
        dy = torch.randn(yref.shape, generator=g)
        xr = x.clone().requires_grad_(True)
        F.conv2d(xr, r16(w), None, s, p).backward(r16(dy))
        wt16 = hip.weight_flip_transpose_bf16(w2, K, k, k, C)
        dx = hip.conv2d_dgrad_bf16(dy.permute(0, 2, 3, 1).contiguous().cuda().to(xdt), wt16, (N, H, W, C), s, p, k, k, out_dtype=ydt)
        dxc = dx.cpu().permute(0, 3, 1, 2)
        assert relerr(dxc, xr.grad) < 2e-4 if ydt == torch.float32 else close16(dxc, xr.grad)


@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (BF, BF), (BF, torch.float32)])
def test_conv_ml_bf16(xdt, ydt):
    from ubteacher import hip
    from ubteacher.ops import LevelMeta
    g = torch.Generator().manual_seed(1)
    N, C, K, k = 2, 64, 80, 3
    level_hw = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    meta = LevelMeta(N, level_hw)
    xs = [torch.randn(N, C, h, w, generator=g) for h, w in level_hw]
    wt = torch.randn(K, C, k, k, generator=g) * 0.1
    big = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, C) for x in xs]).cuda().to(xdt)
    w16 = wt.permute(0, 2, 3, 1).reshape(K, -1).contiguous().cuda().to(torch.bfloat16)
    y = hip.conv2d_ml_fwd_bf16(big, w16, level_hw, N, k=k, pad=1, out_dtype=ydt)
    for l, x in enumerate(xs):
        ref = F.conv2d(r16(x), r16(wt), None, 1, 1)
        got = meta.level_view(y, l).permute(0, 3, 1, 2).cpu()
        assert relerr(got, ref) < 2e-4 if ydt == torch.float32 else close16(got, ref)


def _to_oracle_pseudo(pb):
    out = []
    for i in range(pb.n):
        m = pb["valid"][i].bool()
        out.append(dict(boxes=pb["boxes"][i][m].cpu(), classes=pb["classes"][i][m].long().cpu(), scores=pb["scores"][i][m].cpu(),
                        centerness=pb["centerness"][i][m].cpu(), cls_confid=pb["cls_confid"][i][m].cpu(),
                        reg_pred_std=pb["reg_pred_std"][i][m].cpu()))
    return out


@pytest.mark.parametrize("kind,tol", [("bf16", 1e-2), ("fp16", 2e-3)])
def test_fcos_step_bf16_vs_rounding_oracle(kind, tol, monkeypatch):
    """Full UTv2 FCOS step in AMP mode (16-bit MFMA operands) vs the oracle with the same operand rounding
    emulated in its convs.  bf16: the two agree to ~1e-3 on activations (measured: 7e-4 mean, 4e-3 max on the
    logits: a 1e-6 accumulation-order difference that lands on a bf16 rounding boundary becomes a 4e-3 one),
    so: teacher detections must overlap (IoU-matched) and, given the SAME pseudo labels, every loss must
    be within 1e-2 relative; EMA stays bit exact.  fp16 (UTV2_PRECISION=fp16: the second build of the kernel library, the reference's
    own autocast element type, 3 more mantissa bits; loss scale 65536 on the backward): 2e-3."""
    from oracle import utv2_oracle as O
    from tests.utv2_testutil import FixedLoader, cpu_state, make_batch, small_fcos_cfg, tune_state_for_pseudo_labels
    from ubteacher.engine import UBTeacherTrainer
    from ubteacher import ops
    cfg = small_fcos_cfg()
    cfg.SOLVER.AMP.ENABLED = True
    monkeypatch.setenv("UTV2_PRECISION", kind)
    torch.manual_seed(0)
    prod, orac = make_batch(12, 2, 2, 96, 128, "cuda")
    try:
        O.CONV_ROUND[0] = kind
        tr = UBTeacherTrainer(cfg, data_loader=FixedLoader(prod))
        assert ops.PRECISION[0] == kind and (tr._amp_state is not None) == (kind == "fp16")
        sd_s = tune_state_for_pseudo_labels(cpu_state(tr.model), [d["image"] for d in orac[3]])
        sd_t = dict(sd_s)
        sd_t["proposal_generator.fcos_head.bbox_pred_std.bias"] = torch.full((4,), -3.0)
        tr.model.load_state_dict(sd_s)
        tr.model_teacher.load_state_dict(sd_t)
        tr.iter = 1
        tr.optimizer.param_groups[0]["lr"] = 0.01
        tr.run_step_full_semisup()
        rec = tr.flush_metrics()
        pc, pr = tr._last_pseudo
        override = (_to_oracle_pseudo(pc), _to_oracle_pseudo(pr))
        rec_o, _, new_t, _, _, _ = O.fcos_semisup_step(
            O.FCOSCfg(), sd_s, sd_t, orac, keep_rate=cfg.SEMISUPNET.EMA_KEEP_RATE, lam_u=cfg.SEMISUPNET.UNSUP_LOSS_WEIGHT,
            lam_r=cfg.SEMISUPNET.UNSUP_REG_LOSS_WEIGHT, lr=0.01, mean=sd_s["pixel_mean"], pix_std=sd_s["pixel_std"],
            pseudo_override=override)
        _, _, _, _, _, own = O.fcos_semisup_step(
            O.FCOSCfg(), sd_s, sd_t, orac, keep_rate=cfg.SEMISUPNET.EMA_KEEP_RATE, lr=0.01, mean=sd_s["pixel_mean"], pix_std=sd_s["pixel_std"])
    finally:
        O.CONV_ROUND[0] = None
        ops.set_precision("fp32")
    # the oracle's own teacher detections and the product's mostly coincide
    tot, hit = 0, 0
    for i, p in enumerate(own[0]):
        mine = override[0][i]["boxes"]
        tot += max(len(p["boxes"]), len(mine))
        if len(p["boxes"]) and len(mine):
            hit += int((O.pairwise_iou(p["boxes"], mine).max(dim=1)[0] > 0.9).sum())
    assert tot > 0 and hit >= 0.6 * tot, (hit, tot)
    for k, v in rec_o.items():
        assert abs(rec[k] - v) <= tol * max(abs(v), 1e-6), (k, rec[k], v)
    t_after = cpu_state(tr.model_teacher)
    for k in new_t:
        assert torch.equal(t_after[k], new_t[k]), k
    if kind == "fp16":
        st = tr._amp_state.cpu().tolist()
        assert st == [65536.0, 0.0, 1.0], st       # finite gradients: the step was applied, one clean step counted, flag cleared
        torch.cuda.synchronize()


def test_amp_loss_scaler_matches_gradscaler_semantics():
    """utv2_amp_found_inf / utv2_sgd_momentum_amp / utv2_amp_update_scale against torch.cuda.amp.GradScaler's rules (the reference:
    engine/trainer.py:207,424-426): a non-finite gradient skips the whole step and halves the scale; `growth_interval` clean steps
    double it; the applied step equals plain SGD on grad / scale."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(1)
    n = 10007
    p0 = torch.randn(n, generator=g).cuda()
    m0 = torch.randn(n, generator=g).cuda() * 0.1
    gr = torch.randn(n, generator=g).cuda()
    st = torch.tensor([1024.0, 0.0, 0.0], device="cuda")
    # clean step == sgd on the unscaled gradient
    p, m = p0.clone(), m0.clone()
    hip.amp_found_inf(gr * 1024.0, st)
    hip.sgd_momentum_amp(p, gr * 1024.0, m, 0.01, 0.9, 1e-4, 0.5, st)
    hip.amp_update_scale(st, 2.0, 0.5, 3)
    pr, mr, gz = p0.clone(), m0.clone(), gr.clone()
    hip.sgd_momentum(pr, gz, mr, 0.01, 0.9, 1e-4, 0.5, zero_grad=False)
    assert torch.equal(p, pr) and torch.equal(m, mr) and st.cpu().tolist() == [1024.0, 0.0, 1.0]
    # two more clean steps: the third one grows the scale
    for want in ([1024.0, 0.0, 2.0], [2048.0, 0.0, 0.0]):
        hip.amp_found_inf(gr, st)
        hip.amp_update_scale(st, 2.0, 0.5, 3)
        assert st.cpu().tolist() == want
    # an inf (or a NaN) anywhere - including the tail past the last full quad - skips the step and backs the scale off
    for pos, val in ((5, float("inf")), (n - 1, float("nan")), (n - 3, float("-inf"))):
        bad = gr.clone()
        bad[pos] = val
        p, m = p0.clone(), m0.clone()
        hip.amp_found_inf(bad, st)
        assert float(st[1]) == 1.0
        hip.sgd_momentum_amp(p, bad, m, 0.01, 0.9, 1e-4, 1.0, st)
        before = float(st[0])
        hip.amp_update_scale(st, 2.0, 0.5, 3)
        assert torch.equal(p, p0) and torch.equal(m, m0) and st.cpu().tolist() == [before * 0.5, 0.0, 0.0]


def test_fp16_library_conv_matches_fp32_conv_on_rounded_operands():
    """libutv2_hip_f16.so (the same sources with h16_t = _Float16): forward, dgrad-as-conv and wgrad of a 3x3 256 -> 256 conv equal an
    fp32 conv on the fp16-rounded operands to accumulation order - the only change against the bf16 build is the operand rounding."""
    from ubteacher import hip, ops
    g = torch.Generator().manual_seed(2)
    N, H, W, C, K = 2, 25, 42, 256, 256
    try:
        ops.set_precision("fp16")
        assert hip.h16_dtype() == torch.float16
        x = (torch.randn(N, H, W, C, generator=g)).half().cuda()
        w = (torch.randn(K, 9 * C, generator=g) * 0.02).half().cuda()
        b = torch.randn(K, generator=g).cuda()
        y = hip.conv2d_fwd_bf16(x, w, bias=b, pad=1, kh=3, kw=3, out_dtype=torch.float32)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2).cpu(), w.float().view(K, 3, 3, C).permute(0, 3, 1, 2).cpu(), b.cpu(), 1, 1)
        assert relerr(y.permute(0, 3, 1, 2).cpu(), ref) < 2e-4
        y16 = hip.conv2d_fwd_bf16(x, w, bias=b, pad=1, kh=3, kw=3)
        assert y16.dtype == torch.float16 and relerr(y16.float().permute(0, 3, 1, 2).cpu(), ref) < 2e-3
        dy = (torch.randn(N * H * W, K, generator=g) * 0.1).half().cuda()
        dw = torch.zeros(K, 9 * C, device="cuda")
        ri = hip.rowinfo_nhwc(N, H, W, H, W, 1, 1, 3, 3, "cuda")
        hip.conv2d_wgrad_bf16(x, dy, dw, ri, C, 3, 3, accumulate=True)
        xr = x.float().permute(0, 3, 1, 2).cpu().requires_grad_(False)
        wr = w.float().view(K, 3, 3, C).permute(0, 3, 1, 2).cpu().clone().requires_grad_(True)
        F.conv2d(xr, wr, None, 1, 1).backward(dy.float().view(N, H, W, K).permute(0, 3, 1, 2).cpu())
        assert relerr(dw.cpu(), wr.grad.permute(0, 2, 3, 1).reshape(K, -1)) < 2e-4
    finally:
        ops.set_precision("fp32")


@pytest.mark.parametrize("xdt,dydt", [(torch.float32, torch.float32), (BF, BF), (BF, torch.float32), (torch.float32, BF)])
@pytest.mark.parametrize("case", [(2, 25, 42, 256, 256, 3, 1, 1), (2, 13, 21, 64, 128, 3, 2, 1), (2, 50, 84, 64, 256, 1, 1, 0),
                                  (2, 30, 40, 256, 512, 1, 2, 0), (1, 20, 20, 128, 80, 3, 1, 1), (3, 9, 9, 32, 40, 3, 1, 1),
                                  (1, 11, 7, 24, 16, 3, 1, 1), (2, 8, 8, 200, 136, 1, 1, 0),
                                  # shapes of the 256 x 256-tile LDS-DMA kernel (bf16 x and dY, C % 256 == 0, K % 256 == 0, >= 16 chunks of
                                  # 64 pixels): two k-tiles per tap, two co-tiles, a 1x1, a strided 3x3, a ragged pixel tail
                                  (1, 40, 40, 512, 256, 3, 1, 1), (2, 30, 30, 256, 512, 3, 1, 1), (2, 40, 40, 256, 256, 1, 1, 0),
                                  (2, 41, 41, 256, 256, 3, 2, 1), (1, 33, 37, 256, 256, 3, 1, 1)])
def test_conv_bf16_wgrad(case, xdt, dydt):
    from ubteacher import hip
    N, H, W, C, K, k, s, p = case
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, C, H, W, generator=g)
    w = (torch.randn(K, C, k, k, generator=g) * 0.05).requires_grad_(True)
    y = F.conv2d(r16(x), w, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(r16(dy))
    ref = w.grad.permute(0, 2, 3, 1).reshape(K, -1)
    xh = x.permute(0, 2, 3, 1).contiguous().cuda().to(xdt)
    dyh = dy.permute(0, 2, 3, 1).contiguous().cuda().to(dydt)
    ri = hip.rowinfo_nhwc(N, H, W, dyh.shape[1], dyh.shape[2], s, p, k, k, "cuda")
    dw = torch.zeros(K, k * k * C, device="cuda")
    db = torch.zeros(K, device="cuda")
    hip.conv2d_wgrad_bf16(xh, dyh.reshape(-1, K), dw, ri, C, k, k, accumulate=True, db=db)
    hip.conv2d_wgrad_bf16(xh, dyh.reshape(-1, K), dw, ri, C, k, k, accumulate=True, db=db)
    assert relerr(dw.cpu() / 2, ref) < 2e-4
    assert relerr(db.cpu() / 2, r16(dy).sum((0, 2, 3))) < 2e-5      # bias gradient: fp32 column sums of the bf16 dy operand
    dw2 = torch.zeros_like(dw)
    hip.conv2d_wgrad_bf16(xh, dyh.reshape(-1, K), dw2, ri, C, k, k, accumulate=False)
    assert torch.equal(dw2 * 2, dw)                                   # deterministic (fixed-order slab reduction)


@pytest.mark.parametrize("xdt,dydt", [(torch.float32, torch.float32), (BF, BF), (BF, torch.float32)])
def test_conv_ml_bf16_wgrad(xdt, dydt):
    from ubteacher import hip
    g = torch.Generator().manual_seed(3)
    N, C, K, k = 2, 64, 80, 3
    level_hw = [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    xs = [torch.randn(N, C, h, w, generator=g) for h, w in level_hw]
    wt = (torch.randn(K, C, k, k, generator=g) * 0.1).requires_grad_(True)
    ys = [F.conv2d(r16(x), wt, None, 1, 1) for x in xs]
    dys = [torch.randn(y.shape, generator=g) for y in ys]
    torch.autograd.backward(ys, [r16(d) for d in dys])
    big = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, C) for x in xs]).cuda().to(xdt)
    dy = torch.cat([d.permute(0, 2, 3, 1).reshape(-1, K) for d in dys]).cuda().to(dydt)
    dw = torch.zeros(K, k * k * C, device="cuda")
    hip.conv2d_wgrad_bf16(big, dy, dw, hip.rowinfo_ml(N, level_hw, 1, k, "cuda"), C, k, k, accumulate=False)
    assert relerr(dw.cpu(), wt.grad.permute(0, 2, 3, 1).reshape(K, -1)) < 2e-4


def test_conv_ml_bf16_wgrad_big_tile():
    """multi-level (level-first) wgrad on the 256 x 256-tile kernel: tower-shaped C = K = 256 over five levels, with the bias gradient"""
    from ubteacher import hip
    g = torch.Generator().manual_seed(5)
    N, C, K, k = 2, 256, 256, 3
    level_hw = [(25, 42), (13, 21), (7, 11), (4, 6), (2, 3)]
    xs = [torch.randn(N, C, h, w, generator=g) for h, w in level_hw]
    wt = (torch.randn(K, C, k, k, generator=g) * 0.05).requires_grad_(True)
    ys = [F.conv2d(r16(x), wt, None, 1, 1) for x in xs]
    dys = [torch.randn(y.shape, generator=g) for y in ys]
    torch.autograd.backward(ys, [r16(d) for d in dys])
    big = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, C) for x in xs]).cuda().to(BF)
    dy = torch.cat([d.permute(0, 2, 3, 1).reshape(-1, K) for d in dys]).cuda().to(BF)
    assert big.shape[0] >= 16 * 64
    dw = torch.zeros(K, k * k * C, device="cuda")
    db = torch.zeros(K, device="cuda")
    sc = torch.rand(K, generator=g).cuda() + 0.5
    ri = hip.rowinfo_ml(N, level_hw, 1, k, "cuda")
    hip.conv2d_wgrad_bf16(big, dy, dw, ri, C, k, k, accumulate=False, db=db, rowscale=sc)
    ref = wt.grad.permute(0, 2, 3, 1).reshape(K, -1) * sc.cpu()[:, None]
    assert relerr(dw.cpu(), ref) < 2e-4
    refb = torch.cat([r16(d).permute(0, 2, 3, 1).reshape(-1, K) for d in dys]).sum(0) * sc.cpu()
    assert relerr(db.cpu(), refb) < 2e-5
    dw2 = torch.zeros_like(dw)
    hip.conv2d_wgrad_bf16(big, dy, dw2, ri, C, k, k, accumulate=False, rowscale=sc)
    assert torch.equal(dw2, dw)                                       # deterministic


def test_elementwise_bf16():
    """bf16-I/O forms of the glue kernels == the fp32 kernels applied to the same (bf16-representable) values, rounded once."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(4)
    N, H, W, C = 2, 12, 16, 128
    x = r16(torch.randn(N, H, W, C, generator=g)).cuda()
    dy = r16(torch.randn(N, H, W, C, generator=g)).cuda()
    sc = torch.rand(C, generator=g).cuda() + 0.5
    y = torch.relu(x)
    a = hip.relu_bwd_scale(dy.to(BF), y.to(BF), sc)
    assert a.dtype == BF and torch.equal(a, hip.relu_bwd_scale(dy, y, sc).to(BF))
    assert torch.equal(hip.maxpool3x3s2(x, out_dtype=BF), hip.maxpool3x3s2(x).to(BF))
    assert torch.equal(hip.maxpool3x3s2(x.to(BF)), hip.maxpool3x3s2(x).to(BF))
    top = r16(torch.randn(N, H // 2, W // 2, C, generator=g)).cuda()
    assert torch.equal(hip.upsample2x_add(x.to(BF), top.to(BF)), hip.upsample2x_add(x, top).to(BF))
    assert torch.equal(hip.downsample2x_sum(dy.to(BF)), hip.downsample2x_sum(dy).to(BF))
    # GroupNorm: same statistics (fp32 from the same values), one rounding of y / dx
    ga = (torch.rand(C, generator=g) + 0.5).cuda(); be = torch.randn(C, generator=g).cuda()
    seg = [H * W] * N
    y32, m32, r32 = hip.groupnorm_relu_seg_fwd(x.view(-1, C), seg, ga, be, 32, 1e-5, True)
    y16, m16, r16_ = hip.groupnorm_relu_seg_fwd(x.view(-1, C).to(BF), seg, ga, be, 32, 1e-5, True)
    assert torch.equal(m32, m16) and torch.equal(r32, r16_) and torch.equal(y16, y32.to(BF))
    dg32 = torch.zeros(C, device="cuda"); db32 = torch.zeros(C, device="cuda")
    dg16 = torch.zeros(C, device="cuda"); db16 = torch.zeros(C, device="cuda")
    y16f = y16.float()   # the mask (y > 0) and x must be identical in both runs
    dx32 = hip.groupnorm_relu_seg_bwd(dy.view(-1, C), y16f, x.view(-1, C), seg, m32, r32, ga, dg32, db32, 32, True)
    dx16 = hip.groupnorm_relu_seg_bwd(dy.view(-1, C).to(BF), y16, x.view(-1, C).to(BF), seg, m32, r32, ga, dg16, db16, 32, True)
    assert torch.equal(dx16, dx32.to(BF)) and torch.equal(dg16, dg32) and torch.equal(db16, db32)
    # ReLU mask recomputed from x (beta given) == mask read from the stored y
    dg3 = torch.zeros(C, device="cuda"); db3 = torch.zeros(C, device="cuda")
    dx3 = hip.groupnorm_relu_seg_bwd(dy.view(-1, C).to(BF), y16, x.view(-1, C).to(BF), seg, m32, r32, ga, dg3, db3, 32, True, beta=be)
    assert torch.equal(dx3, dx16) and torch.equal(dg3, dg16) and torch.equal(db3, db16)


def test_bn_scale_folding():
    """rowscale / flip-time scale == scaling the result rows (wgrad) / the weights before the bf16 rounding (dgrad image)."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(5)
    N, H, W, C, K, k = 2, 10, 12, 64, 128, 3
    w2 = (torch.randn(K, k * k * C, generator=g) * 0.05).cuda()
    sc = (torch.rand(K, generator=g) + 0.5).cuda()
    a = hip.weight_flip_transpose_bf16(w2, K, k, k, C, sc)
    b = hip.weight_flip_transpose_bf16((w2 * sc[:, None]).contiguous(), K, k, k, C)
    assert torch.equal(a, b)
    x = torch.randn(N, H, W, C, generator=g).cuda().to(BF)
    dy = torch.randn(N * H * W, K, generator=g).cuda().to(BF)
    ri = hip.rowinfo_nhwc(N, H, W, H, W, 1, 1, k, k, "cuda")
    d0 = torch.zeros(K, k * k * C, device="cuda"); d1 = torch.zeros_like(d0)
    b0 = torch.zeros(K, device="cuda"); b1 = torch.zeros_like(b0)
    hip.conv2d_wgrad_bf16(x, dy, d0, ri, C, k, k, accumulate=False, db=b0)
    hip.conv2d_wgrad_bf16(x, dy, d1, ri, C, k, k, accumulate=False, db=b1, rowscale=sc)
    assert torch.equal(d1, d0 * sc[:, None]) and torch.equal(b1, b0 * sc)


def test_stride2_1x1_dgrad_compact_path():
    """compact GEMM + zero interleave == the dilated-gather dgrad kernel for a stride-2 1x1 conv (odd sizes included)"""
    from ubteacher import hip
    g = torch.Generator().manual_seed(6)
    for (N, H, W, C, K) in [(2, 12, 10, 64, 128), (1, 13, 9, 256, 64)]:
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        w2 = (torch.randn(K, C, generator=g) * 0.05).cuda()
        wt16 = hip.weight_flip_transpose_bf16(w2, K, 1, 1, C)
        dy = torch.randn(N, OH, OW, K, generator=g).cuda().to(BF)
        ref = hip.conv2d_dgrad_bf16(dy, wt16, (N, H, W, C), 2, 0, 1, 1, out_dtype=BF)
        got = hip.zero_interleave2x(hip.conv2d_fwd_bf16(dy, wt16, out_dtype=BF), H, W)
        assert torch.equal(ref, got)


def test_stem_bf16():
    """bf16-MFMA image stem (7x7 s2 p3 as KH=7, KW=1, C=32 with a 4-element pixel pitch on the zero-bordered bf16 image)
    == conv2d on the bf16-rounded image and weights, FrozenBN scale/shift + ReLU fused."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(7)
    ims = [torch.randn(3, 50, 70, generator=g) * 50 + 100, torch.randn(3, 64, 61, generator=g) * 50 + 100]
    mean, std = [103.53, 116.28, 123.675], [57.0, 58.0, 59.0]
    x16, sizes = hip.preprocess_images([i.cuda() for i in ims], mean, std, 32, bf16_stem=True)
    x4, _ = hip.preprocess_images([i.cuda() for i in ims], mean, std, 32)
    Hp, Wp = x16.canvas
    assert (Hp, Wp) == tuple(x4.shape[1:3]) and x16.shape == (2, Hp + 6, Wp + 8, 4)
    assert torch.equal(x16[:, 3:3 + Hp, 3:3 + Wp], x4.to(BF))
    border = x16.clone(); border[:, 3:3 + Hp, 3:3 + Wp] = 0
    assert float(border.float().abs().max()) == 0.0
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    w208 = torch.zeros(64, 208)
    w208[:, :196].view(64, 7, 7, 4)[..., :3] = w.permute(0, 2, 3, 1)
    sc = (torch.rand(64, generator=g) + 0.5); sh = torch.randn(64, generator=g) * 0.1
    y = hip.conv2d_stem_fwd_bf16(x16, hip.stem_weight_image(w208.cuda()), sc.cuda(), sh.cuda(), True, BF)
    xin = x4[..., :3].permute(0, 3, 1, 2).cpu()
    ref = torch.relu(F.conv2d(r16(xin), r16(w), None, 2, 3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    assert y.shape == (2, Hp // 2, Wp // 2, 64)
    assert close16(y.cpu().permute(0, 3, 1, 2), ref)
    # second call reuses the cached zero-bordered buffer
    x16b, _ = hip.preprocess_images([i.cuda() for i in ims], mean, std, 32, bf16_stem=True)
    assert x16b.data_ptr() == x16.data_ptr()


def test_dgrad_epilogue_mask_and_residual():
    """fused dgrad epilogue (ReLU mask of the producing layer, residual gradient add) == the separate passes, bit for bit"""
    from ubteacher import hip
    g = torch.Generator().manual_seed(8)
    N, H, W, C, K, k = 2, 14, 18, 64, 128, 3
    w2 = (torch.randn(K, k * k * C, generator=g) * 0.05).cuda()
    wt16 = hip.weight_flip_transpose_bf16(w2, K, k, k, C)
    dy = torch.randn(N, H, W, K, generator=g).cuda().to(BF)
    act = torch.relu(torch.randn(N, H, W, C, generator=g)).cuda().to(BF)   # forward activation the dgrad output belongs to
    res = torch.randn(N, H, W, C, generator=g).cuda().to(BF)
    plain = hip.conv2d_dgrad_bf16(dy, wt16, (N, H, W, C), 1, 1, k, k, out_dtype=torch.float32)
    fused_m = hip.conv2d_dgrad_bf16(dy, wt16, (N, H, W, C), 1, 1, k, k, out_dtype=BF, mask=act)
    assert torch.equal(fused_m, torch.where(act > 0, plain, torch.zeros_like(plain)).to(BF))
    fused_r = hip.conv2d_dgrad_bf16(dy, wt16, (N, H, W, C), 1, 1, k, k, out_dtype=BF, residual=res)
    assert torch.equal(fused_r, (plain + res.float()).to(BF))


def test_w8_tile_kernel_bit_identical_to_128_tile_kernel(monkeypatch):
    """Deep K >= 256 layers run whole rounds of 256 x 256 tiles on the 8-wave LDS-DMA kernel (conv_igemm_bf16_w8) and the remaining
    output rows on the 128 x 128 kernel; both accumulate the same k16 steps in the same order in fp32, so the split launch must equal
    the single-kernel launch (UTV2_W8=0) BIT FOR BIT - multi-level tower conv, masked + residual + ReLU epilogue, fp32 output, wide
    1x1 - and agree with a torch fp32 conv of the same bf16 operands."""
    import torch.nn.functional as F
    from ubteacher import hip
    BF = torch.bfloat16
    torch.manual_seed(0)

    def both(fn):
        monkeypatch.setenv("UTV2_W8", "0")
        a = fn()
        monkeypatch.setenv("UTV2_W8", "1")
        b = fn()
        torch.cuda.synchronize()
        assert torch.equal(a, b)
        return b
    N = 3
    level_hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    P = N * sum(h * w for h, w in level_hw)                      # 67200 rows: one round of 256 tiles + 1664 rows on the 128 kernel
    x = torch.randn(P, 256, device="cuda").to(BF)
    w16 = (torch.randn(256, 9 * 256, device="cuda") * 0.05).to(BF)
    y = both(lambda: hip.conv2d_ml_fwd_bf16(x, w16, level_hw, N, k=3, pad=1, out=torch.empty(P, 256, device="cuda", dtype=BF)).clone())
    r0 = 0
    wt = w16.float().view(256, 3, 3, 256).permute(0, 3, 1, 2)
    for (h, w_) in level_hw[:2]:
        xs = x[r0:r0 + N * h * w_].float().view(N, h, w_, 256).permute(0, 3, 1, 2)
        ref = F.conv2d(xs, wt, padding=1).permute(0, 2, 3, 1).reshape(-1, 256)
        got = y[r0:r0 + N * h * w_].float()
        assert float((got - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
        r0 += N * h * w_
    xn = torch.randn(8, 100, 100, 512, device="cuda").to(BF)
    wn = (torch.randn(512, 9 * 512, device="cuda") * 0.03).to(BF)
    sc = torch.rand(512, device="cuda") + 0.5
    bi = torch.randn(512, device="cuda")
    res = torch.randn(8, 100, 100, 512, device="cuda").to(BF)
    msk = torch.randn(8, 100, 100, 512, device="cuda").to(BF)
    both(lambda: hip.conv2d_fwd_bf16(xn, wn, scale=sc, bias=bi, residual=res, stride=1, pad=1, relu=True, kh=3, kw=3, mask=msk))
    both(lambda: hip.conv2d_fwd_bf16(xn, wn, stride=1, pad=1, kh=3, kw=3, out_dtype=torch.float32))
    x2 = torch.randn(12, 50, 84, 256, device="cuda").to(BF)   # 197 tiles: more than half a round -> all on the big tile, last row tile partial
    w2 = (torch.randn(256, 9 * 256, device="cuda") * 0.05).to(BF)
    both(lambda: hip.conv2d_fwd_bf16(x2, w2, stride=1, pad=1, kh=3, kw=3, relu=True))
    x1 = torch.randn(4, 128, 160, 1024, device="cuda").to(BF)
    w1 = (torch.randn(256, 1024, device="cuda") * 0.03).to(BF)
    both(lambda: hip.conv2d_fwd_bf16(x1, w1))


def test_post_mask_epilogue_and_masked_zero_interleave():
    """post_mask: y = post_mask > 0 ? (conv [* mask] + residual) : 0 - the gradient of a ReLU output masked where it is produced;
    same operand on the stride-2 zero-interleave."""
    from ubteacher import hip
    BF = torch.bfloat16
    torch.manual_seed(1)
    N, Hh, Ww, C, K = 2, 20, 24, 64, 128
    dy = torch.randn(N, Hh, Ww, K, device="cuda").to(BF)
    wt = (torch.randn(C, K, device="cuda") * 0.1).to(BF)          # 1x1 dgrad weight image [C][K]
    res = torch.randn(N, Hh, Ww, C, device="cuda").to(BF)
    mk = torch.randn(N, Hh, Ww, C, device="cuda").to(BF)
    pm = torch.randn(N, Hh, Ww, C, device="cuda").to(BF)
    base = hip.conv2d_dgrad_bf16(dy, wt, (N, Hh, Ww, C), 1, 0, 1, 1, out_dtype=torch.float32)
    got = hip.conv2d_dgrad_bf16(dy, wt, (N, Hh, Ww, C), 1, 0, 1, 1, out_dtype=BF, mask=mk, residual=res, post_mask=pm)
    want = torch.where(pm.float() > 0, torch.where(mk.float() > 0, base, torch.zeros_like(base)) + res.float(), torch.zeros_like(base)).to(BF)
    assert torch.equal(got, want)
    f32 = hip.conv2d_dgrad_bf16(dy, wt, (N, Hh, Ww, C), 1, 0, 1, 1, out_dtype=torch.float32, residual=res.float(), post_mask=pm.float())
    assert torch.equal(f32, torch.where(pm.float() > 0, base + res.float(), torch.zeros_like(base)))
    c = torch.randn(N, 10, 12, C, device="cuda").to(BF)
    full_mask = torch.randn(N, 20, 24, C, device="cuda").to(BF)
    other = torch.randn(N, 20, 24, C, device="cuda").to(BF)       # fused add of another producer's gradient, mask as a bit plane
    za = hip.zero_interleave2x(c, 20, 24, mask_bits=_pack_bits(full_mask), add=other)
    refa = other.float().clone()
    refa[:, ::2, ::2] += torch.where(full_mask[:, ::2, ::2].float() > 0, c, torch.zeros_like(c)).float()
    assert torch.equal(za, refa.to(BF))
    assert torch.equal(hip.zero_interleave2x(c, 20, 24, add=other)[:, 1::2], other[:, 1::2])
    z = hip.zero_interleave2x(c, 20, 24, mask=full_mask)
    ref = torch.zeros(N, 20, 24, C, device="cuda", dtype=BF)
    ref[:, ::2, ::2] = torch.where(full_mask[:, ::2, ::2].float() > 0, c, torch.zeros_like(c))
    assert torch.equal(z, ref)
    assert torch.equal(hip.zero_interleave2x(c, 20, 24)[:, ::2, ::2], c)


def _pack_bits(t):
    """uint8 [..., K / 8]: bit q of byte c = t[..., 8c + q] > 0"""
    b = (t.float() > 0).reshape(t.shape[:-1] + (t.shape[-1] // 8, 8)).to(torch.int32)
    return (b << torch.arange(8, device=t.device, dtype=torch.int32)).sum(-1).to(torch.uint8)


@pytest.mark.parametrize("case", [(2, 20, 24, 64, 128, 1, "v2"), (2, 14, 18, 64, 64, 3, "small"), (4, 64, 80, 256, 256, 3, "tile256"),
                                  (2, 50, 84, 256, 1024, 1, "wide")])
def test_relu_bit_planes_written_and_read_by_the_conv_epilogues(case):
    """utv2_conv2d_nhwc_fwd_bf16_bits: the forward epilogue writes the bit plane `output > 0`; a dgrad that reads bit planes in place of
    the 16-bit mask / post_mask tensors returns the same bits (every tile kernel: 64- and 128-wide tiles, the 256 x 256 tile)."""
    from ubteacher import hip
    N, H, W, C, K, k, _ = case
    g = torch.Generator().manual_seed(C + K + k)
    x = torch.randn(N, H, W, C, generator=g).cuda().to(BF)
    w16 = (torch.randn(K, k * k * C, generator=g) * (1.0 / (k * k * C) ** 0.5)).cuda().to(BF)
    bias = torch.randn(K, generator=g).cuda() * 0.1
    res = torch.randn(N, H, W, K, generator=g).cuda().to(BF)
    y0 = hip.conv2d_fwd_bf16(x, w16, bias=bias, residual=res, pad=k // 2, relu=True, kh=k, kw=k, out_dtype=BF)
    bits = hip.relu_bits_buffer((N, H, W, K), x.device)
    bits.fill_(0xA5)
    y1 = hip.conv2d_fwd_bf16(x, w16, bias=bias, residual=res, pad=k // 2, relu=True, kh=k, kw=k, out_dtype=BF, relu_bits=bits)
    assert torch.equal(y0, y1)
    assert torch.equal(bits, _pack_bits(y1))
    assert 0.2 < float((y1 > 0).float().mean()) < 0.8
    # dgrad of a K -> C layer: the output has C channels; masks as 16-bit tensors vs as bit planes
    wt16 = (torch.randn(C, k * k * K, generator=g) * 0.05).cuda().to(BF)
    dy = torch.randn(N, H, W, K, generator=g).cuda().to(BF)
    mk = torch.relu(torch.randn(N, H, W, C, generator=g)).cuda().to(BF)
    pm = torch.relu(torch.randn(N, H, W, C, generator=g)).cuda().to(BF)
    rs = torch.randn(N, H, W, C, generator=g).cuda().to(BF)
    for kw16, kwb in ((dict(mask=mk), dict(mask_bits=_pack_bits(mk))),
                      (dict(post_mask=pm, residual=rs), dict(post_mask_bits=_pack_bits(pm), residual=rs)),
                      (dict(mask=mk, post_mask=pm, residual=rs), dict(mask_bits=_pack_bits(mk), post_mask_bits=_pack_bits(pm), residual=rs)),
                      (dict(mask=mk, post_mask=pm), dict(mask=mk, post_mask_bits=_pack_bits(pm)))):
        a = hip.conv2d_dgrad_bf16(dy, wt16, (N, H, W, C), 1, k // 2, k, k, out_dtype=BF, **kw16)
        b = hip.conv2d_dgrad_bf16(dy, wt16, (N, H, W, C), 1, k // 2, k, k, out_dtype=BF, **kwb)
        assert torch.equal(a, b), sorted(kwb)
    # both forms of one mask, an fp32 output or K % 8 != 0 are rejected
    with pytest.raises(Exception):
        hip.conv2d_dgrad_bf16(dy, wt16, (N, H, W, C), 1, k // 2, k, k, out_dtype=torch.float32, mask_bits=_pack_bits(mk))


def test_batched_weight_flip_equals_per_layer_flip():
    """one launch for all layers (tiled transpose) == the per-layer element-wise kernel, bit for bit, incl. ragged K / C, scales and a
    zero-padded output-channel axis"""
    import struct
    from ubteacher import hip
    torch.manual_seed(3)
    layers = [(256, 3, 3, 256, True), (80, 3, 3, 256, False), (72, 1, 1, 40, True), (512, 1, 1, 2048, False), (64, 7, 1, 32, True)]
    arena, scales, recs, refs = [], [], [], []
    w_off = dst_off = sc_off = 0
    for (K, KH, KW, C, has_scale) in layers:
        w = torch.randn(K, KH, KW, C, device="cuda")
        sc = (torch.rand(K, device="cuda") + 0.5) if has_scale else None
        arena.append(w.reshape(-1))
        if sc is not None:
            scales.append(sc)
        Kpad = 96 if K == 80 else K          # the 80-channel prediction convs: image zero-padded to 96 output channels
        recs.append(struct.pack("<qqqiiiiii", w_off, dst_off, sc_off if has_scale else -1, K, KH, KW, C, Kpad, 0))
        ref = hip.weight_flip_transpose_bf16(w, K, KH, KW, C, scale=sc)
        refs.append(torch.nn.functional.pad(ref.view(C, KH * KW, K), (0, Kpad - K)).reshape(-1))
        w_off += w.numel(); dst_off += C * KH * KW * Kpad; sc_off += K if has_scale else 0
    arena = torch.cat(arena).contiguous()
    scales = torch.cat(scales).contiguous()
    table = torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).cuda()
    bank = torch.zeros(dst_off, dtype=torch.bfloat16, device="cuda")
    hip.weight_flip_transpose_bf16_batched(arena, scales, bank, table, len(layers))
    torch.cuda.synchronize()
    assert torch.equal(bank, torch.cat(refs))


def test_big_tile_kernels_bit_identical_across_schedules(tmp_path):
    """The kernels a deep bf16 conv can run on - the 128 x 128 tile (UTV2_W8=0), the 256 x 256 ping-pong tile (one tile per workgroup,
    UTV2_PP=1, and the default persistent grid) and its row-span form (default for 3x3; UTV2_PP_RS=0: plain ping-pong) - accumulate every
    output element in the same order: bit-identical outputs on a multi-level
    tower conv, a 3x3 with mask + residual + ReLU epilogue, an fp32-output conv and a wide 1x1 (tools/check_w8.py; the switches are
    read once per process, hence the subprocesses)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = str(tmp_path / "ref.pt")

    def run(env, *args):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_w8.py")] + list(args), env=e, capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stdout

    run({"UTV2_W8": "0"}, "save", ref)
    # ... and the epilogue's branch-free plain path (default) stores the same bits as the general one (UTV2_EPI_PLAIN=0), GroupNorm
    # partial sums and ReLU bit planes included
    # (the default 3x3 path is the row-span form conv_igemm_bf16_rs since round 5; UTV2_PP_RS=0: the ping-pong kernel it derives from)
    for env in ({"UTV2_PP": "1"}, {"UTV2_PP": "1", "UTV2_PP_RS": "0"}, {}, {"UTV2_PP_RS": "0"}, {"UTV2_EPI_PLAIN": "0"},
                {"UTV2_EPI_PLAIN": "0", "UTV2_W8": "0"}):
        lines = [ln for ln in run(env, "cmp", ref).splitlines() if ln.strip()]
        assert len(lines) == 14 and all("bit-identical" in ln and "nan" not in ln for ln in lines), lines


@pytest.mark.parametrize("dt", [torch.float32, BF])
def test_maxpool_backward_first_maximum_rule(dt):
    """utv2_maxpool3x3s2_bwd_nhwc (trainable stem): the gradient of ATen's max_pool2d - ties go to the FIRST maximum of a window in scan
    order, which post-ReLU inputs full of equal zeros exercise - with the ReLU mask of the layer in front fused; odd and even sizes."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(5)
    for (N, H, W, C) in ((2, 14, 18, 64), (1, 15, 17, 8), (2, 9, 12, 4)):
        x = torch.relu(torch.randn(N, H, W, C, generator=g)).mul(4).round().div(4)    # many exact ties, many zeros
        x = x.to(dt).cuda()
        xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
        y = F.max_pool2d(xr, 3, 2, 1)
        dy = torch.randn(y.shape, generator=g).to(dt).cuda()
        y.backward(dy.float())
        want = (xr.grad * (xr > 0)).permute(0, 2, 3, 1)
        got = hip.maxpool3x3s2_bwd(x, dy.permute(0, 2, 3, 1).contiguous(), relu=True)
        assert got.dtype == dt
        if dt == torch.float32:
            assert torch.equal(got, want)
        else:
            assert float((got.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
        raw = hip.maxpool3x3s2_bwd(x, dy.permute(0, 2, 3, 1).contiguous(), relu=False)
        assert float((raw.float() - xr.grad.permute(0, 2, 3, 1)).abs().max()) <= (0 if dt == torch.float32 else 2.0 ** -7 * float(want.abs().max()))


@pytest.mark.parametrize("ydt", [BF, torch.float32])
def test_tall_256x64_tile_on_long_narrow_layers(ydt):
    """K <= 64 layers with a real K loop and >= 65536 output pixels (res2's 3x3 64 -> 64 at benchmark size, the stem) run on the
    256 x 64 tile of conv_igemm_bf16_v2<64, ..., TALL> (round 4): == an fp32 conv on the rounded operands, with FrozenBN scale / shift,
    residual and ReLU in the epilogue, a ragged last row tile, and the ReLU bit plane written from the same stored values."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(5)
    N, H, W, C, K = 2, 203, 167, 64, 64                  # M = 67 802: not a multiple of 256
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.05
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.1
    xh = x.permute(0, 2, 3, 1).contiguous().cuda().to(BF)
    w16 = w.permute(0, 2, 3, 1).reshape(K, -1).contiguous().cuda().to(BF)
    res = torch.randn(N, H, W, K, generator=g).cuda().to(ydt)
    y = hip.conv2d_fwd_bf16(xh, w16, scale=sc.cuda(), bias=sh.cuda(), residual=res, stride=1, pad=1, kh=3, kw=3, relu=True, out_dtype=ydt)
    ref = torch.relu(F.conv2d(r16(x), r16(w), None, 1, 1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res.float().cpu().permute(0, 3, 1, 2))
    yc = y.cpu().permute(0, 3, 1, 2)
    assert relerr(yc, ref) < 2e-4 if ydt == torch.float32 else close16(yc, ref)
    if ydt == BF:
        bits = hip.relu_bits_buffer((N, H, W, K), "cuda")
        y2 = hip.conv2d_fwd_bf16(xh, w16, scale=sc.cuda(), bias=sh.cuda(), stride=1, pad=1, kh=3, kw=3, relu=True, out_dtype=ydt, relu_bits=bits)
        want = (y2.float() > 0).view(-1, K // 8, 8)
        got = ((bits.view(-1, K // 8, 1).int() >> torch.arange(8, device="cuda").view(1, 1, 8)) & 1).bool()
        assert torch.equal(got, want)


def test_stem_on_the_tall_tile_at_benchmark_size():
    """the image stem at a size whose 400 x 336 output grid (134 400 pixels) reaches the 256 x 64 tile"""
    from ubteacher import hip
    g = torch.Generator().manual_seed(8)
    ims = [torch.randn(3, 800, 672, generator=g) * 50 + 100]
    mean, std = [103.53, 116.28, 123.675], [57.0, 58.0, 59.0]
    x16, _ = hip.preprocess_images([i.cuda() for i in ims], mean, std, 32, bf16_stem=True)
    x4, _ = hip.preprocess_images([i.cuda() for i in ims], mean, std, 32)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    w208 = torch.zeros(64, 208)
    w208[:, :196].view(64, 7, 7, 4)[..., :3] = w.permute(0, 2, 3, 1)
    sc = (torch.rand(64, generator=g) + 0.5); sh = torch.randn(64, generator=g) * 0.1
    y = hip.conv2d_stem_fwd_bf16(x16, hip.stem_weight_image(w208.cuda()), sc.cuda(), sh.cuda(), True, BF)
    xin = x4[..., :3].permute(0, 3, 1, 2).cpu()
    ref = torch.relu(F.conv2d(r16(xin), r16(w), None, 2, 3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    assert y.shape[1] * y.shape[2] >= 65536 and close16(y.cpu().permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("shortcut", [False, True])
@pytest.mark.parametrize("shape", [(2, 24, 48), (1, 21, 37), (3, 8, 16), (1, 5, 7)])
def test_fused_frozen_bottleneck_equals_the_conv_chain(shape, shortcut):
    """csrc/bottleneck.hip (round 4): a frozen stride-1 bottleneck of R-50's res2 (D2 BottleneckBlock with FrozenBN as scale / shift,
    reference backbone fpn.py:21-22) as ONE kernel with the 64-channel intermediates in LDS - the identity blocks (x has 256 channels) and
    the stage's first block (x has 64, 1x1 shortcut conv) - against the three / four conv launches it replaces (same 16-bit roundings of
    c1 / c2 / the shortcut, fp32 accumulation in another chunk order) and against torch on the same rounded operands.  Ragged tiles
    (H % 8, W % 16 != 0), images smaller than a tile, several images."""
    from ubteacher import hip
    N, H, W = shape
    C, MID, K = (64 if shortcut else 256), 64, 256
    h16 = hip.h16_dtype()
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(N, H, W, C, generator=g).clamp(min=0) * 0.7).to(h16).cuda()          # a ReLU / max-pool output, as in the network
    w1 = (torch.randn(MID, C, generator=g) * (0.06 if not shortcut else 0.12)).to(h16).cuda()
    w2 = (torch.randn(MID, 9 * MID, generator=g) * 0.05).to(h16).cuda()
    w3 = (torch.randn(K, MID, generator=g) * 0.1).to(h16).cuda()
    wsc = (torch.randn(K, C, generator=g) * 0.1).to(h16).cuda() if shortcut else None
    sb = [(torch.rand(k, generator=g) + 0.5).cuda() if i % 2 == 0 else (torch.randn(k, generator=g) * 0.2).cuda()
          for i, k in enumerate((MID, MID, MID, MID, K, K, K, K))]
    s1, b1, s2, b2, s3, b3, ssc, bsc = sb
    assert hip.bottleneck_supported(C, MID, shortcut) and not hip.bottleneck_supported(512, 128, False) and not hip.bottleneck_supported(256, 64, True)
    kw = dict(wsc=wsc, ssc=ssc, bsc=bsc) if shortcut else {}
    y = hip.bottleneck_fwd_bf16(x, w1, w2, w3, s1, b1, s2, b2, s3, b3, **kw)
    c1 = hip.conv2d_fwd_bf16(x, w1, scale=s1, bias=b1, relu=True)
    c2 = hip.conv2d_fwd_bf16(c1, w2, scale=s2, bias=b2, relu=True, kh=3, kw=3, pad=1)
    r = hip.conv2d_fwd_bf16(x, wsc, scale=ssc, bias=bsc) if shortcut else x
    y3 = hip.conv2d_fwd_bf16(c2, w3, scale=s3, bias=b3, relu=True, residual=r)
    assert y.shape == y3.shape == (N, H, W, K) and y.dtype == h16
    d = (y.float() - y3.float()).abs()
    ulp = 2 ** -7 if h16 == torch.bfloat16 else 2 ** -10
    # c1 / c2 may flip by one 16-bit ulp between the two accumulation orders, which then propagates: bounded by a few output ulps
    assert float(d.max()) <= 4 * ulp * float(y3.float().abs().max()), float(d.max())
    assert float((d > 0).float().mean()) < 0.05
    # torch on the same rounded operands, fp32 math with the same 16-bit roundings of the intermediates
    xf = x.float().permute(0, 3, 1, 2)
    v = lambda t: t.view(1, -1, 1, 1)
    t1 = F.relu(F.conv2d(xf, w1.float().view(MID, C, 1, 1)) * v(s1) + v(b1)).to(h16).float()
    t2 = F.relu(F.conv2d(t1, w2.float().view(MID, 3, 3, MID).permute(0, 3, 1, 2), padding=1) * v(s2) + v(b2)).to(h16).float()
    rt = (F.conv2d(xf, wsc.float().view(K, C, 1, 1)) * v(ssc) + v(bsc)).to(h16).float() if shortcut else xf
    t3 = F.relu(F.conv2d(t2, w3.float().view(K, MID, 1, 1)) * v(s3) + v(b3) + rt)
    assert relerr(y.float().permute(0, 3, 1, 2), t3) < (2e-2 if h16 == torch.bfloat16 else 3e-3)
    assert torch.equal(y, hip.bottleneck_fwd_bf16(x, w1, w2, w3, s1, b1, s2, b2, s3, b3, **kw))     # deterministic


@pytest.mark.parametrize("hw", [(64, 96), (70, 66), (800, 1344), (33, 18)])
def test_fused_stem_pool_equals_stem_conv_then_maxpool(hw):
    """csrc/stem_pool.hip (round 4): the frozen BasicStem conv (7x7 s2 p3 + FrozenBN + ReLU) and max_pool2d(3, 2, 1) as one kernel with
    the conv output in LDS - against the two launches it replaces (same 16-bit rounding of the conv output, fp32 sums in another k
    order; the max is exact) and against torch on the same rounded operands.  Odd conv / pooled extents, tiles cut by the image edge,
    the benchmark size."""
    from ubteacher import hip
    H, W = hw
    N = 2 if H < 400 else 1
    h16 = hip.h16_dtype()
    g = torch.Generator().manual_seed(3)
    imgs = [(torch.rand(3, H, W, generator=g) * 255).cuda() for _ in range(N)]
    mean, std = [103.53, 116.28, 123.675], [57.0, 57.0, 58.0]
    x4, sizes = hip.preprocess_images(imgs, mean, std, 2, bf16_stem=True)
    assert x4.dtype == h16 and x4.canvas == (H + H % 2, W + W % 2)
    w208 = torch.zeros(64, 208)
    w208[:, :196] = torch.randn(64, 196, generator=g) * 0.05
    w16s = hip.stem_weight_image(w208.cuda())
    sc = (torch.rand(64, generator=g) + 0.5).cuda()
    sh = (torch.randn(64, generator=g) * 0.3).cuda()
    y = hip.stem_pool_fwd_bf16(x4, w16s, sc, sh)
    c = hip.conv2d_stem_fwd_bf16(x4, w16s, sc, sh, True, h16)
    ref = hip.maxpool3x3s2(c)
    assert y.shape == ref.shape and y.dtype == h16
    d = (y.float() - ref.float()).abs()
    ulp = 2 ** -7 if h16 == torch.bfloat16 else 2 ** -10
    assert float(d.max()) <= 2 * ulp * float(ref.float().abs().max()), float(d.max())      # a conv value one ulp off may win a window
    assert float((d > 0).float().mean()) < 0.02
    # torch on the same rounded operands
    Hc, Wc = x4.canvas
    xi = x4[:, 3:3 + Hc, 3:3 + Wc, :3].float().permute(0, 3, 1, 2)
    wt = w16s.float().view(64, 7, 8, 4)[:, :, :7, :3].permute(0, 3, 1, 2)
    t = F.relu(F.conv2d(xi, wt, stride=2, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).to(h16).float()
    t = F.max_pool2d(t, 3, 2, 1)
    assert relerr(y.float().permute(0, 3, 1, 2), t) < (2e-2 if h16 == torch.bfloat16 else 3e-3)
    assert torch.equal(y, hip.stem_pool_fwd_bf16(x4, w16s, sc, sh))


@pytest.mark.parametrize("case", [(40, 300, 3, 128, 256), (48, 350, 2, 128, 256), (34, 1000, 1, 128, 256), (8, 37, 131, 192, 384), (3, 111, 107, 256, 512)])
def test_row_span_kernel_on_narrow_and_ragged_geometry(case):
    """conv_igemm_bf16_rs (round 5: one LDS-resident span of input rows per kernel row serves its three taps; column borders by switching
    a lane's fragment address to zero bytes) on the geometries that stress exactly that: images 3, 2 and 1 pixels wide (every output
    pixel sits on a column border; at W = 1 both side taps of every row are masked; a 256-row tile spans > 80 image rows and several
    images), an odd number of span groups (C = 192: 9), K that is not a multiple of the 256-column tile (384), a ragged last row tile -
    forward and dgrad, against an fp32 conv on the rounded operands.  (Bit-identity with the ping-pong and the 128-tile kernels on the
    benchmark shapes: test_big_tile_kernels_bit_identical_across_schedules.)"""
    from ubteacher import hip
    N, H, W, C, K = case
    g = torch.Generator().manual_seed(11)
    x = torch.relu(torch.randn(N, C, H, W, generator=g))
    w = torch.randn(K, C, 3, 3, generator=g) * 0.05
    xh = x.permute(0, 2, 3, 1).contiguous().cuda().to(BF)
    w16 = w.permute(0, 2, 3, 1).reshape(K, -1).contiguous().cuda().to(BF)
    y = hip.conv2d_fwd_bf16(xh, w16, stride=1, pad=1, kh=3, kw=3, out_dtype=BF)
    ref = F.conv2d(r16(x), r16(w), None, 1, 1)
    assert close16(y.cpu().permute(0, 3, 1, 2), ref)
    # dgrad = the same kernel over dY with the flipped / transposed weights
    dy = torch.randn(N, K, H, W, generator=g)
    dyh = dy.permute(0, 2, 3, 1).contiguous().cuda().to(BF)
    wt16 = hip.weight_flip_transpose_bf16(w.permute(0, 2, 3, 1).contiguous().cuda(), K, 3, 3, C)
    dx = hip.conv2d_dgrad_bf16(dyh, wt16, (N, H, W, C), 1, 1, 3, 3, out_dtype=BF)
    refd = F.conv_transpose2d(r16(dy), r16(w), None, 1, 1)
    assert close16(dx.cpu().permute(0, 3, 1, 2), refd)


def test_big_tile_weight_gradients_are_deterministic_across_processes(tmp_path):
    """conv_wgrad_bf16_pp (the 256-tile weight-gradient kernel on the ping-pong schedule: 32-pixel chunks in a ring of four stages, the two
    wave groups one slot apart) + the fixed-order slab reduction: two processes produce the SAME bits - single and paired towers, a wide
    layer, accumulate on / off, a ragged NHWC 3x3 (tools/check_wgrad_pp.py).  (Until round 5 this compared the kernel with the lock-step
    schedule it replaced, bit for bit; that kernel left the library in round 6 - tools/probe/conv_lockstep.h - and
    test_conv_ml_bf16_wgrad_big_tile pins the values against the fp32 reference.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = str(tmp_path / "wref.pt")

    def run(env, *args):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_wgrad_pp.py")] + list(args), env=e, capture_output=True, text=True,
                             timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return out.stdout

    run({}, "save", ref)
    lines = [ln for ln in run({}, "cmp", ref).splitlines() if ln.strip() and "amdgpu" not in ln]
    assert len(lines) == 7 and all("bit-identical" in ln and "nan" not in ln for ln in lines), lines


def _rowinfo_host(N, H, W, OH, OW, stride, pad, kh, kw, start):
    """the geometry table as the host built it until round 5 (torch CPU ops): {input pixel index of tap (0,0), (W << 16) | tap mask}"""
    n = torch.arange(N, dtype=torch.int64).view(N, 1, 1)
    ih0 = (torch.arange(OH, dtype=torch.int64) * stride - pad).view(1, OH, 1)
    iw0 = (torch.arange(OW, dtype=torch.int64) * stride - pad).view(1, 1, OW)
    anchor = (start + n * (H * W) + ih0 * W + iw0).expand(N, OH, OW)
    mask = torch.zeros((1, OH, OW), dtype=torch.int64)
    for a in range(kh):
        for b in range(kw):
            ok = ((ih0 + a >= 0) & (ih0 + a < H)) & ((iw0 + b >= 0) & (iw0 + b < W))
            mask = mask | (ok.to(torch.int64) << (a * kw + b))
    word = ((W << 16) | mask).expand(N, OH, OW)
    return torch.stack((anchor, word), dim=-1).reshape(-1, 2).to(torch.int32)


def test_device_built_geometry_tables_equal_the_host_formula():
    """utv2_rowinfo_nhwc (round 6: the per-output-pixel geometry table of the 16-bit convs / weight gradients built on the device, so that a
    new canvas - every batch of the reference recipes' ResizeShortestEdge range - costs no host arithmetic and no copy) against the host
    formula it replaces, bit for bit: 3x3 / 1x1 / 7x7, strides 1 and 2, with and without padding, odd extents, a level-concatenated table."""
    from ubteacher import hip
    hip._rowinfo_cache.clear()
    for (N, H, W, s, p, k) in ((2, 25, 42, 1, 1, 3), (3, 13, 21, 2, 1, 3), (2, 50, 83, 1, 0, 1), (1, 51, 85, 2, 0, 1), (2, 37, 29, 2, 3, 7 if False else 3),
                               (4, 200, 336, 1, 1, 3), (1, 7, 11, 1, 1, 3)):
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        got = hip.rowinfo_nhwc(N, H, W, OH, OW, s, p, k, k, "cuda")
        assert got.dtype == torch.int32 and tuple(got.shape) == (N * OH * OW, 2)
        assert torch.equal(got.cpu(), _rowinfo_host(N, H, W, OH, OW, s, p, k, k, 0)), (N, H, W, s, p, k)
    level_hw = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    N = 3
    got = hip.rowinfo_ml(N, level_hw, 1, 3, "cuda")
    parts, start = [], 0
    for h, w in level_hw:
        parts.append(_rowinfo_host(N, h, w, h, w, 1, 1, 3, 3, start))
        start += N * h * w
    assert torch.equal(got.cpu(), torch.cat(parts))
    assert hip.rowinfo_ml(N, level_hw, 1, 3, "cuda") is got       # cached per geometry
