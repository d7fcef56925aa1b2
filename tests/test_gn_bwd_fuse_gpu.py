"""GroupNorm backward's first reduction taken in the producing dgrad's epilogue (round 6).

Inside the FCOS towers' conv -> GN -> ReLU -> conv chain (reference fcos/fcos.py:252-304) the gradient of a GroupNorm + ReLU output is made by
the NEXT conv's dgrad.  With UTV2_GN_BWD_FUSE=1 that dgrad (utv2_conv2d_ml_fwd_bf16_gnb) applies the ReLU mask - a bit plane the
GroupNorm's apply pass wrote (utv2_groupnorm_relu_seg_fwd_p32b) - and leaves per 64-row block and channel {sum g, sum g * x} while its rows
are in registers; utv2_groupnorm_seg_bwd_p64 finishes the backward from them.  Checked here against the unfused kernels on the same inputs:
the bit plane and the masked gradient bit for bit, the partial sums against float64 sums of the stored values, dx / dgamma / dbeta to the
rounding of sums taken in another order (stated below), on the three kernel routes (row-span 256-tile alone, row-span + 128-tile
remainder, 128-tile alone) - then a whole FCOS step with the switch on and off."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

LEVELS_BIG = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]     # the 1333 x 800 canvas: 22 400 rows per image
LEVELS_SMALL = [(28, 40), (14, 20), (7, 10), (4, 5), (2, 3)]


def _pack_bits(t):
    b = (t.float() > 0).reshape(-1, 8).to(torch.int32)
    return (b << torch.arange(8, device=t.device, dtype=torch.int32)).sum(-1).to(torch.uint8)


def _chain(level_hw, N, groups, seed):
    """conv_a -> GN -> ReLU (with bit plane) on random level-first features, and a random gradient of the NEXT conv's output"""
    from ubteacher import hip
    h16 = hip.h16_dtype()
    g = torch.Generator(device="cuda").manual_seed(seed)
    C = 256
    K = C * groups
    P = N * sum(h * w for h, w in level_hw)
    seg_rows = [h * w for h, w in level_hw for _ in range(N)]
    x0 = torch.relu(torch.randn(P, K, device="cuda", generator=g)).to(h16)
    wa = (torch.randn(K, 9 * C, device="cuda", generator=g) * 0.02).to(h16)
    part32 = hip.gn_part_buffer(P, K, x0.device)
    xa = hip.conv2d_ml_fwd_bf16(x0, wa, level_hw, N, bias=torch.randn(K, device="cuda", generator=g) * 0.1, k=3, pad=1, groups=groups,
                                gn_part=part32)
    gamma = torch.rand(K, device="cuda", generator=g) + 0.5
    beta = torch.randn(K, device="cuda", generator=g) * 0.3
    G = K // 8
    bits = torch.zeros(P * K // 8, dtype=torch.uint8, device="cuda")
    y, mean, rstd = hip.groupnorm_relu_seg_fwd_p32(xa, seg_rows, gamma, beta, part32, G, relu_bits=bits)
    y_plain, mean2, rstd2 = hip.groupnorm_relu_seg_fwd_p32(xa, seg_rows, gamma, beta, part32, G)
    assert torch.equal(y, y_plain) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    gout = (torch.randn(P, K, device="cuda", generator=g) * 0.05).to(h16)       # gradient of the next conv's output
    wt = (torch.randn(K, 9 * C, device="cuda", generator=g) * 0.02).to(h16)      # its dgrad weight image
    return dict(P=P, K=K, G=G, seg_rows=seg_rows, xa=xa, y=y, mean=mean, rstd=rstd, gamma=gamma, beta=beta, bits=bits, gout=gout, wt=wt)


@pytest.fixture(params=["fp16", "bf16"])
def h16(request):
    from ubteacher import ops
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision("fp32")


@pytest.mark.parametrize("route,level_hw,N", [("rs", LEVELS_BIG, 1), ("rs+128", LEVELS_BIG, 3), ("128", LEVELS_SMALL, 2)])
@pytest.mark.parametrize("groups", [2, 1])
def test_dgrad_epilogue_partials_and_groupnorm_backward(route, level_hw, N, groups, h16):
    from ubteacher import hip
    c = _chain(level_hw, N, groups, seed=3 + N)
    P, K, G = c["P"], c["K"], c["G"]
    # the bit plane is the sign of the fp32 output (as the unfused backward recomputes it from x): every stored positive has its bit, and
    # a bit without a stored positive is a value that rounded to zero in 16 bits - next to none
    stored = _pack_bits(c["y"])
    assert torch.equal(c["bits"] & stored, stored)
    extra = (c["bits"] ^ stored).to(torch.int32)
    assert int((extra != 0).sum()) <= max(4, P * K // 1000000), int((extra != 0).sum())
    # unfused: dgrad, then GroupNorm backward (mask recomputed from x) in three launches
    dxc = hip.conv2d_ml_fwd_bf16(c["gout"], c["wt"], level_hw, N, k=3, pad=1, groups=groups)
    dga_ref, dbe_ref = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
    dx_ref, col_ref = hip.groupnorm_relu_seg_bwd(dxc, c["y"], c["xa"], c["seg_rows"], c["mean"], c["rstd"], c["gamma"], dga_ref, dbe_ref, G,
                                                 True, beta=c["beta"], want_colsum=True)
    # fused
    assert hip.gnb_eligible(c["gout"], c["wt"], 3, 1, groups)
    part = hip.gnb_part_buffer(P, K, "cuda")
    part.fill_(float("nan"))
    gm = hip.conv2d_ml_fwd_bf16(c["gout"], c["wt"], level_hw, N, k=3, pad=1, groups=groups, gnb=(c["bits"], c["xa"], part))
    mask = ((c["bits"].view(-1, 1).to(torch.int32) >> torch.arange(8, device="cuda", dtype=torch.int32)) & 1).bool().view(P, K)
    assert torch.equal(gm, torch.where(mask, dxc, torch.zeros_like(dxc))), "masked gradient differs from dgrad x mask"
    # partial sums of the STORED values, against float64
    nb = (P + 63) // 64
    pad_rows = nb * 64 - P
    g64 = torch.cat([gm.double(), torch.zeros(pad_rows, K, dtype=torch.float64, device="cuda")]).view(nb, 64, K)
    x64 = torch.cat([c["xa"].double(), torch.zeros(pad_rows, K, dtype=torch.float64, device="cuda")]).view(nb, 64, K)
    s0, s1 = g64.sum(1), (g64 * x64).sum(1)
    a0, a1 = g64.abs().sum(1), (g64 * x64).abs().sum(1)
    assert not torch.isnan(part).any(), "a (block, channel) pair was left unwritten"
    # 64 fp32 additions of 16-bit values: a few ulp of the sum of magnitudes
    assert float(((part[:, :, 0].double() - s0).abs() / (a0 + 1e-30)).max()) < 4e-6
    assert float(((part[:, :, 1].double() - s1).abs() / (a1 + 1e-30)).max()) < 4e-6
    dga, dbe = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
    dx, col = hip.groupnorm_seg_bwd_p64(gm, c["xa"], c["seg_rows"], c["mean"], c["rstd"], c["gamma"], dga, dbe, G, part, want_colsum=True)
    torch.cuda.synchronize()
    # parameter gradients: sums of signed terms over all rows (they cancel: |sum| << sum of magnitudes), taken in another order (fp32
    # chunks of 256 rows there, fp32 64-row blocks + double here) - both against the float64 sums of the same stored values, relative to
    # the sum of magnitudes; and against each other
    seg_of_row = torch.repeat_interleave(torch.arange(len(c["seg_rows"]), device="cuda"), torch.tensor(c["seg_rows"], device="cuda"))
    m_row = c["mean"].double()[seg_of_row].repeat_interleave(8, dim=1)
    r_row = c["rstd"].double()[seg_of_row].repeat_interleave(8, dim=1)
    gd = gm.double()
    xhat = (c["xa"].double() - m_row) * r_row
    for got, ref, t64, name in ((dga, dga_ref, gd * xhat, "dgamma"), (dbe, dbe_ref, gd, "dbeta")):
        exact, mag = t64.sum(0), t64.abs().sum(0)
        dev = float(((got.double() - exact).abs() / mag).max())
        dev_ref = float(((ref.double() - exact).abs() / mag).max())
        assert dev < 2e-6 and dev <= max(dev_ref, 5e-7), (name, dev, dev_ref)
        assert float((got - ref).abs().max() / mag.max()) < 4e-6, name
    # dx = rstd * (g * gamma - (s2 + xhat * s1) / cnt): the same expression per element with s1 / s2 equal to ~1e-6 relative - a 16-bit
    # result may land on the neighbouring value (never further) on a small fraction of the elements; where the two terms cancel (masked
    # elements: g = 0, s2 ~ -xhat s1) the difference is that of the terms, 1e-5 of the tensor's range at most
    af, bf = dx.float(), dx_ref.float()
    d = (af - bf).abs()
    ulp = torch.maximum(af.abs(), bf.abs()) * (2.0 ** -7 if dx.dtype == torch.bfloat16 else 2.0 ** -10)
    atol = 1e-5 * float(bf.abs().max())
    worst = float(((d - atol).clamp(min=0) / (ulp + 1e-30)).max())
    assert worst <= 1.01 and float((d > 0).float().mean()) < 2e-2, (worst, float((d > 0).float().mean()), float(d.max()) / float(bf.abs().max()))
    assert float((col - col_ref).abs().max() / col_ref.abs().max()) < 2e-3       # per-chunk column sums of dx as stored (follow dx)


def test_gnb_entry_rejects_what_it_cannot_take(h16):
    from ubteacher import hip
    c = _chain(LEVELS_SMALL, 1, 2, seed=9)
    part = hip.gnb_part_buffer(c["P"], c["K"], "cuda")
    H = hip._iarr([h for h, _ in LEVELS_SMALL]); W = hip._iarr([w for _, w in LEVELS_SMALL])
    import ctypes
    args = lambda bits, gx, pp, C=256: ("utv2_conv2d_ml_fwd_bf16_gnb", hip._p(c["gout"]), c["K"], hip._p(c["wt"]), hip._p(torch.empty_like(c["gout"])),
                                        len(LEVELS_SMALL), ctypes.cast(H, hip.c_p), ctypes.cast(W, hip.c_p), 1, C, c["K"], 3, 3, 1, 2, None,
                                        bits, gx, pp, hip._stream())
    for bad in (args(None, hip._p(c["xa"]), hip._p(part)), args(hip._p(c["bits"]), None, hip._p(part)), args(hip._p(c["bits"]), hip._p(c["xa"]), None)):
        with pytest.raises(RuntimeError):
            hip.call(*bad)


def test_fcos_step_with_the_fused_groupnorm_backward(monkeypatch):
    """a whole FCOS semi-supervised AMP step with UTV2_GN_BWD_FUSE=1 against the default: the forward is untouched (losses bit-identical);
    three of the four tower GroupNorms per student pass take the fused backward; the student after one SGD step agrees to the rounding of
    16-bit gradients whose GroupNorm sums were taken in another order"""
    from tests.utv2_testutil import FixedLoader, cpu_state, make_batch, small_fcos_cfg, tune_state_for_pseudo_labels
    H, W = 96, 128
    from ubteacher import ops
    from ubteacher.engine import UBTeacherTrainer
    monkeypatch.setenv("UTV2_PRECISION", "fp16")
    res = {}
    try:
        for fuse in ("1", "0"):
            monkeypatch.setenv("UTV2_GN_BWD_FUSE", fuse)
            torch.manual_seed(0)
            cfg = small_fcos_cfg()
            cfg.SOLVER.AMP.ENABLED = True
            prod, orac = make_batch(12, 2, 2, H, W, "cuda")
            tr = UBTeacherTrainer(cfg, data_loader=FixedLoader(prod))
            sd_s = tune_state_for_pseudo_labels(cpu_state(tr.model), [d["image"] for d in orac[3]])
            tr.model.load_state_dict(sd_s); tr.model_teacher.load_state_dict(sd_s)
            tr.iter = 1
            tr.optimizer.param_groups[0]["lr"] = 1e-3
            before = tr.model.store.flat.detach().clone()
            n0 = ops.GNB_STATS["fused"]
            tr.run_step_full_semisup()
            tr.flush_metrics()
            torch.cuda.synchronize()
            res[fuse] = dict(after=tr.model.store.flat.detach().clone(), before=before, fused=ops.GNB_STATS["fused"] - n0,
                             metrics=dict(tr._last_metrics))
    finally:
        ops.set_precision("fp32")
    assert res["0"]["fused"] == 0 and res["1"]["fused"] >= 3 and res["1"]["fused"] % 3 == 0, (res["0"]["fused"], res["1"]["fused"])
    assert torch.equal(res["0"]["before"], res["1"]["before"])
    for k, v in res["0"]["metrics"].items():
        if k.startswith("loss"):
            assert res["1"]["metrics"][k] == v, (k, res["1"]["metrics"][k], v)
    upd0, upd1 = res["0"]["after"] - res["0"]["before"], res["1"]["after"] - res["1"]["before"]
    assert float(upd0.abs().max()) > 0
    dev = float((upd1 - upd0).abs().max() / upd0.abs().max())
    assert dev < 2e-3, dev
