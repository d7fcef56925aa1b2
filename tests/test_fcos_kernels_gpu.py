"""HIP FCOS kernels (targets, focal, fused location losses, decode, NMS, EMA, SGD, GroupNorm ...)
vs the golden vectors generated from the reference and vs the CPU oracle, through the C-ABI.
Tolerances: integer / index outputs exact; fp32 losses rtol 2e-5; gradients rtol 1e-4 (different
summation order + device expf/logf)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import utv2_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = b.detach().cpu().numpy() if torch.is_tensor(b) else np.asarray(b)
    a, b = a.astype(np.float64), b.astype(np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.allclose(a, b, rtol=rtol, atol=atol), (float(np.abs(a - b).max()), float(np.abs(b).max()))


@pytest.fixture(scope="module")
def fc():
    return dict(np.load(os.path.join(G, "fcos_outputs.npz")))


def fcos_cfg():
    from ubteacher import add_ubteacher_config
    from ubteacher.d2 import get_cfg
    cfg = get_cfg()
    add_ubteacher_config(cfg)
    f = cfg.MODEL.FCOS
    f.CENTER_SAMPLE = False; f.REG_DISCRETE = True; f.KL_LOSS = True; f.KLLOSS_WEIGHT = 0.05; f.KL_LOSS_TYPE = "nlloss"
    f.YIELD_PROPOSAL = True
    cfg.SEMISUPNET.CONSIST_REG_LOSS = "ts_locvar_better_nms_nll_l1"
    return cfg


def build_head_out(fc, requires_grad=False):
    """golden NCHW head tensors -> the product's level-first [P, 80] buffers."""
    from ubteacher.ops import LevelMeta
    N = int(fc["N"])
    level_hw = [tuple(fc["logits%d" % l].shape[2:]) for l in range(5)]
    meta = LevelMeta(N, level_hw)
    logits_all = torch.zeros((meta.P, 80), device=DEV)
    box_all = torch.zeros((meta.P, 80), device=DEV)
    for l, (r0, r1) in enumerate(meta.rows):
        logits_all[r0:r1] = T(fc["logits%d" % l]).permute(0, 2, 3, 1).reshape(-1, 80).to(DEV)
        box_all[r0:r1, :68] = T(fc["reg%d" % l]).permute(0, 2, 3, 1).reshape(-1, 68).to(DEV)
        box_all[r0:r1, 68:72] = T(fc["std%d" % l]).permute(0, 2, 3, 1).reshape(-1, 4).to(DEV)
        box_all[r0:r1, 72] = T(fc["ctr%d" % l]).permute(0, 2, 3, 1).reshape(-1).to(DEV)
    if requires_grad:
        logits_all.requires_grad_(True)
        box_all.requires_grad_(True)
    return {"logits": logits_all, "box": box_all, "meta": meta}, level_hw


def padded_gt(fc, prefix, N):
    from ubteacher.d2.structures import Boxes, Instances
    from ubteacher.modeling.fcos import PaddedBoxes
    insts = []
    for i in range(N):
        x = Instances((int(fc["H"]), int(fc["W"])))
        x.gt_boxes = Boxes(T(fc["%s%d_boxes" % (prefix, i)]).float().reshape(-1, 4))
        x.gt_classes = T(fc["%s%d_classes" % (prefix, i)]).long()
        if "%s%d_std" % (prefix, i) in fc:
            x.reg_pred_std = T(fc["%s%d_std" % (prefix, i)]).float().reshape(-1, 4)
            x.scores = T(fc["%s%d_scores" % (prefix, i)])
        insts.append(x)
    return PaddedBoxes.from_instances(insts, DEV)


def level_grads(fc, case, head_out):
    """compare d/d(head outputs) with the golden NCHW grads"""
    meta = head_out["meta"]
    for l in range(5):
        gl = meta.level_view(head_out["logits"].grad, l)
        close(gl.permute(0, 3, 1, 2), fc["%s_glogits%d" % (case, l)], rtol=1e-4, atol=2e-7)
        gb = meta.level_view(head_out["box"].grad, l)
        close(gb[..., :68].permute(0, 3, 1, 2), fc["%s_greg%d" % (case, l)], rtol=1e-4, atol=2e-7)
        close(gb[..., 68:72].permute(0, 3, 1, 2), fc["%s_gstd%d" % (case, l)], rtol=1e-4, atol=2e-7)
        close(gb[..., 72:73].permute(0, 3, 1, 2), fc["%s_gctr%d" % (case, l)], rtol=1e-4, atol=2e-7)
        assert float(gb[..., 73:].abs().max()) == 0.0


@pytest.mark.parametrize("case", ["sup", "supempty"])
def test_supervised_losses(fc, case):
    from ubteacher.modeling.fcos import FCOSOutputs
    outm = FCOSOutputs(fcos_cfg())
    head_out, level_hw = build_head_out(fc, True)
    gt = padded_gt(fc, case + "_gt", int(fc["N"]))
    extras, losses = outm.losses(head_out, level_hw, gt)
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(losses[k], fc["%s_%s" % (case, k)], rtol=2e-5)
    # targets: exact labels (dropped locations of empty images compare as "not kept")
    lab = extras["labels"].cpu().numpy()
    N = int(fc["N"])
    r = 0
    for l, (h, w) in enumerate(level_hw):
        gl = fc["%s_labels%d" % (case, l)]
        mine = lab[r:r + N * h * w]
        keep = mine >= 0
        assert np.array_equal(mine[keep], gl[keep])
        if case == "supempty":
            assert (~keep).sum() == h * w  # exactly the empty image's locations
        close(extras["reg_targets"][r:r + N * h * w][torch.from_numpy(keep).to(DEV)], fc["%s_regt%d" % (case, l)][keep])
        r += N * h * w
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    level_grads(fc, case, head_out)


def test_pseudo_losses(fc):
    from ubteacher.modeling.fcos import FCOSOutputs
    outm = FCOSOutputs(fcos_cfg())
    head_out, level_hw = build_head_out(fc, True)
    N = int(fc["N"])
    gt = {"cls": padded_gt(fc, "pcls_gt", N), "reg": padded_gt(fc, "preg_gt", N)}
    extras, losses = outm.pseudo_losses(head_out, level_hw, gt)
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr", "teacher_better_student"):
        close(losses[k], fc["pseudo_%s" % k], rtol=2e-5)
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    level_grads(fc, "pseudo", head_out)


def test_joint_losses_fused_tail_equals_op_chain(fc):
    """FCOSOutputs.joint_losses (range form of the target kernel: no padded ground-truth copies, no activity mask; raw kernel sums ->
    normalised losses, the trainer's weighting and the backward coefficients in ONE utv2_fcos_loss_combine launch) against losses() +
    pseudo_losses() on padded ground truth with activity masks + the trainer's per-key arithmetic as ATen ops: every loss, the weighted
    total and the gradients of both head buffers.  The first image(s) of the batch are the labeled branch, the rest the pseudo branch."""
    from ubteacher.modeling.fcos import FCOSOutputs, PaddedBoxes
    outm = FCOSOutputs(fcos_cfg())
    N = int(fc["N"])
    nl = max(N // 2, 1)

    def rows(pb, a, b):
        return PaddedBoxes(list(pb.image_sizes)[a:b], **{k: v[a:b].contiguous() for k, v in pb.f.items() if k != "count"})

    gtl_full = padded_gt(fc, "sup_gt", N)
    gtu_full = {"cls": padded_gt(fc, "pcls_gt", N), "reg": padded_gt(fc, "preg_gt", N)}
    gtl = rows(gtl_full, 0, nl)                                  # what the fused pass hands over: each branch's own images only
    gtu = {k: rows(v, nl, N) for k, v in gtu_full.items()}
    act = torch.zeros(N, dtype=torch.uint8, device=DEV); act[:nl] = 1
    lu, lr = 4.0, 1.5
    lw = {"loss_fcos_cls": (1.0, lu + 1.0), "loss_fcos_ctr": (1.0, lu + 1.0), "loss_fcos_loc": (1.0, lr + 1.0),
          "loss_fcos_cls_pseudo": (lu, lu + 1.0), "loss_fcos_ctr_pseudo": (lu, lu + 1.0), "loss_fcos_loc_pseudo": (lr, lr + 1.0)}
    ha, level_hw = build_head_out(fc, True)
    ls, lun, total = outm.joint_losses(ha, level_hw, gtl, gtu, nl, N, lw)
    total.backward()
    hb, _ = build_head_out(fc, True)
    _, rs = outm.losses(hb, level_hw, gtl.pad_images(0, N - nl), active=act)
    _, ru = outm.pseudo_losses(hb, level_hw, {k: v.pad_images(nl, 0) for k, v in gtu.items()}, active=(1 - act))
    ref = (rs["loss_fcos_cls"] / (lu + 1.0) + rs["loss_fcos_loc"] / (lr + 1.0) + rs["loss_fcos_ctr"] / (lu + 1.0)
           + ru["loss_fcos_cls"] * lu / (lu + 1.0) + ru["loss_fcos_ctr"] * lu / (lu + 1.0) + ru["loss_fcos_loc"] * lr / (lr + 1.0))
    ref.backward()
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(ls[k], rs[k].detach(), rtol=1e-6)
        close(lun[k], ru[k].detach(), rtol=1e-6)
        assert not ls[k].requires_grad
    close(lun["teacher_better_student"], ru["teacher_better_student"], rtol=0)
    close(total.detach(), ref.detach(), rtol=1e-6)
    for key in ("logits", "box"):
        ga, gb = ha[key].grad, hb[key].grad
        assert float(gb.abs().max()) > 0
        assert float((ga - gb).abs().max()) <= 1e-6 * float(gb.abs().max())


@pytest.mark.parametrize("method", ["cls", "cls_n_ctr", "cls_n_loc"])
def test_decode_nms(fc, method):
    from ubteacher.modeling.fcos import FCOSOutputs
    outm = FCOSOutputs(fcos_cfg())
    outm.training = False
    head_out, level_hw = build_head_out(fc)
    N, H, W = int(fc["N"]), int(fc["H"]), int(fc["W"])
    det = outm.predict_proposals(head_out, level_hw, [(H, W)] * N, method)
    insts = det.to_instances()
    for i, r in enumerate(insts):
        gcls = fc["det_%s_%d_classes" % (method, i)]
        assert len(r) == len(gcls)
        assert np.array_equal(r.pred_classes.cpu().numpy(), gcls)  # same detections, same (descending score) order
        close(r.pred_boxes.tensor, fc["det_%s_%d_boxes" % (method, i)], rtol=1e-5, atol=2e-4)
        close(r.scores, fc["det_%s_%d_scores" % (method, i)], rtol=2e-5)
        close(r.centerness, fc["det_%s_%d_ctr" % (method, i)], rtol=2e-5)
        close(r.cls_confid, fc["det_%s_%d_conf" % (method, i)], rtol=2e-5)
        close(r.reg_pred_std, fc["det_%s_%d_std" % (method, i)])
    th = det.threshold(0.3).to_instances(as_gt=True)
    for i, r in enumerate(th):
        close(r.gt_boxes.tensor, fc["thr_%s_%d_boxes" % (method, i)], rtol=1e-5, atol=2e-4)
    # SEMISUPNET.PSEUDO_BBOX_SAMPLE "thresholding_cls_ctr" through the product's PseudoGenerator (reference
    # pseudo_generator.py:49-52,107-131): kept set exact against the reference's own output
    from ubteacher.modeling.pseudo_generator import PseudoGenerator
    thr = tuple(float(v) for v in fc["thrcc_thresholds"])
    out, num = PseudoGenerator(fcos_cfg()).process_pseudo_label(det, thr, "roih", "thresholding_cls_ctr")
    want_num = np.mean([float(fc["thrcc_%s_%d_num" % (method, i)]) for i in range(N)])
    assert abs(float(num) - want_num) < 1e-6
    for i, r in enumerate(out.to_instances(as_gt=True)):
        assert np.array_equal(r.gt_classes.cpu().numpy(), fc["thrcc_%s_%d_classes" % (method, i)])
        close(r.gt_boxes.tensor, fc["thrcc_%s_%d_boxes" % (method, i)], rtol=1e-5, atol=2e-4)
        close(r.scores, fc["thrcc_%s_%d_scores" % (method, i)], rtol=2e-5)
        close(r.centerness, fc["thrcc_%s_%d_ctr" % (method, i)], rtol=2e-5)
        close(r.cls_confid, fc["thrcc_%s_%d_conf" % (method, i)], rtol=2e-5)
        close(r.reg_pred_std, fc["thrcc_%s_%d_std" % (method, i)])


def test_nms_bit_exact_vs_oracle():
    """NMS index selection must be bit-exact (north star): random overlapping boxes incl. score ties."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(5)
    for (N, M, thr, aware) in [(2, 700, 0.6, True), (1, 3000, 0.7, False), (3, 130, 0.5, True), (1, 64, 0.3, True)]:
        ctr = torch.rand(N, M, 2, generator=g) * 300
        wh = torch.rand(N, M, 2, generator=g) * 80 + 4
        boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=2)
        scores = torch.rand(N, M, generator=g)
        scores[:, ::7] = 0.5  # ties
        cls = torch.randint(0, 5, (N, M), generator=g, dtype=torch.int32)
        valid = (torch.rand(N, M, generator=g) > 0.1).to(torch.uint8)
        keep, cnt = hip.nms_batched(boxes.to(DEV), scores.to(DEV), cls.to(DEV), valid.to(DEV), thr, class_aware=aware,
                                    post_topk=-1, max_out=M)
        keep, cnt = keep.cpu(), cnt.cpu()
        for n in range(N):
            vi = valid[n].bool().nonzero().squeeze(1)
            if aware:
                ref = O.batched_nms(boxes[n][vi], scores[n][vi], cls[n][vi].long(), thr)
            else:
                ref = O.nms(boxes[n][vi], scores[n][vi], thr)
            ref = vi[ref]
            assert int(cnt[n]) == len(ref)
            assert torch.equal(keep[n, : len(ref)].long(), ref)


def test_ema_bit_exact():
    from ubteacher import hip
    d = dict(np.load(os.path.join(G, "ema.npz")))
    for keep in (0.0, 0.9996, 0.9999):
        tag = str(keep).replace(".", "p")
        keys = sorted(k[len(tag) + 3:] for k in d if k.startswith(tag + "_s_"))
        s = torch.cat([T(d["%s_s_%s" % (tag, k)]) for k in keys]).to(DEV)
        t = torch.cat([T(d["%s_t_%s" % (tag, k)]) for k in keys]).to(DEV)
        want = np.concatenate([d["%s_out_%s" % (tag, k)] for k in keys])
        hip.ema_axpby(t, s, keep)
        assert np.array_equal(t.cpu().numpy(), want)  # bit exact vs the reference's own output


def test_sgd_matches_torch():
    from ubteacher import hip
    g = torch.Generator().manual_seed(2)
    p = torch.randn(10007, generator=g); gr = torch.randn(10007, generator=g)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.01, momentum=0.9, weight_decay=1e-4)
    pd, gd, md = p.to(DEV), gr.to(DEV), torch.zeros(10007, device=DEV)
    for it in range(3):
        pr.grad = gr.clone() * (it + 1)
        opt.step()
        gd.copy_((gr * (it + 1)).to(DEV))
        hip.sgd_momentum(pd, gd, md, 0.01, 0.9, 1e-4, 1.0, zero_grad=True)
    close(pd, pr.detach(), rtol=1e-6, atol=1e-7)
    assert float(gd.abs().max()) == 0.0


def test_groupnorm_relu_fwd_bwd():
    from ubteacher import hip
    g = torch.Generator().manual_seed(4)
    N, H, W, C = 2, 13, 21, 256
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    ga = torch.rand(C, generator=g) + 0.5; be = torch.randn(C, generator=g) * 0.1
    xr = x.clone().requires_grad_(True); gar = ga.clone().requires_grad_(True); ber = be.clone().requires_grad_(True)
    y = F.relu(F.group_norm(xr, 32, gar, ber, 1e-5))
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xh = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    yh, mean, rstd = hip.groupnorm_relu_fwd(xh, ga.to(DEV), be.to(DEV))
    close(yh.permute(0, 3, 1, 2), y.detach(), rtol=1e-4, atol=1e-5)
    dga = torch.zeros(C, device=DEV); dbe = torch.zeros(C, device=DEV)
    dx = hip.groupnorm_relu_bwd(dy.permute(0, 2, 3, 1).contiguous().to(DEV), yh, xh, mean, rstd, ga.to(DEV), dga, dbe)
    close(dx.permute(0, 3, 1, 2), xr.grad, rtol=1e-3, atol=1e-5)
    close(dga, gar.grad, rtol=1e-4, atol=1e-4)
    close(dbe, ber.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_groupnorm_bwd_column_sums_of_dx(dtype):
    """utv2_groupnorm_relu_seg_bwd_colsum: the per-chunk column sums written next to dx add up to the column sums of dx AS STORED (the
    bias gradient of the conv in front of the GroupNorm: what the separate colsum pass over its dY computed), dx / dgamma / dbeta are those
    of the plain entry; ragged (image, level) segments, chunks that end mid-way."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(8)
    C, seg_rows = 256, [700, 256, 1, 513, 90]
    rows = sum(seg_rows)
    x = (torch.randn(rows, C, generator=g) * 2 + 0.3).to(dtype).to(DEV)
    dy = torch.randn(rows, C, generator=g).to(dtype).to(DEV)
    ga = (torch.rand(C, generator=g) + 0.5).to(DEV); be = (torch.randn(C, generator=g) * 0.1).to(DEV)
    y, mean, rstd = hip.groupnorm_relu_seg_fwd(x, seg_rows, ga, be)
    d0 = [torch.zeros(C, device=DEV) for _ in range(2)]
    d1 = [torch.zeros(C, device=DEV) for _ in range(2)]
    dx0 = hip.groupnorm_relu_seg_bwd(dy, y, x, seg_rows, mean, rstd, ga, d0[0], d0[1], beta=be)
    dx1, part = hip.groupnorm_relu_seg_bwd(dy, y, x, seg_rows, mean, rstd, ga, d1[0], d1[1], beta=be, want_colsum=True)
    assert torch.equal(dx0, dx1) and torch.equal(d0[0], d1[0]) and torch.equal(d0[1], d1[1])
    assert part.shape == (sum((r + 255) // 256 for r in seg_rows), C)
    ref = dx1.double().sum(0)
    got = part.double().sum(0)
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    db = torch.full((C,), 0.5, device=DEV)
    hip.colsum_partials(part, db, accumulate=True)
    assert float((db.double() - 0.5 - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_pool_upsample_preprocess_fold():
    from ubteacher import hip
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 64, 18, 26, generator=g)
    y = hip.maxpool3x3s2(x.permute(0, 2, 3, 1).contiguous().to(DEV))
    close(y.permute(0, 3, 1, 2), F.max_pool2d(x, 3, 2, 1), rtol=0, atol=0)
    lat = torch.randn(2, 32, 8, 12, generator=g); top = torch.randn(2, 32, 4, 6, generator=g)
    o = hip.upsample2x_add(lat.permute(0, 2, 3, 1).contiguous().to(DEV), top.permute(0, 2, 3, 1).contiguous().to(DEV))
    close(o.permute(0, 3, 1, 2), lat + F.interpolate(top, scale_factor=2.0, mode="nearest"), rtol=0, atol=0)
    gg = torch.randn(2, 8, 12, 32, generator=g)
    dt = hip.downsample2x_sum(gg.to(DEV))
    ref = F.avg_pool2d(gg.permute(0, 3, 1, 2), 2) * 4
    close(dt.permute(0, 3, 1, 2), ref, rtol=1e-6, atol=1e-6)
    ims = [torch.randint(0, 256, (3, 37, 50), generator=g, dtype=torch.uint8), torch.randint(0, 256, (3, 40, 45), generator=g, dtype=torch.uint8)]
    mean, std = [103.53, 116.28, 123.675], [1.0, 57.0, 2.0]
    x4, sizes = hip.preprocess_images([i.to(DEV) for i in ims], mean, std, 32)
    ref, rs = O.preprocess(ims, torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1), 32)
    assert sizes == rs and tuple(x4.shape) == (2, 64, 64, 4)
    close(x4[..., :3].permute(0, 3, 1, 2), ref, rtol=1e-6, atol=1e-6)
    assert float(x4[..., 3].abs().max()) == 0.0
    n = 300
    w, b, m, v = (torch.rand(n, generator=g) + 0.5 for _ in range(4))
    sc, sh = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    hip.frozenbn_fold(w.to(DEV), b.to(DEV), m.to(DEV), v.to(DEV), sc, sh)
    rsc = w * (v + 1e-5).rsqrt()
    close(sc, rsc, rtol=1e-6); close(sh, b - m * rsc, rtol=1e-5, atol=1e-6)


def test_topk_rows_exact():
    """utv2_topk_rows_i64 == torch.topk(sorted=True) on ragged rows of unique keys with -1 holes: heavy ties in the high
    (score) bits, rows shorter than k, an all-empty row."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(11)
    widths = [50000, 7000, 300, 64, 5000]
    k = 1000
    rows = []
    for i, wd in enumerate(widths):
        idx = torch.arange(wd, dtype=torch.int64)
        if i == 0:      # few distinct scores -> long runs of equal high words, ties broken by the low (index) word
            hi = torch.randint(0x3D000000, 0x3D000008, (wd,), generator=g, dtype=torch.int64)
        elif i == 4:    # saturated: every candidate has the same score
            hi = torch.full((wd,), 0x3F800000, dtype=torch.int64)
        else:
            hi = torch.randint(0x3C000000, 0x3F800000, (wd,), generator=g, dtype=torch.int64)
        key = (hi << 32) | (0xFFFFFFFF - idx)
        hole = torch.rand(wd, generator=g) < (0.5 if i != 3 else 2.0)   # row 3: all empty
        key[hole] = -1
        rows.append(key)
    flat = torch.cat(rows).to(DEV)
    offs = torch.tensor([0] + list(torch.tensor(widths).cumsum(0)), dtype=torch.int64, device=DEV)
    got = hip.topk_rows(flat, offs, len(widths), max(widths), k)
    for i, r in enumerate(rows):
        kk = min(k, r.numel())
        ref = torch.topk(r, kk, sorted=True).values
        assert torch.equal(got[i, :kk].cpu(), ref), i
        assert bool((got[i, kk:] == -1).all())
    got2 = hip.topk_rows(flat, offs, len(widths), max(widths), k)
    assert torch.equal(got, got2)


def test_nms_class_parallel_scan_post_topk():
    """class-parallel scan (independent per-class chains on 8 waves) with the post-NMS kthvalue rule and its early exit:
    bit-exact index selection vs the oracle for RPN-like (5 levels, thousands of candidates) and FCOS-like (80 classes) inputs."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(9)
    for (N, M, thr, ncls, post) in [(2, 6000, 0.7, 5, 1000), (1, 4000, 0.6, 80, 100), (2, 900, 0.5, 3, 50), (1, 200, 0.6, 80, 100)]:
        ctr = torch.rand(N, M, 2, generator=g) * 400
        wh = torch.rand(N, M, 2, generator=g) * 90 + 4
        boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=2)
        scores = torch.rand(N, M, generator=g)
        scores[:, ::5] = torch.round(scores[:, ::5] * 20) / 20   # many exact ties, also around the k-th score
        cls = torch.randint(0, ncls, (N, M), generator=g, dtype=torch.int32)
        valid = (torch.rand(N, M, generator=g) > 0.05).to(torch.uint8)
        keep, cnt = hip.nms_batched(boxes.to(DEV), scores.to(DEV), cls.to(DEV), valid.to(DEV), thr, class_aware=True,
                                    post_topk=post, max_out=M)
        keep, cnt = keep.cpu(), cnt.cpu()
        for n in range(N):
            vi = valid[n].bool().nonzero().squeeze(1)
            ref = vi[O.batched_nms(boxes[n][vi], scores[n][vi], cls[n][vi].long(), thr)]
            if len(ref) > post:
                kth = scores[n][ref[post - 1]]
                ref = ref[scores[n][ref] >= kth]
            assert int(cnt[n]) == len(ref), (N, M, ncls, int(cnt[n]), len(ref))
            assert torch.equal(keep[n, : len(ref)].long(), ref)


def test_nms_bucketed_max_out_truncation_and_sparse_class_ids():
    """the bucketed class-aware pipeline (per-bucket IoU matrices and scans, global-order emit) when only the first max_out kept
    candidates are wanted (the RPN: 1000 of ~9000, a bucket stops once it has kept max_out on its own), with class ids far apart
    that collide in a bucket (ids 3, 35, 67 share bucket 3), an empty image and a single-candidate image: bit-exact vs the oracle."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(19)
    for (N, M, thr, ids, max_out) in [(2, 9000, 0.7, [0, 1, 2, 3, 4], 1000), (2, 3000, 0.5, [3, 35, 67, 1000, 64], 300),
                                      (3, 500, 0.6, [7], 64)]:
        ctr = torch.rand(N, M, 2, generator=g) * 500
        wh = torch.rand(N, M, 2, generator=g) * 100 + 4
        boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=2)
        scores = torch.rand(N, M, generator=g)
        scores[:, ::9] = 0.25
        cls = torch.tensor(ids, dtype=torch.int32)[torch.randint(0, len(ids), (N, M), generator=g)]
        valid = (torch.rand(N, M, generator=g) > 0.05).to(torch.uint8)
        if N == 3:
            valid[1] = 0                       # no candidates at all
            valid[2] = 0; valid[2, 17] = 1     # exactly one
        keep, cnt = hip.nms_batched(boxes.to(DEV), scores.to(DEV), cls.to(DEV), valid.to(DEV), thr, class_aware=True, post_topk=-1,
                                    max_out=max_out)
        keep, cnt = keep.cpu(), cnt.cpu()
        for n in range(N):
            vi = valid[n].bool().nonzero().squeeze(1)
            ref = vi[O.batched_nms(boxes[n][vi], scores[n][vi], cls[n][vi].long(), thr)][:max_out] if len(vi) else vi
            assert int(cnt[n]) == len(ref), (N, M, int(cnt[n]), len(ref))
            assert torch.equal(keep[n, : len(ref)].long(), ref)
            assert bool((keep[n, len(ref):] == -1).all())


def test_center_sample_variant_vs_reference_golden():
    """MODEL.FCOS.CENTER_SAMPLE True through the product (target kernel with the radius*stride centre region): labels, regression
    targets, losses and head gradients vs the reference's own outputs (fcos_center_sample.npz); also the first-box-centre quirk."""
    from ubteacher import hip
    from ubteacher.modeling.fcos import FCOSOutputs
    cs = dict(np.load(os.path.join(G, "fcos_center_sample.npz")))
    cfg = fcos_cfg()
    cfg.MODEL.FCOS.CENTER_SAMPLE = True
    cfg.MODEL.FCOS.POS_RADIUS = float(cs["radius"])
    outm = FCOSOutputs(cfg)
    head_out, level_hw = build_head_out(cs, True)
    N = int(cs["N"])
    gt = padded_gt(cs, "gt", N)
    extras, losses = outm.losses(head_out, level_hw, gt)
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(losses[k], cs["loss_%s" % k], rtol=2e-5)
    lab = extras["labels"].cpu().numpy()
    r = 0
    for l, (h, w) in enumerate(level_hw):
        gl = cs["labels%d" % l]
        mine = lab[r:r + N * h * w]
        keep = mine >= 0                       # the empty third image is dropped (-1) in the supervised branch
        assert np.array_equal(mine[keep], gl[keep]) and (~keep).sum() == h * w
        close(extras["reg_targets"][r:r + N * h * w][torch.from_numpy(keep).to(DEV)], cs["regt%d" % l][keep])
        r += N * h * w
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    meta = head_out["meta"]
    for l in range(5):
        close(meta.level_view(head_out["logits"].grad, l).permute(0, 3, 1, 2), cs["glogits%d" % l], rtol=1e-4, atol=2e-7)
        gb = meta.level_view(head_out["box"].grad, l)
        close(gb[..., :68].permute(0, 3, 1, 2), cs["greg%d" % l], rtol=1e-4, atol=2e-7)
        close(gb[..., 72:73].permute(0, 3, 1, 2), cs["gctr%d" % l], rtol=1e-4, atol=2e-7)
    # quirk (`center_x[..., 0].sum() == 0`): a first box centred at x = 0 switches every positive off - vs the oracle
    boxes = torch.tensor([[[-10.0, 20.0, 10.0, 60.0], [30.0, 30.0, 90.0, 100.0]]], device=DEV)
    classes = torch.tensor([[3, 5]], dtype=torch.int32, device=DEV)
    valid = torch.ones((1, 2), dtype=torch.uint8, device=DEV)
    soi = [[-1, 64], [64, 128], [128, 256], [256, 512], [512, 1e8]]
    labels, _, _, _ = hip.fcos_targets(level_hw, [8, 16, 32, 64, 128], soi, boxes, classes, valid, None, 80, 0, center_radius=1.5)
    assert int((labels < 80).sum()) == 0
    locs = [O.compute_locations(h, w, s) for (h, w), s in zip(level_hw, (8, 16, 32, 64, 128))]
    tg = O.fcos_targets(O.FCOSCfg(center_sample=True), locs, [dict(boxes=boxes[0].cpu(), classes=classes[0].long().cpu())])
    assert sum(int((x < 80).sum()) for x in tg["labels"]) == 0


def test_ignore_near_variant_vs_reference_golden():
    """SEMISUPNET.PSEUDO_CLS_IGNORE_NEAR through the product (fcos_outputs.py:841-851 in the target kernel): dropped locations, supervised
    losses and head gradients vs the reference's own (fcos_center_sample.npz, ign_*); the pseudo branch takes the switch and, like the
    reference (which never reads keep_locations there), returns the same losses."""
    from ubteacher.modeling.fcos import FCOSOutputs
    cs = dict(np.load(os.path.join(G, "fcos_center_sample.npz")))
    cfg = fcos_cfg()
    cfg.MODEL.FCOS.CENTER_SAMPLE = True
    cfg.MODEL.FCOS.POS_RADIUS = float(cs["radius"])
    outm = FCOSOutputs(cfg)
    head_out, level_hw = build_head_out(cs, True)
    N = int(cs["N"])
    extras, losses = outm.losses(head_out, level_hw, padded_gt(cs, "gt", N), ignore_near=True)
    lab = extras["labels"].cpu().numpy()
    r = 0
    for l, (h, w) in enumerate(level_hw):
        assert np.array_equal(lab[r:r + N * h * w] >= 0, cs["ign_keep%d" % l].astype(bool))
        r += N * h * w
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(losses[k], cs["ign_loss_%s" % k], rtol=2e-5)
    (losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]).backward()
    meta = head_out["meta"]
    for l in range(5):
        close(meta.level_view(head_out["logits"].grad, l).permute(0, 3, 1, 2), cs["ign_glogits%d" % l], rtol=1e-4, atol=2e-7)
        gb = meta.level_view(head_out["box"].grad, l)
        close(gb[..., :68].permute(0, 3, 1, 2), cs["ign_greg%d" % l], rtol=1e-4, atol=2e-7)
        close(gb[..., 72:73].permute(0, 3, 1, 2), cs["ign_gctr%d" % l], rtol=1e-4, atol=2e-7)
    head_out, level_hw = build_head_out(cs, False)
    pg = padded_gt(cs, "ign_pgt", N)
    _, pl = outm.pseudo_losses(head_out, level_hw, {"cls": pg, "reg": pg})
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(pl[k], cs["ign_pseudo_%s" % k], rtol=2e-5)


LOSS_VARIANTS = {
    "klloss": dict(KL_LOSS_TYPE="klloss"),
    "nokl": dict(KL_LOSS=False),
    "iouq": dict(QUALITY_EST="iou"),
    "lociou": dict(LOC_LOSS_TYPE="iou"),
    "loclinear": dict(LOC_LOSS_TYPE="linear_iou"),
    "klloss_iouq_linear": dict(KL_LOSS_TYPE="klloss", QUALITY_EST="iou", LOC_LOSS_TYPE="linear_iou"),
    "klloss_sum": dict(KL_LOSS_TYPE="klloss", LOC_FUN_ALL="sum"),                       # MODEL.FCOS.LOC_FUN_ALL (kl_loss.py:48-64)
    "klloss_wsum": dict(KL_LOSS_TYPE="klloss", LOC_FUN_ALL="weight_ctr_sum"),
    "klloss_wmean_iouq": dict(KL_LOSS_TYPE="klloss", LOC_FUN_ALL="weight_ctr_mean", QUALITY_EST="iou"),
}


def _variant_grads(lv, case, head_out):
    meta = head_out["meta"]
    for l in range(5):
        gb = meta.level_view(head_out["box"].grad, l)
        close(gb[..., :68].permute(0, 3, 1, 2), lv["%s_greg%d" % (case, l)], rtol=1e-4, atol=2e-7)
        close(gb[..., 68:72].permute(0, 3, 1, 2), lv["%s_gstd%d" % (case, l)], rtol=1e-4, atol=2e-7)
        close(gb[..., 72:73].permute(0, 3, 1, 2), lv["%s_gctr%d" % (case, l)], rtol=1e-4, atol=2e-7)


@pytest.mark.parametrize("case", sorted(LOSS_VARIANTS))
def test_supervised_loss_variants_vs_reference_golden(case):
    """KL_LOSS_TYPE "klloss" / KL_LOSS False / QUALITY_EST "iou" / LOC_LOSS_TYPE "iou", "linear_iou" through the product (flags of
    the fused positive-location kernels) vs the reference's own FCOSOutputs.losses under that config (fcos_loss_variants.npz)."""
    from ubteacher.modeling.fcos import FCOSOutputs
    lv = dict(np.load(os.path.join(G, "fcos_loss_variants.npz")))
    cfg = fcos_cfg()
    for k, v in LOSS_VARIANTS[case].items():
        setattr(cfg.MODEL.FCOS, k, v)
    outm = FCOSOutputs(cfg)
    head_out, level_hw = build_head_out(lv, True)
    extras, losses = outm.losses(head_out, level_hw, padded_gt(lv, "gt", int(lv["N"])))
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(losses[k], lv["%s_%s" % (case, k)], rtol=2e-5)
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    _variant_grads(lv, case, head_out)


@pytest.mark.parametrize("case,kl_type,fun", [("pseudo_nll", "nlloss", "mean"), ("pseudo_kl", "klloss", "mean"),
                                              ("pseudo_kl_wmean", "klloss", "weight_ctr_mean")])
def test_pseudo_regression_kl_term_vs_reference_golden(case, kl_type, fun):
    """CONSIST_REG_LOSS other than the TS-better selection: loss_fcos_loc = KLLOSS_WEIGHT * (NLL | KL) on the regression pseudo set."""
    from ubteacher.modeling.fcos import FCOSOutputs
    lv = dict(np.load(os.path.join(G, "fcos_loss_variants.npz")))
    cfg = fcos_cfg()
    cfg.SEMISUPNET.CONSIST_REG_LOSS = "mse_loss_all_raw"
    cfg.MODEL.FCOS.KL_LOSS_TYPE = kl_type
    cfg.MODEL.FCOS.LOC_FUN_ALL = fun
    outm = FCOSOutputs(cfg)
    head_out, level_hw = build_head_out(lv, True)
    N = int(lv["N"])
    gt = {"cls": padded_gt(lv, "pcls_gt", N), "reg": padded_gt(lv, "preg_gt", N)}
    extras, losses = outm.pseudo_losses(head_out, level_hw, gt)
    assert "teacher_better_student" not in losses
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(losses[k], lv["%s_%s" % (case, k)], rtol=2e-5)
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    _variant_grads(lv, case, head_out)
    cfg.MODEL.FCOS.KL_LOSS = False
    with pytest.raises(ValueError):  # fcos_outputs.py:587-588
        FCOSOutputs(cfg).pseudo_losses(build_head_out(lv)[0], level_hw, gt)


def test_two_criteria_in_one_launch_set_equal_two_calls(fc):
    """predict_proposals with a tuple of ranking criteria (one set of rank-key / top-k / decode / NMS launches over (criterion, image)
    pairs) returns exactly what the separate calls return - every field, bit for bit."""
    from ubteacher.modeling.fcos import FCOSOutputs
    outm = FCOSOutputs(fcos_cfg())
    outm.training = False
    head_out, level_hw = build_head_out(fc)
    N, H, W = int(fc["N"]), int(fc["H"]), int(fc["W"])
    sizes = [(H, W)] * N
    both = outm.predict_proposals(head_out, level_hw, sizes, ("cls_n_ctr", "cls_n_loc", "cls"))
    assert isinstance(both, list) and len(both) == 3
    for res, m in zip(both, ("cls_n_ctr", "cls_n_loc", "cls")):
        one = outm.predict_proposals(head_out, level_hw, sizes, m)
        assert set(res.f) == set(one.f)
        for k in one.f:
            assert torch.equal(res[k], one[k]), (m, k)
    with pytest.raises(ValueError):
        outm.predict_proposals(head_out, level_hw, sizes, ("cls", "nope"))


def test_scale_cols_ml_forward_backward_vs_torch():
    """the per-level Scale layers (reference fcos/fcos.py:22-41,338-364: bbox_pred * scale_l) of all levels in one launch forward / two
    backward, against plain torch arithmetic: y[:, :68] *= s_l; dL/dx = g * s_l; dL/ds_l = sum(g * x) (= sum(g * y) / s_l)"""
    from ubteacher import hip
    g = torch.Generator().manual_seed(21)
    rows = [(0, 4033), (4033, 5050), (5050, 5301), (5301, 5367), (5367, 5387)]
    P, BS, nc = rows[-1][1], 80, 68
    x = torch.randn(P, BS, generator=g).to(DEV)
    gy = torch.randn(P, BS, generator=g).to(DEV)
    scales = [torch.tensor([0.5 + 0.37 * l], device=DEV) for l in range(5)]
    y = x.clone()
    hip.scale_cols_ml(y, rows, nc, scales)
    want = x.clone()
    for (a, b), s in zip(rows, scales):
        want[a:b, :nc] *= s
    assert torch.equal(y, want)
    sgr = [torch.full((1,), 0.25, device=DEV) for _ in range(5)]          # accumulated into
    gin = gy.clone()
    hip.scale_cols_bwd_ml(gin, y, rows, nc, scales, sgr)
    for l, ((a, b), s) in enumerate(zip(rows, scales)):
        assert torch.equal(gin[a:b, :nc], gy[a:b, :nc] * s) and torch.equal(gin[a:b, nc:], gy[a:b, nc:])
        ds = (gy[a:b, :nc].double() * x[a:b, :nc].double()).sum()
        assert abs(float(sgr[l]) - 0.25 - float(ds)) <= 2e-5 * float((gy[a:b, :nc].double() * x[a:b, :nc].double()).abs().sum()) + 1e-6
    # bit-deterministic (fixed-order partial sums)
    sgr2 = [torch.full((1,), 0.25, device=DEV) for _ in range(5)]
    gin2 = gy.clone()
    hip.scale_cols_bwd_ml(gin2, y, rows, nc, scales, sgr2)
    assert all(torch.equal(a, b) for a, b in zip(sgr, sgr2))


def test_scale_cols_bwd_pad16_equals_scale_then_pad():
    """utv2_scale_cols_bwd_ml_pad16 (Scale backward + conversion + zero padding in one out-of-place pass) == utv2_scale_cols_bwd_ml on
    a clone followed by utv2_pad_cols_bf16, bit for bit (values and the Scale gradients), and leaves the incoming gradient untouched"""
    from ubteacher import hip
    g = torch.Generator().manual_seed(22)
    rows = [(0, 4033), (4033, 5050), (5050, 5301), (5301, 5367), (5367, 5387)]
    P, BS, nc, cpad = rows[-1][1], 80, 68, 96
    y = torch.randn(P, BS, generator=g).to(DEV)
    gy = torch.randn(P, BS, generator=g).to(DEV)
    scales = [torch.tensor([0.5 + 0.37 * l], device=DEV) for l in range(5)]
    sg_a = [torch.zeros(1, device=DEV) for _ in range(5)]
    sg_b = [torch.zeros(1, device=DEV) for _ in range(5)]
    ref = gy.clone()
    hip.scale_cols_bwd_ml(ref, y, rows, nc, scales, sg_a)
    want = hip.pad_cols_bf16(ref, cpad)
    keep = gy.clone()
    got = hip.scale_cols_bwd_ml_pad16(gy, y, rows, nc, scales, sg_b, cpad)
    assert torch.equal(gy, keep)
    assert got.dtype == hip.h16_dtype() and tuple(got.shape) == (P, cpad) and torch.equal(got, want)
    assert float(got[:, BS:].abs().max()) == 0.0
    assert all(abs(float(a) - float(b)) <= 1e-5 * max(abs(float(a)), 1.0) for a, b in zip(sg_a, sg_b))   # another partition of the same sum
