"""HIP Faster-RCNN pieces (RoIAlign fwd/bwd, matcher, softmax focal, predictor losses/inference, RPN
pseudo losses, proposal sampling) vs golden vectors from the reference and vs the CPU oracle."""
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
def rc():
    return dict(np.load(os.path.join(G, "rcnn.npz")))


def rcnn_cfg():
    from ubteacher.presets import get_config
    return get_config("rcnn", 1, ["MODEL.DEVICE", DEV])


def test_roi_align_fwd_bwd_vs_oracle():
    from ubteacher import ops
    g = torch.Generator().manual_seed(3)
    N, C = 2, 64
    shapes = [(24, 32), (12, 16), (6, 8), (3, 4)]
    feats = [torch.randn(N, C, h, w, generator=g) for h, w in shapes]
    R = 37
    xy = torch.rand(R, 2, generator=g) * torch.tensor([90.0, 60.0])
    wh = torch.exp(torch.rand(R, 2, generator=g) * 4.0 + 1.0)
    rois = torch.cat([xy, xy + wh], 1)
    rois[0] = torch.tensor([-20.0, -10.0, 200.0, 150.0])  # sticks out of the image
    batch = torch.randint(0, N, (R,), generator=g)
    fr = [f.clone().requires_grad_(True) for f in feats]
    outs = []
    for r in range(R):  # oracle, roi by roi (its pooler wants per-image lists)
        outs.append(O.roi_pool([f[batch[r]:batch[r] + 1] for f in fr], [rois[r:r + 1]]))
    ref = torch.cat(outs)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    fh = [f.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True) for f in feats]
    y = ops.roi_align(fh, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, rois.to(DEV), batch.to(torch.int32).to(DEV),
                      torch.ones(R, dtype=torch.uint8, device=DEV), 7)
    close(y.permute(0, 3, 1, 2), ref.detach(), rtol=1e-4, atol=1e-5)
    y.backward(dy.permute(0, 2, 3, 1).contiguous().to(DEV))
    for a, b in zip(fh, fr):
        close(a.grad.permute(0, 3, 1, 2), b.grad if b.grad is not None else torch.zeros_like(b), rtol=1e-3, atol=1e-5)


def test_match_boxes_and_lowq_vs_oracle():
    from ubteacher import hip
    g = torch.Generator().manual_seed(8)
    anchors = torch.cat(O.make_anchors([(12, 16), (6, 8)], [8, 16], sizes=(32, 64)))
    N, Gm = 3, 16
    gb = torch.zeros(N, Gm, 4); gv = torch.zeros(N, Gm, dtype=torch.uint8)
    for n, k in enumerate((5, 0, 9)):
        c = torch.rand(k, 2, generator=g) * torch.tensor([100.0, 70.0])
        gb[n, :k] = torch.cat([c, c + torch.rand(k, 2, generator=g) * 60 + 8], 1)
        gv[n, :k] = 1
    gv[2, 3] = 0  # a hole: thresholded-away pseudo box
    mx, arg, gmax = hip.match_boxes(anchors.to(DEV), gb.to(DEV), gv.to(DEV), want_gt_max=True)
    lowq = hip.match_lowq(anchors.to(DEV), gb.to(DEV), gv.to(DEV), gmax)
    for n in range(N):
        vi = gv[n].bool().nonzero().squeeze(1)
        if len(vi) == 0:
            assert float(mx[n].max()) == -1.0
            continue
        iou = O.pairwise_iou(gb[n][vi], anchors)
        v, i = iou.max(dim=0)
        assert torch.equal(mx[n].cpu(), v)                       # same fp32 formula -> bit exact
        assert torch.equal(vi[i], arg[n].cpu().long())
        _, lab = O.matcher(iou, [0.3, 0.7], [0, -1, 1], True)
        _, lab_nolq = O.matcher(iou, [0.3, 0.7], [0, -1, 1], False)
        assert torch.equal(lowq[n].cpu().bool(), (iou == iou.max(dim=1)[0][:, None]).any(dim=0))


@pytest.mark.parametrize("branch", ["supervised", "unsup_data_train"])
# MODEL.ROI_HEADS.LOSS FocalLoss_BoundaryVar | CrossEntropy_BoundaryVar | round 4 "rcc": predicted deltas beyond the +-62.5 clamp of
# Box2BoxXYXYTransform.apply_deltas (reference box_regression.py:115-118) - the decode inside the nlloss IoU weight of rcnn.hip
@pytest.mark.parametrize("pre", ["rc", "rcce", "rcc"])
def test_predictor_losses_vs_reference_golden(rc, branch, pre):
    from ubteacher.modeling import rcnn as R_
    from ubteacher.params import ParamStore
    st = ParamStore()
    klass = R_.FastRCNNCrossEntropyBoundaryVarOutputLayers if pre == "rcce" else R_.FastRCNNFocaltLossBoundaryVarOutputLayers
    pred = klass(rcnn_cfg(), st, 1024, "roi_heads.box_predictor")
    src = "rcc" if pre == "rcc" else "rc"
    R = rc[src + "_cls"].shape[0]
    pad = 4  # empty slots must be ignored
    def padded(x, fill=0.0):
        x = T(x).float()
        return torch.cat([x, torch.full((pad,) + tuple(x.shape[1:]), fill)]).to(DEV)
    scores, deltas, std = (padded(rc["%s_%s_%s" % (src, branch, k)]).requires_grad_(True) for k in ("scores", "deltas", "std"))
    sampled = dict(gt_classes=torch.cat([T(rc[src + "_cls"]).long(), torch.full((pad,), -1)]).to(DEV)[None],
                   proposal_boxes=padded(rc[src + "_prop"])[None], gt_boxes=padded(rc[src + "_gtb"])[None], gt_loc_std=padded(rc[src + "_gstd"])[None])
    ls = pred.losses((scores, deltas, std), sampled, branch)
    close(ls["loss_cls"], rc["%s_%s_loss_cls" % (pre, branch)], rtol=2e-5)
    close(ls["loss_box_reg"], rc["%s_%s_loss_box_reg" % (pre, branch)], rtol=2e-5)
    (ls["loss_cls"] + 2.0 * ls["loss_box_reg"]).backward()
    for k, v in (("scores", scores), ("deltas", deltas), ("std", std)):
        gv = v.grad if v.grad is not None else torch.zeros_like(v)
        close(gv[:R], rc["%s_%s_g%s" % (pre, branch, k)], rtol=1e-4, atol=2e-7)
        assert float(gv[R:].abs().max()) == 0.0


V1 = {"spec": (False, 0.0, False), "spec_conf": (True, 0.0, False), "agn_conf_beta": (True, 0.5, True), "spec_giou": (False, 0.0, False)}


def utv1_predictor(agnostic=False, beta=0.0, box_loss="smooth_l1"):
    from ubteacher.modeling.rcnn import FastRCNNFocaltLossOutputLayers
    from ubteacher.params import ParamStore
    from ubteacher.presets import get_config
    cfg = get_config("rcnn", 1, ["MODEL.DEVICE", DEV, "MODEL.ROI_HEADS.LOSS", "FocalLoss", "MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG", agnostic,
                                 "MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA", beta, "MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE", box_loss])
    return FastRCNNFocaltLossOutputLayers(cfg, ParamStore(), 1024, "roi_heads.box_predictor")


@pytest.mark.parametrize("name", sorted(V1))
def test_utv1_focal_predictor_losses_vs_reference_golden(rc, name):
    """MODEL.ROI_HEADS.LOSS "FocalLoss": the UTv1 predictor's losses (reference fast_rcnn.py:1296-1429 executed by gen_golden.py),
    class-specific / class-agnostic deltas, gt_confid weighting, smooth-L1 beta; empty slots ignored"""
    conf, beta, agn = V1[name]
    pred = utv1_predictor(agn, beta, "giou" if name.endswith("giou") else "smooth_l1")
    R, pad = rc["v1_cls"].shape[0], 4
    def padded(x, fill=0.0):
        x = T(x).float()
        return torch.cat([x, torch.full((pad,) + tuple(x.shape[1:]), fill)]).to(DEV)
    scores, deltas = (padded(rc["v1_%s_%s" % (name, k)]).requires_grad_(True) for k in ("scores", "deltas"))
    sampled = dict(gt_classes=torch.cat([T(rc["v1_cls"]).long(), torch.full((pad,), -1)]).to(DEV)[None],
                   proposal_boxes=padded(rc["v1_prop"])[None], gt_boxes=padded(rc["v1_gtb"])[None])
    if conf:
        sampled["gt_confid"] = padded(rc["v1_conf"])[None]
    ls = pred.losses((scores, deltas, None), sampled, "supervised")
    close(ls["loss_cls"], rc["v1_%s_loss_cls" % name], rtol=2e-5); close(ls["loss_box_reg"], rc["v1_%s_loss_box_reg" % name], rtol=2e-5)
    (ls["loss_cls"] + 2.0 * ls["loss_box_reg"]).backward()
    for k, v in (("scores", scores), ("deltas", deltas)):
        close(v.grad[:R], rc["v1_%s_g%s" % (name, k)], rtol=1e-4, atol=2e-7)
        assert float(v.grad[R:].abs().max()) == 0.0 and bool(torch.isfinite(v.grad).all())
    raw = pred.losses((scores.detach(), deltas.detach(), None), sampled, "supervised", raw=True)   # the fused scalar tail's inputs
    close(raw["focal"][0] / R, rc["v1_%s_loss_cls" % name], rtol=2e-5); close(raw["box"][0] / R, rc["v1_%s_loss_box_reg" % name], rtol=2e-5)
    assert int((raw["tgt"] >= 0).sum()) == R


@pytest.mark.parametrize("agnostic", [False, True])
def test_utv1_focal_predictor_inference_vs_oracle(agnostic):
    """Detectron2's FastRCNNOutputLayers.inference with per-class boxes (class-specific apply_deltas with BBOX_REG_WEIGHTS, clip, score
    threshold, class-aware NMS, top-k), two images, invalid proposal slots, a non-finite row: vs the oracle's restatement"""
    pred = utv1_predictor(agnostic)
    from ubteacher.modeling.fcos import PaddedBoxes
    g = torch.Generator().manual_seed(5)
    N, P, K = 2, 40, 80
    p0 = torch.rand(N, P, 2, generator=g) * 200
    prop = torch.cat([p0, p0 + torch.rand(N, P, 2, generator=g) * 90 + 4], -1)
    scores = torch.randn(N * P, K + 1, generator=g) * 3
    deltas = torch.randn(N * P, 4 * pred.nbox, generator=g) * 1.5
    deltas[3, :] = float("nan")
    valid = torch.ones(N, P, dtype=torch.uint8); valid[1, 30:] = 0
    sizes = [(300, 280), (250, 300)]
    props = PaddedBoxes(sizes, boxes=prop.to(DEV), valid=valid.to(DEV))
    dets, rows = pred.inference((scores.to(DEV), deltas.to(DEV), None), props)
    for i in range(N):
        m = valid[i].bool()
        bx = O.d2_apply_deltas(deltas.view(N, P, -1)[i][m], prop[i][m], (10.0, 10.0, 5.0, 5.0))
        want, wrows = O.fast_rcnn_inference_per_class(bx, F.softmax(scores.view(N, P, -1)[i][m], dim=-1), sizes[i])
        n = int(dets["count"][i])
        assert n == len(wrows) and n > 5
        assert np.array_equal(rows[i, :n].cpu().numpy(), torch.nonzero(m).squeeze(1)[wrows].numpy())
        assert np.array_equal(dets["classes"][i, :n].cpu().numpy(), want["classes"].numpy())
        close(dets["boxes"][i, :n], want["boxes"], atol=2e-4); close(dets["scores"][i, :n], want["scores"], rtol=2e-5)
        assert 3 not in rows[0, :int(dets["count"][0])].tolist()


@pytest.mark.parametrize("pre,prop_key", [("inf", "rc_prop"), ("infc", "infc_prop")])
def test_predictor_inference_vs_reference_golden(rc, pre, prop_key):
    """infc (round 4): rows 0-11 are tiny proposals whose deltas lie far beyond the +-62.5 clamp of
    Box2BoxXYXYTransform.apply_deltas (box_regression.py:88-128): the kept boxes are the clamped decode (inside the image),
    through the fused decode of the product's inference path."""
    from ubteacher.modeling.fcos import PaddedBoxes
    from ubteacher.modeling.rcnn import FastRCNNFocaltLossBoundaryVarOutputLayers
    from ubteacher.params import ParamStore
    pred = FastRCNNFocaltLossBoundaryVarOutputLayers(rcnn_cfg(), ParamStore(), 1024, "roi_heads.box_predictor")
    R = rc[prop_key].shape[0]
    props = PaddedBoxes([(300, 300)], boxes=T(rc[prop_key]).float()[None].to(DEV), valid=torch.ones(1, R, dtype=torch.uint8, device=DEV))
    dets, rows = pred.inference((T(rc[pre + "_scores"]).to(DEV), T(rc[pre + "_deltas"]).to(DEV), T(rc[pre + "_std"]).to(DEV)), props)
    n = int(dets["count"][0])
    assert n == len(rc[pre + "_keep"])
    assert np.array_equal(rows[0, :n].cpu().numpy(), rc[pre + "_keep"])
    assert np.array_equal(dets["classes"][0, :n].cpu().numpy(), rc[pre + "_cls"])
    close(dets["boxes"][0, :n], rc[pre + "_boxes"], atol=2e-4); close(dets["scores"][0, :n], rc[pre + "_sc"], rtol=2e-5)
    close(dets["pred_boxes_std"][0, :n], rc[pre + "_bstd"])


def test_rpn_pseudo_losses_vs_reference_golden(rc):
    from ubteacher.modeling.fcos import PaddedBoxes
    from ubteacher.modeling.rcnn import PseudoLabRPN
    from ubteacher.params import ParamStore
    cfg = rcnn_cfg()
    cfg.MODEL.RPN.BATCH_SIZE_PER_IMAGE = 16
    rpn = PseudoLabRPN(cfg, ParamStore(), 256)
    anchors = torch.cat(O.make_anchors([(6, 8), (3, 4)], [16, 32], sizes=(32, 64))).to(DEV)
    obj = torch.cat([T(rc["rpn_obj%d" % l]) for l in range(2)], 1).to(DEV).requires_grad_(True)
    dl = torch.cat([T(rc["rpn_dl%d" % l]) for l in range(2)], 1).to(DEV).requires_grad_(True)
    gb = torch.zeros(2, 16, 4); gs = torch.zeros(2, 16); gv = torch.zeros(2, 16, dtype=torch.uint8)
    for i in range(2):
        b = T(rc["rpn_gt%d" % i]).float().reshape(-1, 4)
        gb[i, :len(b)] = b; gs[i, :len(b)] = T(rc["rpn_sc%d" % i]).float(); gv[i, :len(b)] = 1
    gt = PaddedBoxes([(96, 128)] * 2, boxes=gb.to(DEV), scores=gs.to(DEV), valid=gv.to(DEV), classes=torch.zeros(2, 16, dtype=torch.int32, device=DEV))
    rpn.sample_keys = T(rc["rpn_keys"]).float().to(DEV)
    ls = rpn.losses(anchors, obj, dl, gt)
    close(ls["loss_rpn_cls"], rc["rpn_loss_cls"], rtol=2e-5); close(ls["loss_rpn_loc"], rc["rpn_loss_loc"], rtol=2e-5)
    (ls["loss_rpn_cls"] + ls["loss_rpn_loc"]).backward()
    n0 = rc["rpn_obj0"].shape[1]
    close(obj.grad[:, :n0], rc["rpn_gobj0"], rtol=1e-4, atol=1e-8); close(obj.grad[:, n0:], rc["rpn_gobj1"], rtol=1e-4, atol=1e-8)
    close(dl.grad[:, :n0], rc["rpn_gdl0"], rtol=1e-4, atol=1e-8); close(dl.grad[:, n0:], rc["rpn_gdl1"], rtol=1e-4, atol=1e-8)


def test_rpn_losses_from_head_output_equal_dense_form():
    """PseudoLabRPN.losses reading logits / deltas straight from the level-first head output (the product path: no per-image copies)
    against the same fused kernel on the dense per-image tensors the reference-golden test above pins: identical sums, and the
    gradient scattered into the head buffer equals autograd through the per-image views.  Pseudo-label weights, an image without gt."""
    from ubteacher import ops
    from ubteacher.modeling.fcos import PaddedBoxes
    from ubteacher.modeling.rcnn import RPN_CH
    from ubteacher.modeling import build_model
    from ubteacher.presets import get_config
    cfg = get_config("rcnn", 1, ["MODEL.DEVICE", "cuda"])
    torch.manual_seed(0)
    rpn = build_model(cfg).proposal_generator
    N, hw = 3, [(40, 56), (20, 28), (10, 14), (5, 7), (3, 4)]
    meta = ops.LevelMeta(N, hw)
    g = torch.Generator(device="cuda").manual_seed(3)
    big0 = torch.randn(meta.P, RPN_CH, device="cuda", generator=g)
    anchors = rpn.anchor_generator(hw, big0.device)
    gb = torch.zeros(N, 8, 4, device="cuda"); gv = torch.zeros(N, 8, dtype=torch.uint8, device="cuda")
    gb[0, :3] = torch.tensor([[10., 12., 90., 70.], [100., 40., 180., 150.], [30., 60., 60., 100.]], device="cuda"); gv[0, :3] = 1
    gb[2, :1] = torch.tensor([[50., 20., 200., 140.]], device="cuda"); gv[2, :1] = 1          # image 1 has no gt
    gs = torch.rand(N, 8, device="cuda", generator=g)
    keys = torch.rand(N, torch.cat(anchors).shape[0], device="cuda", generator=g)
    for pseudo in (False, True):
        f = dict(boxes=gb, valid=gv, classes=torch.zeros(N, 8, dtype=torch.int32, device="cuda"))
        if pseudo:
            f["scores"] = gs
        gt = PaddedBoxes([(160, 224)] * N, **f)
        rpn.sample_keys = keys
        big_a = big0.clone().requires_grad_(True)
        la = rpn.losses(torch.cat(anchors), big_a, None, gt, head_hw=hw)
        (la["loss_rpn_cls"] * 1.7 + la["loss_rpn_loc"] * 0.6).backward()
        big_b = big0.clone().requires_grad_(True)
        obj, dl = rpn._per_image_views(big_b, N, hw)
        lb = rpn.losses(torch.cat(anchors), torch.cat(obj, 1), torch.cat(dl, 1), gt)
        (lb["loss_rpn_cls"] * 1.7 + lb["loss_rpn_loc"] * 0.6).backward()
        assert la["loss_rpn_cls"].item() > 0 and la["loss_rpn_loc"].item() > 0
        assert torch.equal(la["loss_rpn_cls"], lb["loss_rpn_cls"]) and torch.equal(la["loss_rpn_loc"], lb["loss_rpn_loc"])
        assert torch.equal(big_a.grad, big_b.grad) and float(big_a.grad.abs().sum()) > 0
    rpn.sample_keys = None


def test_roi_sampling_vs_reference_golden(rc):
    from ubteacher.modeling.fcos import PaddedBoxes
    from ubteacher.modeling.rcnn import StandardROIHeadsPseudoLab
    from ubteacher.params import ParamStore
    cfg = rcnn_cfg()
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 16
    heads = StandardROIHeadsPseudoLab(cfg, ParamStore(), 256)
    P, Gn, MG = rc["roi_prop"].shape[0], rc["roi_gtb"].shape[0], 16
    props = PaddedBoxes([(300, 300)], boxes=T(rc["roi_prop"]).float()[None].to(DEV), valid=torch.ones(1, P, dtype=torch.uint8, device=DEV))
    gb = torch.zeros(1, MG, 4); gc = torch.zeros(1, MG, dtype=torch.int32); gs = torch.zeros(1, MG); gst = torch.zeros(1, MG, 4)
    gv = torch.zeros(1, MG, dtype=torch.uint8)
    gb[0, :Gn] = T(rc["roi_gtb"]); gc[0, :Gn] = T(rc["roi_gtc"]).to(torch.int32); gs[0, :Gn] = T(rc["roi_gts"]); gst[0, :Gn] = T(rc["roi_gtstd"]); gv[0, :Gn] = 1
    gt = PaddedBoxes([(300, 300)], boxes=gb.to(DEV), classes=gc.to(DEV), scores=gs.to(DEV), pred_boxes_std=gst.to(DEV), valid=gv.to(DEV))
    keys = torch.full((1, P + MG), 0.5)
    keys[0, :P] = T(rc["roi_keys"])[:P]; keys[0, P:P + Gn] = T(rc["roi_keys"])[P:]
    heads.sample_keys = keys.to(DEV)
    out = heads.label_and_sample_proposals(props, gt, "x")
    n = int(out["valid"].sum())
    assert n == len(rc["roi_out_cls"])
    assert np.array_equal(out["gt_classes"][0, :n].cpu().numpy(), rc["roi_out_cls"])
    close(out["proposal_boxes"][0, :n], rc["roi_out_prop"]); close(out["gt_boxes"][0, :n], rc["roi_out_gtb"])
    close(out["gt_confid"][0, :n], rc["roi_out_conf"]); close(out["gt_loc_std"][0, :n], rc["roi_out_std"])


def test_roi_align_bf16_io():
    """bf16 features / output / dy: forward == the fp32 kernel on the same (bf16-representable) features rounded once;
    backward scatters the same fp32 values."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(21)
    feats32 = [torch.randn(2, h, w, 256, generator=g).to(torch.bfloat16).float().cuda() for h, w in ((50, 64), (25, 32), (13, 16), (7, 8))]
    feats16 = [f.to(torch.bfloat16) for f in feats32]
    R = 64
    x1 = torch.rand(R, generator=g) * 180; y1 = torch.rand(R, generator=g) * 140
    wh = torch.exp(torch.rand(R, 2, generator=g) * 4.5 + 1.0)
    rois = torch.stack((x1, y1, (x1 + wh[:, 0]).clamp(max=255), (y1 + wh[:, 1]).clamp(max=199)), 1).cuda()
    batch = torch.randint(0, 2, (R,), generator=g).to(torch.int32).cuda()
    valid = torch.ones(R, dtype=torch.uint8).cuda()
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    y32 = hip.roi_align_fwd(feats32, scales, 2, rois, batch, valid, 7)
    y16 = hip.roi_align_fwd(feats16, scales, 2, rois, batch, valid, 7)
    assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.to(torch.bfloat16))
    dy = torch.randn(R, 7, 7, 256, generator=g).to(torch.bfloat16).cuda()
    d32 = [torch.zeros_like(f) for f in feats32]
    d16 = [torch.zeros_like(f) for f in feats32]
    hip.roi_align_bwd(d32, scales, 2, rois, batch, valid, dy.float())
    hip.roi_align_bwd(d16, scales, 2, rois, batch, valid, dy)
    for a, b in zip(d16, d32):   # fp32 atomics: the summation order differs between launches
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max() + 1e-6)


def test_roi_align_fwd_per_roi_kernel_matches_per_bin_kernel(tmp_path):
    """utv2_roi_align_fwd runs one workgroup per ROI (tap tables built once per ROI in LDS, 16-byte loads); UTV2_ROI_FWD_PER_ROI=0 keeps
    the one-wave-per-(roi, bin) kernel: the same formula tap for tap - the compiler contracts the sample coordinate `start + k * bin`
    into an FMA in one and not in the other, so the outputs agree to the last bits, not always bit for bit: <= 1e-6 of the tensor's scale in
    fp32, <= one 16-bit rounding step in 16-bit, on ROIs of every level incl. clipped / out-of-image / degenerate / invalid ones (the
    switch is read once per process: subprocess).  Both are checked against the oracle by test_roi_align_fwd_bwd_vs_oracle."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, torch
sys.path.insert(0, %r)
from ubteacher import hip
g = torch.Generator().manual_seed(77)
feats32 = [torch.randn(3, h, w, 256, generator=g).to(torch.bfloat16).float().cuda() for h, w in ((100, 168), (50, 84), (25, 42), (13, 21))]
feats16 = [f.to(torch.bfloat16) for f in feats32]
R = 3 * 200
x1 = torch.rand(R, generator=g) * 620 - 20; y1 = torch.rand(R, generator=g) * 380 - 20
wh = torch.exp(torch.rand(R, 2, generator=g) * 6.5 + 0.5)
rois = torch.stack((x1, y1, x1 + wh[:, 0], y1 + wh[:, 1]), 1)
rois[0] = torch.tensor([10.0, 10.0, 10.0, 40.0]); rois[1] = torch.tensor([-300.0, -300.0, 900.0, 700.0]); rois[2] = torch.tensor([650.0, 390.0, 700.0, 420.0])
rois = rois.cuda()
batch = (torch.arange(R) // 200).to(torch.int32).cuda()
valid = (torch.rand(R, generator=g) > 0.1).to(torch.uint8).cuda()
scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
out = {"f32": hip.roi_align_fwd(feats32, scales, 2, rois, batch, valid, 7).cpu(), "h16": hip.roi_align_fwd(feats16, scales, 2, rois, batch, valid, 7).cpu(),
       "novalid": hip.roi_align_fwd(feats16, scales, 2, rois, batch, None, 7).cpu(), "p5": hip.roi_align_fwd(feats16, scales, 2, rois, batch, valid, 5).cpu()}
torch.save(out, sys.argv[1])
""" % os.path.join(root, "unbiased-teacher-v2_amd")
    outs = {}
    for flag in ("0", "1"):
        path = str(tmp_path / ("roi%s.pt" % flag))
        e = dict(os.environ); e["UTV2_ROI_FWD_PER_ROI"] = flag
        r = subprocess.run([sys.executable, "-c", code, path], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[flag] = torch.load(path)
    for k in outs["0"]:
        a, b = outs["0"][k].float(), outs["1"][k].float()
        assert torch.isfinite(b).all() and float(b.abs().max()) > 0
        scale = float(a.abs().max())
        tol = 1e-6 * scale if outs["0"][k].dtype == torch.float32 else 2.0 ** -7 * a.abs() + 1e-6 * scale
        assert bool(((a - b).abs() <= tol).all()), (k, float((a - b).abs().max()))
        assert float(((a - b) != 0).float().mean()) < 0.01       # and almost everywhere identical
        assert torch.equal(a == 0, b == 0)                       # invalid / empty ROIs: the same zeros


def test_box_iou_and_add_entry_points():
    """utv2_box_iou (D2 pairwise_iou [D2-recall]) against the oracle's restatement, including degenerate and disjoint boxes;
    utv2_add bit-exact against a + b."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(3)
    xy = torch.rand(37, 2, generator=g) * 100
    a = torch.cat([xy, xy + torch.rand(37, 2, generator=g) * 60], 1)
    xy = torch.rand(53, 2, generator=g) * 100
    b = torch.cat([xy, xy + torch.rand(53, 2, generator=g) * 60], 1)
    a[0] = torch.tensor([5.0, 5.0, 5.0, 9.0])          # zero area
    b[1] = torch.tensor([500.0, 500.0, 510.0, 510.0])  # disjoint from everything
    b[2] = a[3]                                         # identical
    got = hip.box_iou(a.cuda(), b.cuda()).cpu()
    ref = O.pairwise_iou(a, b)
    assert got.shape == ref.shape and float((got - ref).abs().max()) < 1e-6
    assert float(got[3, 2]) == 1.0 and float(got[:, 1].max()) == 0.0
    x, y = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    assert torch.equal(hip.add(x.cuda(), y.cuda()).cpu(), x + y)


def test_rpn_pre_nms_topk_radix_select_equals_torch_topk():
    """PseudoLabRPN._pre_nms_topk (one utv2_topk_rows_i64 call for all levels and images) returns exactly the anchor indices of the
    per-level torch.topk on (score desc, index asc) keys - negative logits, ties and levels narrower than PRE_NMS_TOPK included."""
    from ubteacher import ops
    from ubteacher.modeling.rcnn import RPN_CH, float_order_key
    from ubteacher.modeling import build_model
    from ubteacher.presets import get_config
    cfg = get_config("rcnn", 1, ["MODEL.DEVICE", "cuda"])
    torch.manual_seed(0)
    rpn = build_model(cfg).proposal_generator
    N, hw = 3, [(40, 56), (20, 28), (10, 14), (5, 7), (3, 4)]
    meta = ops.LevelMeta(N, hw)
    g = torch.Generator(device="cuda").manual_seed(1)
    big = torch.randn(meta.P, RPN_CH, device="cuda", generator=g) * 3
    big[:, :3] = torch.round(big[:, :3] * 4) / 4            # many exact ties
    top, ks = rpn._pre_nms_topk(big, N, hw)
    obj, _ = rpn._per_image_views(big, N, hw)
    pre = rpn.pre_nms_topk[rpn.training]
    for l, o in enumerate(obj):
        k = min(pre, o.shape[1])
        assert ks[l] == k
        ref = torch.topk(float_order_key(o), k, dim=1, sorted=True).values
        ref_idx = 4294967295 - (ref & 4294967295)
        got = 2147483647 - (top[l * N:(l + 1) * N, :k] & 2147483647)
        assert torch.equal(got, ref_idx), l


def test_rpn_fused_decode_equals_elementwise_chain():
    """utv2_rpn_rank_keys + utv2_rpn_decode (one launch for gather / apply_deltas / clip / keep over all levels) against
    PseudoLabRPN.predict_proposals' per-level ATen chain on the same head output: identical candidate boxes, scores, levels and keep
    masks, hence identical proposals after the NMS.  Includes deltas beyond SCALE_CLAMP, boxes pushed outside the image, boxes
    that shrink below MIN_SIZE after clipping, a NaN and an inf logit / delta, and ragged image sizes."""
    from ubteacher import hip, ops
    from ubteacher.modeling.rcnn import RPN_CH, SCALE_CLAMP
    from ubteacher.modeling import build_model
    from ubteacher.presets import get_config
    cfg = get_config("rcnn", 1, ["MODEL.DEVICE", "cuda"])
    torch.manual_seed(0)
    rpn = build_model(cfg).proposal_generator
    N, hw = 3, [(40, 56), (20, 28), (10, 14), (5, 7), (3, 4)]
    sizes = [(160, 224), (150, 200), (97, 131)]
    meta = ops.LevelMeta(N, hw)
    g = torch.Generator(device="cuda").manual_seed(5)
    big = torch.randn(meta.P, RPN_CH, device="cuda", generator=g)
    big[:, 3:15] *= 1.5                                       # some dw / dh beyond log(1000 / 16) after the next line
    big[::37, 5] = 9.0
    big[::41, 3] = -40.0                                      # pushed far outside: clipped to zero width
    big[11, 0] = float("nan"); big[13, 4] = float("inf"); big[17, 1] = float("-inf")
    anchors = rpn.anchor_generator(hw, big.device)
    for training in (True, False):
        rpn.train(training)
        obj, dl = rpn._per_image_views(big, N, hw)
        ref = rpn.predict_proposals(anchors, obj, dl, sizes)
        sel = rpn._pre_nms_topk(big, N, hw)
        got = rpn._proposals_fused(big, anchors, sel, hw, N, sizes)
        assert torch.equal(got["count"], ref["count"])
        assert torch.equal(got["valid"], ref["valid"])
        m = ref["valid"].bool()
        assert torch.equal(got["boxes"][m], ref["boxes"][m])
        assert torch.equal(got["objectness_logits"][m], ref["objectness_logits"][m])
        # the candidate tensors themselves
        top, ks = sel
        boxes, scores, lvls, keep = hip.rpn_decode(top, big, torch.cat(anchors).contiguous(),
                                                   torch.tensor([[s[0], s[1]] for s in sizes], dtype=torch.float32, device="cuda"),
                                                   [h * w for h, w in hw], ks, N, rpn.A, rpn.box_weights, SCALE_CLAMP, rpn.min_box_size)
        assert int(keep.sum()) > 0 and int((keep == 0).sum()) > 0
        assert torch.equal(lvls[0].cpu(), torch.repeat_interleave(torch.arange(len(hw), dtype=torch.int32), torch.tensor(ks)))
    rpn.train(True)


def test_roi_align_bwd_tiled_gather_equals_scatter_and_is_deterministic():
    """utv2_roi_align_bwd_tiled (the deterministic gather over 8 x 8 pixel tiles the ROI heads' backward runs) against the atomic
    scatter kernel and, through autograd, against the oracle: ROIs image by image (P slots each, some invalid), boxes that stick out of
    the image, elongated boxes with many samples per bin, all four levels; bit-identical between runs; bf16 in / out forms."""
    from ubteacher import hip, ops
    g = torch.Generator().manual_seed(33)
    N, P, C = 3, 40, 256
    shapes = [(N, 50, 64, C), (N, 25, 32, C), (N, 13, 16, C), (N, 7, 8, C)]
    x1 = torch.rand(N * P, generator=g) * 220 - 10; y1 = torch.rand(N * P, generator=g) * 170 - 10
    wh = torch.exp(torch.rand(N * P, 2, generator=g) * 5.0 + 0.5)
    rois = torch.stack((x1, y1, x1 + wh[:, 0], y1 + wh[:, 1]), 1)
    rois[0] = torch.tensor([-300.0, -200.0, 500.0, 460.0])    # far outside on every side, large enough for the coarsest level
    rois[1] = torch.tensor([5.0, 3.0, 9.0, 190.0])            # tall and thin: many samples per bin row
    rois[2] = torch.tensor([2.0, 100.0, 250.0, 104.0])        # wide and flat
    rois[3] = torch.tensor([255.5, 199.5, 256.0, 200.0])      # at the far corner
    rois[4] = torch.tensor([-40.0, -40.0, 260.0, 240.0])      # sqrt(area) in [224, 448): the third level
    rois = rois.cuda()
    valid = (torch.rand(N * P, generator=g) > 0.15).to(torch.uint8).cuda()
    valid[:5] = 1
    batch = torch.arange(N, dtype=torch.int32).repeat_interleave(P).cuda()
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    dy = torch.randn(N * P, 7, 7, C, generator=g).cuda()
    ref = [torch.zeros(s, device="cuda") for s in shapes]
    hip.roi_align_bwd(ref, scales, 2, rois, batch, valid, dy)
    got = hip.roi_align_bwd_tiled(shapes, torch.float32, scales, 2, rois, valid, dy, P)
    again = hip.roi_align_bwd_tiled(shapes, torch.float32, scales, 2, rois, valid, dy, P)
    for a, b, c in zip(got, ref, again):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max() + 1e-6)
        assert torch.equal(a, c)                                # deterministic
    assert all(float(b.abs().max()) > 0 for b in ref)          # every level is exercised
    # bf16 dy / bf16 gradient maps: the fp32 result of the bf16 dy, rounded once
    dy16 = dy.to(torch.bfloat16)
    g32 = hip.roi_align_bwd_tiled(shapes, torch.float32, scales, 2, rois, valid, dy16.float(), P)
    g16 = hip.roi_align_bwd_tiled(shapes, torch.bfloat16, scales, 2, rois, valid, dy16, P)
    for a, b in zip(g16, g32):
        assert a.dtype == torch.bfloat16 and torch.equal(a, b.to(torch.bfloat16))
    # through autograd (the product's path) against the oracle's autograd
    feats = [torch.randn(N, C, s[1], s[2], generator=g) * 0.5 for s in shapes]
    fr = [f.clone().requires_grad_(True) for f in feats]
    rc, vc = rois.cpu(), valid.cpu().bool()
    outs = [O.roi_pool([f[n:n + 1] for f in fr], [rc[n * P + i:n * P + i + 1]]) for n in range(N) for i in range(P)]
    refy = torch.cat(outs)
    dyc = dy.cpu().permute(0, 3, 1, 2) * vc[:, None, None, None]
    refy.backward(dyc)
    fh = [f.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True) for f in feats]
    y = ops.roi_align(fh, scales, 2, rois, batch, valid, 7, rois_per_image=P)
    y.backward(dy)
    for a, b in zip(fh, fr):
        close(a.grad.permute(0, 3, 1, 2), b.grad, rtol=1e-3, atol=2e-5)


def test_fused_samplers_equal_the_elementwise_chains(monkeypatch):
    """utv2_rpn_sample_keys + radix select + utv2_rpn_sample_unpack, and utv2_roi_sample, against the chains of elementwise / topk /
    sort / gather launches they replace (UTV2_FUSED_SAMPLERS=0, the form the reference goldens above pin): identical sampled sets in
    identical order on every valid slot - images with many, few and no ground-truth boxes, more candidates than slots and fewer."""
    from ubteacher.modeling.fcos import PaddedBoxes
    from ubteacher.modeling.rcnn import PseudoLabRPN, StandardROIHeadsPseudoLab
    from ubteacher.params import ParamStore
    cfg = rcnn_cfg()
    g = torch.Generator().manual_seed(5)
    N, M = 4, 16
    sizes = [(160, 224)] * N
    gb = torch.zeros(N, M, 4); gv = torch.zeros(N, M, dtype=torch.uint8)
    for n, k in enumerate((9, 2, 0, 16)):
        xy = torch.rand(k, 2, generator=g) * torch.tensor([150.0, 100.0])
        wh = torch.rand(k, 2, generator=g) * 60 + 8
        gb[n, :k] = torch.cat((xy, xy + wh), 1); gv[n, :k] = 1
    gt = PaddedBoxes(sizes, boxes=gb.to(DEV), classes=torch.randint(0, 80, (N, M), generator=g).to(torch.int32).to(DEV), valid=gv.to(DEV),
                     scores=torch.rand(N, M, generator=g).to(DEV), pred_boxes_std=torch.randn(N, M, 4, generator=g).to(DEV))
    rpn = PseudoLabRPN(cfg, ParamStore(), 256)
    hw = [(40, 56), (20, 28), (10, 14), (5, 7), (3, 4)]
    anchors = torch.cat(rpn.anchor_generator(hw, torch.device(DEV)))
    R = anchors.shape[0]
    rkeys = torch.rand(N, R, generator=g).to(DEV)
    rpn.sample_keys = rkeys
    heads = StandardROIHeadsPseudoLab(cfg, ParamStore(), 256)
    P = 300
    xy = torch.rand(N, P, 2, generator=g) * torch.tensor([150.0, 100.0])
    wh = torch.rand(N, P, 2, generator=g) * 60 + 8
    pbx = torch.cat((xy, xy + wh), 2)
    pbx[:, :40] = gb[:, torch.arange(40) % M] + torch.randn(N, 40, 4, generator=g) * 2     # proposals near the gt boxes: foreground
    pvalid = (torch.rand(N, P, generator=g) > 0.1).to(torch.uint8)
    props = PaddedBoxes(sizes, boxes=pbx.to(DEV).contiguous(), valid=pvalid.to(DEV))
    outs = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("UTV2_FUSED_SAMPLERS", fused)
        for bs in (256, 32):
            rpn.batch_size_per_image = bs
            outs[("rpn", bs, fused)] = rpn.label_and_sample(anchors, gt)
        for bs in (512, 64):
            heads.batch_size_per_image = bs
            heads.sample_keys = torch.rand(N, P + M, generator=torch.Generator().manual_seed(bs)).to(DEV)
            outs[("roi", bs, fused)] = heads.label_and_sample_proposals(props, gt, "unsup_data_train")
    for bs in (256, 32):
        a, b = outs[("rpn", bs, "1")], outs[("rpn", bs, "0")]
        assert torch.equal(a["has_gt"].bool(), b["has_gt"]) and torch.equal(a["matched32"], b["matched32"])
        for part in ("pos", "neg"):
            va, vb = a[part + "_valid"].bool(), b[part + "_valid"].bool()
            assert torch.equal(va, vb), (bs, part)
            assert torch.equal(a[part + "_idx"][va], b[part + "_idx"][vb]), (bs, part)
        assert int(a["pos_valid"].sum()) > 0 and int(a["neg_valid"].sum()) > 0
        assert int(a["pos_valid"][2].sum()) == 0 and int(a["neg_valid"][2].sum()) == bs      # image without gt: negatives only
    for bs in (512, 64):
        a, b = outs[("roi", bs, "1")], outs[("roi", bs, "0")]
        w = b["valid"].shape[1]      # the chain's width is min(batch, nfg_max + #slots) (here 128 + 316 < 512); the kernel always pads to batch
        assert not bool(a["valid"][:, w:].any())
        va = a["valid"].bool()[:, :w]
        assert torch.equal(va, b["valid"].bool()) and int(va.sum()) > 0
        assert set(a) == set(b)
        for k in b:
            if k != "valid":
                assert torch.equal(a[k][:, :w][va], b[k][va]), (bs, k)
        va = a["valid"].bool()
        assert bool((a["gt_classes"][~va] == -1).all())
        fg = (a["gt_classes"] >= 0) & (a["gt_classes"] < 80)
        assert int(fg.sum()) > 0 and int(fg[2].sum()) == 0


def test_fused_roi_inference_equals_the_aten_chain(monkeypatch):
    """utv2_roi_infer_keys / _gather / _pack (round 4) around softmax, top-k and the class-aware NMS == the ~55-op ATen chain they replace
    (reference roi_heads/fast_rcnn.py:1094-1125,1162-1225 + D2 fast_rcnn_inference): identical padded detections bit for bit - boxes,
    scores, classes, pred_boxes_std, kept proposal rows, counts - with invalid proposal slots, deltas beyond the clamp, a NaN delta and an
    infinite score among the inputs, and probability ties (duplicated rows: the flat-index tie rule decides)."""
    from ubteacher.modeling.fcos import PaddedBoxes
    from ubteacher.modeling.rcnn import FastRCNNFocaltLossBoundaryVarOutputLayers
    from ubteacher.params import ParamStore
    pred = FastRCNNFocaltLossBoundaryVarOutputLayers(rcnn_cfg(), ParamStore(), 1024, "roi_heads.box_predictor")
    g = torch.Generator().manual_seed(17)
    N, P = 3, 333
    xy = torch.rand(N, P, 2, generator=g) * 200
    boxes = torch.cat([xy, xy + torch.rand(N, P, 2, generator=g) * 90 + 2], dim=2)
    valid = (torch.rand(N, P, generator=g) > 0.1).to(torch.uint8)
    scores = torch.randn(N * P, 81, generator=g) * 3
    deltas = torch.randn(N * P, 4, generator=g) * 3
    std = torch.randn(N * P, 4, generator=g)
    deltas[5] = torch.tensor([900.0, -900.0, 700.0, -2000.0])
    deltas[17, 2] = float("nan")
    scores[23, 4] = float("inf")
    scores[40:44] = scores[40]; deltas[40:44] = deltas[40]; boxes.view(-1, 4)[40:44] = boxes.view(-1, 4)[40]     # exact probability ties
    props = PaddedBoxes([(240, 300), (200, 280), (300, 300)], boxes=boxes.to(DEV), valid=valid.to(DEV))
    args = (scores.to(DEV), deltas.to(DEV), std.to(DEV))
    monkeypatch.setenv("UTV2_FUSED_ROI_INFERENCE", "0")
    ref, rows_ref = pred.inference(args, props)
    monkeypatch.setenv("UTV2_FUSED_ROI_INFERENCE", "1")
    got, rows = pred.inference(args, props)
    assert int(ref["count"].sum()) > 50
    assert torch.equal(got["count"], ref["count"]) and torch.equal(got["valid"], ref["valid"])
    m = ref["valid"].bool()
    for k in ("boxes", "scores", "classes", "pred_boxes_std"):
        assert torch.equal(got[k][m], ref[k][m]), k
    assert torch.equal(rows[m], rows_ref[m])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_box_losses_ignore_what_the_reference_never_indexes(mode):
    """box_reg_loss / box_reg_pseudo_loss (fast_rcnn.py:938-1090) index the FOREGROUND rows before any arithmetic: whatever a background or
    empty row holds - a std logit of -200 (sigmoid = 0: nll = inf - inf), an infinite delta of a diverged model - never reaches the
    sum.  The kernel walks every row: the loss value and the gradients with such rows must equal those with benign ones (found by a
    launcher run from random weights: metric NaN, gradients fine)."""
    from ubteacher import hip
    g = torch.Generator().manual_seed(0)
    R, K = 96, 80
    cls = torch.randint(0, K + 1, (R,), generator=g)
    cls[::7] = -1                                   # empty slots
    cls[:8] = torch.arange(8)                       # some foreground for sure
    prop = torch.rand(R, 4, generator=g) * 50
    prop[:, 2:] += prop[:, :2] + 8
    gtb = prop + torch.randn(R, 4, generator=g) * 2
    pred = torch.randn(R, 8, generator=g) * 0.3
    gstd = torch.randn(R, 4, generator=g)
    fg = (cls >= 0) & (cls < K)
    bad = pred.clone()
    bad[~fg, 4:] = -200.0
    bad[~fg, 0] = float("inf")
    bad[~fg, 1] = float("-inf")
    outs = []
    for p in (pred, bad):
        pc = p.cuda()
        outs.append(hip.roi_box_loss(pc[:, :4], pc[:, 4:], cls.cuda(), prop.cuda(), gtb.cuda(), gstd.cuda() if mode == 2 else None, K, mode,
                                     10.0, 5.0, 4.135, 0.1, 0.5))
    (s0, gd0, gs0), (s1, gd1, gs1) = outs
    assert torch.isfinite(s0).all() and float(s0) > 0
    assert torch.equal(s0, s1) and torch.equal(gd0, gd1) and torch.equal(gs0, gs1)
    assert float(gd1[~fg.cuda()].abs().max()) == 0 and float(gs1[~fg.cuda()].abs().max()) == 0
