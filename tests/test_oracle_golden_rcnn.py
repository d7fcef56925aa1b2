"""Pins the oracle's Faster-RCNN restatement against golden vectors produced by executing the
reference's own fast_rcnn.py / rpn.py / roi_heads.py / box_regression.py (tests/golden/gen_golden.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import utv2_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    a, b = a.astype(np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.allclose(a, b, rtol=rtol, atol=atol), float(np.abs(a - b).max())


@pytest.fixture(scope="module")
def rc():
    return dict(np.load(os.path.join(G, "rcnn.npz")))


def test_box2box_xyxy(rc):
    close(O.xyxy_get_deltas(T(rc["bx_src"]), T(rc["bx_tgt"])), rc["bx_get"])
    close(O.xyxy_apply_deltas(T(rc["bx_deltas"]), T(rc["bx_src"])), rc["bx_apply"], rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("branch", ["supervised", "unsup_data_train"])
# FocalLoss_BoundaryVar | CrossEntropy_BoundaryVar predictor | round 4: "rcc" = predicted deltas beyond the +-62.5 clamp of
# Box2BoxXYXYTransform.apply_deltas (box_regression.py:115-118) inside the nlloss IoU weight
@pytest.mark.parametrize("pre,gamma,src", [("rc", 1.5, "rc"), ("rcce", 0.0, "rc"), ("rcc", 1.5, "rcc")])
def test_roi_losses(rc, branch, pre, gamma, src):
    scores, deltas, std = (T(rc["%s_%s_%s" % (src, branch, k)]).clone().requires_grad_(True) for k in ("scores", "deltas", "std"))
    cls, prop, gtb, gstd = (T(rc["%s_%s" % (src, k)]) for k in ("cls", "prop", "gtb", "gstd"))
    lc = O.softmax_focal(scores, cls, gamma)
    lb = O.roi_box_reg_loss(prop, gtb, deltas, std, cls) if branch == "supervised" else O.roi_box_reg_pseudo_loss(prop, gtb, deltas, std, gstd, cls)
    close(lc, rc["%s_%s_loss_cls" % (pre, branch)]); close(lb, rc["%s_%s_loss_box_reg" % (pre, branch)])
    (lc + 2.0 * lb).backward()
    for k, v in (("scores", scores), ("deltas", deltas), ("std", std)):
        close(v.grad if v.grad is not None else torch.zeros_like(v), rc["%s_%s_g%s" % (pre, branch, k)], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("pre,prop_key", [("inf", "rc_prop"), ("infc", "infc_prop")])   # infc: rows 0-11 decode through the clamp
def test_roi_inference(rc, pre, prop_key):
    prop = T(rc[prop_key])
    boxes = O.xyxy_apply_deltas(T(rc[pre + "_deltas"]), prop)
    dets, rows = O.fast_rcnn_inference(boxes, F.softmax(T(rc[pre + "_scores"]), dim=-1), (300, 300))
    assert np.array_equal(rows.numpy(), rc[pre + "_keep"])
    assert np.array_equal(dets["classes"].numpy(), rc[pre + "_cls"])
    close(dets["boxes"], rc[pre + "_boxes"], atol=1e-4); close(dets["scores"], rc[pre + "_sc"])
    close(T(rc[pre + "_std"])[rows], rc[pre + "_bstd"])
    if pre == "infc":
        # the clamp decided these boxes: |delta / 10| > 62.5 on a 2 px proposal moves an edge by exactly 62.5 widths, inside the image
        kept = {int(r): i for i, r in enumerate(rc["infc_keep"])}
        w = rc["infc_prop"][0, 2] - rc["infc_prop"][0, 0]
        assert abs((rc["infc_boxes"][kept[0], 0] - rc["infc_prop"][0, 0]) - 62.5 * w) < 1e-3   # delta +900 -> +62.5 widths, not +90
        assert 0.0 < rc["infc_boxes"][kept[0], 0] < 300.0


V1 = {"spec": (False, 0.0), "spec_conf": (True, 0.0), "agn_conf_beta": (True, 0.5), "spec_giou": (False, 0.0)}


@pytest.mark.parametrize("name", sorted(V1))
def test_utv1_focal_predictor_losses(rc, name):
    """MODEL.ROI_HEADS.LOSS "FocalLoss" (the UTv1 predictor, fast_rcnn.py:1296-1429): class-specific and class-agnostic deltas, with and
    without the gt_confid weighting of the pseudo-labeled branch, smooth-L1 beta 0 and 0.5; losses and gradients"""
    conf, beta = V1[name]
    sc = T(rc["v1_%s_scores" % name]).clone().requires_grad_(True)
    de = T(rc["v1_%s_deltas" % name]).clone().requires_grad_(True)
    lc, lb = O.utv1_roi_losses(sc, de, T(rc["v1_prop"]), T(rc["v1_gtb"]), T(rc["v1_cls"]), T(rc["v1_conf"]) if conf else None, beta=beta,
                               box_reg_loss_type="giou" if name.endswith("giou") else "smooth_l1")
    close(lc, rc["v1_%s_loss_cls" % name]); close(lb, rc["v1_%s_loss_box_reg" % name])
    (lc + 2.0 * lb).backward()
    close(sc.grad, rc["v1_%s_gscores" % name], rtol=1e-4, atol=1e-8); close(de.grad, rc["v1_%s_gdeltas" % name], rtol=1e-4, atol=1e-8)
    if conf:    # the weighting bites: the unweighted loss differs
        lc0, _ = O.utv1_roi_losses(sc.detach(), de.detach(), T(rc["v1_prop"]), T(rc["v1_gtb"]), T(rc["v1_cls"]), None, beta=beta)
        assert abs(float(lc0) - float(lc.detach())) > 0.1 * float(lc.detach())


def test_d2_box2box_transform_round_trip(rc):
    """the stand-in of Detectron2's Box2BoxTransform with BBOX_REG_WEIGHTS (10, 10, 5, 5): apply(get(src, tgt), src) == tgt"""
    W = (10.0, 10.0, 5.0, 5.0)
    src, tgt = T(rc["v1_prop"]), T(rc["v1_gtb"])
    d = O.d2_get_deltas(src, tgt, W)
    close(O.d2_apply_deltas(d, src, W), tgt, rtol=1e-5, atol=1e-3)
    close(O.d2_apply_deltas(torch.cat([d, d], 1), src, W), torch.cat([tgt, tgt], 1), rtol=1e-5, atol=1e-3)


def test_rpn_pseudo_losses(rc):
    anchors = O.make_anchors([(6, 8), (3, 4)], [16, 32], sizes=(32, 64))
    obj = [T(rc["rpn_obj%d" % l]).clone().requires_grad_(True) for l in range(2)]
    dl = [T(rc["rpn_dl%d" % l]).clone().requires_grad_(True) for l in range(2)]
    gts = [dict(boxes=T(rc["rpn_gt%d" % i]).float().reshape(-1, 4), scores=T(rc["rpn_sc%d" % i]).float()) for i in range(2)]
    ls, samples = O.rpn_losses(torch.cat(anchors), torch.cat(obj, 1), torch.cat(dl, 1), gts, T(rc["rpn_keys"]), True, batch=16, frac=0.25)
    close(ls["loss_rpn_cls"], rc["rpn_loss_cls"]); close(ls["loss_rpn_loc"], rc["rpn_loss_loc"])
    lab = rc["rpn_labels"]
    for n, (pos, neg) in enumerate(samples):
        assert set(pos.tolist()) == set(np.nonzero(lab[n] == 1)[0].tolist())
        assert set(neg.tolist()) == set(np.nonzero(lab[n] == 0)[0].tolist())
    (ls["loss_rpn_cls"] + ls["loss_rpn_loc"]).backward()
    for l in range(2):
        close(obj[l].grad, rc["rpn_gobj%d" % l], rtol=1e-4, atol=1e-8); close(dl[l].grad, rc["rpn_gdl%d" % l], rtol=1e-4, atol=1e-8)


def test_roi_label_and_sample_pseudo(rc):
    gt = dict(boxes=T(rc["roi_gtb"]), classes=T(rc["roi_gtc"]), scores=T(rc["roi_gts"]), pred_boxes_std=T(rc["roi_gtstd"]))
    out = O.roi_label_and_sample(T(rc["roi_prop"]), gt, T(rc["roi_keys"]), True, batch=16, frac=0.25)
    close(out["proposal_boxes"], rc["roi_out_prop"]); assert np.array_equal(out["gt_classes"].numpy(), rc["roi_out_cls"])
    close(out["gt_boxes"], rc["roi_out_gtb"]); close(out["gt_confid"], rc["roi_out_conf"]); close(out["gt_loc_std"], rc["roi_out_std"])


def test_known_answers_d2_primitives():
    """Textbook known-answer cases for the [D2-recall] primitives (no upstream vectors exist)."""
    # RoIAlign aligned=True on a ramp image: bilinear interpolation of a linear function is exact
    H, W = 12, 16
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    feat = (2.0 * xx + 3.0 * yy)[None]
    roi = torch.tensor([[2.0, 1.0, 10.0, 8.0]])
    out = O.roi_align(feat, roi, 1.0, 7)[0, 0]
    bw, bh = 8.0 / 7, 7.0 / 7
    exp = torch.tensor([[2.0 * (2.0 - 0.5 + (j + 0.5) * bw) + 3.0 * (1.0 - 0.5 + (i + 0.5) * bh) for j in range(7)] for i in range(7)])
    assert torch.allclose(out, exp, atol=1e-4)
    # NMS on hand-made boxes with a score tie: lower index wins, strict '>' threshold
    b = torch.tensor([[0, 0, 10, 10], [0, 0, 10, 10], [20, 20, 30, 30], [0, 0, 10, 5.0]])
    s = torch.tensor([0.5, 0.5, 0.9, 0.4])
    assert O.nms(b, s, 0.5).tolist() == [2, 0, 3]        # box 3 has IoU exactly 0.5 with box 0: 0.5 > 0.5 is False -> kept
    assert O.nms(b, s, 0.49).tolist() == [2, 0]
    assert O.nms(b[[0, 3]], s[[0, 3]], 0.5).tolist() == [0, 1]
    # focal loss closed form at x = 0: ce = ln 2, p_t = .5 -> alpha_t * .25 * ln2
    v = O.sigmoid_focal_loss(torch.zeros(2), torch.tensor([1.0, 0.0]), 0.25, 2.0)
    assert torch.allclose(v, torch.tensor([0.25, 0.75]) * 0.25 * np.log(2.0), atol=1e-7)
    # matcher: low-quality matches promote the best anchor of every gt
    iou = torch.tensor([[0.1, 0.4, 0.75], [0.2, 0.05, 0.1]])
    idx, lab = O.matcher(iou, [0.3, 0.7], [0, -1, 1], True)
    assert idx.tolist() == [1, 0, 0] and lab.tolist() == [1, -1, 1]
    # anchors: size 32, ratio 0.5 -> w = 45.2548, h = 22.6274 centred on the shift
    a = O.make_anchors([(1, 2)], [16], sizes=(32,))[0]
    assert torch.allclose(a[0], torch.tensor([-22.6274, -11.3137, 22.6274, 11.3137]), atol=1e-3)
    assert torch.allclose(a[3], a[0] + torch.tensor([16.0, 0, 16.0, 0]))


def test_fast_roi_align_equals_pinned_form():
    """bench.py's cpu_baseline leg switches the oracle's RoIAlign to its separable form (same arithmetic regrouped, a backward that
    touches each ROI's window once): forward and the gradient w.r.t. the feature map equal the pinned per-sample form."""
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(6, 23, 31, generator=g)
    rois = torch.tensor([[3.0, 4.0, 60.0, 50.0], [-20.0, -8.0, 30.0, 200.0], [100.0, 70.0, 124.0, 92.0], [5.0, 5.0, 5.5, 5.2],
                         [110.0, 80.0, 140.0, 100.0], [0.0, 0.0, 124.0, 92.0], [40.0, 40.0, 40.0, 40.0]])
    w = torch.randn(len(rois), 6, 7, 7, generator=g)
    fa = feat.clone().requires_grad_(True)
    ya = O.roi_align(fa, rois, 0.25)
    (ya * w).sum().backward()
    fb = feat.clone().requires_grad_(True)
    yb = O._RoiAlignSeparable.apply(fb, rois, 0.25, 7)
    (yb * w).sum().backward()
    assert torch.allclose(ya, yb, rtol=1e-5, atol=1e-6)
    assert torch.allclose(fa.grad, fb.grad, rtol=1e-5, atol=1e-5)
    assert float(ya.abs().sum()) > 0
