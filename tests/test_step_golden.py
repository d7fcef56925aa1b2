"""The oracle's whole-iteration restatements (`fcos_semisup_step`, `rcnn_semisup_step`) against the goldens produced by EXECUTING
the reference's own `UBTeacherTrainer.run_step_full_semisup` / `UBRCNNTeacherTrainer.run_step_full_semisup`
(engine/trainer.py:181-429, :786-912; tests/golden/gen_golden_step.py): every record_dict entry, the weighted loss the
reference back-propagates, the pseudo-label sets, the teacher after EMA (bit exact) and the student after SGD.
This pins the orchestration (EMA placement, thresholds, label surgery, loss weights, key renaming) that rows a1 / a2 of
SURVEY 8(a) own; the GPU product is checked against the same files in tests/test_*_step_gpu.py."""
import numpy as np
import torch

from oracle import utv2_oracle as O
from tests.utv2_testutil import (check_state_fingerprints, golden_batches, golden_init_state, golden_record, load_step_golden,
                                 rcnn_tune, tune_state_for_pseudo_labels)


def _weighted_fcos(rec, lu, lr):
    tot = 0.0
    for k, v in rec.items():
        if k[:4] != "loss":
            continue
        if k in ("loss_fcos_loc",):
            tot += v / (lr + 1.0)
        elif k == "loss_fcos_loc_pseudo":
            tot += v * lr / (lr + 1.0)
        elif k.endswith("_pseudo"):
            tot += v * lu / (lu + 1.0)
        else:
            tot += v / (lu + 1.0)
    return tot


def test_fcos_step_oracle_vs_reference_trainer():
    d = load_step_golden("fcos")
    cfg, sd0 = golden_init_state("fcos", d)
    _, orac = golden_batches(d, "cpu")
    sd_s = tune_state_for_pseudo_labels(sd0, [x["image"] for x in orac[3]])
    sd_t = dict(sd_s)
    sd_t["proposal_generator.fcos_head.bbox_pred_std.bias"] = torch.full((4,), -3.0)
    S = cfg.SEMISUPNET
    rec, new_s, new_t, grads, bufs, pseudo = O.fcos_semisup_step(
        O.FCOSCfg(), sd_s, sd_t, orac, keep_rate=S.EMA_KEEP_RATE, lam_u=S.UNSUP_LOSS_WEIGHT, lam_r=S.UNSUP_REG_LOSS_WEIGHT,
        thr_cls=S.BBOX_THRESHOLD, thr_reg=S.BBOX_THRESHOLD_REG, lr=float(d["lr"]), momentum=cfg.SOLVER.MOMENTUM,
        wd=cfg.SOLVER.WEIGHT_DECAY, mean=sd0["pixel_mean"], pix_std=sd0["pixel_std"])
    ref = golden_record(d)
    for k, v in ref.items():
        if k in ("data_time", "total_loss"):
            continue
        assert k in rec, k
        assert abs(rec[k] - v) <= 1e-5 * max(abs(v), 1e-6), (k, rec[k], v)
    assert set(rec) == set(ref) - {"data_time", "total_loss"}
    # the reference's logged total (plain sum of the loss entries) and the weighted objective it back-propagates
    assert abs(sum(v for k, v in rec.items() if k[:4] == "loss") - ref["total_loss"]) <= 1e-5 * ref["total_loss"]
    assert abs(_weighted_fcos(rec, S.UNSUP_LOSS_WEIGHT, S.UNSUP_REG_LOSS_WEIGHT) - float(d["losses"])) <= 1e-5 * float(d["losses"])
    for name, sets in zip(("pcls", "preg"), pseudo):
        for i, p in enumerate(sets):
            assert np.array_equal(p["classes"].numpy(), d["%s%d_classes" % (name, i)])
            np.testing.assert_allclose(p["boxes"].numpy(), d["%s%d_boxes" % (name, i)], rtol=0, atol=1e-4)
            np.testing.assert_allclose(p["scores"].numpy(), d["%s%d_scores" % (name, i)], rtol=2e-5)
            np.testing.assert_allclose(p["reg_pred_std"].numpy(), d["%s%d_std" % (name, i)], rtol=1e-4, atol=1e-5)
    check_state_fingerprints(d, "teacher", new_t, 0.0, exact=True)
    check_state_fingerprints(d, "student", new_s, 2e-6)


def test_rcnn_step_oracle_vs_reference_trainer():
    d = load_step_golden("rcnn")
    cfg, sd0 = golden_init_state("rcnn", d)
    _, orac = golden_batches(d, "cpu")
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
    pstd = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    sd_s = rcnn_tune(sd0, [x["image"] for x in orac[3]], mean, pstd)
    sd_t = dict(sd_s)
    sd_t["roi_heads.box_predictor.bbox_pred_std.bias"] = torch.full((4,), -3.0)
    post = int(cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN)
    K = {k: torch.from_numpy(d["keys_" + k]) for k in ("rpn_sup", "roi_sup", "rpn_unsup", "roi_unsup")}

    def compact(name):
        return [(lambda i: lambda nprop, ngt: torch.cat((K[name][i, :nprop], K[name][i, post:post + ngt])))(i)
                for i in range(K[name].shape[0])]
    keys = dict(rpn_sup=K["rpn_sup"], rpn_unsup=K["rpn_unsup"], roi_sup=compact("roi_sup"), roi_unsup=compact("roi_unsup"))
    S = cfg.SEMISUPNET
    rec, new_s, new_t, grads, pseudo = O.rcnn_semisup_step(
        sd_s, sd_t, orac, keys, keep_rate=S.EMA_KEEP_RATE, lam_u=S.UNSUP_LOSS_WEIGHT, lam_r=S.UNSUP_REG_LOSS_WEIGHT,
        thr=S.BBOX_THRESHOLD, lr=float(d["lr"]), momentum=cfg.SOLVER.MOMENTUM, wd=cfg.SOLVER.WEIGHT_DECAY, mean=mean, pix_std=pstd)
    ref = golden_record(d)
    for k, v in ref.items():
        if k in ("data_time", "total_loss"):
            continue
        assert abs(rec[k] - v) <= 1e-5 * max(abs(v), 1e-6), (k, rec[k], v)
    assert set(rec) == set(ref) - {"data_time", "total_loss"}
    assert abs(sum(v for k, v in rec.items() if k[:4] == "loss") - ref["total_loss"]) <= 1e-5 * ref["total_loss"]
    lu, lr_ = S.UNSUP_LOSS_WEIGHT, S.UNSUP_REG_LOSS_WEIGHT
    tot = sum(v * (0.0 if k == "loss_rpn_loc_pseudo" else lr_ if k == "loss_box_reg_pseudo" else lu if k.endswith("pseudo") else 1.0)
              for k, v in rec.items() if k[:4] == "loss")
    assert abs(tot - float(d["losses"])) <= 1e-5 * float(d["losses"])
    for i, p in enumerate(pseudo):
        assert np.array_equal(p["classes"].numpy(), d["pseudo%d_classes" % i])
        np.testing.assert_allclose(p["boxes"].numpy(), d["pseudo%d_boxes" % i], rtol=0, atol=1e-4)
        np.testing.assert_allclose(p["scores"].numpy(), d["pseudo%d_scores" % i], rtol=2e-5)
        np.testing.assert_allclose(p["pred_boxes_std"].numpy(), d["pseudo%d_std" % i], rtol=1e-4, atol=1e-5)
    check_state_fingerprints(d, "teacher", new_t, 0.0, exact=True)
    check_state_fingerprints(d, "student", new_s, 2e-6)


def _eval_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fcos_eval.npz"), allow_pickle=False)


def eval_golden_state(d):
    """the weights the eval golden ran on: product CPU init (fingerprint-checked) + the stored cls_logits rescale"""
    from tests.utv2_testutil import golden_init_state
    cfg, sd = golden_init_state("fcos", d)
    p = "proposal_generator.fcos_head.cls_logits"
    g = torch.Generator().manual_seed(0)
    sd = dict(sd)
    sd[p + ".weight"] = torch.randn(sd[p + ".weight"].shape, generator=g) * 0.01 * float(d["cls_scale"])
    sd[p + ".bias"] = torch.full_like(sd[p + ".bias"], float(d["cls_bias"]))
    return cfg, sd


EVAL_VARIANTS = {"default": dict(nms="cls_n_ctr"), "testth": dict(nms="cls_n_loc", pre_nms_thresh=0.2, pre_nms_topk=60, post_nms_topk=12)}


def test_fcos_eval_oracle_vs_reference_golden():
    """FCOS test-mode inference (tests/golden/gen_golden_eval.py: the reference's eval-mode OneStageDetector.forward +
    detector_postprocess, one_stage_detector.py:16-43,136-145,230-240): the oracle's forward + predict at the *_TEST thresholds +
    rescale reproduces the kept detections (classes exact, scores 1e-5, boxes 1e-4)."""
    d = _eval_golden()
    _, sd = eval_golden_state(d)
    images = [torch.from_numpy(d["img%d" % i]) for i in range(2)]
    with torch.no_grad():
        out = O.fcos_forward(sd, images, sd["pixel_mean"], sd["pixel_std"])
    for name, v in EVAL_VARIANTS.items():
        v = dict(v)
        nms = v.pop("nms")
        dets = O.fcos_predict(O.FCOSCfg(**v), *out[:4], out[4], out[5], nms)
        for i, det in enumerate(dets):
            oh, ow = [int(x) for x in d["orig%d" % i]]
            h, w = det["image_size"]
            b = det["boxes"].clone()
            b[:, 0::2] = (b[:, 0::2] * (ow / w)).clamp(0, ow)
            b[:, 1::2] = (b[:, 1::2] * (oh / h)).clamp(0, oh)
            keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
            assert np.array_equal(det["classes"][keep].numpy(), d["%s_classes%d" % (name, i)]), (name, i)
            np.testing.assert_allclose(det["scores"][keep].numpy(), d["%s_scores%d" % (name, i)], rtol=1e-5)
            np.testing.assert_allclose(b[keep].numpy(), d["%s_boxes%d" % (name, i)], rtol=1e-4, atol=1e-3)


def _rcnn_eval_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rcnn_eval.npz"), allow_pickle=False)


def rcnn_eval_golden_state(d):
    """the weights the Faster-RCNN eval golden ran on: product CPU init (fingerprint-checked) + the stored rescaled prediction layers"""
    from tests.utv2_testutil import golden_init_state
    cfg, sd = golden_init_state("rcnn", d)
    sd = dict(sd)
    for k in d["changed"]:
        sd[str(k)] = torch.from_numpy(d["state::" + str(k)].copy())
    return cfg, sd


def test_rcnn_eval_oracle_vs_reference_golden():
    """Faster-RCNN test-mode inference (tests/golden/gen_golden_eval.py::gen_rcnn_eval: the reference's eval-mode
    TwoStagePseudoLabGeneralizedRCNN.forward -> PseudoLabRPN.forward -> StandardROIHeadsPseudoLab.forward -> predictor.inference,
    meta_arch/rcnn.py:8-13, proposal_generator/rpn.py:21-76, roi_heads/roi_heads.py:75-139, roi_heads/fast_rcnn.py:1094-1125): the oracle's
    rcnn_inference reproduces the kept detections - classes and order exact, scores 1e-5, boxes 1e-4, pred_boxes_std 1e-5."""
    d = _rcnn_eval_golden()
    cfg, sd = rcnn_eval_golden_state(d)
    images = [torch.from_numpy(d["img%d" % i]) for i in range(2)]
    mean, pstd = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1), torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    origs = [tuple(int(v) for v in d["orig%d" % i]) for i in range(2)]
    with torch.no_grad():
        dets, props = O.rcnn_inference(sd, images, mean, pstd, origs, int(d["pre_topk"]), int(d["post_topk"]))
    for i, det in enumerate(dets):
        assert len(props[i]["boxes"]) == int(d["nprop%d" % i])
        assert np.array_equal(det["classes"].numpy(), d["classes%d" % i]) and len(det["classes"]) > 0
        np.testing.assert_allclose(det["scores"].numpy(), d["scores%d" % i], rtol=1e-5)
        np.testing.assert_allclose(det["boxes"].numpy(), d["boxes%d" % i], rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(det["pred_boxes_std"].numpy(), d["std%d" % i], rtol=1e-5, atol=1e-6)
