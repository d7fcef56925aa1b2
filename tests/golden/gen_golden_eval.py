"""Golden vectors of the FCOS evaluation path: the reference's OWN eval-mode `OneStageDetector.forward` executed here (CPU, build
container only), SURVEY 8(f) rank 3.

The reference's `inference_on_dataset` (evaluation/evaluator.py:14-104) switches the model to eval mode and calls
`model(inputs, nms_method=cfg.MODEL.FCOS.NMS_CRITERIA_TEST)`; `OneStageDetector.forward` (one_stage_detector.py:230-240) falls through to
`PseudoProposalNetwork.forward` (:70-145): backbone -> `FCOS.forward` in eval mode (fcos/fcos.py: `predict_proposals` with the `*_TEST`
thresholds, fcos_outputs.py:1058-1065) -> `detector_postprocess` per image to the dict's `height` / `width` (:16-43,136-145) ->
`{"instances": ...}`.  All of that is the reference's code, imported in place; the Detectron2 backbone is the oracle's functional
ResNet-50 + FPN (as in gen_golden_step.py) and Detectron2's own `detector_postprocess` [D2-recall: boxes scaled by
(out_w / in_w, out_h / in_h), clipped to the output size, empty boxes dropped] is a stand-in written here.

Stored (tests/golden/fcos_eval.npz): the input images (two different sizes - ImageList padding - with original sizes that differ
from the network input), per image the returned pred_boxes / scores / pred_classes, for two configs: the shipped defaults and a
variant whose *_TEST thresholds differ from the *_TRAIN ones (proves the eval path reads *_TEST).  Initial weights: the product's
CPU initialisation under the stored seed + the cls_logits rescale of the parity tests.

    python tests/golden/gen_golden_eval.py
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import gen_golden as G  # noqa: E402
import gen_golden_step as S  # noqa: E402
from oracle import utv2_oracle as O  # noqa: E402

VARIANTS = {
    "default": [],
    "testth": ["MODEL.FCOS.INFERENCE_TH_TEST", 0.2, "MODEL.FCOS.PRE_NMS_TOPK_TEST", 60, "MODEL.FCOS.POST_NMS_TOPK_TEST", 12,
               "MODEL.FCOS.NMS_CRITERIA_TEST", "cls_n_loc"],
}


def d2_detector_postprocess(structures):
    def post(results, output_height, output_width, mask_threshold=0.5):
        sx, sy = output_width / results.image_size[1], output_height / results.image_size[0]
        out = structures.Instances((output_height, output_width))
        for k, v in results.get_fields().items():
            out.set(k, v)
        b = out.pred_boxes.tensor.clone()
        b[:, 0::2] *= sx
        b[:, 1::2] *= sy
        b[:, 0::2] = b[:, 0::2].clamp(min=0, max=output_width)
        b[:, 1::2] = b[:, 1::2].clamp(min=0, max=output_height)
        out.pred_boxes = structures.Boxes(b)
        keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
        return out[keep]
    return post


def main():
    structures, fo, pg, tr = G.install_shims()
    fcos_mod, osd, _ = S.load_ref_fcos_modules()
    osd.d2_postprocesss = d2_detector_postprocess(structures)
    cfg, sd0 = S.product_cfg_and_state("fcos", seed=0)
    g = torch.Generator().manual_seed(41)

    def img(H, W):
        base = torch.rand(3, H // 8 + 1, W // 8 + 1, generator=g)
        im = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear", align_corners=False)[0]
        return (im * 255 + torch.randn(3, H, W, generator=g) * 20).clamp(0, 255).to(torch.uint8)
    sizes = [(96, 128), (80, 112)]
    origs = [(144, 192), (100, 140)]        # height / width the detections are rescaled to (1.5x, 1.25x)
    images = [img(*s) for s in sizes]
    mean, pstd = sd0["pixel_mean"], sd0["pixel_std"]
    p = "proposal_generator.fcos_head.cls_logits"
    gw = torch.Generator().manual_seed(0)
    sd = OrderedDict(sd0)
    sd[p + ".weight"] = torch.randn(sd0[p + ".weight"].shape, generator=gw) * 0.01
    sd[p + ".bias"] = torch.zeros_like(sd0[p + ".bias"])
    with torch.no_grad():
        s = torch.cat([x.reshape(-1) for x in O.fcos_forward(sd, images, mean, pstd)[0]]).std().item()
    sd[p + ".weight"] = sd[p + ".weight"] * (1.5 / max(s, 1e-12))
    sd[p + ".bias"] = torch.full_like(sd[p + ".bias"], -3.0)
    d = {"seed_state": 0, "cls_scale": 1.5 / max(s, 1e-12), "cls_bias": -3.0}
    S.state_fingerprints("init", sd0, d)
    for i, (im, (oh, ow)) in enumerate(zip(images, origs)):
        d["img%d" % i] = im.numpy()
        d["orig%d" % i] = np.array([oh, ow])
    for name, over in VARIANTS.items():
        c = cfg.clone()
        c.defrost()
        c.merge_from_list(over)
        model = S.build_ref_one_stage(c, sd, fcos_mod, osd)
        model.eval()
        batch = []
        for im, (oh, ow) in zip(images, origs):
            inst = structures.Instances((oh, ow))
            inst.gt_boxes = structures.Boxes(torch.tensor([[4.0, 5.0, 60.0, 70.0]]))
            inst.gt_classes = torch.tensor([3])
            batch.append({"image": im, "height": oh, "width": ow, "instances": inst})
        with torch.no_grad():
            out = model(batch, nms_method=c.MODEL.FCOS.NMS_CRITERIA_TEST)
        for i, r in enumerate(out):
            x = r["instances"]
            d["%s_boxes%d" % (name, i)] = G.npy(x.pred_boxes.tensor)
            d["%s_scores%d" % (name, i)] = G.npy(x.scores)
            d["%s_classes%d" % (name, i)] = G.npy(x.pred_classes)
            d["%s_size%d" % (name, i)] = np.array(x.image_size)
            print(name, i, "detections:", len(x), "image_size", x.image_size, "top score %.4f" % float(x.scores.max()) if len(x) else "")
            assert len(x) > 0
        d[name + "_overrides"] = np.array([str(v) for v in over])
    np.savez_compressed(os.path.join(HERE, "fcos_eval.npz"), **d)
    print("fcos_eval.npz:", len(d), "arrays")


def gen_rcnn_eval():
    """Faster-RCNN evaluation path (round 4): the reference's own eval-mode call chain executed here -
    `TwoStagePseudoLabGeneralizedRCNN.forward` (meta_arch/rcnn.py:8-13: not training and not val_mode -> self.inference) ->
    `PseudoLabRPN.forward` (proposal_generator/rpn.py:21-76, inference branch: no losses, predict_proposals) ->
    `StandardROIHeadsPseudoLab.forward` / `_forward_box` (roi_heads/roi_heads.py:75-139, the not-training branch) ->
    `FastRCNNFocaltLossBoundaryVarOutputLayers.inference` / predict_boxes / predict_probs / predict_boxes_std
    (roi_heads/fast_rcnn.py:1094-1125,1162-1225).  Detectron2 pieces are stand-ins [D2-recall]: `GeneralizedRCNN.inference` (preprocess,
    backbone, proposal generator, roi heads, `_postprocess` = detector_postprocess per image), the RPN's anchor generator / head /
    `predict_proposals` (find_top_rpn_proposals with the *_TEST top-k) and the box pooler / head run on the oracle's functional
    Faster-RCNN.  -> tests/golden/rcnn_eval.npz"""
    import types
    structures, fo, pg, tr = G.install_shims()
    REF = G.REF
    import detectron2.modeling.proposal_generator as d2pg
    import detectron2.modeling.meta_arch.rcnn as d2rcnn
    d2pg.RPN = type("RPN", (), {})
    d2rcnn.GeneralizedRCNN = type("GeneralizedRCNN", (), {})
    br = G._load("ubteacher.modeling.box_regression", REF + "/ubteacher/modeling/box_regression.py")
    fr = G._load("ubteacher.modeling.roi_heads.fast_rcnn", REF + "/ubteacher/modeling/roi_heads/fast_rcnn.py")
    rh = G._load("ubteacher.modeling.roi_heads.roi_heads", REF + "/ubteacher/modeling/roi_heads/roi_heads.py")
    rp = G._load("ubteacher.modeling.proposal_generator.rpn", REF + "/ubteacher/modeling/proposal_generator/rpn.py")
    ma = G._load("ubteacher.modeling.meta_arch.rcnn", REF + "/ubteacher/modeling/meta_arch/rcnn.py")
    Boxes, Instances = structures.Boxes, structures.Instances
    cfg, sd0 = S.product_cfg_and_state("rcnn", seed=0)
    g = torch.Generator().manual_seed(43)

    def img(H, W):
        base = torch.rand(3, H // 8 + 1, W // 8 + 1, generator=g)
        im = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear", align_corners=False)[0]
        return (im * 255 + torch.randn(3, H, W, generator=g) * 20).clamp(0, 255).to(torch.uint8)
    sizes = [(96, 128), (80, 112)]
    origs = [(144, 192), (100, 140)]
    images = [img(*s) for s in sizes]
    mean, pstd = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1), torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    sd = S.rcnn_tune(sd0, images, mean, pstd)
    PRE, POST = cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, cfg.MODEL.RPN.POST_NMS_TOPK_TEST
    F = torch.nn.functional

    # ---- Detectron2 stand-ins around the reference's classes ---------------------------------------------------------------------
    tf = br.Box2BoxXYXYTransform(weights=cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS)
    pred = types.SimpleNamespace(num_classes=80, box2box_transform=tf, test_score_thresh=cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST,
                                 test_nms_thresh=cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST, test_topk_per_image=cfg.TEST.DETECTIONS_PER_IMAGE)
    cls_ = fr.FastRCNNFocaltLossBoundaryVarOutputLayers
    for name in ("inference", "predict_boxes", "predict_boxes_std", "predict_probs"):
        setattr(pred, name, types.MethodType(getattr(cls_, name), pred))
    p = "roi_heads.box_predictor."

    class Predictor:
        """the three Linear heads of the reference predictor (fast_rcnn.py:812-829) as a callable + its own inference methods"""
        def __call__(self, x):
            return (F.linear(x, sd[p + "cls_score.weight"], sd[p + "cls_score.bias"]), F.linear(x, sd[p + "bbox_pred.weight"], sd[p + "bbox_pred.bias"]),
                    F.linear(x, sd[p + "bbox_pred_std.weight"], sd[p + "bbox_pred_std.bias"]))
        inference = staticmethod(pred.inference)

    def box_head(x):
        x = x.flatten(1)
        x = F.relu(F.linear(x, sd["roi_heads.box_head.fc1.weight"], sd["roi_heads.box_head.fc1.bias"]))
        return F.relu(F.linear(x, sd["roi_heads.box_head.fc2.weight"], sd["roi_heads.box_head.fc2.bias"]))
    roi = types.SimpleNamespace(training=False, box_in_features=["p2", "p3", "p4", "p5"], train_on_pred_boxes=False,
                                box_pooler=lambda feats, boxes: O.roi_pool(feats, [b.tensor for b in boxes]), box_head=box_head,
                                box_predictor=Predictor())
    S.bind(roi, rh.StandardROIHeadsPseudoLab, ("forward", "_forward_box"))

    class AnchorGen:
        box_dim = 4

        def __call__(self, feats):
            return [Boxes(a) for a in O.make_anchors([(f.shape[2], f.shape[3]) for f in feats], [4, 8, 16, 32, 64])]

    def rpn_head(feats):
        obj, dl = [], []
        for f in feats:
            t = F.relu(F.conv2d(f, sd["proposal_generator.rpn_head.conv.weight"], sd["proposal_generator.rpn_head.conv.bias"], 1, 1))
            obj.append(F.conv2d(t, sd["proposal_generator.rpn_head.objectness_logits.weight"], sd["proposal_generator.rpn_head.objectness_logits.bias"]))
            dl.append(F.conv2d(t, sd["proposal_generator.rpn_head.anchor_deltas.weight"], sd["proposal_generator.rpn_head.anchor_deltas.bias"]))
        return obj, dl

    def predict_proposals(anchors, logits, deltas, image_sizes):
        props = O.find_top_rpn_proposals([a.tensor for a in anchors], logits, deltas, image_sizes, PRE, POST)
        out = []
        for q, sz in zip(props, image_sizes):
            x = Instances(sz)
            x.proposal_boxes = Boxes(q["boxes"]); x.objectness_logits = q["logits"] if "logits" in q else torch.zeros(len(q["boxes"]))
            out.append(x)
        return out
    rpn = types.SimpleNamespace(training=False, in_features=["p2", "p3", "p4", "p5", "p6"], anchor_generator=AnchorGen(), rpn_head=rpn_head,
                                predict_proposals=predict_proposals, loss_weight={})
    S.bind(rpn, rp.PseudoLabRPN, ("forward",))
    post = d2_detector_postprocess(structures)

    def d2_inference(batched_inputs):
        """Detectron2 GeneralizedRCNN.inference(detected_instances=None, do_postprocess=True) [D2-recall]"""
        feats, image_sizes = O.rcnn_backbone(sd, [x["image"] for x in batched_inputs], mean, pstd)
        imgs = types.SimpleNamespace(image_sizes=image_sizes)
        proposals, _ = rpn.forward(imgs, feats, None)
        results, _ = roi.forward(imgs, feats, proposals, None)
        return [{"instances": post(r, x.get("height", sz[0]), x.get("width", sz[1]))} for r, x, sz in zip(results, batched_inputs, image_sizes)], proposals
    model = types.SimpleNamespace(training=False)
    cap = {}

    def inference(batched_inputs):
        out, cap["proposals"] = d2_inference(batched_inputs)
        return out
    model.inference = inference
    batch = [{"image": im, "height": oh, "width": ow} for im, (oh, ow) in zip(images, origs)]
    with torch.no_grad():
        out = ma.TwoStagePseudoLabGeneralizedRCNN.forward(model, batch)       # eval mode, val_mode False -> self.inference
    d = {"seed_state": 0, "pre_topk": PRE, "post_topk": POST}
    S.state_fingerprints("init", sd0, d)
    changed = [k for k in sd0 if not torch.equal(sd0[k], sd[k])]
    d["changed"] = np.asarray(changed)
    for k in changed:
        d["state::" + k] = G.npy(sd[k])
    for i, (im, (oh, ow)) in enumerate(zip(images, origs)):
        d["img%d" % i] = im.numpy()
        d["orig%d" % i] = np.array([oh, ow])
        x = out[i]["instances"]
        d["boxes%d" % i], d["scores%d" % i], d["classes%d" % i] = G.npy(x.pred_boxes.tensor), G.npy(x.scores), G.npy(x.pred_classes)
        d["std%d" % i] = G.npy(x.pred_boxes_std)
        d["size%d" % i] = np.array(x.image_size)
        d["nprop%d" % i] = np.int64(len(cap["proposals"][i]))
        print("rcnn eval image", i, "detections:", len(x), "image_size", x.image_size, "proposals", len(cap["proposals"][i]),
              "top score %.4f" % float(x.scores.max()))
        assert len(x) > 0
    np.savez_compressed(os.path.join(HERE, "rcnn_eval.npz"), **d)
    print("rcnn_eval.npz:", len(d), "arrays")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rcnn":
        gen_rcnn_eval()
    else:
        main()
        gen_rcnn_eval()
