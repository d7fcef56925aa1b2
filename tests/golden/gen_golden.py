"""Generate golden vectors by EXECUTING THE REFERENCE'S OWN MODULES (CPU) in the build container.

Runs only where /root/reference exists (never on the GPU box).  Nothing from the reference is
copied: its hot-path modules are imported in place by file path, Detectron2 / fvcore (absent here)
are replaced by minimal stand-ins injected into sys.modules, and the hard-coded `.cuda()` calls
 are neutralised with a `torch.Tensor.cuda = identity` shim.  The outputs
(inputs + expected outputs, plain arrays) are written to tests/golden/*.npz.

    python tests/golden/gen_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import utv2_oracle as O  # noqa: E402  (stand-ins for the D2 primitives come from the oracle)


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _Stub(types.ModuleType):
    """Any attribute is an empty class (good enough for unused imports / base classes)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {})
        setattr(self, name, cls)
        return cls


def install_shims():
    torch.Tensor.cuda = lambda self, *a, **k: self
    structures = _load("_utv2_structs", os.path.join(ROOT, "unbiased-teacher-v2_amd", "ubteacher", "d2", "structures.py"))
    d2 = _Stub("detectron2")
    d2.__path__ = []
    layers = _Stub("detectron2.layers")
    layers.cat = lambda tensors, dim=0: tensors[0] if len(tensors) == 1 else torch.cat(tensors, dim)
    layers.batched_nms = O.batched_nms
    layers.nonzero_tuple = lambda x: x.nonzero().unbind(1)
    layers.cross_entropy = torch.nn.functional.cross_entropy
    st = _Stub("detectron2.structures")
    st.Boxes, st.Instances, st.ImageList = structures.Boxes, structures.Instances, structures.ImageList
    st.pairwise_iou = lambda a, b: O.pairwise_iou(a.tensor, b.tensor)
    sti = _Stub("detectron2.structures.instances")
    sti.Instances = structures.Instances
    utils = _Stub("detectron2.utils")
    utils.__path__ = []
    comm = _Stub("detectron2.utils.comm")
    comm.get_world_size = lambda: 1
    comm.get_local_rank = lambda: 0
    comm.is_main_process = lambda: True
    mods = {"detectron2": d2, "detectron2.layers": layers, "detectron2.structures": st,
            "detectron2.structures.instances": sti, "detectron2.utils": utils, "detectron2.utils.comm": comm}
    for n in ("detectron2.engine", "detectron2.engine.train_loop", "detectron2.evaluation", "detectron2.utils.events",
              "detectron2.config", "detectron2.modeling", "detectron2.modeling.proposal_generator",
              "detectron2.modeling.proposal_generator.build", "detectron2.checkpoint", "detectron2.data",
              "detectron2.utils.logger", "detectron2.utils.memory", "detectron2.modeling.box_regression",
              "detectron2.modeling.roi_heads", "detectron2.modeling.roi_heads.fast_rcnn",
              "detectron2.modeling.proposal_generator.rpn", "detectron2.modeling.proposal_generator.proposal_utils",
              "detectron2.modeling.sampling", "detectron2.modeling.matcher", "detectron2.modeling.poolers",
              "detectron2.modeling.roi_heads.box_head", "detectron2.modeling.roi_heads.roi_heads",
              "detectron2.modeling.meta_arch", "detectron2.modeling.meta_arch.build", "detectron2.modeling.meta_arch.rcnn"):
        m = _Stub(n)
        m.__path__ = []
        mods[n] = m
    class _Reg:
        def register(self, obj=None):
            return (lambda o: o) if obj is None else obj
    mods["detectron2.config"].configurable = lambda f=None, **kw: f if f is not None else (lambda g: g)
    mods["detectron2.modeling.roi_heads"].ROI_HEADS_REGISTRY = _Reg()
    mods["detectron2.modeling.proposal_generator.build"].PROPOSAL_GENERATOR_REGISTRY = _Reg()
    mods["detectron2.modeling.meta_arch.build"].META_ARCH_REGISTRY = _Reg()
    mods["detectron2.modeling.roi_heads.fast_rcnn"].FastRCNNOutputLayers = torch.nn.Module
    mods["detectron2.modeling.roi_heads.fast_rcnn"]._log_classification_stats = lambda *a, **k: None
    mods["detectron2.modeling.roi_heads.fast_rcnn"].fast_rcnn_inference = _d2_fast_rcnn_inference
    mods["detectron2.modeling.box_regression"]._dense_box_regression_loss = _d2_dense_box_regression_loss
    mods["detectron2.modeling.proposal_generator.proposal_utils"].add_ground_truth_to_proposals = _d2_add_gt(structures)
    mods["detectron2.utils.memory"].retry_if_cuda_oom = lambda f: f

    class _Storage:
        def put_scalar(self, *a, **k):
            pass
    mods["detectron2.utils.events"].get_event_storage = lambda: _Storage()
    fv = _Stub("fvcore")
    fv.__path__ = []
    fvnn = _Stub("fvcore.nn")
    fvnn.__path__ = []
    fvnn.sigmoid_focal_loss_jit = lambda inputs, targets, alpha=-1, gamma=2, reduction="none": O.sigmoid_focal_loss(inputs, targets, alpha, gamma)
    fvnn.smooth_l1_loss = _fv_smooth_l1
    fvnn.giou_loss = _fv_giou
    mods.update({"fvcore": fv, "fvcore.nn": fvnn, "fvcore.nn.precise_bn": _Stub("fvcore.nn.precise_bn")})
    sys.modules.update(mods)
    for n, m in mods.items():  # parent.attr = child, so `import a.b.c as x` resolves
        if "." in n:
            parent, child = n.rsplit(".", 1)
            setattr(mods[parent], child, m)
    # the reference package skeleton (its __init__ files import detectron2.config -> not executed)
    for n in ("ubteacher", "ubteacher.layers", "ubteacher.utils", "ubteacher.modeling", "ubteacher.modeling.fcos",
              "ubteacher.modeling.meta_arch", "ubteacher.engine"):
        m = types.ModuleType(n)
        m.__path__ = [os.path.join(REF, *n.split("."))]
        sys.modules[n] = m
    for n in ("ubteacher.checkpoint", "ubteacher.checkpoint.detection_checkpoint", "ubteacher.data", "ubteacher.data.build",
              "ubteacher.data.dataset_mapper", "ubteacher.evaluation", "ubteacher.evaluation.evaluator", "ubteacher.solver",
              "ubteacher.solver.build"):
        m = _Stub(n)
        m.__path__ = []
        sys.modules[n] = m
    il = _load("ubteacher.layers.iou_loss", REF + "/ubteacher/layers/iou_loss.py")
    kl = _load("ubteacher.layers.kl_loss", REF + "/ubteacher/layers/kl_loss.py")
    ml = _load("ubteacher.layers.ml_nms", REF + "/ubteacher/layers/ml_nms.py")
    L = sys.modules["ubteacher.layers"]
    L.IOULoss, L.KLLoss, L.NLLoss, L.ml_nms = il.IOULoss, kl.KLLoss, kl.NLLoss, ml.ml_nms
    _load("ubteacher.utils.comm", REF + "/ubteacher/utils/comm.py")
    fo = _load("ubteacher.modeling.fcos.fcos_outputs", REF + "/ubteacher/modeling/fcos/fcos_outputs.py")
    pg = _load("ubteacher.modeling.pseudo_generator", REF + "/ubteacher/modeling/pseudo_generator.py")
    _load("ubteacher.modeling.meta_arch.ts_ensemble", REF + "/ubteacher/modeling/meta_arch/ts_ensemble.py")
    tr = _load("ubteacher.engine.trainer", REF + "/ubteacher/engine/trainer.py")
    sys.modules["ubteacher.modeling.roi_heads"] = types.ModuleType("ubteacher.modeling.roi_heads")
    sys.modules["ubteacher.modeling.roi_heads"].__path__ = [REF + "/ubteacher/modeling/roi_heads"]
    sys.modules["ubteacher.modeling.proposal_generator"] = types.ModuleType("ubteacher.modeling.proposal_generator")
    sys.modules["ubteacher.modeling.proposal_generator"].__path__ = [REF + "/ubteacher/modeling/proposal_generator"]
    return structures, fo, pg, tr


def _d2_fast_rcnn_inference(boxes, scores, image_shapes, score_thresh, nms_thresh, topk_per_image):
    """stand-in for D2 fast_rcnn_inference built on the oracle's restatement"""
    structures = sys.modules["_utv2_structs"]
    res, kept = [], []
    for b, s, shp in zip(boxes, scores, image_shapes):
        d, rows = O.fast_rcnn_inference(b, s, shp, score_thresh, nms_thresh, topk_per_image)
        inst = structures.Instances(shp)
        inst.pred_boxes = structures.Boxes(d["boxes"]); inst.scores = d["scores"]; inst.pred_classes = d["classes"]
        res.append(inst); kept.append(rows)
    return res, kept


def _d2_dense_box_regression_loss(anchors, box2box_transform, pred_anchor_deltas, gt_boxes, fg_mask, box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0):
    """D2 _dense_box_regression_loss (smooth_l1 branch) [D2-recall]"""
    a = type(anchors[0]).cat(anchors).tensor
    gt = torch.stack([O.rpn_get_deltas(a, k) for k in gt_boxes])
    return _fv_smooth_l1(torch.cat(pred_anchor_deltas, dim=1)[fg_mask], gt[fg_mask], smooth_l1_beta, reduction="sum")


def _d2_add_gt(structures):
    def add(gt_boxes, proposals):
        out = []
        for g, p in zip(gt_boxes, proposals):
            n = structures.Instances(p.image_size)
            n.proposal_boxes = structures.Boxes.cat([p.proposal_boxes, g])
            n.objectness_logits = torch.cat([p.objectness_logits, torch.full((len(g),), 23.025850929940457)])
            out.append(n)
        return out
    return add


def _fv_smooth_l1(input, target, beta, reduction="none"):
    if beta < 1e-5:
        loss = torch.abs(input - target)
    else:
        n = torch.abs(input - target)
        loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    return loss.sum() if reduction == "sum" else (loss.mean() if reduction == "mean" else loss)


def _fv_giou(boxes1, boxes2, reduction="none", eps=1e-7):
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xkis1, ykis1 = torch.max(x1, x1g), torch.max(y1, y1g)
    xkis2, ykis2 = torch.min(x2, x2g), torch.min(y2, y2g)
    intsctk = torch.zeros_like(x1)
    mask = (ykis2 > ykis1) & (xkis2 > xkis1)
    intsctk[mask] = (xkis2[mask] - xkis1[mask]) * (ykis2[mask] - ykis1[mask])
    unionk = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - intsctk
    iouk = intsctk / (unionk + eps)
    xc1, yc1 = torch.min(x1, x1g), torch.min(y1, y1g)
    xc2, yc2 = torch.max(x2, x2g), torch.max(y2, y2g)
    area_c = (xc2 - xc1) * (yc2 - yc1)
    miouk = iouk - ((area_c - unionk) / (area_c + eps))
    loss = 1 - miouk
    return loss.sum() if reduction == "sum" else (loss.mean() if reduction == "mean" else loss)


def fcos_cfg():
    """A cfg object carrying exactly the keys FCOSOutputs.__init__ reads, with the values of
    configs/FCOS/coco-standard/fcos_R_50_ut2_sup1_run0.yaml (+ config.py defaults)."""
    ns = types.SimpleNamespace
    F = ns(LOSS_ALPHA=0.25, LOSS_GAMMA=2.0, CENTER_SAMPLE=False, POS_RADIUS=1.5, INFERENCE_TH_TRAIN=0.05,
           PRE_NMS_TOPK_TRAIN=1000, POST_NMS_TOPK_TRAIN=100, INFERENCE_TH_TEST=0.05, PRE_NMS_TOPK_TEST=1000,
           POST_NMS_TOPK_TEST=100, NMS_TH=0.6, THRESH_WITH_CTR=False, NUM_CLASSES=80, FPN_STRIDES=[8, 16, 32, 64, 128],
           REG_DISCRETE=True, REG_MAX=16, DFL_WEIGHT=0.0, UNIFY_CTRCLS=False, KL_LOSS=True, KL_LOSS_TYPE="nlloss",
           KLLOSS_WEIGHT=0.05, LOC_FUN_ALL="mean", LOC_LOSS_TYPE="giou", QUALITY_EST="centerness", TSBETTER_CLS_SIGMA=0.0,
           SIZES_OF_INTEREST=[64, 128, 256, 512])
    S = ns(SOFT_CLS_LABEL=False, CLS_LOSS_METHOD="focal", CONSIST_REG_LOSS="ts_locvar_better_nms_nll_l1",
           CLS_LOSS_PSEUDO_METHOD="focal", TS_BETTER=0.1, TS_BETTER_CERT=0.8)
    return ns(MODEL=ns(FCOS=F), SEMISUPNET=S)


def make_head_outputs(g, N, H, W, strides, bias=-2.0):
    logits, reg, std, ctr, locs = [], [], [], [], []
    for s in strides:
        h, w = -(-H // s), -(-W // s)
        logits.append(torch.randn(N, 80, h, w, generator=g) * 1.5 + bias)
        reg.append(torch.randn(N, 68, h, w, generator=g) * 2.0)
        std.append(torch.randn(N, 4, h, w, generator=g) * 1.5)
        ctr.append(torch.randn(N, 1, h, w, generator=g))
        locs.append(O.compute_locations(h, w, s))
    return logits, reg, std, ctr, locs


def make_gts(g, N, H, W, structures, with_scores=False, empty_image=None):
    out = []
    for i in range(N):
        G = 0 if i == empty_image else int(torch.randint(1, 6, (1,), generator=g))
        cx = torch.rand(G, generator=g) * W
        cy = torch.rand(G, generator=g) * H
        bw = torch.exp(torch.rand(G, generator=g) * 3.0 + 2.0)
        bh = torch.exp(torch.rand(G, generator=g) * 3.0 + 2.0)
        boxes = torch.stack([(cx - bw / 2).clamp(0, W - 2), (cy - bh / 2).clamp(0, H - 2), (cx + bw / 2).clamp(2, W), (cy + bh / 2).clamp(2, H)], 1)
        inst = structures.Instances((H, W))
        inst.gt_boxes = structures.Boxes(boxes)
        inst.gt_classes = torch.randint(0, 80, (G,), generator=g)
        if with_scores:
            inst.scores = torch.rand(G, generator=g)
            inst.reg_pred_std = torch.randn(G, 4, generator=g) * 2.0 - 1.0
        out.append(inst)
    return out


def npy(x):
    return x.detach().cpu().numpy()


def gts_to_arrays(prefix, gts, d):
    for i, x in enumerate(gts):
        d["%s%d_boxes" % (prefix, i)] = npy(x.gt_boxes.tensor)
        d["%s%d_classes" % (prefix, i)] = npy(x.gt_classes)
        if x.has("reg_pred_std"):
            d["%s%d_std" % (prefix, i)] = npy(x.reg_pred_std)
            d["%s%d_scores" % (prefix, i)] = npy(x.scores)


THR_CLS_CTR = (0.85, 0.7)


def gen_fcos(structures, fo, pg):
    cfg = fcos_cfg()
    outm = fo.FCOSOutputs(cfg)
    g = torch.Generator().manual_seed(1234)
    N, H, W = 2, 128, 160
    strides = [8, 16, 32, 64, 128]
    d = {"N": N, "H": H, "W": W}
    logits, reg, std, ctr, locs = make_head_outputs(g, N, H, W, strides)
    for l in range(5):
        d["logits%d" % l], d["reg%d" % l], d["std%d" % l], d["ctr%d" % l] = map(npy, (logits[l], reg[l], std[l], ctr[l]))

    # ---- supervised losses + grads (second image has no gt: exercises keep_locations) ----
    for case, empty in (("sup", None), ("supempty", 1)):
        gts = make_gts(g, N, H, W, structures, empty_image=empty)
        gts_to_arrays(case + "_gt", gts, d)
        leaves = [[t.clone().requires_grad_(True) for t in lst] for lst in (logits, reg, std, ctr)]
        extras, losses = outm.losses(leaves[0], leaves[1], leaves[3], locs, gts, leaves[2], [], False, branch="labeled")
        tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
        tot.backward()
        for k, v in losses.items():
            d["%s_%s" % (case, k)] = npy(v)
        for nm, lst in zip(("logits", "reg", "std", "ctr"), leaves):
            for l in range(5):
                d["%s_g%s%d" % (case, nm, l)] = npy(lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l]))
        tt = outm._get_ground_truth(locs, gts)
        for l in range(5):
            d["%s_labels%d" % (case, l)] = npy(tt["labels"][l])
            d["%s_regt%d" % (case, l)] = npy(tt["reg_targets"][l])
            d["%s_tinds%d" % (case, l)] = npy(tt["target_inds"][l])

    # ---- pseudo losses + grads ----
    gcls = make_gts(g, N, H, W, structures, with_scores=True)
    greg = make_gts(g, N, H, W, structures, with_scores=True)
    for x in greg:  # make some teacher boundaries confident so the TS-better selection is non-empty
        x.reg_pred_std[:, :2] = -4.0
    gts_to_arrays("pcls_gt", gcls, d)
    gts_to_arrays("preg_gt", greg, d)
    leaves = [[t.clone().requires_grad_(True) for t in lst] for lst in (logits, reg, std, ctr)]
    extras, losses = outm.pseudo_losses(leaves[0], leaves[1], leaves[3], locs, {"cls": gcls, "reg": greg}, leaves[2], [], False, branch="unlabeled")
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    for k, v in losses.items():
        d["pseudo_%s" % k] = npy(v.float() if torch.is_tensor(v) else torch.tensor(float(v)))
    for nm, lst in zip(("logits", "reg", "std", "ctr"), leaves):
        for l in range(5):
            d["pseudo_g%s%d" % (nm, l)] = npy(lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l]))
    tt = outm._get_ground_truth(locs, greg)
    for l in range(5):
        d["preg_bvars%d" % l] = npy(tt["boundary_vars"][l])
        d["preg_labels%d" % l] = npy(tt["labels"][l])

    # ---- decode + NMS for the three criteria used by the trainer, and thresholding ----
    gen = pg.PseudoGenerator(cfg)
    outm.eval()
    image_sizes = [(H, W)] * N
    with torch.no_grad():
        for m in ("cls", "cls_n_ctr", "cls_n_loc"):
            res = outm.predict_proposals(logits, reg, ctr, locs, image_sizes, std, [], m)
            for i, r in enumerate(res):
                d["det_%s_%d_boxes" % (m, i)] = npy(r.pred_boxes.tensor)
                d["det_%s_%d_scores" % (m, i)] = npy(r.scores)
                d["det_%s_%d_classes" % (m, i)] = npy(r.pred_classes)
                d["det_%s_%d_ctr" % (m, i)] = npy(r.centerness)
                d["det_%s_%d_conf" % (m, i)] = npy(r.cls_confid)
                d["det_%s_%d_std" % (m, i)] = npy(r.reg_pred_std)
                th, _ = gen.process_pseudo_label([r], 0.3, "roih", "thresholding")
                d["thr_%s_%d_boxes" % (m, i)] = npy(th[0].gt_boxes.tensor)
                d["thr_%s_%d_scores" % (m, i)] = npy(th[0].scores)
                # round 4: the two-threshold selection (pseudo_generator.py:107-131, reached through process_pseudo_label :49-52):
                # cls_confid > thr[0] AND centerness > thr[1]; thresholds chosen so that each of the two tests alone rejects some
                # of the detections the other accepts
                thcc, ncc = gen.process_pseudo_label([r], THR_CLS_CTR, "roih", "thresholding_cls_ctr")
                d["thrcc_%s_%d_boxes" % (m, i)] = npy(thcc[0].gt_boxes.tensor)
                d["thrcc_%s_%d_classes" % (m, i)] = npy(thcc[0].gt_classes)
                d["thrcc_%s_%d_scores" % (m, i)] = npy(thcc[0].scores)
                d["thrcc_%s_%d_ctr" % (m, i)] = npy(thcc[0].centerness)
                d["thrcc_%s_%d_conf" % (m, i)] = npy(thcc[0].cls_confid)
                d["thrcc_%s_%d_std" % (m, i)] = npy(thcc[0].reg_pred_std)
                d["thrcc_%s_%d_num" % (m, i)] = np.float64(ncc)
    d["thrcc_thresholds"] = np.asarray(THR_CLS_CTR, np.float64)
    np.savez_compressed(os.path.join(HERE, "fcos_outputs.npz"), **d)
    print("fcos_outputs.npz:", len(d), "arrays")


def gen_fcos_center_sample(structures, fo):
    """CENTER_SAMPLE True (config-reachable variant, fcos_outputs.py:700-770): targets + supervised losses / gradients."""
    cfg = fcos_cfg()
    cfg.MODEL.FCOS.CENTER_SAMPLE = True
    outm = fo.FCOSOutputs(cfg)
    g = torch.Generator().manual_seed(4321)
    N, H, W = 3, 128, 160
    strides = [8, 16, 32, 64, 128]
    d = {"N": N, "H": H, "W": W, "radius": 1.5}
    logits, reg, std, ctr, locs = make_head_outputs(g, N, H, W, strides)
    for l in range(5):
        d["logits%d" % l], d["reg%d" % l], d["std%d" % l], d["ctr%d" % l] = map(npy, (logits[l], reg[l], std[l], ctr[l]))
    gts = make_gts(g, N, H, W, structures, empty_image=2)
    gts_to_arrays("gt", gts, d)
    leaves = [[t.clone().requires_grad_(True) for t in lst] for lst in (logits, reg, std, ctr)]
    extras, losses = outm.losses(leaves[0], leaves[1], leaves[3], locs, gts, leaves[2], [], False, branch="labeled")
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    for k, v in losses.items():
        d["loss_%s" % k] = npy(v)
    for nm, lst in zip(("logits", "reg", "std", "ctr"), leaves):
        for l in range(5):
            d["g%s%d" % (nm, l)] = npy(lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l]))
    tt = outm._get_ground_truth(locs, gts)
    for l in range(5):
        d["labels%d" % l] = npy(tt["labels"][l])
        d["regt%d" % l] = npy(tt["reg_targets"][l])
        d["tinds%d" % l] = npy(tt["target_inds"][l])
    # SEMISUPNET.PSEUDO_CLS_IGNORE_NEAR (fcos_outputs.py:841-851): with centre sampling, locations inside a box but outside its
    # sampling region are dropped from the supervised losses (:310-311) ...
    leaves = [[t.clone().requires_grad_(True) for t in lst] for lst in (logits, reg, std, ctr)]
    extras, losses = outm.losses(leaves[0], leaves[1], leaves[3], locs, gts, leaves[2], [], True, branch="labeled")
    (losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]).backward()
    for k, v in losses.items():
        d["ign_loss_%s" % k] = npy(v)
    for nm, lst in zip(("logits", "reg", "std", "ctr"), leaves):
        for l in range(5):
            d["ign_g%s%d" % (nm, l)] = npy(lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l]))
    tt = outm._get_ground_truth(locs, gts, True)
    for l in range(5):
        d["ign_keep%d" % l] = npy(tt["keep_locations"][l].to(torch.uint8))
    # ... and is never consulted by the pseudo losses (:487-631), the only branch the trainer hands the switch to (trainer.py:340,347)
    pg_ = make_gts(g, N, H, W, structures, with_scores=True)
    pl = []
    for ign in (False, True):
        ex, lo = outm.pseudo_losses(logits, reg, ctr, locs, {"cls": pg_, "reg": pg_}, std, [], ign, branch="unlabeled")
        pl.append({k: npy(v.float() if torch.is_tensor(v) else torch.tensor(float(v))) for k, v in lo.items()})
    assert all(np.array_equal(pl[0][k], pl[1][k]) for k in pl[0]), "ignore_near changed a pseudo loss"
    gts_to_arrays("ign_pgt", pg_, d)
    for k, v in pl[1].items():
        d["ign_pseudo_%s" % k] = v
    np.savez_compressed(os.path.join(HERE, "fcos_center_sample.npz"), **d)
    print("fcos_center_sample.npz:", len(d), "arrays")


def gen_fcos_loss_variants(structures, fo):
    """Config-reachable FCOS loss variants no shipped YAML selects: KL_LOSS_TYPE "klloss", KL_LOSS False,
    QUALITY_EST "iou", LOC_LOSS_TYPE "iou" / "linear_iou", and a CONSIST_REG_LOSS other than the TS-better one (the pseudo
    regression loss then is the KL / NLL term, fcos_outputs.py:571-585).  One set of inputs, one set of outputs per case."""
    g = torch.Generator().manual_seed(2468)
    N, H, W = 2, 128, 160
    strides = [8, 16, 32, 64, 128]
    d = {"N": N, "H": H, "W": W}
    logits, reg, std, ctr, locs = make_head_outputs(g, N, H, W, strides)
    for l in range(5):
        d["logits%d" % l], d["reg%d" % l], d["std%d" % l], d["ctr%d" % l] = map(npy, (logits[l], reg[l], std[l], ctr[l]))
    gts = make_gts(g, N, H, W, structures)
    gts_to_arrays("gt", gts, d)
    gcls = make_gts(g, N, H, W, structures, with_scores=True)
    greg = make_gts(g, N, H, W, structures, with_scores=True)
    gts_to_arrays("pcls_gt", gcls, d)
    gts_to_arrays("preg_gt", greg, d)
    cases = {
        "klloss": dict(KL_LOSS_TYPE="klloss"),
        "nokl": dict(KL_LOSS=False),
        "iouq": dict(QUALITY_EST="iou"),
        "lociou": dict(LOC_LOSS_TYPE="iou"),
        "loclinear": dict(LOC_LOSS_TYPE="linear_iou"),
        "klloss_iouq_linear": dict(KL_LOSS_TYPE="klloss", QUALITY_EST="iou", LOC_LOSS_TYPE="linear_iou"),
        # MODEL.FCOS.LOC_FUN_ALL: the reduction of the KLLoss term (kl_loss.py:48-64); the default "mean" is the cases above
        "klloss_sum": dict(KL_LOSS_TYPE="klloss", LOC_FUN_ALL="sum"),
        "klloss_wsum": dict(KL_LOSS_TYPE="klloss", LOC_FUN_ALL="weight_ctr_sum"),
        "klloss_wmean_iouq": dict(KL_LOSS_TYPE="klloss", LOC_FUN_ALL="weight_ctr_mean", QUALITY_EST="iou"),
    }
    for case, over in cases.items():
        cfg = fcos_cfg()
        for k, v in over.items():
            setattr(cfg.MODEL.FCOS, k, v)
        outm = fo.FCOSOutputs(cfg)
        leaves = [[t.clone().requires_grad_(True) for t in lst] for lst in (logits, reg, std, ctr)]
        extras, losses = outm.losses(leaves[0], leaves[1], leaves[3], locs, gts, leaves[2], [], False, branch="labeled")
        tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
        tot.backward()
        for k, v in losses.items():
            d["%s_%s" % (case, k)] = npy(v)
        for nm, lst in zip(("logits", "reg", "std", "ctr"), leaves):
            if nm == "logits" and case != "klloss":
                continue  # the classification path does not depend on the variant: one copy is enough
            for l in range(5):
                d["%s_g%s%d" % (case, nm, l)] = npy(lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l]))
    # pseudo regression loss = weight * KL / NLL term when CONSIST_REG_LOSS is not the TS-better selection
    for case, over in (("pseudo_nll", dict()), ("pseudo_kl", dict(KL_LOSS_TYPE="klloss")),
                       ("pseudo_kl_wmean", dict(KL_LOSS_TYPE="klloss", LOC_FUN_ALL="weight_ctr_mean"))):
        cfg = fcos_cfg()
        cfg.SEMISUPNET.CONSIST_REG_LOSS = "mse_loss_all_raw"  # config.py:191 default
        for k, v in over.items():
            setattr(cfg.MODEL.FCOS, k, v)
        outm = fo.FCOSOutputs(cfg)
        leaves = [[t.clone().requires_grad_(True) for t in lst] for lst in (logits, reg, std, ctr)]
        extras, losses = outm.pseudo_losses(leaves[0], leaves[1], leaves[3], locs, {"cls": gcls, "reg": greg}, leaves[2], [], False,
                                            branch="unlabeled")
        tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
        tot.backward()
        for k, v in losses.items():
            d["%s_%s" % (case, k)] = npy(v.float() if torch.is_tensor(v) else torch.tensor(float(v)))
        for nm, lst in zip(("logits", "reg", "std", "ctr"), leaves):
            if nm == "logits" and case != "pseudo_nll":
                continue
            for l in range(5):
                d["%s_g%s%d" % (case, nm, l)] = npy(lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l]))
    np.savez_compressed(os.path.join(HERE, "fcos_loss_variants.npz"), **d)
    print("fcos_loss_variants.npz:", len(d), "arrays")


def gen_data_pipeline():
    """Host logic of the two-crop loader, executed from the reference's own sources:
    `AspectRatioGroupedSemiSupDatasetTwoCrop` (data/common.py:93-167, imported with a stand-in for its Detectron2 base class) on seeded
    streams of (width, height) pairs, and `divide_label_unlabel` (data/build.py:30-53, the function's AST node executed on its own -
    the module imports most of Detectron2) on a small seed table and on dataseed/COCO_supervision.txt."""
    import ast
    import json
    d2c = types.ModuleType("detectron2.data.common")

    class _Base:
        def __init__(self, dataset, batch_size):
            self.dataset, self.batch_size = dataset, batch_size
    d2c.AspectRatioGroupedDataset = _Base
    d2c.MapDataset = _Base
    for name in ("detectron2", "detectron2.data"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["detectron2.data.common"] = d2c
    cm = _load("ubteacher.data.common_ref", REF + "/ubteacher/data/common.py")
    rng = np.random.default_rng(2024)
    out = {}
    for case, (bl, bu, n) in enumerate([(2, 2, 60), (4, 4, 90), (1, 3, 50), (3, 1, 50)]):
        lab = [(int(rng.integers(300, 700)), int(rng.integers(300, 700))) for _ in range(n)]
        unl = [(int(rng.integers(300, 700)), int(rng.integers(300, 700))) for _ in range(n)]

        def stream(sizes, tag):
            for i, (w, h) in enumerate(sizes):
                yield ({"width": w, "height": h, "id": i, "view": tag + "s"}, {"width": w, "height": h, "id": i, "view": tag + "w"})
        ds = cm.AspectRatioGroupedSemiSupDatasetTwoCrop((stream(lab, "l"), stream(unl, "u")), (bl, bu))
        batches = []
        for ls, lw, us, uw in ds:
            assert [d["id"] for d in ls] == [d["id"] for d in lw] and [d["id"] for d in us] == [d["id"] for d in uw]
            batches.append([[d["id"] for d in ls], [d["id"] for d in us]])
        out["batcher_%d" % case] = dict(bl=bl, bu=bu, label_wh=lab, unlabel_wh=unl, batches=batches)
    # divide_label_unlabel
    src = open(REF + "/ubteacher/data/build.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "divide_label_unlabel"][0]
    ns = {"np": np, "json": json, "PathManager": types.SimpleNamespace(open=open)}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "build.py", "exec"), ns)
    table = {"10.0": {"0": sorted(int(v) for v in rng.choice(50, 5, replace=False)), "1": [int(v) for v in rng.choice(50, 5, replace=False)]},
             "30.0": {"0": [int(v) for v in rng.choice(50, 15, replace=False)]}}
    seed_path = os.path.join(HERE, "supervision_small.json")
    json.dump(table, open(seed_path, "w"))
    dicts = [{"image_id": i} for i in range(50)]
    small = {}
    for pct, sd in (("10.0", 0), ("10.0", 1), ("30.0", 0)):
        lab, unl = ns["divide_label_unlabel"](dicts, float(pct), sd, seed_path)
        small["%s_%d" % (pct, sd)] = dict(label=[d["image_id"] for d in lab], unlabel=[d["image_id"] for d in unl])
    out["divide_small"] = small
    # the shipped table: COCO train2017 after Detectron2's empty-annotation filter has 117266 images (README / data seeds)
    dicts = list(range(117266))
    big = {}
    for pct, sd in ((1.0, 0), (5.0, 3), (10.0, 1)):
        lab, unl = ns["divide_label_unlabel"](dicts, pct, sd, REF + "/dataseed/COCO_supervision.txt")
        big["%s_%d" % (pct, sd)] = dict(n_label=len(lab), n_unlabel=len(unl), label_sum=int(np.sum(lab)), label_head=[int(v) for v in lab[:8]],
                                        unlabel_head=[int(v) for v in unl[:8]])
    out["divide_coco"] = big
    json.dump(out, open(os.path.join(HERE, "data_pipeline.json"), "w"))
    print("data_pipeline.json:", len(out), "entries")


def gen_small_ops(fo):
    from ubteacher.layers import IOULoss, NLLoss
    g = torch.Generator().manual_seed(7)
    P = 64
    pred = (torch.rand(P, 4, generator=g) * 10 + 0.1).requires_grad_(True)
    tgt = torch.rand(P, 4, generator=g) * 10 + 0.1
    w = torch.rand(P, generator=g)
    std = torch.randn(P, 4, generator=g).requires_grad_(True)
    d = {"pred": npy(pred), "tgt": npy(tgt), "w": npy(w), "std": npy(std)}
    l = IOULoss("giou")(pred, tgt, w)
    l.backward()
    d["giou"] = npy(l); d["giou_gpred"] = npy(pred.grad); pred.grad = None
    iw = fo.compute_iou_targets(pred.detach(), tgt)
    l = NLLoss()(pred, std, tgt, weight=w, iou_weight=iw, loss_denorm=1.0, method="mean")
    l.backward()
    d["iou_targets"] = npy(iw); d["nll"] = npy(l); d["nll_gpred"] = npy(pred.grad); d["nll_gstd"] = npy(std.grad)
    d["ctr_targets"] = npy(fo.compute_ctrness_targets(tgt))
    x = torch.randn(P, 68, generator=g)
    d["integral_in"] = npy(x); d["integral_out"] = npy(fo.Integral(16)(x))
    np.savez_compressed(os.path.join(HERE, "small_ops.npz"), **d)
    print("small_ops.npz:", len(d), "arrays")


def gen_ema(tr):
    g = torch.Generator().manual_seed(3)

    class M:
        def __init__(self, sd):
            self.sd = sd

        def state_dict(self):
            return self.sd

        def load_state_dict(self, sd):
            self.sd = sd

    d = {}
    for keep in (0.0, 0.9996, 0.9999):
        s = {"a.weight": torch.randn(1000, generator=g), "b.running_var": torch.rand(37, generator=g) + 0.5}
        t = {k: torch.randn(v.shape, generator=g) for k, v in s.items()}
        duck = types.SimpleNamespace(model=M(s), model_teacher=M({k: v.clone() for k, v in t.items()}))
        tr.UBTeacherTrainer._update_teacher_model(duck, keep_rate=keep)
        tag = str(keep).replace(".", "p")
        for k in s:
            d["%s_s_%s" % (tag, k)] = npy(s[k]); d["%s_t_%s" % (tag, k)] = npy(t[k])
            d["%s_out_%s" % (tag, k)] = npy(duck.model_teacher.sd[k])
    np.savez_compressed(os.path.join(HERE, "ema.npz"), **d)
    print("ema.npz:", len(d), "arrays")


def gen_rcnn(structures):
    br = _load("ubteacher.modeling.box_regression", REF + "/ubteacher/modeling/box_regression.py")
    fr = _load("ubteacher.modeling.roi_heads.fast_rcnn", REF + "/ubteacher/modeling/roi_heads/fast_rcnn.py")
    rh = _load("ubteacher.modeling.roi_heads.roi_heads", REF + "/ubteacher/modeling/roi_heads/roi_heads.py")
    rp = _load("ubteacher.modeling.proposal_generator.rpn", REF + "/ubteacher/modeling/proposal_generator/rpn.py")
    g = torch.Generator().manual_seed(77)
    d = {}
    Boxes, Instances = structures.Boxes, structures.Instances
    # ---- Box2BoxXYXYTransform -----------------------------------------------------------------
    tf = br.Box2BoxXYXYTransform(weights=(10.0, 10.0, 5.0, 5.0))
    n = 50
    xy = torch.rand(n, 2, generator=g) * 200
    src = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 100 + 1], 1)
    xy2 = torch.rand(n, 2, generator=g) * 200
    tgt = torch.cat([xy2, xy2 + torch.rand(n, 2, generator=g) * 100 + 1], 1)
    dl = torch.randn(n, 4, generator=g) * 3
    dl[0] = torch.tensor([900.0, -900.0, 400.0, -400.0])  # exercises the +-62.5 clamp
    d["bx_src"], d["bx_tgt"], d["bx_deltas"] = npy(src), npy(tgt), npy(dl)
    d["bx_get"] = npy(tf.get_deltas(src, tgt)); d["bx_apply"] = npy(tf.apply_deltas(dl, src))
    # ---- FastRCNNFocaltLossBoundaryVarOutputLayers.losses ------------------------------------------
    cls_ = fr.FastRCNNFocaltLossBoundaryVarOutputLayers
    duck = types.SimpleNamespace(num_classes=80, box2box_transform=tf, smooth_l1_beta=0.0, box_reg_loss_type="nlloss",
                                 box_pseudo_reg_loss_type="tsbetter", loss_weight={"loss_box_reg": 1.0}, ts_better=0.1, t_cert=0.5)
    for name in ("comput_focal_loss", "box_reg_loss", "box_reg_pseudo_loss"):
        setattr(duck, name, types.MethodType(getattr(cls_, name), duck))
    R = 60
    gt_classes = torch.randint(0, 81, (R,), generator=g)
    gt_classes[: R // 2] = 80
    gt_classes[-5:] = torch.tensor([3, 7, 7, 12, 0])
    pxy = torch.rand(R, 2, generator=g) * 150
    prop = torch.cat([pxy, pxy + torch.rand(R, 2, generator=g) * 80 + 4], 1)
    gtb = prop + torch.randn(R, 4, generator=g) * 4
    gstd = torch.randn(R, 4, generator=g) * 2 - 1.0
    inst = Instances((300, 300))
    inst.proposal_boxes = Boxes(prop); inst.gt_boxes = Boxes(gtb); inst.gt_classes = gt_classes; inst.gt_loc_std = gstd
    d.update(rc_prop=npy(prop), rc_gtb=npy(gtb), rc_cls=npy(gt_classes), rc_gstd=npy(gstd))
    for branch in ("supervised", "unsup_data_train"):
        scores = (torch.randn(R, 81, generator=g) * 2).requires_grad_(True)
        deltas = (torch.randn(R, 4, generator=g) * 0.5).requires_grad_(True)
        std = (torch.randn(R, 4, generator=g) * 1.5).requires_grad_(True)
        if branch == "unsup_data_train":
            std = (std.detach() + 1.0).requires_grad_(True)
        ls = cls_.losses(duck, (scores, deltas, std), [inst], branch)
        (ls["loss_cls"] + 2.0 * ls["loss_box_reg"]).backward()
        for k, v in (("scores", scores), ("deltas", deltas), ("std", std)):
            d["rc_%s_%s" % (branch, k)] = npy(v)
            d["rc_%s_g%s" % (branch, k)] = npy(v.grad if v.grad is not None else torch.zeros_like(v))
        d["rc_%s_loss_cls" % branch] = npy(ls["loss_cls"]); d["rc_%s_loss_box_reg" % branch] = npy(ls["loss_box_reg"])
        # MODEL.ROI_HEADS.LOSS "CrossEntropy_BoundaryVar" (fast_rcnn.py:214-712) on the same inputs (no further random draws)
        ce_ = fr.FastRCNNCrossEntropyBoundaryVarOutputLayers
        cduck = types.SimpleNamespace(**vars(duck))
        for name in ("box_reg_loss", "box_reg_pseudo_loss"):
            setattr(cduck, name, types.MethodType(getattr(ce_, name), cduck))
        leaves = [v.detach().clone().requires_grad_(True) for v in (scores, deltas, std)]
        ls = ce_.losses(cduck, tuple(leaves), [inst], branch)
        (ls["loss_cls"] + 2.0 * ls["loss_box_reg"]).backward()
        for k, v in zip(("scores", "deltas", "std"), leaves):
            d["rcce_%s_g%s" % (branch, k)] = npy(v.grad if v.grad is not None else torch.zeros_like(v))
        d["rcce_%s_loss_cls" % branch] = npy(ls["loss_cls"]); d["rcce_%s_loss_box_reg" % branch] = npy(ls["loss_box_reg"])
    # ---- inference (predict_boxes / probs + D2 fast_rcnn_inference stand-in + pred_boxes_std gather) -----------
    for name in ("predict_boxes", "predict_boxes_std", "predict_probs"):
        setattr(duck, name, types.MethodType(getattr(cls_, name), duck))
    duck.test_score_thresh, duck.test_nms_thresh, duck.test_topk_per_image = 0.05, 0.5, 100
    pi = Instances((300, 300)); pi.proposal_boxes = Boxes(prop)
    sc = torch.randn(R, 81, generator=g) * 3; de = torch.randn(R, 4, generator=g) * 2; sd_ = torch.randn(R, 4, generator=g)
    res, keep = cls_.inference(duck, (sc, de, sd_), [pi])
    d.update(inf_scores=npy(sc), inf_deltas=npy(de), inf_std=npy(sd_), inf_boxes=npy(res[0].pred_boxes.tensor),
             inf_sc=npy(res[0].scores), inf_cls=npy(res[0].pred_classes), inf_bstd=npy(res[0].pred_boxes_std), inf_keep=npy(keep[0]))
    # ---- round 4: the +-62.5 clamp of Box2BoxXYXYTransform.apply_deltas (box_regression.py:88-128) driven through the predictor's
    # inference and through the supervised nlloss (its IoU weight decodes the predicted deltas: fast_rcnn.py:938-1016).  Own generator:
    # the arrays above and below keep their values.  Rows 0-11 are 2-4 px proposals in the middle of the image with |delta / 10|
    # far beyond 62.5, so the clamped boxes stay inside the image (an unclamped decode would hit the image border instead).
    g2 = torch.Generator().manual_seed(78)
    Rc = 40
    cxy = torch.rand(Rc, 2, generator=g2) * 150 + 20
    propc = torch.cat([cxy, cxy + torch.rand(Rc, 2, generator=g2) * 60 + 4], 1)
    propc[:12, :2] = 150 + torch.rand(12, 2, generator=g2) * 4
    propc[:12, 2:] = propc[:12, :2] + 2 + torch.rand(12, 2, generator=g2) * 0.2
    dec = torch.randn(Rc, 4, generator=g2) * 2
    big = torch.tensor([[900.0, -900.0, 2000.0, -2000.0], [-900.0, 900.0, -2000.0, 2000.0], [-700.0, -650.0, -626.0, -1e4],
                        [626.0, 700.0, 1e4, 650.0], [-624.0, 624.0, -625.5, 625.5], [0.0, 900.0, 0.0, 900.0]])
    dec[:6] = big; dec[6:12] = -big
    scc = torch.randn(Rc, 81, generator=g2) * 3
    scc[:12, 80] = -9.0                                 # the clamped rows are confident foreground: they survive score threshold and NMS
    scc[torch.arange(12), torch.arange(12)] = 9.0
    sdc = torch.randn(Rc, 4, generator=g2)
    pic = Instances((300, 300)); pic.proposal_boxes = Boxes(propc)
    resc, keepc = cls_.inference(duck, (scc, dec, sdc), [pic])
    assert set(range(12)) <= set(keepc[0].tolist()), "the clamped rows must be among the kept detections"
    d.update(infc_prop=npy(propc), infc_scores=npy(scc), infc_deltas=npy(dec), infc_std=npy(sdc), infc_boxes=npy(resc[0].pred_boxes.tensor),
             infc_sc=npy(resc[0].scores), infc_cls=npy(resc[0].pred_classes), infc_bstd=npy(resc[0].pred_boxes_std), infc_keep=npy(keepc[0]))
    clsc = torch.randint(0, 80, (Rc,), generator=g2)
    clsc[20:] = 80
    gtbc = propc + torch.randn(Rc, 4, generator=g2) * 1.5
    gstdc = torch.randn(Rc, 4, generator=g2) * 2 - 1.0
    instc = Instances((300, 300))
    instc.proposal_boxes = Boxes(propc); instc.gt_boxes = Boxes(gtbc); instc.gt_classes = clsc; instc.gt_loc_std = gstdc
    d.update(rcc_prop=npy(propc), rcc_gtb=npy(gtbc), rcc_cls=npy(clsc), rcc_gstd=npy(gstdc))
    for branch in ("supervised", "unsup_data_train"):
        leaves = [scc.clone().requires_grad_(True), dec.clone().requires_grad_(True), (sdc * 1.5 + (1.0 if branch != "supervised" else 0.0)).requires_grad_(True)]
        ls = cls_.losses(duck, tuple(leaves), [instc], branch)
        (ls["loss_cls"] + 2.0 * ls["loss_box_reg"]).backward()
        for k, v in zip(("scores", "deltas", "std"), leaves):
            d["rcc_%s_%s" % (branch, k)] = npy(v)
            d["rcc_%s_g%s" % (branch, k)] = npy(v.grad if v.grad is not None else torch.zeros_like(v))
        d["rcc_%s_loss_cls" % branch] = npy(ls["loss_cls"]); d["rcc_%s_loss_box_reg" % branch] = npy(ls["loss_box_reg"])
    # ---- round 4: the UTv1 predictor, MODEL.ROI_HEADS.LOSS "FocalLoss": FastRCNNFocaltLossOutputLayers.losses -> FastRCNNFocalLoss
    # (fast_rcnn.py:1296-1429; box_reg_loss inherited from FastRCNNOutputs :134-194), class-specific and class-agnostic deltas, with and
    # without gt_confid (the pseudo-labeled branch weights the focal term by the matched pseudo box's score).  Detectron2's
    # Box2BoxTransform is a stand-in on the oracle's restatement [D2-recall].  Own generator.
    g3 = torch.Generator().manual_seed(79)
    W4 = (10.0, 10.0, 5.0, 5.0)
    d2tf = types.SimpleNamespace(get_deltas=lambda src, tgt: O.d2_get_deltas(src, tgt, W4), apply_deltas=lambda dl, bx: O.d2_apply_deltas(dl, bx, W4))
    R1 = 48
    cls1 = torch.randint(0, 81, (R1,), generator=g3)
    cls1[:16] = 80
    cls1[-4:] = torch.tensor([0, 79, 5, 5])
    p1 = torch.rand(R1, 2, generator=g3) * 150
    prop1 = torch.cat([p1, p1 + torch.rand(R1, 2, generator=g3) * 80 + 4], 1)
    gtb1 = prop1 + torch.randn(R1, 4, generator=g3) * 4
    gtb1[:, 2:] = torch.maximum(gtb1[:, 2:], gtb1[:, :2] + 2.0)
    conf1 = torch.rand(R1, generator=g3)
    d.update(v1_prop=npy(prop1), v1_gtb=npy(gtb1), v1_cls=npy(cls1), v1_conf=npy(conf1))
    lay = types.SimpleNamespace(box2box_transform=d2tf, smooth_l1_beta=0.0, box_reg_loss_type="smooth_l1", num_classes=80)
    for name, nbx, with_conf, beta in (("spec", 80, False, 0.0), ("spec_conf", 80, True, 0.0), ("agn_conf_beta", 1, True, 0.5), ("spec_giou", 80, False, 0.0)):
        sc1 = (torch.randn(R1, 81, generator=g3) * 2).requires_grad_(True)
        de1 = (torch.randn(R1, 4 * nbx, generator=g3) * 0.5).requires_grad_(True)
        i1 = Instances((300, 300))
        i1.proposal_boxes = Boxes(prop1); i1.gt_boxes = Boxes(gtb1); i1.gt_classes = cls1
        if with_conf:
            i1.gt_confid = conf1
        lay.smooth_l1_beta = beta
        lay.box_reg_loss_type = "giou" if name.endswith("giou") else "smooth_l1"     # fast_rcnn.py:163-186
        ls = fr.FastRCNNFocaltLossOutputLayers.losses(lay, (sc1, de1), [i1], "supervised")
        (ls["loss_cls"] + 2.0 * ls["loss_box_reg"]).backward()
        d["v1_%s_scores" % name], d["v1_%s_deltas" % name] = npy(sc1), npy(de1)
        d["v1_%s_gscores" % name], d["v1_%s_gdeltas" % name] = npy(sc1.grad), npy(de1.grad)
        d["v1_%s_loss_cls" % name], d["v1_%s_loss_box_reg" % name] = npy(ls["loss_cls"]), npy(ls["loss_box_reg"])
    # ---- PseudoLabRPN: label_and_sample_anchors_pseudo + losses (weights on all valid anchors) ----
    hw = [(6, 8), (3, 4)]
    anchors = O.make_anchors(hw, [16, 32], sizes=(32, 64))
    A = torch.cat(anchors)
    Rn = A.shape[0]
    keys = torch.rand(2, Rn, generator=g)
    kit = iter([keys[0], keys[1]])

    def sub(labels):
        k = next(kit)
        pos, neg = O.subsample_by_keys(labels, k, 16, 0.25, 0)
        labels.fill_(-1); labels.scatter_(0, pos, 1); labels.scatter_(0, neg, 0)
        return labels
    rduck = types.SimpleNamespace(anchor_matcher=lambda m: O.matcher(m, [0.3, 0.7], [0, -1, 1], True), anchor_boundary_thresh=-1,
                                  _subsample_labels=sub, box2box_transform=None, box_reg_loss_type="smooth_l1", smooth_l1_beta=0.0,
                                  batch_size_per_image=16, loss_weight={"loss_rpn_cls": 1.0, "loss_rpn_loc": 1.0})
    gi = []
    for i in range(2):
        G = 3 if i == 0 else 0
        c = torch.rand(G, 2, generator=g) * torch.tensor([100.0, 70.0])
        b = torch.cat([c, c + torch.rand(G, 2, generator=g) * 60 + 10], 1)
        x = Instances((96, 128)); x.gt_boxes = Boxes(b); x.scores = torch.rand(G, generator=g) * 0.3 + 0.7
        gi.append(x)
        d["rpn_gt%d" % i] = npy(b); d["rpn_sc%d" % i] = npy(x.scores)
    labs, mboxes, confs = rp.PseudoLabRPN.label_and_sample_anchors_pseudo(rduck, [Boxes(a) for a in anchors], gi)
    obj = [(torch.randn(2, a.shape[0], generator=g)).requires_grad_(True) for a in anchors]
    dls = [(torch.randn(2, a.shape[0], 4, generator=g) * 0.3).requires_grad_(True) for a in anchors]
    ls = rp.PseudoLabRPN.losses(rduck, [Boxes(a) for a in anchors], obj, labs, dls, mboxes, confs)
    (ls["loss_rpn_cls"] + ls["loss_rpn_loc"]).backward()
    d.update(rpn_keys=npy(keys), rpn_labels=npy(torch.stack(labs)), rpn_conf=npy(torch.stack(confs).float()),
             rpn_loss_cls=npy(ls["loss_rpn_cls"]), rpn_loss_loc=npy(ls["loss_rpn_loc"]))
    for l in range(2):
        d["rpn_obj%d" % l], d["rpn_dl%d" % l] = npy(obj[l]), npy(dls[l])
        d["rpn_gobj%d" % l], d["rpn_gdl%d" % l] = npy(obj[l].grad), npy(dls[l].grad)
    # ---- StandardROIHeadsPseudoLab.label_and_sample_proposals_pseudo ---------------------------------------
    P = 40
    pc = torch.rand(P, 2, generator=g) * 200
    pbx = torch.cat([pc, pc + torch.rand(P, 2, generator=g) * 80 + 5], 1)
    G = 4
    gb = pbx[:G] + torch.randn(G, 4, generator=g) * 3
    t = Instances((300, 300)); t.gt_boxes = Boxes(gb); t.gt_classes = torch.randint(0, 80, (G,), generator=g)
    t.scores = torch.rand(G, generator=g); t.pred_boxes_std = torch.randn(G, 4, generator=g)
    pr = Instances((300, 300)); pr.proposal_boxes = Boxes(pbx); pr.objectness_logits = torch.randn(P, generator=g)
    rkeys = torch.rand(P + G, generator=g)

    def samp(matched_idxs, matched_labels, gt_classes):
        cls = gt_classes[matched_idxs].clone()
        cls[matched_labels == 0] = 80
        fgi, bgi = O.subsample_by_keys(cls, rkeys, 16, 0.25, 80)
        s = torch.cat([fgi, bgi])
        return s, cls[s]
    hduck = types.SimpleNamespace(proposal_append_gt=True, proposal_matcher=lambda m: O.matcher(m, [0.5], [0, 1], False),
                                  _sample_proposals=samp, num_classes=80)
    out = rh.StandardROIHeadsPseudoLab.label_and_sample_proposals_pseudo.__wrapped__(hduck, [pr], [t], branch="x") \
        if hasattr(rh.StandardROIHeadsPseudoLab.label_and_sample_proposals_pseudo, "__wrapped__") else \
        rh.StandardROIHeadsPseudoLab.label_and_sample_proposals_pseudo(hduck, [pr], [t], branch="x")
    o = out[0]
    d.update(roi_prop=npy(pbx), roi_gtb=npy(gb), roi_gtc=npy(t.gt_classes), roi_gts=npy(t.scores), roi_gtstd=npy(t.pred_boxes_std),
             roi_keys=npy(rkeys), roi_out_prop=npy(o.proposal_boxes.tensor), roi_out_cls=npy(o.gt_classes),
             roi_out_gtb=npy(o.gt_boxes.tensor), roi_out_conf=npy(o.gt_confid), roi_out_std=npy(o.gt_loc_std))
    np.savez_compressed(os.path.join(HERE, "rcnn.npz"), **d)
    print("rcnn.npz:", len(d), "arrays")


if __name__ == "__main__":
    structures, fo, pg, tr = install_shims()
    if len(sys.argv) > 1:  # python gen_golden.py center_sample loss_variants ... : only those files
        for name in sys.argv[1:]:
            fn = globals()["gen_" + name] if name in ("rcnn", "data_pipeline", "fcos") else globals()["gen_fcos_" + name]
            fn(*{"rcnn": (structures,), "data_pipeline": (), "fcos": (structures, fo, pg)}.get(name, (structures, fo)))
        sys.exit(0)
    gen_rcnn(structures)
    gen_fcos(structures, fo, pg)
    gen_fcos_center_sample(structures, fo)
    gen_fcos_loss_variants(structures, fo)
    gen_data_pipeline()
    gen_small_ops(fo)
    gen_ema(tr)
