"""CPU ORACLE for the Unbiased-Teacher-v2 hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference's algorithm for the
training-step path, used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
as the checker / the timed CPU port.  The product (unbiased-teacher-v2_amd/) never imports it.

Pinning status
  * reference-owned arithmetic (FCOS targets / losses / decode, IOULoss, NLLoss, Integral,
    pseudo-label thresholding, EMA, loss weighting): PINNED against golden vectors produced by
    executing the reference's own modules in the build container
    (tests/golden/gen_golden.py -> tests/golden/*.npz, checked by tests/test_oracle_golden.py).
  * Detectron2 / fvcore / torchvision primitives (ResNet-50, FPN, FrozenBN, batched_nms, focal
    loss, SGD): those libraries are NOT in /root/reference nor installed (detectron2 >= 0.6
    unpinned commit, fvcore unpinned, torchvision "matching torch" - reference README.md:26-27,44),
    so they are restated here from their published behaviour and are
    "parity unpinned" beyond textbook known-answer tests and stock torch.nn.functional.

Every function cites the reference file:line it follows (paths relative to the reference root).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

INF = 100000000

# Emulation of the product's mixed-precision (AMP) conv path for parity tests: when set to "bf16", the
# operands of every conv / linear whose input-channel count is a multiple of 8 are rounded to bf16
# (round-to-nearest-even) before an fp32 convolution - exactly what the bf16-MFMA kernels compute
# (bf16 operands, fp32 accumulate).  None = the reference's fp32 CPU arithmetic.
CONV_ROUND = [None]


def _r16(t):
    """round to the emulated 16-bit type (CONV_ROUND "bf16": bfloat16, "fp16": IEEE half - the product's second kernel build)"""
    return t.to(torch.float16 if CONV_ROUND[0] == "fp16" else torch.bfloat16).to(torch.float32)


def _conv2d(x, w, b=None, stride=1, padding=0):
    if CONV_ROUND[0] in ("bf16", "fp16") and x.shape[1] % 8 == 0:
        x, w = _r16(x), _r16(w)
    return F.conv2d(x, w, b, stride, padding)


def _linear(x, w, b=None):
    if CONV_ROUND[0] in ("bf16", "fp16") and x.shape[1] % 8 == 0:
        x, w = _r16(x), _r16(w)
    return F.linear(x, w, b)


# =================================================================================================
# Third-party primitives (restated)
# =================================================================================================
def sigmoid_focal_loss(inputs, targets, alpha=0.25, gamma=2.0):
    """fvcore.nn.sigmoid_focal_loss (reduction none) - called at fcos_outputs.py:329-335,619-625."""
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss


def nms(boxes, scores, thr):
    """torchvision.ops.nms semantics with the tie rule declared for this project:
    order = (score desc, index asc); suppress when inter/(a_i+a_j-inter) > thr (strict).
    Returns kept indices in descending-score order (int64)."""
    b = boxes.detach().cpu().numpy().astype(np.float32)
    s = scores.detach().cpu().numpy().astype(np.float32)
    n = len(s)
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    order = np.lexsort((np.arange(n), -s.astype(np.float64)))  # primary -s, secondary index
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for ii in range(n):
        i = order[ii]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[ii + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1); h = np.maximum(np.float32(0), yy2 - yy1)
        inter = (w * h).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr > np.float32(thr)]] = True
    return torch.as_tensor(np.array(keep, dtype=np.int64))


def batched_nms(boxes, scores, idxs, thr):
    """torchvision batched_nms, coordinate-trick form (D2 batched_nms -> ml_nms.py:27):
    boxes + idxs * (max_coordinate + 1) in fp32, then plain nms."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, thr)


def pairwise_iou(a, b):
    """D2 pairwise_iou [D2-recall]."""
    area1 = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = (torch.min(a[:, None, 2:], b[:, 2:]) - torch.max(a[:, None, :2], b[:, :2])).clamp(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (area1[:, None] + area2 - inter), torch.zeros(1, dtype=inter.dtype))


def frozen_bn(x, sd, prefix, eps=1e-5):
    """D2 FrozenBatchNorm2d.forward [D2-recall]."""
    scale = sd[prefix + ".weight"] * (sd[prefix + ".running_var"] + eps).rsqrt()
    bias = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * scale
    return x * scale.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)


def resnet50(sd, x, prefix, out_features):
    """D2 ResNet-50, STRIDE_IN_1X1 True, FrozenBN [D2-recall, SURVEY appendix C]."""
    outs = {}
    x = _conv2d(x, sd[prefix + ".stem.conv1.weight"], None, 2, 3)
    x = F.relu(frozen_bn(x, sd, prefix + ".stem.conv1.norm"))
    x = F.max_pool2d(x, 3, 2, 1)
    for name, nblocks in (("res2", 3), ("res3", 4), ("res4", 6), ("res5", 3)):
        for b in range(nblocks):
            p = "%s.%s.%d" % (prefix, name, b)
            stride = 2 if (b == 0 and name != "res2") else 1
            if (p + ".shortcut.weight") in sd:
                sc = frozen_bn(_conv2d(x, sd[p + ".shortcut.weight"], None, stride), sd, p + ".shortcut.norm")
            else:
                sc = x
            o = F.relu(frozen_bn(_conv2d(x, sd[p + ".conv1.weight"], None, stride), sd, p + ".conv1.norm"))
            o = F.relu(frozen_bn(_conv2d(o, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".conv2.norm"))
            o = frozen_bn(_conv2d(o, sd[p + ".conv3.weight"], None, 1), sd, p + ".conv3.norm")
            x = F.relu(o + sc)
        if name in out_features:
            outs[name] = x
    return outs


def fpn(sd, feats, in_features, top="p6p7", prefix="backbone"):
    """D2 FPN (sum fuse, no norm) + LastLevelP6P7 from P5 (backbone/fpn.py:11-29,65) or LastLevelMaxPool."""
    stage = {"res2": 2, "res3": 3, "res4": 4, "res5": 5}
    res = {}
    prev = None
    for f in reversed(in_features):
        s = stage[f]
        lat = _conv2d(feats[f], sd["%s.fpn_lateral%d.weight" % (prefix, s)], sd["%s.fpn_lateral%d.bias" % (prefix, s)])
        if prev is not None:
            lat = lat + F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = lat
        res["p%d" % s] = _conv2d(lat, sd["%s.fpn_output%d.weight" % (prefix, s)], sd["%s.fpn_output%d.bias" % (prefix, s)], 1, 1)
    last = stage[in_features[-1]]
    if top == "p6p7":
        p6 = _conv2d(res["p%d" % last], sd[prefix + ".top_block.p6.weight"], sd[prefix + ".top_block.p6.bias"], 2, 1)
        p7 = _conv2d(F.relu(p6), sd[prefix + ".top_block.p7.weight"], sd[prefix + ".top_block.p7.bias"], 2, 1)
        res["p%d" % (last + 1)] = p6
        res["p%d" % (last + 2)] = p7
    elif top == "maxpool":
        res["p%d" % (last + 1)] = F.max_pool2d(res["p%d" % last], kernel_size=1, stride=2, padding=0)
    return res


def preprocess(images, mean, std, div):
    """one_stage_detector.py:88-90: normalise, ImageList.from_tensors(pad to divisibility)."""
    ims = [(x.float() - mean) / std for x in images]
    sizes = [(t.shape[-2], t.shape[-1]) for t in ims]
    hm, wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
    hm, wm = (hm + div - 1) // div * div, (wm + div - 1) // div * div
    out = ims[0].new_zeros((len(ims), 3, hm, wm))
    for i, t in enumerate(ims):
        out[i, :, : t.shape[-2], : t.shape[-1]] = t
    return out, sizes


# =================================================================================================
# FCOS head (fcos/fcos.py:220-376)
# =================================================================================================
def fcos_head(sd, feats, num_levels=5, prefix="proposal_generator.fcos_head"):
    logits, reg, std, ctr = [], [], [], []
    for l, f in enumerate(feats):
        def tower(name, x):
            i = 0
            while "%s.%s_tower.%d.weight" % (prefix, name, 3 * i) in sd:
                x = _conv2d(x, sd["%s.%s_tower.%d.weight" % (prefix, name, 3 * i)], sd["%s.%s_tower.%d.bias" % (prefix, name, 3 * i)], 1, 1)
                x = F.group_norm(x, 32, sd["%s.%s_tower.%d.weight" % (prefix, name, 3 * i + 1)], sd["%s.%s_tower.%d.bias" % (prefix, name, 3 * i + 1)])
                x = F.relu(x)
                i += 1
            return x
        f = tower("share", f)
        ct, bt = tower("cls", f), tower("bbox", f)
        logits.append(_conv2d(ct, sd[prefix + ".cls_logits.weight"], sd[prefix + ".cls_logits.bias"], 1, 1))
        ctr.append(_conv2d(bt, sd[prefix + ".ctrness.weight"], sd[prefix + ".ctrness.bias"], 1, 1))
        r = _conv2d(bt, sd[prefix + ".bbox_pred.weight"], sd[prefix + ".bbox_pred.bias"], 1, 1)
        key = "%s.scales.%d.scale" % (prefix, l)
        if key in sd:
            r = r * sd[key]
        reg.append(r)  # REG_DISCRETE: no relu (fcos.py:360-364)
        std.append(_conv2d(bt, sd[prefix + ".bbox_pred_std.weight"], sd[prefix + ".bbox_pred_std.bias"], 1, 1))
    return logits, reg, std, ctr


def compute_locations(h, w, stride):
    """utils/comm.py:34-45."""
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    return torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1) + stride // 2


# =================================================================================================
# FCOSOutputs restatement (fcos/fcos_outputs.py)
# =================================================================================================
def integral(x, reg_max=16):
    """fcos_outputs.py:44-77."""
    p = F.softmax(x.reshape(-1, reg_max + 1), dim=1)
    return F.linear(p, torch.linspace(0, reg_max, reg_max + 1)).reshape(-1, 4)


def ctrness_targets(t):
    """fcos_outputs.py:80-88."""
    if len(t) == 0:
        return t.new_zeros(len(t))
    lr, tb = t[:, [0, 2]], t[:, [1, 3]]
    return torch.sqrt((lr.min(dim=-1)[0] / lr.max(dim=-1)[0]) * (tb.min(dim=-1)[0] / tb.max(dim=-1)[0]))


def iou_targets(pred, target):
    """fcos_outputs.py:91-129."""
    if len(target) == 0:
        return target.new_zeros(len(target))
    ta = (target[:, 0] + target[:, 2]) * (target[:, 1] + target[:, 3])
    pa = (pred[:, 0] + pred[:, 2]) * (pred[:, 1] + pred[:, 3])
    wi = torch.min(pred[:, 0], target[:, 0]) + torch.min(pred[:, 2], target[:, 2])
    hi = torch.min(pred[:, 3], target[:, 3]) + torch.min(pred[:, 1], target[:, 1])
    ai = wi * hi
    return (ai + 1.0) / (ta + pa - ai + 1.0)


def giou_loss_ltrb(pred, target, weight=None, loc_loss_type="giou"):
    """layers/iou_loss.py:20-76; loc_loss_type 'giou' (every shipped config), 'iou' or 'linear_iou' (:64-69)."""
    ta = (target[:, 0] + target[:, 2]) * (target[:, 1] + target[:, 3])
    pa = (pred[:, 0] + pred[:, 2]) * (pred[:, 1] + pred[:, 3])
    wi = torch.min(pred[:, 0], target[:, 0]) + torch.min(pred[:, 2], target[:, 2])
    hi = torch.min(pred[:, 3], target[:, 3]) + torch.min(pred[:, 1], target[:, 1])
    gw = torch.max(pred[:, 0], target[:, 0]) + torch.max(pred[:, 2], target[:, 2])
    gh = torch.max(pred[:, 3], target[:, 3]) + torch.max(pred[:, 1], target[:, 1])
    ac = gw * gh
    ai = wi * hi
    au = ta + pa - ai
    ious = (ai + 1.0) / (au + 1.0)
    gious = ious - (ac - au) / ac
    if loc_loss_type == "iou":
        losses = -torch.log(ious)
    elif loc_loss_type == "linear_iou":
        losses = 1 - ious
    elif loc_loss_type == "giou":
        losses = 1 - gious
    else:
        raise NotImplementedError
    return (losses * weight).sum() if weight is not None else losses.sum()


def nl_loss(inp, inp_std, target, iou_weight):
    """layers/kl_loss.py:69-105 (NLLoss)."""
    sigma = inp_std.sigmoid()
    sq = torch.square(sigma)
    first = torch.square(target - inp) / (2 * sq)
    second = 0.5 * torch.log(sq)
    s = (first + second).sum(dim=1) + 2 * torch.log(2 * torch.Tensor([math.pi]))
    return (s * iou_weight).mean()


def kl_loss(inp, inp_std, target, beta=1.0, method="mean", weight=None, loss_denorm=None):
    """layers/kl_loss.py:11-66 (KLLoss) as fcos_outputs.py calls it: default beta 1.0, method = MODEL.FCOS.LOC_FUN_ALL (config.py
    default "mean"); input_std enters raw (no sigmoid); `weight` (the centerness / quality targets) and `loss_denorm` are used by the
    weight_ctr_* methods only (:52-58), the IoU weight by none."""
    n = torch.abs(inp - target)
    l1_smooth = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    loss = torch.exp(-inp_std) * l1_smooth + 0.5 * inp_std
    if method == "weight_ctr_sum":
        return (loss.sum(dim=1) * weight).sum()
    if method == "weight_ctr_mean":
        return (loss.sum(dim=1) * weight).sum() / loss_denorm
    if method == "sum":
        return loss.sum()
    if method == "mean":
        return loss.mean()
    raise ValueError("No defined regression loss method")


class FCOSCfg:
    """The FCOSOutputs constructor state (fcos_outputs.py:133-208) for the shipped UTv2 FCOS configs."""

    def __init__(self, **kw):
        self.alpha, self.gamma = 0.25, 2.0
        self.num_classes = 80
        self.strides = [8, 16, 32, 64, 128]
        self.soi_edges = [64, 128, 256, 512]
        self.reg_max = 16
        self.kl_weight = 0.05
        self.ts_better, self.ts_better_cert = 0.1, 0.8
        self.pre_nms_thresh, self.pre_nms_topk, self.post_nms_topk = 0.05, 1000, 100
        self.nms_thresh = 0.6
        self.unify_ctrcls = False
        self.center_sample, self.radius = False, 1.5  # MODEL.FCOS.CENTER_SAMPLE / POS_RADIUS (config.py defaults)
        # config-reachable variants (config.py:153,168,196-198; SEMISUPNET.CONSIST_REG_LOSS :191)
        self.kl_loss, self.kl_loss_type, self.quality_est, self.loc_loss_type = True, "nlloss", "centerness", "giou"
        self.reg_unsup_loss = "ts_locvar_better_nms_nll_l1"
        self.loc_fun_all = "mean"   # MODEL.FCOS.LOC_FUN_ALL: the reduction of the KLLoss variant (kl_loss.py:48-64); NLLoss ignores it
        self.__dict__.update(kw)
        soi, prev = [], -1
        for s in self.soi_edges:
            soi.append([prev, s])
            prev = s
        soi.append([prev, INF])
        self.soi = soi


def center_sample_region(cfg, boxes, num_loc, xs, ys):
    """fcos_outputs.py:700-770 get_sample_region (no bitmasks): a location is positive for a box only inside the square of
    half-width radius*stride around the box centre, clipped to the box.  Quirk kept: an all-False mask when the FIRST box's
    centre x is 0 (`center_x[..., 0].sum() == 0`)."""
    cx = boxes[:, [0, 2]].sum(dim=-1) * 0.5
    cy = boxes[:, [1, 3]].sum(dim=-1) * 0.5
    K, G = len(xs), boxes.shape[0]
    if cx.numel() == 0 or float(cx[0]) * K == 0:
        return torch.zeros((K, G), dtype=torch.bool)
    b = boxes[None].expand(K, G, 4)
    cxe, cye = cx[None].expand(K, G), cy[None].expand(K, G)
    cg = torch.zeros((K, G, 4), dtype=boxes.dtype)
    beg = 0
    for level, n in enumerate(num_loc):
        end = beg + n
        st = cfg.strides[level] * cfg.radius
        xmin, ymin, xmax, ymax = cxe[beg:end] - st, cye[beg:end] - st, cxe[beg:end] + st, cye[beg:end] + st
        cg[beg:end, :, 0] = torch.where(xmin > b[beg:end, :, 0], xmin, b[beg:end, :, 0])
        cg[beg:end, :, 1] = torch.where(ymin > b[beg:end, :, 1], ymin, b[beg:end, :, 1])
        cg[beg:end, :, 2] = torch.where(xmax > b[beg:end, :, 2], b[beg:end, :, 2], xmax)
        cg[beg:end, :, 3] = torch.where(ymax > b[beg:end, :, 3], b[beg:end, :, 3], ymax)
        beg = end
    left, right = xs[:, None] - cg[..., 0], cg[..., 2] - xs[:, None]
    top, bottom = ys[:, None] - cg[..., 1], cg[..., 3] - ys[:, None]
    return torch.stack((left, top, right, bottom), -1).min(-1)[0] > 0


def fcos_targets(cfg, locations, gts, ignore_near=False):
    """fcos_outputs.py:649-698 + 772-906 (CENTER_SAMPLE per cfg.center_sample; ignore_near = SEMISUPNET.PSEUDO_CLS_IGNORE_NEAR, :841-851).
    gts: list of dict(boxes [G,4], classes [G] long, reg_pred_std [G,4] optional).
    Returns level-first dict of lists (labels, reg_targets (stride-normalised), boundary_vars, target_inds,
    keep_locations)."""
    num_loc = [len(l) for l in locations]
    size_ranges = torch.cat([torch.tensor(cfg.soi[i], dtype=torch.float32)[None].expand(n, -1) for i, n in enumerate(num_loc)])
    locs = torch.cat(locations, dim=0)
    xs, ys = locs[:, 0], locs[:, 1]
    out = {k: [] for k in ("labels", "reg_targets", "target_inds", "keep_locations", "boundary_vars")}
    num_targets = 0
    for g in gts:
        bboxes, labels_im = g["boxes"], g["classes"]
        bvar_im = g["reg_pred_std"] if g.get("reg_pred_std") is not None else torch.zeros_like(bboxes)
        L = locs.size(0)
        if bboxes.numel() == 0:
            out["labels"].append(labels_im.new_zeros(L) + cfg.num_classes)
            out["reg_targets"].append(locs.new_zeros((L, 4)))
            out["boundary_vars"].append(locs.new_zeros((L, 4)))
            out["target_inds"].append(labels_im.new_zeros(L) - 1)
            out["keep_locations"].append(torch.zeros(L, dtype=torch.bool))
            continue
        area = (bboxes[:, 2] - bboxes[:, 0]) * (bboxes[:, 3] - bboxes[:, 1])
        l = xs[:, None] - bboxes[:, 0][None]
        t = ys[:, None] - bboxes[:, 1][None]
        r = bboxes[:, 2][None] - xs[:, None]
        b = bboxes[:, 3][None] - ys[:, None]
        reg = reg_all = torch.stack([l, t, r, b], dim=2)
        if cfg.center_sample:
            is_in = center_sample_region(cfg, bboxes, num_loc, xs, ys)
        else:
            is_in = reg.min(dim=2)[0] > 0
        mx = reg.max(dim=2)[0]
        cared = (mx >= size_ranges[:, [0]]) & (mx <= size_ranges[:, [1]])
        a = area[None].repeat(L, 1)
        a[is_in == 0] = INF
        a[cared == 0] = INF
        amin, ginds = a.min(dim=1)
        reg = reg[range(L), ginds]
        tinds = ginds + num_targets
        num_targets += len(bboxes)
        lab = labels_im[ginds].clone()
        lab[amin == INF] = cfg.num_classes
        bv = bvar_im[ginds].clone()
        bv[amin == INF] = 99999.0
        out["labels"].append(lab)
        out["reg_targets"].append(reg)
        out["target_inds"].append(tinds)
        if ignore_near:  # :841-848: a location inside some box is kept only if it is inside some box's sampling region
            keep = ~((reg_all.min(dim=2)[0] > 0).sum(1) > 0) | (is_in.sum(1) > 0)
        else:
            keep = torch.ones(L, dtype=torch.bool)
        out["keep_locations"].append(keep)
        out["boundary_vars"].append(bv)

    def transpose(lst):
        per_im = [torch.split(x, num_loc, dim=0) for x in lst]
        return [torch.cat(lv, dim=0) for lv in zip(*per_im)]

    out = {k: transpose(v) for k, v in out.items()}
    for la in range(len(out["reg_targets"])):
        out["reg_targets"][la] = out["reg_targets"][la] / float(cfg.strides[la])
    return out


def _flatten_preds(cfg, logits, reg, std, ctr):
    """fcos_outputs.py:261-290: NCHW per level -> level-first [P, C]."""
    R = 4 * (cfg.reg_max + 1)
    lg = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, cfg.num_classes) for x in logits])
    rg = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, R) for x in reg])
    sd_ = torch.cat([x.permute(0, 2, 3, 1).reshape(-1, 4) for x in std])
    ct = torch.cat([x.permute(0, 2, 3, 1).reshape(-1) for x in ctr])
    return lg, rg, sd_, ct


def fcos_losses(cfg, logits, reg, std, ctr, locations, gts, world_size=1, ignore_near=False):
    """Supervised branch: fcos_outputs.py:212-444 (branch 'labeled')."""
    tg = fcos_targets(cfg, locations, gts, ignore_near)
    labels = torch.cat([x.reshape(-1) for x in tg["labels"]])
    keep = torch.cat([x.reshape(-1) for x in tg["keep_locations"]])
    regt = torch.cat([x.reshape(-1, 4) for x in tg["reg_targets"]])
    lg, rg, sdv, ct = _flatten_preds(cfg, logits, reg, std, ctr)
    any_keep = keep.sum() > 0
    if any_keep:  # :310-311
        labels, regt, lg, rg, sdv, ct = labels[keep], regt[keep], lg[keep], rg[keep], sdv[keep], ct[keep]
    pos = torch.nonzero(labels != cfg.num_classes).squeeze(1)
    num_pos_avg = max(pos.numel() / world_size, 1.0)
    tgt = torch.zeros_like(lg)
    tgt[pos, labels[pos]] = 1
    class_loss = sigmoid_focal_loss(lg, tgt, cfg.alpha, cfg.gamma).sum(1).sum() / num_pos_avg
    rg, sdv, ct, regt = rg[pos], sdv[pos], ct[pos], regt[pos]
    reg_pred = integral(rg, cfg.reg_max) if pos.numel() > 0 else rg
    if cfg.quality_est == "centerness":  # :353-359
        ctr_t = ctrness_targets(regt)
    elif cfg.quality_est == "iou":
        ctr_t = iou_targets(reg_pred.detach(), regt)
    else:
        raise NotImplementedError
    loss_denorm = max(ctr_t.sum().item() / world_size, 1e-6)
    if pos.numel() > 0:
        iou_t = iou_targets(reg_pred.detach(), regt)
        ctr_loss = F.binary_cross_entropy_with_logits(ct, ctr_t, reduction="sum") / num_pos_avg
        iou_loss = giou_loss_ltrb(reg_pred, regt, ctr_t, cfg.loc_loss_type) / loss_denorm
        if cfg.kl_loss:
            if cfg.kl_loss_type == "nlloss":
                kl = cfg.kl_weight * nl_loss(reg_pred, sdv, regt, iou_t)  # :400
            elif cfg.kl_loss_type == "klloss":
                kl = cfg.kl_weight * kl_loss(reg_pred, sdv, regt, method=cfg.loc_fun_all, weight=ctr_t, loss_denorm=loss_denorm)  # :381
            else:
                raise NotImplementedError
            reg_loss = cfg.kl_weight * kl + iou_loss  # :397,:416 (weight applied twice)
        else:
            reg_loss = iou_loss  # :418-423
    else:
        reg_loss = torch.tensor(0.0)
        ctr_loss = torch.tensor(0.0)
    if not any_keep:  # :430-434
        class_loss, reg_loss, ctr_loss = class_loss * 0, reg_loss * 0, ctr_loss * 0
    return {"loss_fcos_cls": class_loss, "loss_fcos_loc": reg_loss, "loss_fcos_ctr": ctr_loss}, tg


def fcos_pseudo_losses(cfg, logits, reg, std, ctr, locations, gt_dict, world_size=1, ignore_near=False):
    """Unsupervised branch: fcos_outputs.py:447-631 (cls set -> cls+ctr; reg set -> loc).  ignore_near only fills keep_locations (:474),
    which this branch never reads (:487-631): PSEUDO_CLS_IGNORE_NEAR does not change a pseudo loss."""
    losses, extras = {}, {}
    lg, rg, sdv, ct = _flatten_preds(cfg, logits, reg, std, ctr)
    for labeltype, gts in gt_dict.items():
        tg = fcos_targets(cfg, locations, gts, ignore_near)
        extras[labeltype] = tg
        labels = torch.cat([x.reshape(-1) for x in tg["labels"]])
        regt = torch.cat([x.reshape(-1, 4) for x in tg["reg_targets"]])
        bvar = torch.cat([x.reshape(-1, 4) for x in tg["boundary_vars"]])
        pos = torch.nonzero(labels != cfg.num_classes).squeeze(1)
        num_pos_avg = max(pos.numel() / world_size, 1.0)
        if labeltype == "cls":
            tgt = torch.zeros_like(lg)
            tgt[pos, labels[pos]] = 1
            losses["loss_fcos_cls"] = sigmoid_focal_loss(lg, tgt, cfg.alpha, cfg.gamma).sum(1).sum() / num_pos_avg
        ctr_t = ctrness_targets(regt[pos])
        if pos.numel() > 0:
            if labeltype == "cls":
                cl = F.binary_cross_entropy_with_logits(ct[pos], ctr_t, reduction="sum") / num_pos_avg
                losses["loss_fcos_ctr"] = cl * 0 if cfg.unify_ctrcls else cl
            elif not cfg.kl_loss:
                raise ValueError  # :587-588
            elif cfg.reg_unsup_loss != "ts_locvar_better_nms_nll_l1":  # :571-585: weight * (KL | NLL) term
                reg_pred = integral(rg[pos], cfg.reg_max)
                if cfg.kl_loss_type == "nlloss":
                    term = nl_loss(reg_pred, sdv[pos], regt[pos], iou_targets(reg_pred.detach(), regt[pos]))
                else:
                    term = kl_loss(reg_pred, sdv[pos], regt[pos], method=cfg.loc_fun_all, weight=ctr_t,
                                   loss_denorm=max(ctr_t.sum().item() / world_size, 1e-6))   # :521,:577-584
                losses["loss_fcos_loc"] = cfg.kl_weight * term
            else:
                reg_pred = integral(rg[pos], cfg.reg_max)
                conf_s = 1 - sdv[pos].sigmoid()
                conf_t = 1 - bvar[pos].sigmoid()
                select = (conf_t > cfg.ts_better_cert) * (conf_t > conf_s + cfg.ts_better)
                losses["teacher_better_student"] = select.sum()
                if select.sum() > 0:
                    losses["loss_fcos_loc"] = F.smooth_l1_loss(reg_pred[select], regt[pos][select], beta=0.0)
                else:
                    losses["loss_fcos_loc"] = torch.tensor(0.0)
        else:
            if labeltype == "cls":
                losses["loss_fcos_ctr"] = torch.tensor(0.0)
            else:
                losses["loss_fcos_loc"] = torch.tensor(0.0)
                losses["teacher_better_student"] = torch.tensor(0.0)
    return losses, extras


def fcos_predict(cfg, logits, reg, std, ctr, locations, image_sizes, nms_method):
    """fcos_outputs.py:1046-1320.  Returns per image a dict of tensors (post NMS, kthvalue top-k rule).
    Pre-NMS top-k: (ranking score desc, flat (loc,class) index asc) - the order upstream leaves open."""
    N = logits[0].shape[0]
    per_im = [[] for _ in range(N)]
    for lvl, (loc, o, r, c, sd_) in enumerate(zip(locations, logits, reg, ctr, std)):
        n, C, H, W = o.shape
        s = cfg.strides[lvl]
        rs = integral(r.permute(0, 2, 3, 1).reshape(-1, 4 * (cfg.reg_max + 1)), cfg.reg_max).reshape(n, H * W, 4) * s
        p = o.permute(0, 2, 3, 1).reshape(n, -1, C).sigmoid()
        cs = c.permute(0, 2, 3, 1).reshape(n, -1).sigmoid()
        st = sd_.permute(0, 2, 3, 1).reshape(n, -1, 4)
        cand = p > cfg.pre_nms_thresh
        if nms_method == "cls_n_ctr":
            rank = p * cs[:, :, None]
        elif nms_method == "cls":
            rank = p
        elif nms_method == "ctr":
            rank = cs[:, :, None].expand_as(p)
        elif nms_method == "cls_n_loc":
            rank = p * (1 - st.sigmoid()).mean(2)[:, :, None]
        else:
            raise ValueError("Undefined nms criteria")
        for i in range(n):
            nz = cand[i].nonzero()
            bl, cl = nz[:, 0], nz[:, 1]
            rk = rank[i][cand[i]]
            k = min(int(cand[i].sum()), cfg.pre_nms_topk)
            if len(rk) > k:
                flat = bl * C + cl
                order = np.lexsort((flat.numpy(), -rk.numpy().astype(np.float64)))[:k]
                order = torch.as_tensor(order)
                bl, cl, rk = bl[order], cl[order], rk[order]
            box = torch.stack([loc[bl, 0] - rs[i][bl, 0], loc[bl, 1] - rs[i][bl, 1],
                               loc[bl, 0] + rs[i][bl, 2], loc[bl, 1] + rs[i][bl, 3]], dim=1)
            sc = torch.sqrt(rk) if nms_method in ("cls_n_ctr", "cls_n_loc") else rk
            per_im[i].append(dict(boxes=box, scores=sc, classes=cl, locations=loc[bl], centerness=cs[i][bl],
                                  cls_confid=p[i][bl, cl], reg_pred_std=st[i][bl],
                                  fpn_levels=torch.full((len(bl),), lvl, dtype=torch.long)))
    results = []
    for i in range(N):
        d = {k: torch.cat([x[k] for x in per_im[i]]) for k in per_im[i][0]}
        keep = batched_nms(d["boxes"], d["scores"], d["classes"], cfg.nms_thresh)
        d = {k: v[keep] for k, v in d.items()}
        nd = len(keep)
        if nd > cfg.post_nms_topk > 0:
            thr, _ = torch.kthvalue(d["scores"], nd - cfg.post_nms_topk + 1)
            kk = torch.nonzero(d["scores"] >= thr.item()).squeeze(1)
            d = {k: v[kk] for k, v in d.items()}
        d["image_size"] = image_sizes[i]
        results.append(d)
    return results


def threshold_bbox(det, thr):
    """pseudo_generator.py:62-105 ('roih')."""
    m = det["scores"] > thr
    return dict(boxes=det["boxes"][m], classes=det["classes"][m], scores=det["scores"][m],
                centerness=det["centerness"][m], cls_confid=det["cls_confid"][m], reg_pred_std=det["reg_pred_std"][m])


def threshold_cls_ctr_bbox(det, thr):
    """pseudo_generator.py:107-131: keep a detection when BOTH its classification confidence and its centerness pass
    (`cls_confid > thr[0]` and `centerness > thr[1]`, strict); the ranking score is carried along, not tested."""
    m = (det["cls_confid"] > thr[0]) & (det["centerness"] > thr[1])
    return dict(boxes=det["boxes"][m], classes=det["classes"][m], scores=det["scores"][m],
                centerness=det["centerness"][m], cls_confid=det["cls_confid"][m], reg_pred_std=det["reg_pred_std"][m])


def process_pseudo_label(dets, cur_threshold, method):
    """pseudo_generator.py:39-60: per-image thresholding by the configured method + the mean number of kept boxes."""
    if method == "thresholding":
        out = [threshold_bbox(d, cur_threshold) for d in dets]
    elif method == "thresholding_cls_ctr":
        out = [threshold_cls_ctr_bbox(d, cur_threshold) for d in dets]
    else:
        raise ValueError("Unkown pseudo label boxes methods")
    return out, sum(len(o["scores"]) for o in out) / float(len(dets))


# =================================================================================================
# EMA / SGD / step
# =================================================================================================
def ema_update(student_sd, teacher_sd, keep_rate):
    """engine/trainer.py:468-486."""
    new = OrderedDict()
    for k, v in teacher_sd.items():
        if k not in student_sd:
            raise Exception("{} is not found in student model".format(k))
        new[k] = student_sd[k] * (1 - keep_rate) + v * keep_rate
    return new


def sgd_step(params, grads, bufs, lr, momentum, wd):
    """torch.optim.SGD (momentum, dampening 0, no nesterov) on dicts of tensors; wd: dict name->float."""
    for k, p in params.items():
        g = grads[k] + wd[k] * p
        b = momentum * bufs[k] + g if k in bufs else g.clone()
        bufs[k] = b
        params[k] = p - lr * b
    return params, bufs


def is_norm_param(key):
    return ("_tower." in key and int(key.split(".")[-2]) % 3 == 1)


def fcos_forward(sd, images, mean, std_pix, trainable_keys=None):
    x, sizes = preprocess(images, mean, std_pix, 32)
    c = resnet50(sd, x, "backbone.bottom_up", ("res3", "res4", "res5"))
    p = fpn(sd, c, ["res3", "res4", "res5"], "p6p7")
    feats = [p[k] for k in ("p3", "p4", "p5", "p6", "p7")]
    logits, reg, sdv, ctr = fcos_head(sd, feats)
    locations = [compute_locations(f.shape[2], f.shape[3], s) for f, s in zip(feats, [8, 16, 32, 64, 128])]
    return logits, reg, sdv, ctr, locations, sizes


def fcos_semisup_step(cfg, student_sd, teacher_sd, batch, keep_rate, lam_u=3.0, lam_r=0.2, thr_cls=0.5, thr_reg=0.5,
                      lr=0.01, momentum=0.9, wd=1e-4, bufs=None, mean=None, pix_std=None, frozen_prefixes=("backbone.bottom_up.stem", "backbone.bottom_up.res2"),
                      pseudo_override=None, phase_times=None):
    """One post-burn-in iteration of UBTeacherTrainer.run_step_full_semisup (engine/trainer.py:212-429),
    fp32 (no AMP).  batch = (label_q, label_k, unlabel_q, unlabel_k) lists of dicts with 'image' (+ 'gt').
    phase_times (optional dict): seconds per phase are ADDED to its entries (bench.py's cpu_baseline phase split)."""
    import time as _time
    _t = [_time.perf_counter()]

    def _mark(name):
        if phase_times is not None:
            now = _time.perf_counter()
            phase_times[name] = phase_times.get(name, 0.0) + now - _t[0]
            _t[0] = now
    mean = mean if mean is not None else torch.tensor([103.53, 116.28, 123.675]).view(3, 1, 1)
    pix_std = pix_std if pix_std is not None else torch.ones(3, 1, 1)
    lq, lk, uq, uk = batch
    teacher_sd = ema_update(student_sd, teacher_sd, keep_rate)
    _mark("ema")
    rec = {"ema_rate_1000x": keep_rate * 1000}
    with torch.no_grad():
        tl = fcos_forward(teacher_sd, [d["image"] for d in uk], mean, pix_std)
        _mark("teacher_forward")
        det_cls = fcos_predict(cfg, *tl[:4], tl[4], tl[5], "cls")
        det_loc = fcos_predict(cfg, *tl[:4], tl[4], tl[5], "cls_n_loc")
    # a pair of thresholds selects SEMISUPNET.PSEUDO_BBOX_SAMPLE(_REG) = "thresholding_cls_ctr" (engine/trainer.py:253-276)
    pseudo_cls, _ = process_pseudo_label(det_cls, thr_cls, "thresholding_cls_ctr" if isinstance(thr_cls, (tuple, list)) else "thresholding")
    pseudo_reg, _ = process_pseudo_label(det_loc, thr_reg, "thresholding_cls_ctr" if isinstance(thr_reg, (tuple, list)) else "thresholding")
    _mark("decode_nms_threshold")
    if pseudo_override is not None:  # (mixed-precision tests: decouple the student check from teacher selection noise)
        pseudo_cls, pseudo_reg = pseudo_override
    params = {k: v.clone().requires_grad_(True) for k, v in student_sd.items()
              if v.dtype.is_floating_point and "norm." not in k and not k.startswith(frozen_prefixes)
              and k not in ("pixel_mean", "pixel_std") and not k.endswith("integral.project")}
    sd = dict(student_sd)
    sd.update(params)
    out = fcos_forward(sd, [d["image"] for d in lq + lk], mean, pix_std)
    sup, _ = fcos_losses(cfg, *out[:4], out[4], [d["gt"] for d in lq + lk])
    rec.update(sup)
    out = fcos_forward(sd, [d["image"] for d in uq], mean, pix_std)
    uns, _ = fcos_pseudo_losses(cfg, *out[:4], out[4], {"cls": pseudo_cls, "reg": pseudo_reg})
    for k, v in uns.items():
        rec[k + "_pseudo"] = v
    _mark("student_forward_losses")
    total = 0.0
    for k, v in rec.items():
        if k[:4] != "loss":
            continue
        if k in ("loss_fcos_ctr", "loss_fcos_cls"):
            total = total + v / (lam_u + 1.0)
        elif k in ("loss_fcos_ctr_pseudo", "loss_fcos_cls_pseudo"):
            total = total + v * lam_u / (lam_u + 1.0)
        elif k == "loss_fcos_loc":
            total = total + v / (lam_r + 1.0)
        elif k == "loss_fcos_loc_pseudo":
            total = total + v * lam_r / (lam_r + 1.0)
        else:
            total = total + v / (lam_u + 1.0)
    grads_l = torch.autograd.grad(total, list(params.values()), allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(params.keys(), grads_l)}
    _mark("backward")
    wds = {k: (0.0 if is_norm_param(k) else wd) for k in params}
    bufs = bufs if bufs is not None else {}
    newp, bufs = sgd_step({k: v.detach() for k, v in params.items()}, grads, bufs, lr, momentum, wds)
    _mark("sgd")
    new_student = OrderedDict(student_sd)
    new_student.update(newp)
    rec = {k: (float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in rec.items()}
    return rec, new_student, teacher_sd, grads, bufs, (pseudo_cls, pseudo_reg)


# =================================================================================================
# Faster-RCNN path
# =================================================================================================
SCALE_CLAMP = math.log(1000.0 / 16)


def xyxy_get_deltas(src, tgt, weights=(10.0, 10.0, 5.0, 5.0)):
    """modeling/box_regression.py:38-73 (order l, r, d, u; +1 on the source size; only weights[0:2])."""
    sw = src[:, 2] - src[:, 0] + 1.0
    sh = src[:, 3] - src[:, 1] + 1.0
    wx, wy = weights[0], weights[1]
    return torch.stack((wx * (tgt[:, 0] - src[:, 0]) / sw, wx * (tgt[:, 2] - src[:, 2]) / sw,
                        wy * (tgt[:, 1] - src[:, 1]) / sh, wy * (tgt[:, 3] - src[:, 3]) / sh), dim=1)


def xyxy_apply_deltas(deltas, boxes, weights=(10.0, 10.0, 5.0, 5.0), clamp=1000.0 / 16):
    """modeling/box_regression.py:75-129 (no +1, clamp +-62.5, class-agnostic 4 columns)."""
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    wx, wy = weights[0], weights[1]
    dl = torch.clamp(deltas[:, 0] / wx, max=clamp, min=-clamp)
    dr = torch.clamp(deltas[:, 1] / wx, max=clamp, min=-clamp)
    dd = torch.clamp(deltas[:, 2] / wy, max=clamp, min=-clamp)
    du = torch.clamp(deltas[:, 3] / wy, max=clamp, min=-clamp)
    return torch.stack((dl * w + boxes[:, 0], dd * h + boxes[:, 1], dr * w + boxes[:, 2], du * h + boxes[:, 3]), dim=1)


def rpn_get_deltas(src, tgt):
    """D2 Box2BoxTransform.get_deltas, weights (1,1,1,1) [D2-recall]."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    return torch.stack(((tx - sx) / sw, (ty - sy) / sh, torch.log(tw / sw), torch.log(th / sh)), dim=1)


def rpn_apply_deltas(deltas, boxes):
    w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
    dw, dh = torch.clamp(deltas[:, 2], max=SCALE_CLAMP), torch.clamp(deltas[:, 3], max=SCALE_CLAMP)
    pcx, pcy = deltas[:, 0] * w + cx, deltas[:, 1] * h + cy
    pw, ph = torch.exp(dw) * w, torch.exp(dh) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=1)


def matcher(iou, thresholds, labels, allow_low_quality):
    """D2 Matcher [D2-recall].  iou [G, P] -> (matched idx [P], labels [P])."""
    if iou.numel() == 0:
        P = iou.shape[1]
        return torch.zeros(P, dtype=torch.int64), torch.full((P,), labels[0], dtype=torch.int8)
    vals, idx = iou.max(dim=0)
    out = torch.full(vals.shape, 1, dtype=torch.int8)
    th = [-float("inf")] + list(thresholds) + [float("inf")]
    for l, lo, hi in zip(labels, th[:-1], th[1:]):
        out[(vals >= lo) & (vals < hi)] = l
    if allow_low_quality:
        best, _ = iou.max(dim=1)
        out[(iou == best[:, None]).any(dim=0)] = 1
    return idx, out


def subsample_by_keys(labels, keys, num, frac, bg):
    """D2 subsample_labels with randperm[:k] replaced by 'k smallest per-slot keys' (same distribution; lets
    the product and the oracle draw the same sample).  Returns (pos idx, neg idx)."""
    pos = torch.nonzero((labels != -1) & (labels != bg)).squeeze(1)
    neg = torch.nonzero(labels == bg).squeeze(1)
    npos = min(int(num * frac), pos.numel())
    nneg = min(num - npos, neg.numel())
    pos = pos[torch.argsort(keys[pos], stable=True)[:npos]]
    neg = neg[torch.argsort(keys[neg], stable=True)[:nneg]]
    return pos, neg


def make_anchors(level_hw, strides, sizes=(32, 64, 128, 256, 512), ratios=(0.5, 1.0, 2.0)):
    """D2 DefaultAnchorGenerator, offset 0, order (H, W, A) [D2-recall]."""
    out = []
    for (h, w), s, sz in zip(level_hw, strides, sizes):
        cell = []
        for r in ratios:
            ww = math.sqrt(sz ** 2.0 / r)
            hh = r * ww
            cell.append([-ww / 2.0, -hh / 2.0, ww / 2.0, hh / 2.0])
        cell = torch.tensor(cell, dtype=torch.float32)
        sx = torch.arange(0, w * s, step=s, dtype=torch.float32)
        sy = torch.arange(0, h * s, step=s, dtype=torch.float32)
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
        out.append((shifts.view(-1, 1, 4) + cell.view(1, -1, 4)).reshape(-1, 4))
    return out


def rpn_head(sd, feats, prefix="proposal_generator.rpn_head"):
    """D2 StandardRPNHead [D2-recall]; outputs flattened to the (H, W, A) anchor order of rpn.py:33-46."""
    obj, dl = [], []
    for f in feats:
        t = F.relu(_conv2d(f, sd[prefix + ".conv.weight"], sd[prefix + ".conv.bias"], 1, 1))
        o = _conv2d(t, sd[prefix + ".objectness_logits.weight"], sd[prefix + ".objectness_logits.bias"])
        d = _conv2d(t, sd[prefix + ".anchor_deltas.weight"], sd[prefix + ".anchor_deltas.bias"])
        obj.append(o.permute(0, 2, 3, 1).flatten(1))
        dl.append(d.view(d.shape[0], -1, 4, d.shape[-2], d.shape[-1]).permute(0, 3, 4, 1, 2).flatten(1, -2))
    return obj, dl


def find_top_rpn_proposals(anchors, obj, deltas, image_sizes, pre_topk, post_topk, nms_thresh=0.7):
    """D2 find_top_rpn_proposals [D2-recall]; per-level order (logit desc, anchor index asc)."""
    N = obj[0].shape[0]
    res = []
    for n in range(N):
        bs, ss, ls = [], [], []
        for l, (a, o, d) in enumerate(zip(anchors, obj, deltas)):
            k = min(pre_topk, o.shape[1])
            order = np.lexsort((np.arange(o.shape[1]), -o[n].detach().numpy().astype(np.float64)))[:k]
            order = torch.as_tensor(order)
            bs.append(rpn_apply_deltas(d[n][order].detach(), a[order])); ss.append(o[n][order].detach())
            ls.append(torch.full((k,), l, dtype=torch.int64))
        b, s, lv = torch.cat(bs), torch.cat(ss), torch.cat(ls)
        ok = torch.isfinite(b).all(dim=1) & torch.isfinite(s)
        b, s, lv = b[ok], s[ok], lv[ok]
        h, w = image_sizes[n]
        b = torch.stack((b[:, 0].clamp(0, w), b[:, 1].clamp(0, h), b[:, 2].clamp(0, w), b[:, 3].clamp(0, h)), dim=1)
        keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
        b, s, lv = b[keep], s[keep], lv[keep]
        k = batched_nms(b, s, lv, nms_thresh)[:post_topk]
        res.append(dict(boxes=b[k], logits=s[k]))
    return res


def rpn_losses(anchors_cat, obj_cat, deltas_cat, gts, keys, pseudo, batch=256, frac=0.25):
    """rpn.py:78-225 + D2 label_and_sample_anchors.  gts: list of dict(boxes, scores?).  keys [N, R]."""
    N = obj_cat.shape[0]
    cls_sum, loc_sum = 0.0, 0.0
    samples = []
    for n in range(N):
        gb = gts[n]["boxes"]
        iou = pairwise_iou(gb, anchors_cat) if len(gb) else torch.zeros((0, anchors_cat.shape[0]))
        midx, lab = matcher(iou, [0.3, 0.7], [0, -1, 1], True)
        pos, neg = subsample_by_keys(lab, keys[n], batch, frac, 0)
        samples.append((pos, neg))
        idx = torch.cat((pos, neg))
        tgt = torch.cat((torch.ones(len(pos)), torch.zeros(len(neg))))
        if pseudo:
            wgt = gts[n]["scores"][midx][idx] if len(gb) else torch.zeros(len(idx))  # rpn.py:135-144
            cls_sum = cls_sum + F.binary_cross_entropy_with_logits(obj_cat[n][idx], tgt, weight=wgt, reduction="sum")
        else:
            cls_sum = cls_sum + F.binary_cross_entropy_with_logits(obj_cat[n][idx], tgt, reduction="sum")
        if len(pos) and len(gb):
            t = rpn_get_deltas(anchors_cat[pos], gb[midx[pos]])
            loc_sum = loc_sum + (deltas_cat[n][pos] - t).abs().sum()
    norm = batch * N
    return {"loss_rpn_cls": cls_sum / norm, "loss_rpn_loc": loc_sum / norm}, samples


def roi_align(feat, rois, scale, out=7):
    """torchvision roi_align(aligned=True, sampling_ratio=0) on one NCHW level [D2/torchvision-recall].
    feat [C,H,W] of ONE image; rois [R,4] -> [R,C,out,out]."""
    C, H, W = feat.shape
    R = rois.shape[0]
    res = feat.new_zeros((R, C, out, out))
    for r in range(R):
        x1, y1, x2, y2 = [float(v) * scale - 0.5 for v in rois[r]]
        rw, rh = x2 - x1, y2 - y1
        bw, bh = rw / out, rh / out
        gh, gw = int(math.ceil(rh / out)), int(math.ceil(rw / out))
        if gh <= 0 or gw <= 0:
            continue
        ys = y1 + (torch.arange(out)[:, None] * bh + (torch.arange(gh)[None, :] + 0.5) * bh / gh).reshape(-1)
        xs = x1 + (torch.arange(out)[:, None] * bw + (torch.arange(gw)[None, :] + 0.5) * bw / gw).reshape(-1)

        def prep(v, L):
            okk = (v >= -1.0) & (v <= L)
            v = v.clamp(min=0)
            lo = v.floor().long()
            hi_edge = lo >= L - 1
            lo = torch.where(hi_edge, torch.full_like(lo, L - 1), lo)
            hi = torch.where(hi_edge, lo, lo + 1)
            v = torch.where(hi_edge, lo.to(v.dtype), v)
            frac = v - lo.to(v.dtype)
            return okk, lo, hi, frac
        oky, yl, yh, ly = prep(ys.float(), H)
        okx, xl, xh, lx = prep(xs.float(), W)
        hy, hx = 1 - ly, 1 - lx
        v = (feat[:, yl][:, :, xl] * (hy[:, None] * hx[None, :]) + feat[:, yl][:, :, xh] * (hy[:, None] * lx[None, :]) +
             feat[:, yh][:, :, xl] * (ly[:, None] * hx[None, :]) + feat[:, yh][:, :, xh] * (ly[:, None] * lx[None, :]))
        v = v * (oky[:, None] & okx[None, :]).to(v.dtype)
        res[r] = v.view(C, out, gh, out, gw).sum(dim=(2, 4)) / max(gh * gw, 1)
    return res


# bench.py's cpu_baseline leg sets this: the per-sample form above is the pinned restatement (known-answer tests, step goldens) but
# its autograd backward materialises a full [C,H,W] zero map per ROI and per tap (45 s of a 60 s CPU step at 1333x800) - a cost the
# reference's CPU path (torchvision's C++ roi_align) does not have.  The separable form below is the same arithmetic regrouped
# (tests/test_oracle_golden_rcnn.py::test_fast_roi_align_equals_pinned_form), with a backward that touches each ROI's window once.
FAST_ROI_ALIGN = [False]


def _roi_axis_weights(start, bin_sz, g, out, L):
    """[out, window] weights of one axis: mean over the g samples of a bin of the bilinear taps, torchvision's rules
    (sample skipped outside [-1, L], clamped at 0, last-pixel clamp); returns (P, first pixel, one past the last pixel)."""
    v = (start + (torch.arange(out)[:, None] * bin_sz + (torch.arange(g)[None, :] + 0.5) * bin_sz / g).reshape(-1)).float()
    ok = ((v >= -1.0) & (v <= L)).float()
    v = v.clamp(min=0)
    lo = v.floor().long()
    edge = lo >= L - 1
    lo = torch.where(edge, torch.full_like(lo, L - 1), lo)
    hi = torch.where(edge, lo, lo + 1)
    v = torch.where(edge, lo.to(v.dtype), v)
    frac = v - lo.to(v.dtype)
    a, b = int(lo.min()), int(hi.max()) + 1
    A = torch.zeros(out * g, b - a)
    rows = torch.arange(out * g)
    A[rows, lo - a] += (1 - frac) * ok
    A[rows, hi - a] += frac * ok
    return A.view(out, g, b - a).sum(1) / g, a, b


class _RoiAlignSeparable(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, scale, out):
        C, H, W = feat.shape
        res = feat.new_zeros((rois.shape[0], C, out, out))
        saved = []
        for r in range(rois.shape[0]):
            x1, y1, x2, y2 = [float(v) * scale - 0.5 for v in rois[r]]
            rw, rh = x2 - x1, y2 - y1
            gh, gw = int(math.ceil(rh / out)), int(math.ceil(rw / out))
            if gh <= 0 or gw <= 0:
                saved.append(None)
                continue
            Py, ya, yb = _roi_axis_weights(y1, rh / out, gh, out, H)
            Px, xa, xb = _roi_axis_weights(x1, rw / out, gw, out, W)
            res[r] = torch.einsum("oy,cyx,px->cop", Py, feat[:, ya:yb, xa:xb], Px)
            saved.append((Py, Px, ya, yb, xa, xb))
        ctx.saved, ctx.shape = saved, feat.shape
        return res

    @staticmethod
    def backward(ctx, g):
        gf = g.new_zeros(ctx.shape)
        for r, s in enumerate(ctx.saved):
            if s is not None:
                Py, Px, ya, yb, xa, xb = s
                gf[:, ya:yb, xa:xb] += torch.einsum("oy,cop,px->cyx", Py, g[r], Px)
        return gf, None, None, None


def roi_pool(feats, boxes_per_im, out=7):
    """D2 ROIPooler (levels 2..5, canonical 224 @ level 4) [D2-recall]; feats NCHW list p2..p5."""
    roi_align_ = (lambda f, b, s, o: _RoiAlignSeparable.apply(f, b, s, o)) if FAST_ROI_ALIGN[0] else roi_align
    outs = []
    for n, boxes in enumerate(boxes_per_im):
        area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
        lv = torch.floor(4 + torch.log2(torch.sqrt(area) / 224 + 1e-8)).clamp(2, 5).long() - 2
        o = feats[0].new_zeros((len(boxes), feats[0].shape[1], out, out))
        for l in range(4):
            m = torch.nonzero(lv == l).squeeze(1)
            if len(m):
                o[m] = roi_align_(feats[l][n], boxes[m], 1.0 / (4 * 2 ** l), out)
        outs.append(o)
    return torch.cat(outs) if outs else feats[0].new_zeros((0, feats[0].shape[1], out, out))


def box_head(sd, x, prefix="roi_heads"):
    x = x.flatten(1)
    x = F.relu(_linear(x, sd[prefix + ".box_head.fc1.weight"], sd[prefix + ".box_head.fc1.bias"]))
    x = F.relu(_linear(x, sd[prefix + ".box_head.fc2.weight"], sd[prefix + ".box_head.fc2.bias"]))
    p = prefix + ".box_predictor"
    # no bbox_pred_std head = the UTv1 predictor (MODEL.ROI_HEADS.LOSS "FocalLoss", fast_rcnn.py:1296-1402): Detectron2's two Linear heads
    std = _linear(x, sd[p + ".bbox_pred_std.weight"], sd[p + ".bbox_pred_std.bias"]) if p + ".bbox_pred_std.weight" in sd else None
    return (_linear(x, sd[p + ".cls_score.weight"], sd[p + ".cls_score.bias"]),
            _linear(x, sd[p + ".bbox_pred.weight"], sd[p + ".bbox_pred.bias"]), std)


def softmax_focal(scores, gt_classes, gamma=1.5):
    """roi_heads/fast_rcnn.py:925-936 + FocalLoss :1405-1429."""
    if gt_classes.numel() == 0:
        return 0.0 * scores.sum()
    ce = F.cross_entropy(scores, gt_classes, reduction="none")
    p = torch.exp(-ce)
    return ((1 - p) ** gamma * ce).sum() / gt_classes.shape[0]


def d2_get_deltas(src, tgt, weights):
    """Detectron2 Box2BoxTransform.get_deltas with weights (wx, wy, ww, wh) [D2-recall]"""
    d = rpn_get_deltas(src, tgt)
    return d * torch.tensor(weights, dtype=d.dtype)


def d2_apply_deltas(deltas, boxes, weights):
    """Detectron2 Box2BoxTransform.apply_deltas with weights [D2-recall]; deltas [R, 4k] -> boxes [R, 4k]"""
    R = deltas.shape[0]
    d = deltas.reshape(R, -1, 4) / torch.tensor(weights, dtype=deltas.dtype)
    out = torch.stack([rpn_apply_deltas(d[:, j], boxes) for j in range(d.shape[1])], dim=1)
    return out.reshape(R, -1)


def fv_giou_loss(b1, b2, eps=1e-7):
    """fvcore.nn.giou_loss, reduction none [D2-recall]"""
    x1, y1, x2, y2 = b1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = b2.unbind(dim=-1)
    xk1, yk1, xk2, yk2 = torch.max(x1, x1g), torch.max(y1, y1g), torch.min(x2, x2g), torch.min(y2, y2g)
    inter = torch.where((yk2 > yk1) & (xk2 > xk1), (xk2 - xk1) * (yk2 - yk1), torch.zeros_like(x1))
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    area_c = (torch.max(x2, x2g) - torch.min(x1, x1g)) * (torch.max(y2, y2g) - torch.min(y1, y1g))
    return 1 - (inter / (union + eps) - (area_c - union) / (area_c + eps))


def utv1_roi_losses(scores, deltas, prop, gtb, gt_classes, gt_confid=None, weights=(10.0, 10.0, 5.0, 5.0), num_classes=80, beta=0.0, gamma=1.5,
                    box_reg_loss_type="smooth_l1"):
    """The UTv1 predictor's losses, MODEL.ROI_HEADS.LOSS "FocalLoss" (roi_heads/fast_rcnn.py:1296-1429 FastRCNNFocaltLossOutputLayers ->
    FastRCNNFocalLoss): loss_cls = sum((1 - p)^1.5 * CE [* gt_confid]) / R; loss_box_reg (inherited, :134-194) = smooth-L1 between the
    foreground rows' deltas of their gt class (or the 4 class-agnostic ones) and get_deltas(proposal, gt), summed, / R."""
    R = gt_classes.shape[0]
    ce = F.cross_entropy(scores, gt_classes, reduction="none")
    loss = (1 - torch.exp(-ce)) ** gamma * ce
    if gt_confid is not None:
        loss = loss * gt_confid
    loss_cls = loss.sum() / R
    fg = torch.nonzero((gt_classes >= 0) & (gt_classes < num_classes)).squeeze(1)
    if deltas.shape[1] == 4:
        cols = torch.arange(4)[None, :].expand(len(fg), 4)
    else:
        cols = 4 * gt_classes[fg, None] + torch.arange(4)
    if box_reg_loss_type == "giou":               # fast_rcnn.py:174-183
        return loss_cls, fv_giou_loss(d2_apply_deltas(deltas[fg[:, None], cols], prop[fg], weights), gtb[fg]).sum() / R
    tgt = d2_get_deltas(prop, gtb, weights)
    diff = (deltas[fg[:, None], cols] - tgt[fg]).abs()
    if beta >= 1e-5:
        diff = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)
    return loss_cls, diff.sum() / R


def fast_rcnn_inference_per_class(boxes, probs, image_size, score_thresh=0.05, nms_thresh=0.5, topk=100):
    """D2 fast_rcnn_inference_single_image [D2-recall] with per-class boxes [R, 4K] (or [R, 4]).  Returns (dets, kept proposal row)."""
    ok = torch.isfinite(boxes).all(dim=1) & torch.isfinite(probs).all(dim=1)
    rows = torch.nonzero(ok).squeeze(1)
    boxes, probs = boxes[ok], probs[ok][:, :-1]
    K = probs.shape[1]
    h, w = image_size
    b = boxes.reshape(boxes.shape[0], -1, 4)
    b = torch.stack((b[..., 0].clamp(0, w), b[..., 1].clamp(0, h), b[..., 2].clamp(0, w), b[..., 3].clamp(0, h)), dim=-1)
    m = probs > score_thresh
    fi = m.nonzero()
    bsel = b[fi[:, 0], 0] if b.shape[1] == 1 else b[fi[:, 0], fi[:, 1]]
    s = probs[m]
    keep = batched_nms(bsel, s, fi[:, 1], nms_thresh)[:topk]
    return dict(boxes=bsel[keep], scores=s[keep], classes=fi[keep, 1]), rows[fi[keep, 0]]


def matched_iou(b1, b2):
    """roi_heads/fast_rcnn.py:20-44."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    lt, rb = torch.max(b1[:, :2], b2[:, :2]), torch.min(b1[:, 2:], b2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    return inter / (a1 + a2 - inter)


def roi_box_reg_loss(prop, gtb, deltas, std, gt_classes, num_classes=80):
    """fast_rcnn.py:938-1016, type 'nlloss' + nl_loss :1228-1292 (reduction sum)."""
    fg = torch.nonzero((gt_classes >= 0) & (gt_classes < num_classes)).squeeze(1)
    d, s = deltas[fg], std[fg]
    pred = xyxy_apply_deltas(d, prop[fg])
    iou = matched_iou(gtb[fg], pred)
    t = xyxy_get_deltas(prop[fg], gtb[fg])
    sig = s.sigmoid()
    sq = torch.square(sig)
    nll = ((torch.square(t - d) / (2 * sq) + 0.5 * torch.log(sq)).sum(dim=1) + 2 * torch.log(2 * torch.Tensor([math.pi]))) * iou
    l1 = (d - t).abs().sum()
    return (l1 + 0.05 * nll.sum()) / max(gt_classes.numel(), 1.0)


def roi_box_reg_pseudo_loss(prop, gtb, deltas, std, gt_loc_std, gt_classes, ts_better=0.1, t_cert=0.5, num_classes=80):
    """fast_rcnn.py:1018-1092, type 'tsbetter'."""
    fg = torch.nonzero((gt_classes >= 0) & (gt_classes < num_classes)).squeeze(1)
    d, s = deltas[fg], std[fg]
    t = xyxy_get_deltas(prop[fg], gtb[fg])
    ct = 1 - gt_loc_std[fg].sigmoid()
    cs = 1 - s.sigmoid()
    sel = (ct > cs + ts_better) * (ct > t_cert)
    return (d[sel] - t[sel]).abs().sum() / max(gt_classes.numel(), 1.0)


def roi_label_and_sample(prop_boxes, gt, keys, pseudo, num_classes=80, batch=512, frac=0.25):
    """roi_heads.py:141-270 for one image (+ D2 add_ground_truth_to_proposals, Matcher([0.5],[0,1]),
    _sample_proposals); keys: one per (proposal ++ gt) slot."""
    gb = gt["boxes"]
    pb = torch.cat((prop_boxes, gb))
    has_gt = len(gb) > 0
    iou = pairwise_iou(gb, pb) if has_gt else torch.zeros((0, len(pb)))
    midx, lab = matcher(iou, [0.5], [0, 1], False)
    if has_gt:
        cls = gt["classes"][midx].clone()
        cls[lab == 0] = num_classes
    else:
        cls = torch.zeros_like(midx) + num_classes
    fgi, bgi = subsample_by_keys(cls, keys, batch, frac, num_classes)
    sidx = torch.cat((fgi, bgi))
    out = dict(proposal_boxes=pb[sidx], gt_classes=cls[sidx])
    if has_gt:
        out["gt_boxes"] = gb[midx[sidx]]
        if pseudo:
            out["gt_confid"] = gt["scores"][midx[sidx]]
            if "pred_boxes_std" in gt:            # roi_heads.py:200-203: only when the pseudo labels carry one
                out["gt_loc_std"] = gt["pred_boxes_std"][midx[sidx]]
    else:
        out["gt_boxes"] = gb.new_zeros((len(sidx), 4))
        if pseudo:
            out["gt_confid"] = torch.zeros(len(sidx))
            if "pred_boxes_std" in gt:
                out["gt_loc_std"] = gb.new_zeros((len(sidx), 4))
    return out


def fast_rcnn_inference(boxes, probs, image_size, score_thresh=0.05, nms_thresh=0.5, topk=100):
    """D2 fast_rcnn_inference_single_image [D2-recall] for class-agnostic boxes.  Returns (dets, kept proposal row)."""
    ok = torch.isfinite(boxes).all(dim=1) & torch.isfinite(probs).all(dim=1)
    rows = torch.nonzero(ok).squeeze(1)
    boxes, probs = boxes[ok], probs[ok][:, :-1]
    h, w = image_size
    boxes = torch.stack((boxes[:, 0].clamp(0, w), boxes[:, 1].clamp(0, h), boxes[:, 2].clamp(0, w), boxes[:, 3].clamp(0, h)), dim=1)
    m = probs > score_thresh
    fi = m.nonzero()
    b, s = boxes[fi[:, 0]], probs[m]
    keep = batched_nms(b, s, fi[:, 1], nms_thresh)[:topk]
    return dict(boxes=b[keep], scores=s[keep], classes=fi[keep, 1]), rows[fi[keep, 0]]


def rcnn_backbone(sd, images, mean, pix_std):
    x, sizes = preprocess(images, mean, pix_std, 32)
    c = resnet50(sd, x, "backbone.bottom_up", ("res2", "res3", "res4", "res5"))
    p = fpn(sd, c, ["res2", "res3", "res4", "res5"], "maxpool")
    return p, sizes


UTV1_BOX_WEIGHTS = (10.0, 10.0, 5.0, 5.0)     # MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS (Detectron2 default)


def _predictor_inference(scores, deltas, std, prop, size):
    """one image of the box predictor's inference: boundary-variance predictors (fast_rcnn.py:1094-1125) decode with Box2BoxXYXYTransform,
    class-agnostic, and report pred_boxes_std of the kept rows; the UTv1 predictor (std None; Detectron2 FastRCNNOutputLayers.inference
    [D2-recall]) decodes per-class centre-size deltas"""
    probs = F.softmax(scores, dim=-1)
    if std is None:
        dets, _ = fast_rcnn_inference_per_class(d2_apply_deltas(deltas, prop, UTV1_BOX_WEIGHTS), probs, size)
        return dets
    dets, rows = fast_rcnn_inference(xyxy_apply_deltas(deltas, prop), probs, size)
    dets["pred_boxes_std"] = std[rows]
    return dets


def rcnn_teacher(sd, images, mean, pix_std, pre_topk=2000, post_topk=1000, thr=0.7):
    """meta_arch/rcnn.py:39-55 (branch unsup_data_weak, teacher left in train mode - SURVEY B13) +
    trainer.py:727-751 thresholding."""
    p, sizes = rcnn_backbone(sd, images, mean, pix_std)
    feats = [p[k] for k in ("p2", "p3", "p4", "p5", "p6")]
    hw = [(f.shape[2], f.shape[3]) for f in feats]
    anchors = make_anchors(hw, [4, 8, 16, 32, 64])
    obj, dl = rpn_head(sd, feats)
    props = find_top_rpn_proposals(anchors, obj, dl, sizes, pre_topk, post_topk)
    pooled = roi_pool(feats[:4], [q["boxes"] for q in props])
    scores, deltas, std = box_head(sd, pooled)
    out, r = [], 0
    for n, q in enumerate(props):
        k = len(q["boxes"])
        dets = _predictor_inference(scores[r:r + k], deltas[r:r + k], None if std is None else std[r:r + k], q["boxes"], sizes[n])
        m = dets["scores"] > thr
        out.append({key: v[m] for key, v in dets.items()})     # trainer.py:727-751: pred_boxes_std only when the predictor has one
        r += k
    return out, props


def detector_postprocess(det, size, out_hw):
    """Detectron2 detector_postprocess [D2-recall]: boxes scaled by (out_w / in_w, out_h / in_h), clipped to the output size, empty boxes
    dropped (the reference calls it per image, one_stage_detector.py:16-43,136-145; D2 GeneralizedRCNN._postprocess for the two-stage model)"""
    oh, ow = out_hw
    sx, sy = ow / size[1], oh / size[0]
    b = det["boxes"].clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b[:, 0::2] = b[:, 0::2].clamp(min=0, max=ow)
    b[:, 1::2] = b[:, 1::2].clamp(min=0, max=oh)
    keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    out = {k: (v[keep] if torch.is_tensor(v) and v.shape[:1] == keep.shape else v) for k, v in det.items()}
    out["boxes"] = b[keep]
    return out


def rcnn_inference(sd, images, mean, pix_std, out_sizes=None, pre_topk=1000, post_topk=1000):
    """The eval-mode two-stage detector (meta_arch/rcnn.py:12-13 -> Detectron2 GeneralizedRCNN.inference [D2-recall]): RPN with the *_TEST
    top-k (proposal_generator/rpn.py:21-76, inference branch), box head on the pooled proposals, the predictor's inference
    (roi_heads/roi_heads.py:118-139, roi_heads/fast_rcnn.py:1094-1125: XYXY decode, softmax, score threshold 0.05, class-aware NMS 0.5,
    top 100, pred_boxes_std of the kept rows), detector_postprocess to the requested output sizes."""
    p, sizes = rcnn_backbone(sd, images, mean, pix_std)
    feats = [p[k] for k in ("p2", "p3", "p4", "p5", "p6")]
    hw = [(f.shape[2], f.shape[3]) for f in feats]
    anchors = make_anchors(hw, [4, 8, 16, 32, 64])
    obj, dl = rpn_head(sd, feats)
    props = find_top_rpn_proposals(anchors, obj, dl, sizes, pre_topk, post_topk)
    pooled = roi_pool(feats[:4], [q["boxes"] for q in props])
    scores, deltas, std = box_head(sd, pooled)
    out, r = [], 0
    for n, q in enumerate(props):
        k = len(q["boxes"])
        dets = _predictor_inference(scores[r:r + k], deltas[r:r + k], None if std is None else std[r:r + k], q["boxes"], sizes[n])
        if out_sizes is not None:
            dets = detector_postprocess(dets, sizes[n], out_sizes[n])
        out.append(dets)
        r += k
    return out, props


def rcnn_student_losses(sd, images, gts, rpn_keys, roi_keys, pseudo, mean, pix_std, pre_topk=2000, post_topk=1000, props_override=None):
    """meta_arch/rcnn.py:23-37 / :57-72.  props_override (mixed-precision tests): RPN proposals to use instead of this forward's
    own (a discrete top-k + NMS selection, decoupled from rounding noise the same way pseudo_override decouples the teacher)."""
    p, sizes = rcnn_backbone(sd, images, mean, pix_std)
    feats = [p[k] for k in ("p2", "p3", "p4", "p5", "p6")]
    hw = [(f.shape[2], f.shape[3]) for f in feats]
    anchors = make_anchors(hw, [4, 8, 16, 32, 64])
    obj, dl = rpn_head(sd, feats)
    rl, _ = rpn_losses(torch.cat(anchors), torch.cat(obj, 1), torch.cat(dl, 1), gts, rpn_keys, pseudo)
    with torch.no_grad():
        props = find_top_rpn_proposals(anchors, obj, dl, sizes, pre_topk, post_topk)
    if props_override is not None:
        props = props_override
    # a key entry may be a callable (n_proposals, n_gt) -> keys: the proposal count is only known here
    sampled = [roi_label_and_sample(q["boxes"], g, k(len(q["boxes"]), len(g["boxes"])) if callable(k) else k, pseudo)
               for q, g, k in zip(props, gts, roi_keys)]
    pooled = roi_pool(feats[:4], [s["proposal_boxes"] for s in sampled])
    scores, deltas, std = box_head(sd, pooled)
    cls = torch.cat([s["gt_classes"] for s in sampled])
    pb = torch.cat([s["proposal_boxes"] for s in sampled])
    gb = torch.cat([s["gt_boxes"] for s in sampled])
    if std is None:                               # the UTv1 predictor: confidence-weighted focal + class-specific smooth-L1
        conf = torch.cat([s["gt_confid"].float() for s in sampled]) if pseudo else None
        lc, lb = utv1_roi_losses(scores, deltas, pb, gb, cls, conf, UTV1_BOX_WEIGHTS)
        losses = {"loss_cls": lc, "loss_box_reg": lb}
        losses.update(rl)
        return losses, props, sampled
    losses = {"loss_cls": softmax_focal(scores, cls)}
    if pseudo:
        gstd = torch.cat([s["gt_loc_std"] for s in sampled])
        losses["loss_box_reg"] = roi_box_reg_pseudo_loss(pb, gb, deltas, std, gstd, cls)
    else:
        losses["loss_box_reg"] = roi_box_reg_loss(pb, gb, deltas, std, cls)
    losses.update(rl)
    return losses, props, sampled


def rcnn_semisup_step(student_sd, teacher_sd, batch, keys, keep_rate=0.9996, lam_u=4.0, lam_r=1.0, thr=0.7, lr=0.01,
                      momentum=0.9, wd=1e-4, mean=None, pix_std=None, pre_topk=2000, post_topk=1000,
                      frozen_prefixes=("backbone.bottom_up.stem", "backbone.bottom_up.res2"), pseudo_override=None,
                      props_override=(None, None)):
    """One post-burn-in UBRCNNTeacherTrainer.run_step_full_semisup (engine/trainer.py:814-912).
    keys = dict(rpn_sup [N,R], roi_sup [list], rpn_unsup, roi_unsup): injected sampling keys."""
    mean = mean if mean is not None else torch.tensor([103.53, 116.28, 123.675]).view(3, 1, 1)
    pix_std = pix_std if pix_std is not None else torch.ones(3, 1, 1)
    lq, lk, uq, uk = batch
    teacher_sd = ema_update(student_sd, teacher_sd, keep_rate)
    rec = {"EMA_rate": keep_rate}
    with torch.no_grad():
        pseudo, _ = rcnn_teacher(teacher_sd, [d["image"] for d in uk], mean, pix_std, pre_topk, post_topk, thr)
    if pseudo_override is not None:
        pseudo = pseudo_override
    params = {k: v.clone().requires_grad_(True) for k, v in student_sd.items()
              if v.dtype.is_floating_point and "norm." not in k and not k.startswith(frozen_prefixes)}
    sd = dict(student_sd)
    sd.update(params)
    sup, _, _ = rcnn_student_losses(sd, [d["image"] for d in lq + lk], [d["gt"] for d in lq + lk], keys["rpn_sup"], keys["roi_sup"], False, mean, pix_std, pre_topk, post_topk,
                                     props_override=props_override[0])
    rec.update(sup)
    uns, _, _ = rcnn_student_losses(sd, [d["image"] for d in uq], pseudo, keys["rpn_unsup"], keys["roi_unsup"], True, mean, pix_std, pre_topk, post_topk,
                                     props_override=props_override[1])
    for k, v in uns.items():
        rec[k + "_pseudo"] = v
    total = 0.0
    for k, v in rec.items():
        if k[:4] != "loss":
            continue
        if k == "loss_rpn_loc_pseudo":
            total = total + v * 0
        elif k == "loss_box_reg_pseudo":
            total = total + v * lam_r
        elif k[-6:] == "pseudo":
            total = total + v * lam_u
        else:
            total = total + v
    gl = torch.autograd.grad(total, list(params.values()), allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(params.keys(), gl)}
    newp, bufs = sgd_step({k: v.detach() for k, v in params.items()}, grads, {}, lr, momentum, {k: wd for k in params})
    new_student = OrderedDict(student_sd)
    new_student.update(newp)
    rec = {k: (float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in rec.items()}
    return rec, new_student, teacher_sd, grads, pseudo
