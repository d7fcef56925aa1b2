"""Faster-RCNN UTv2 model on the HIP path: `TwoStagePseudoLabGeneralizedRCNN`, `PseudoLabRPN`,
`StandardROIHeadsPseudoLab` + `FastRCNNFocaltLossBoundaryVarOutputLayers`, `Box2BoxXYXYTransform`.

Mirrors reference ubteacher/modeling/{meta_arch/rcnn.py:6-72, proposal_generator/rpn.py:15-225,
roi_heads/roi_heads.py:23-270, roi_heads/fast_rcnn.py:715-1292,1405-1429, box_regression.py:11-129}
and the Detectron2 pieces they inherit from [D2-recall, SURVEY appendix C], re-laid-out for MI355X:

  * everything is batched over images on padded slots + validity masks (no per-image Python loops,
    no nonzero()/item() host syncs); random subsampling uses per-slot random keys (the k smallest
    keys of a class == randperm[:k] in distribution; tests inject the keys);
  * RPN objectness + anchor deltas are one fused 1x1 conv (16 ch: 3 | 12 | pad); the box predictor's
    cls_score / bbox_pred / bbox_pred_std are one fused 1024->96 GEMM; FC layers run on the same
    MFMA implicit-GEMM kernel as the convs; fc1 consumes the NHWC RoIAlign output directly.
"""
import math
import os

import torch
import torch.nn.functional as F

from .. import hip, ops
from ..d2.registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY
from ..d2.structures import Boxes, Instances
from .arena_model import ArenaModel
from .backbone import _nchw_view, _xavier_init
from .fcos import PaddedBoxes

SCALE_CLAMP = math.log(1000.0 / 16)
RPN_CH = 16   # 3 objectness + 12 deltas + 1 pad
PRED_CH = 96  # 81 cls + 4 deltas + 4 std + 7 pad


# ---------------------------------------------------------------------------------------------------
# box transforms
def rpn_get_deltas(src, tgt, weights=(1.0, 1.0, 1.0, 1.0)):
    """D2 Box2BoxTransform.get_deltas [D2-recall]."""
    sw, sh = src[..., 2] - src[..., 0], src[..., 3] - src[..., 1]
    sx, sy = src[..., 0] + 0.5 * sw, src[..., 1] + 0.5 * sh
    tw, th = tgt[..., 2] - tgt[..., 0], tgt[..., 3] - tgt[..., 1]
    tx, ty = tgt[..., 0] + 0.5 * tw, tgt[..., 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=-1)


def rpn_apply_deltas(deltas, boxes, weights=(1.0, 1.0, 1.0, 1.0)):
    """D2 Box2BoxTransform.apply_deltas [D2-recall]."""
    w, h = boxes[..., 2] - boxes[..., 0], boxes[..., 3] - boxes[..., 1]
    cx, cy = boxes[..., 0] + 0.5 * w, boxes[..., 1] + 0.5 * h
    wx, wy, ww, wh = weights
    dx, dy = deltas[..., 0] / wx, deltas[..., 1] / wy
    dw = torch.clamp(deltas[..., 2] / ww, max=SCALE_CLAMP)
    dh = torch.clamp(deltas[..., 3] / wh, max=SCALE_CLAMP)
    pcx, pcy = dx * w + cx, dy * h + cy
    pw, ph = torch.exp(dw) * w, torch.exp(dh) * h
    return torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=-1)


class Box2BoxXYXYTransform:
    """Per-boundary deltas, reference box_regression.py:11-129 (asymmetric +1, order l,r,d,u - SURVEY B3)."""

    def __init__(self, weights, scale_clamp=1000.0 / 16):
        self.weights = weights
        self.scale_clamp = scale_clamp

    def get_deltas(self, src, tgt):
        sw = src[..., 2] - src[..., 0] + 1.0
        sh = src[..., 3] - src[..., 1] + 1.0
        wx, wy = self.weights[0], self.weights[1]
        return torch.stack((wx * (tgt[..., 0] - src[..., 0]) / sw, wx * (tgt[..., 2] - src[..., 2]) / sw,
                            wy * (tgt[..., 1] - src[..., 1]) / sh, wy * (tgt[..., 3] - src[..., 3]) / sh), dim=-1)

    def apply_deltas(self, deltas, boxes):
        w, h = boxes[..., 2] - boxes[..., 0], boxes[..., 3] - boxes[..., 1]
        wx, wy = self.weights[0], self.weights[1]
        c = self.scale_clamp
        dl = torch.clamp(deltas[..., 0] / wx, max=c, min=-c)
        dr = torch.clamp(deltas[..., 1] / wx, max=c, min=-c)
        dd = torch.clamp(deltas[..., 2] / wy, max=c, min=-c)
        du = torch.clamp(deltas[..., 3] / wy, max=c, min=-c)
        return torch.stack((dl * w + boxes[..., 0], dd * h + boxes[..., 1], dr * w + boxes[..., 2], du * h + boxes[..., 3]), dim=-1)


def float_order_key(x):
    """int64 keys whose descending order == (value desc, flat index asc); NaN-free input assumed."""
    i = x.contiguous().view(torch.int32)
    mono = (i ^ ((i >> 31) & 0x7FFFFFFF)).long()
    n = x.shape[-1]
    idx = torch.arange(n, device=x.device, dtype=torch.int64)
    return mono * 4294967296 + (4294967295 - idx)


def _keys(src, n, m, device):
    """per-slot sampling keys in [0,1): device RNG by default; tests inject a tensor or a callable."""
    if src is None:
        return torch.rand((n, m), device=device)
    k = src(n, m, device) if callable(src) else src
    assert tuple(k.shape) == (n, m), (tuple(k.shape), (n, m))
    return k


def sample_k_smallest(keys, mask, k):
    """indices of the <=k smallest keys among mask (per row), ascending; returns (idx [N,k], valid [N,k])."""
    kk = min(k, keys.shape[1])
    v, idx = torch.topk(torch.where(mask, keys, torch.full_like(keys, 2.0)), kk, dim=1, largest=False, sorted=True)
    return idx, v < 2.0


class _RpnLossFn(torch.autograd.Function):
    """[sum BCE, sum L1] of PseudoLabRPN.losses as one node: forward = utv2_rpn_loss_fwd (which also leaves the per-slot derivatives),
    backward = zero-fill + utv2_rpn_loss_bwd scatter."""

    @staticmethod
    def forward(ctx, obj, deltas, rpn, anchors, s, gt, head_hw, N, batch=None, img0=0):
        R = anchors.shape[0]
        hw = [h * w for (h, w) in head_hw] if head_hw is not None else None
        sums, gobj, gdl = hip.rpn_loss_fwd(obj.detach(), deltas.detach(), hw, N, rpn.A, R, anchors, s, gt["boxes"],
                                           gt["scores"] if "scores" in gt else None, rpn.box_weights, batch=batch, img0=img0)
        ctx.meta = (hw, N, rpn.A, R, s, tuple(obj.shape), tuple(deltas.shape), head_hw is not None, batch, img0)
        ctx.save_for_backward(gobj, gdl)
        return sums

    @staticmethod
    def backward(ctx, g):
        gobj, gdl = ctx.saved_tensors
        hw, N, A, R, s, oshape, dshape, head, batch, img0 = ctx.meta
        g = g.contiguous()
        grad_obj = torch.zeros(oshape, dtype=torch.float32, device=g.device)
        grad_dl = grad_obj if head else torch.zeros(dshape, dtype=torch.float32, device=g.device)
        hip.rpn_loss_bwd(gobj, gdl, g[0:1], g[1:2], hw, N, A, oshape[-1] if head else 0, R, s, grad_obj, grad_dl, batch=batch, img0=img0)
        return grad_obj, (None if head else grad_dl), None, None, None, None, None, None, None, None


class _RoiBoxLossFn(torch.autograd.Function):
    """box_reg_loss / box_reg_pseudo_loss (fast_rcnn.py:938-1090) as one node: utv2_roi_box_loss computes the sum and its derivatives
    w.r.t. the deltas and std logits in the forward launch; the backward scales them."""

    @staticmethod
    def forward(ctx, deltas, std, cls, prop, gtb, gstd, pred, mode):
        t = pred.box2box_transform
        out, gd, gs = hip.roi_box_loss(deltas.detach(), std.detach(), cls, prop, gtb, gstd, pred.num_classes, mode, t.weights[0], t.weights[1],
                                       t.scale_clamp, pred.ts_better, pred.t_cert)
        ctx.save_for_backward(gd, gs)
        return out

    @staticmethod
    def backward(ctx, g):
        gd, gs = ctx.saved_tensors
        return gd * g, gs * g, None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------------
class AnchorGenerator:
    """D2 DefaultAnchorGenerator [D2-recall]: sizes per level x aspect ratios, offset 0, order (H, W, A)."""

    def __init__(self, cfg, strides):
        self.sizes = [list(s) for s in cfg.MODEL.ANCHOR_GENERATOR.SIZES]
        self.ratios = list(cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS[0])
        self.strides = strides
        self.A = len(self.sizes[0]) * len(self.ratios)
        self._cache = {}

    def cell(self, sizes):
        out = []
        for s in sizes:
            area = s ** 2.0
            for r in self.ratios:
                w = math.sqrt(area / r)
                h = r * w
                out.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
        return torch.tensor(out, dtype=torch.float32)

    def __call__(self, level_hw, device):
        key = (tuple(level_hw), str(device))
        if key not in self._cache:
            per = []
            for (h, w), s, sz in zip(level_hw, self.strides, self.sizes):
                sx = torch.arange(0, w * s, step=s, dtype=torch.float32)
                sy = torch.arange(0, h * s, step=s, dtype=torch.float32)
                yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
                per.append((shifts.view(-1, 1, 4) + self.cell(sz).view(1, -1, 4)).reshape(-1, 4))
            self._cache[key] = [p.to(device) for p in per]
        return self._cache[key]


@PROPOSAL_GENERATOR_REGISTRY.register()
class PseudoLabRPN:
    def __init__(self, cfg, store, in_channels, prefix="proposal_generator"):
        r = cfg.MODEL.RPN
        self.in_features = list(r.IN_FEATURES)
        self.strides = [2 ** int(f[1:]) for f in self.in_features]
        self.anchor_generator = AnchorGenerator(cfg, self.strides)
        A = self.anchor_generator.A
        assert A == 3
        self.A = A
        C = in_channels
        w = store.new((C, 9 * C), "decay", lambda t: t.normal_(0.0, 0.01)).export(prefix + ".rpn_head.conv.weight", _nchw_view(C, C, 3))
        b = store.new((C,), "decay", lambda t: t.zero_()).export(prefix + ".rpn_head.conv.bias")
        self.conv = ops.Conv(w, C, C, 3, 1, 1, bias=b, relu=True)

        def init_pred(t):
            t.zero_()
            t[:A + 4 * A].normal_(0.0, 0.01)
        wp = store.new((RPN_CH, C), "decay", init_pred)
        wp.export(prefix + ".rpn_head.objectness_logits.weight", lambda t: t[0:A].view(A, 1, 1, C).permute(0, 3, 1, 2))
        wp.export(prefix + ".rpn_head.anchor_deltas.weight", lambda t: t[A:5 * A].view(4 * A, 1, 1, C).permute(0, 3, 1, 2))
        bp = store.new((RPN_CH,), "decay", lambda t: t.zero_())
        bp.export(prefix + ".rpn_head.objectness_logits.bias", lambda t: t[0:A])
        bp.export(prefix + ".rpn_head.anchor_deltas.bias", lambda t: t[A:5 * A])
        self.pred = ops.Conv(wp, C, RPN_CH, 1, 1, 0, bias=bp, out_fp32=True)
        # the 3x3 conv's ReLU output has one consumer, the prediction conv: its dgrad epilogue applies the mask (ops.premask_on) instead of
        # a pass over the [all levels x images, 256] gradient at the top of the 3x3 conv's backward (450 us per step)
        self.pred.premask_input = True
        self.conv.grad_premasked = True
        self.batch_size_per_image = r.BATCH_SIZE_PER_IMAGE
        self.positive_fraction = r.POSITIVE_FRACTION
        self.iou_thresholds = list(r.IOU_THRESHOLDS)
        self.pre_nms_topk = {True: r.PRE_NMS_TOPK_TRAIN, False: r.PRE_NMS_TOPK_TEST}
        self.post_nms_topk = {True: r.POST_NMS_TOPK_TRAIN, False: r.POST_NMS_TOPK_TEST}
        self.nms_thresh = r.NMS_THRESH
        self.min_box_size = cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE
        self.loss_weight = {"loss_rpn_cls": r.LOSS_WEIGHT, "loss_rpn_loc": r.BBOX_REG_LOSS_WEIGHT * r.LOSS_WEIGHT}
        self.box_weights = tuple(r.BBOX_REG_WEIGHTS)
        assert r.BOUNDARY_THRESH < 0 and r.BBOX_REG_LOSS_TYPE == "smooth_l1" and r.SMOOTH_L1_BETA == 0.0
        self.training = True
        self.sample_keys = None  # tests inject [N, R] keys in [0,1)

    def train(self, mode=True):
        self.training = mode

    # -- head: one launch per layer for all levels, output level-first [P,16] ----------------------------------
    def _head(self, features):
        feats = [features[f] for f in self.in_features]
        hw = [(f.shape[1], f.shape[2]) for f in feats]
        lf = features.get("_levelfirst")
        if lf is not None and lf[1].level_hw == hw:
            big, meta = lf
        else:
            meta = ops.LevelMeta(feats[0].shape[0], hw)
            big = torch.cat([f.reshape(-1, f.shape[-1]) for f in feats], dim=0)
        out = self.pred(self.conv(big, meta=meta), meta=meta)
        return out, hw, meta.N

    def _per_image_views(self, big, N, hw):
        """objectness [N, R] and deltas [N, R, 4] in the reference's (level, h, w, a) anchor order."""
        A = self.A
        obj, dl = [], []
        r = 0
        for (h, w) in hw:
            blk = big[r:r + N * h * w].view(N, h * w, RPN_CH)
            obj.append(blk[:, :, :A].reshape(N, -1))
            dl.append(blk[:, :, A:5 * A].reshape(N, h * w * A, 4))
            r += N * h * w
        return obj, dl

    def forward(self, image_sizes, features, gt=None, compute_loss=True, compute_val_loss=False):
        big, hw, N = self._head(features)
        anchors = self.anchor_generator(hw, big.device)
        losses = {}
        if (self.training and compute_loss) or compute_val_loss:
            losses = self.losses(self._anchors_cat(anchors, hw, big.device), big, None, gt, head_hw=hw)
            losses = {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}  # applied twice
        with torch.no_grad():
            sel = self._pre_nms_topk(big.detach(), N, hw)
            if sel is not None:
                proposals = self._proposals_fused(big.detach(), anchors, sel, hw, N, image_sizes)
            else:
                obj, dl = self._per_image_views(big.detach(), N, hw)
                proposals = self.predict_proposals(anchors, obj, dl, image_sizes)
        return proposals, losses

    def forward_joint_begin(self, image_sizes, features, n_labeled, gt_labeled, raw=False):
        """The labeled and the pseudo-labeled images of one iteration as ONE batch (images [0, n_labeled) carry ground truth): head,
        proposals of every image, and the labeled images' losses - everything that does not need the pseudo labels.
        raw: the losses as raw kernel sums for the fused scalar tail (TwoStagePseudoLabGeneralizedRCNN.forward_joint_finish)."""
        big, hw, N = self._head(features)
        anchors = self.anchor_generator(hw, big.device)
        acat = self._anchors_cat(anchors, hw, big.device)
        losses = self.losses(acat, big, None, gt_labeled, head_hw=hw, batch=N, img0=0, raw=raw)
        if not raw:
            losses = {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}  # applied twice, as in forward()
        sample_l = self._last_sample
        with torch.no_grad():
            sel = self._pre_nms_topk(big.detach(), N, hw)
            if sel is not None:
                proposals = self._proposals_fused(big.detach(), anchors, sel, hw, N, image_sizes)
            else:
                obj, dl = self._per_image_views(big.detach(), N, hw)
                proposals = self.predict_proposals(anchors, obj, dl, image_sizes)
        return dict(big=big, hw=hw, N=N, acat=acat, n_labeled=n_labeled, sample_l=sample_l), proposals, losses

    def forward_joint_finish(self, ctx, gt_unlabeled, raw=False):
        """the pseudo-labeled images' losses of a forward_joint_begin batch"""
        losses = self.losses(ctx["acat"], ctx["big"], None, gt_unlabeled, head_hw=ctx["hw"], batch=ctx["N"], img0=ctx["n_labeled"], raw=raw)
        if raw:
            return losses
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in losses.items()}

    def _anchors_cat(self, anchors, hw, dev):
        cache = self.__dict__.setdefault("_anchor_cat_cache", {})
        ck = (tuple(hw), str(dev))
        a = cache.get(ck)
        if a is None:
            if len(cache) >= 16:
                cache.clear()
            a = cache[ck] = torch.cat(anchors).contiguous()
        return a

    @torch.no_grad()
    def _pre_nms_topk(self, big, N, hw):
        """D2 find_top_rpn_proposals' per-(image, level) `topk(pre_nms_topk)` for all levels and images by ONE exact radix select
        (utv2_topk_rows_i64; torch.topk per level costs ~70 launches per forward): utv2_rpn_rank_keys turns the objectness of the
        level-first head output into one flat buffer of sortable keys whose ragged rows are the (level, image) pairs.  Returns
        (selected keys [L*N, max k], k per level) - row l*N+n in (score desc, anchor index asc) order, the order torch.topk on
        float_order_key gives - or None when PRE_NMS_TOPK exceeds the kernel's 2048."""
        pre = self.pre_nms_topk[self.training]
        A = self.A
        widths = [h * w * A for (h, w) in hw]
        ks = [min(pre, wd) for wd in widths]
        if max(ks) > 2048:
            return None
        dev = big.device
        ck = (N, tuple(hw), str(dev))
        geom = self.__dict__.setdefault("_topk_geom", {})   # one entry per batch geometry (teacher / labeled / unlabeled batches differ)
        row_off = geom.get(ck)
        if row_off is None:
            offs, o = [], 0
            for wd in widths:
                offs.extend(o + n * wd for n in range(N))
                o += N * wd
            offs.append(o)
            if len(geom) >= 16:
                geom.clear()
            row_off = geom[ck] = torch.tensor(offs, dtype=torch.int64, device=dev)
        keys = hip.rpn_rank_keys(big, [h * w for (h, w) in hw], N, A)
        return hip.topk_rows(keys, row_off, len(hw) * N, max(widths), max(ks)), ks

    @torch.no_grad()
    def _proposals_fused(self, big, anchors, sel, hw, N, image_sizes):
        """predict_proposals with the gather / apply_deltas / clip / keep chain of all levels in one launch (utv2_rpn_decode)"""
        top, ks = sel
        dev = big.device
        post = self.post_nms_topk[self.training]
        cache = self.__dict__.setdefault("_decode_cache", {})
        ck = (tuple(hw), tuple(image_sizes), str(dev))
        ent = cache.get(ck)
        if ent is None:
            if len(cache) >= 16:
                cache.clear()
            ent = cache[ck] = (torch.cat(anchors).contiguous(),
                               torch.tensor([[s[0], s[1]] for s in image_sizes], dtype=torch.float32, device=dev))
        anchors_cat, image_hw = ent
        boxes, scores, lvls, keep = hip.rpn_decode(top, big, anchors_cat, image_hw, [h * w for (h, w) in hw], ks, N, self.A,
                                                   self.box_weights, SCALE_CLAMP, self.min_box_size)
        return self._nms_and_pack(boxes, scores, lvls, keep, image_sizes, post)

    def _nms_and_pack(self, boxes, scores, lvls, keep, image_sizes, post):
        dev = boxes.device
        kidx, cnt = hip.nms_batched(boxes, scores, lvls, keep, self.nms_thresh, class_aware=True, post_topk=-1, max_out=post)
        if boxes.is_contiguous() and scores.is_contiguous() and boxes.dtype == scores.dtype == torch.float32:
            ob, osc, ov = hip.nms_pack(kidx, cnt, boxes, scores)        # one launch instead of clamp / arange / lt / 2 gathers + casts
            return PaddedBoxes(image_sizes, boxes=ob, objectness_logits=osc, valid=ov, count=cnt)
        ix = kidx.clamp(min=0).long()
        valid = (torch.arange(post, device=dev)[None, :] < cnt[:, None]).to(torch.uint8)
        return PaddedBoxes(image_sizes, boxes=torch.gather(boxes, 1, ix[:, :, None].expand(-1, -1, 4)).contiguous(),
                           objectness_logits=torch.gather(scores, 1, ix).contiguous(), valid=valid, count=cnt)

    __call__ = forward

    # -- labels + sampling (rpn.py:78-150 and D2 label_and_sample_anchors) -----------------------------------
    @torch.no_grad()
    def label_and_sample(self, anchors, gt):
        N = gt.n
        R = anchors.shape[0]
        mx, arg, gmax = hip.match_boxes(anchors, gt["boxes"], gt["valid"], want_gt_max=True)
        lowq = hip.match_lowq(anchors, gt["boxes"], gt["valid"], gmax)
        lo, hi = self.iou_thresholds
        keys = _keys(self.sample_keys, N, R, anchors.device)
        npos_max = int(self.batch_size_per_image * self.positive_fraction)
        if max(npos_max, self.batch_size_per_image) <= 2048 and os.environ.get("UTV2_FUSED_SAMPLERS", "1") != "0":
            # labels (IoU thresholds, allow_low_quality_matches, images without gt) and both "k smallest keys" draws in three launches
            pidx, pval, nidx, nval, has_gt = hip.rpn_sample(mx, lowq, gt["valid"], keys, lo, hi, npos_max, self.batch_size_per_image)
            return dict(pos_idx=pidx, pos_valid=pval, neg_idx=nidx, neg_valid=nval, matched32=arg, has_gt=has_gt)
        has_gt = gt["valid"].bool().any(dim=1, keepdim=True)
        one = torch.ones((), dtype=torch.int8, device=anchors.device)
        labels = torch.where(mx < lo, one * 0, one * -1)
        labels = torch.where(mx >= hi, one, labels)
        labels = torch.where(lowq.bool(), one, labels)   # allow_low_quality_matches
        labels = torch.where(has_gt, labels, torch.zeros_like(labels))
        pidx, pval = sample_k_smallest(keys, labels == 1, npos_max)
        npos = pval.sum(1, keepdim=True)
        nidx, nval = sample_k_smallest(keys, labels == 0, self.batch_size_per_image)
        nval = nval & (torch.arange(nidx.shape[1], device=anchors.device)[None, :] < (self.batch_size_per_image - npos))
        return dict(pos_idx=pidx, pos_valid=pval, neg_idx=nidx, neg_valid=nval, matched32=arg, has_gt=has_gt)

    def losses(self, anchors, obj, deltas, gt, head_hw=None, batch=None, img0=0, raw=False):
        """rpn.py:153-225: BCE(sum) over sampled anchors (optionally weighted by the matched pseudo-box score,
        negatives too - SURVEY B4) + L1 on positives, both / (batch_size_per_image * N), weights applied here too.
        One fused forward launch and one backward launch (utv2_rpn_loss_fwd / _bwd).  obj [N,R] + deltas [N,R,4], or - head_hw given -
        obj = the level-first head output [P, RPN_CH] (deltas ignored): no per-image copies of the logits / deltas are made.
        batch / img0 (head form): the head output covers `batch` images and gt's are images [img0, img0 + gt.n) of it."""
        N = gt.n
        s = self.label_and_sample(anchors, gt)
        sums = _RpnLossFn.apply(obj, obj if head_hw is not None else deltas, self, anchors, s, gt, head_hw, N, batch, img0)
        norm = self.batch_size_per_image * N
        self._last_sample = s
        if raw:   # the fused scalar tail (ops.rcnn_loss_combine) divides, weights (twice: SURVEY B2) and sums
            return {"raw_sums": sums, "norm": float(norm)}
        out = {"loss_rpn_cls": sums[0] / norm, "loss_rpn_loc": sums[1] / norm}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in out.items()}

    # -- proposals (D2 find_top_rpn_proposals) -----------------------------------------------------------
    @torch.no_grad()
    def predict_proposals(self, anchors, obj, deltas, image_sizes, top_idx=None):
        N = obj[0].shape[0]
        dev = obj[0].device
        pre, post = self.pre_nms_topk[self.training], self.post_nms_topk[self.training]
        boxes, scores, lvls = [], [], []
        for l, (a, o, d) in enumerate(zip(anchors, obj, deltas)):
            k = min(pre, o.shape[1])
            if top_idx is not None:
                idx = top_idx[l]
            else:
                top = torch.topk(float_order_key(o), k, dim=1, sorted=True).values
                idx = 4294967295 - (top & 4294967295)
            sc = torch.gather(o, 1, idx)
            bx = rpn_apply_deltas(torch.gather(d, 1, idx[:, :, None].expand(-1, -1, 4)), a[idx], self.box_weights)
            boxes.append(bx); scores.append(sc)
            lvls.append(torch.full((N, k), l, dtype=torch.int32, device=dev))
        boxes, scores, lvls = torch.cat(boxes, 1), torch.cat(scores, 1), torch.cat(lvls, 1)
        hwt = torch.tensor([[s[1], s[0], s[1], s[0]] for s in image_sizes], dtype=torch.float32, device=dev)[:, None, :]
        finite = torch.isfinite(boxes).all(dim=2) & torch.isfinite(scores)
        boxes = torch.minimum(boxes.clamp(min=0), hwt)
        keep = finite & ((boxes[..., 2] - boxes[..., 0]) > self.min_box_size) & ((boxes[..., 3] - boxes[..., 1]) > self.min_box_size)
        boxes = boxes.contiguous()
        scores = torch.where(finite, scores, torch.zeros_like(scores)).contiguous()
        return self._nms_and_pack(boxes, scores, lvls.contiguous(), keep.to(torch.uint8).contiguous(), image_sizes, post)


# ---------------------------------------------------------------------------------------------------
class FastRCNNFocaltLossBoundaryVarOutputLayers:
    """Predictor + losses + inference (reference fast_rcnn.py:715-1292)."""
    focal_gamma = 1.5  # comput_focal_loss, fast_rcnn.py:925-936

    def __init__(self, cfg, store, in_dim, prefix):
        rh, bh = cfg.MODEL.ROI_HEADS, cfg.MODEL.ROI_BOX_HEAD
        assert bh.CLS_AGNOSTIC_BBOX_REG, "shipped UTv2 configs are class-agnostic (…sup1_run0.yaml:18)"
        self.num_classes = rh.NUM_CLASSES
        K = self.num_classes
        self.K = K

        def init_pred(t):
            t.zero_()
            t[:K + 1].normal_(0.0, 0.01)
            t[K + 1:K + 5].normal_(0.0, 0.001)
            t[K + 5:K + 9].normal_(0.0, 0.0001)
        w = store.new((PRED_CH, in_dim), "decay", init_pred)
        w.export(prefix + ".cls_score.weight", lambda t: t[0:K + 1])
        w.export(prefix + ".bbox_pred.weight", lambda t: t[K + 1:K + 5])
        w.export(prefix + ".bbox_pred_std.weight", lambda t: t[K + 5:K + 9])
        b = store.new((PRED_CH,), "decay", lambda t: t.zero_())
        b.export(prefix + ".cls_score.bias", lambda t: t[0:K + 1])
        b.export(prefix + ".bbox_pred.bias", lambda t: t[K + 1:K + 5])
        b.export(prefix + ".bbox_pred_std.bias", lambda t: t[K + 5:K + 9])
        self.linear = ops.Conv(w, in_dim, PRED_CH, 1, 1, 0, bias=b, out_fp32=True)
        self.box2box_transform = Box2BoxXYXYTransform(tuple(bh.BBOX_REG_WEIGHTS))
        self.smooth_l1_beta = bh.SMOOTH_L1_BETA
        self.test_score_thresh = rh.SCORE_THRESH_TEST
        self.test_nms_thresh = rh.NMS_THRESH_TEST
        self.test_topk_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.box_reg_loss_type = bh.BBOX_REG_LOSS_TYPE
        self.box_pseudo_reg_loss_type = bh.BBOX_PSEUDO_REG_LOSS_TYPE
        self.loss_weight = {"loss_box_reg": bh.BBOX_REG_LOSS_WEIGHT}
        self.ts_better = cfg.SEMISUPNET.TS_BETTER
        self.t_cert = cfg.SEMISUPNET.T_CERT
        if self.box_reg_loss_type not in ("nlloss", "smooth_l1"):
            raise ValueError("Invalid bbox reg loss type '{}'".format(self.box_reg_loss_type))
        if self.box_pseudo_reg_loss_type not in ("tsbetter", "smooth_l1"):
            raise ValueError("Invalid bbox pseudo reg loss type '{}'".format(self.box_pseudo_reg_loss_type))

    def __call__(self, x2d):
        y = self.linear(x2d.view(x2d.shape[0], 1, 1, -1)).view(x2d.shape[0], PRED_CH)
        K = self.K
        return y[:, :K + 1], y[:, K + 1:K + 5], y[:, K + 5:K + 9]

    def losses(self, predictions, sampled, branch, raw=False):
        scores, deltas, std = predictions
        cls = sampled["gt_classes"].reshape(-1)           # -1 = empty slot
        tgt = cls.to(torch.int32).contiguous()
        focal = ops.softmax_focal_sum(scores, tgt, self.focal_gamma)
        pseudo = branch == "unsup_data_train"
        mode = (2 if self.box_pseudo_reg_loss_type == "tsbetter" else 3) if pseudo else (0 if self.box_reg_loss_type == "nlloss" else 1)
        gstd = sampled["gt_loc_std"].reshape(-1, 4).contiguous() if (mode == 2 and "gt_loc_std" in sampled) else None
        box = _RoiBoxLossFn.apply(deltas, std, cls.long().contiguous(), sampled["proposal_boxes"].reshape(-1, 4).contiguous(),
                                  sampled["gt_boxes"].reshape(-1, 4).contiguous(), gstd, self, mode)
        if raw:   # the fused scalar tail counts the sampled ROIs (targets >= 0), divides, weights and sums
            return {"focal": focal, "box": box, "tgt": tgt}
        Rn = (cls >= 0).sum().clamp(min=1).float()        # gt_classes.numel() of the reference
        loss_cls = focal[0] / Rn
        box = box[0]
        out = {"loss_cls": loss_cls, "loss_box_reg": box / Rn}
        return {k: v * self.loss_weight.get(k, 1.0) for k, v in out.items()}

    @torch.no_grad()
    def inference(self, predictions, proposals, max_cand=8192):
        """fast_rcnn.py:1094-1125 + D2 fast_rcnn_inference: softmax, score > thr, class-aware NMS, top-k."""
        scores, deltas, std = predictions
        N, P = proposals["valid"].shape
        K = self.K
        pb = proposals["boxes"]
        cache = self.__dict__.setdefault("_hwt_cache", {})   # a fresh torch.tensor(..., device=cuda) is a synchronizing pageable copy
        ck = (tuple(proposals.image_sizes), str(pb.device))
        hwt = cache.get(ck)
        if hwt is None:
            if len(cache) >= 16:
                cache.clear()
            hwt = cache[ck] = torch.tensor([[s[1], s[0], s[1], s[0]] for s in proposals.image_sizes], dtype=torch.float32, device=pb.device)[:, None, :]
        if os.environ.get("UTV2_FUSED_ROI_INFERENCE", "1") != "0" and scores.dtype == deltas.dtype == std.dtype == torch.float32 and P * K < (1 << 32):
            # round 4: decode + clip + candidate keys, the gather behind the top-k and the packing of the NMS survivors as three launches
            # (utv2_roi_infer_*) around softmax / topk / utv2_nms_batched - the chain below is ~55 ATen launches per teacher pass.
            # Same arithmetic per element and the same (probability desc, flat index asc) order: identical detections.
            pr = F.softmax(scores, dim=-1).contiguous()
            wx, wy = self.box2box_transform.weights[0], self.box2box_transform.weights[1]
            dec, keys = hip.roi_infer_keys(pr, deltas.contiguous(), pb.contiguous(), proposals["valid"].contiguous(), hwt.view(N, 4), K, wx, wy,
                                           self.box2box_transform.scale_clamp, self.test_score_thresh)
            # the (probability desc, flat index asc) top candidates of every image: the exact radix select of the FCOS / RPN top-k
            # (utv2_topk_rows_i64, 9 launches) - torch.topk on an [N, P*K = 80 000] matrix with k = 8192 is ~40 launches (multi-block radix
            # select + segmented sort); same keys, same order (distinct keys: the selected set and its order are unique)
            kk = min(max_cand, P * K)
            if os.environ.get("UTV2_ROI_TOPK", "1") != "0" and kk <= 8192:
                ro = cache.get(("row_off", N, P * K, str(pb.device)))
                if ro is None:
                    ro = cache[("row_off", N, P * K, str(pb.device))] = (torch.arange(N + 1, dtype=torch.int64) * (P * K)).to(pb.device)
                top = hip.topk_rows(keys.view(-1), ro, N, P * K, kk)
            else:
                top = torch.topk(keys, kk, dim=1, sorted=True).values
            sc, r, c, cb, valid = hip.roi_infer_gather(top, dec, K, self.test_score_thresh)
            D = self.test_topk_per_image
            kidx, cnt = hip.nms_batched(cb, sc, c, valid, self.test_nms_thresh, class_aware=True, post_topk=-1, max_out=D)
            ob, osc, oc, ostd, keep_rows, ov = hip.roi_infer_pack(kidx, cnt, cb, sc, c, r, std.contiguous(), P, D)
            return PaddedBoxes(proposals.image_sizes, boxes=ob, scores=osc, classes=oc, pred_boxes_std=ostd, valid=ov, count=cnt), keep_rows
        boxes = self.box2box_transform.apply_deltas(deltas.view(N, P, 4), pb)
        probs = F.softmax(scores, dim=-1).view(N, P, K + 1)[:, :, :K]
        ok = proposals["valid"].bool() & torch.isfinite(boxes).all(dim=2) & torch.isfinite(probs).all(dim=2)
        boxes = torch.minimum(boxes.clamp(min=0), hwt)
        cand = (probs > self.test_score_thresh) & ok[:, :, None]
        flat = torch.where(cand, probs, torch.full_like(probs, -1.0)).reshape(N, P * K)
        k = min(max_cand, P * K)
        top = torch.topk(float_order_key(flat), k, dim=1, sorted=True).values
        idx = 4294967295 - (top & 4294967295)
        sc = torch.gather(flat, 1, idx)
        r, c = idx // K, (idx % K).to(torch.int32)
        cb = torch.gather(boxes, 1, r[:, :, None].expand(-1, -1, 4)).contiguous()
        valid = (sc > self.test_score_thresh).to(torch.uint8)
        D = self.test_topk_per_image
        kidx, cnt = hip.nms_batched(cb, sc.contiguous(), c.contiguous(), valid.contiguous(), self.test_nms_thresh,
                                    class_aware=True, post_topk=-1, max_out=D)
        ix = kidx.clamp(min=0).long()
        keep_rows = torch.gather(r, 1, ix)
        out = PaddedBoxes(proposals.image_sizes,
                          boxes=torch.gather(cb, 1, ix[:, :, None].expand(-1, -1, 4)).contiguous(),
                          scores=torch.gather(sc, 1, ix).contiguous(),
                          classes=torch.gather(c, 1, ix).contiguous(),
                          pred_boxes_std=torch.gather(std.view(N, P, 4), 1, keep_rows[:, :, None].expand(-1, -1, 4)).contiguous(),
                          valid=(torch.arange(D, device=pb.device)[None, :] < cnt[:, None]).to(torch.uint8), count=cnt)
        return out, keep_rows


class FastRCNNFocaltLossOutputLayers:
    """MODEL.ROI_HEADS.LOSS "FocalLoss" - the Unbiased-Teacher-v1 predictor the reference still ships (roi_heads/fast_rcnn.py:1296-1402
    `FastRCNNFocaltLossOutputLayers` / `FastRCNNFocalLoss` on Detectron2's `FastRCNNOutputLayers` [D2-recall]): cls_score (K + 1) and
    bbox_pred (4 per class, or 4 with CLS_AGNOSTIC_BBOX_REG) Linear heads, no boundary-variance head; Detectron2's centre-size
    `Box2BoxTransform(BBOX_REG_WEIGHTS)`; loss_cls = sum of (1 - p)^1.5 * CE, times the matched pseudo box's score when the sampled
    proposals carry `gt_confid` (fast_rcnn.py:1379-1429), / R; loss_box_reg = smooth-L1 (SMOOTH_L1_BETA) between the foreground
    rows' deltas of their gt class and `get_deltas(proposal, gt box)` - or, BBOX_REG_LOSS_TYPE "giou", the GIoU loss of the decoded
    boxes -, summed, / R (fast_rcnn.py:134-194).  No shipped UTv2 YAML selects
    it: the small losses run as ATen ops on the device (autograd differentiates them); the heads are one fused GEMM on the same
    kernels as every other Linear.  Inference = Detectron2's `fast_rcnn_inference` with per-class boxes."""
    focal_gamma = 1.5

    def __init__(self, cfg, store, in_dim, prefix):
        rh, bh = cfg.MODEL.ROI_HEADS, cfg.MODEL.ROI_BOX_HEAD
        K = self.K = self.num_classes = rh.NUM_CLASSES
        self.nbox = 1 if bh.CLS_AGNOSTIC_BBOX_REG else K
        nb = 4 * self.nbox
        self.ch = (K + 1 + nb + 7) // 8 * 8

        def init_pred(t):
            t.zero_()
            t[:K + 1].normal_(0.0, 0.01)
            t[K + 1:K + 1 + nb].normal_(0.0, 0.001)
        w = store.new((self.ch, in_dim), "decay", init_pred)
        w.export(prefix + ".cls_score.weight", lambda t: t[0:K + 1])
        w.export(prefix + ".bbox_pred.weight", lambda t: t[K + 1:K + 1 + nb])
        b = store.new((self.ch,), "decay", lambda t: t.zero_())
        b.export(prefix + ".cls_score.bias", lambda t: t[0:K + 1])
        b.export(prefix + ".bbox_pred.bias", lambda t: t[K + 1:K + 1 + nb])
        self.linear = ops.Conv(w, in_dim, self.ch, 1, 1, 0, bias=b, out_fp32=True)
        self.box_weights = tuple(bh.BBOX_REG_WEIGHTS)
        self.smooth_l1_beta = bh.SMOOTH_L1_BETA
        self.test_score_thresh = rh.SCORE_THRESH_TEST
        self.test_nms_thresh = rh.NMS_THRESH_TEST
        self.test_topk_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.box_reg_loss_type = bh.BBOX_REG_LOSS_TYPE
        self.loss_weight = {}                                    # FastRCNNFocalLoss.losses applies none (fast_rcnn.py:1379-1402)
        if self.box_reg_loss_type not in ("smooth_l1", "giou"):  # fast_rcnn.py:163-186 raises the same at the first loss; the UTv2 YAMLs set "nlloss"
            raise ValueError("Invalid bbox reg loss type '{}'".format(self.box_reg_loss_type))

    def __call__(self, x2d):
        y = self.linear(x2d.view(x2d.shape[0], 1, 1, -1)).view(x2d.shape[0], self.ch)
        K = self.K
        return y[:, :K + 1], y[:, K + 1:K + 1 + 4 * self.nbox], None

    @staticmethod
    def _giou_loss(b1, b2, eps=1e-7):
        """fvcore.nn.giou_loss, reduction none [D2-recall]"""
        x1, y1, x2, y2 = b1.unbind(dim=-1)
        x1g, y1g, x2g, y2g = b2.unbind(dim=-1)
        xk1, yk1, xk2, yk2 = torch.max(x1, x1g), torch.max(y1, y1g), torch.min(x2, x2g), torch.min(y2, y2g)
        inter = torch.where((yk2 > yk1) & (xk2 > xk1), (xk2 - xk1) * (yk2 - yk1), torch.zeros_like(x1))
        union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
        area_c = (torch.max(x2, x2g) - torch.min(x1, x1g)) * (torch.max(y2, y2g) - torch.min(y1, y1g))
        return 1 - (inter / (union + eps) - (area_c - union) / (area_c + eps))

    def losses(self, predictions, sampled, branch, raw=False):
        scores, deltas = predictions[0], predictions[1]
        K = self.K
        cls = sampled["gt_classes"].reshape(-1).long()          # -1 = empty slot
        ce = F.cross_entropy(scores.float(), cls, ignore_index=-1, reduction="none")
        loss = (1.0 - torch.exp(-ce)) ** self.focal_gamma * ce
        if "gt_confid" in sampled:                               # fast_rcnn.py:1424-1427: the pseudo-labeled branch
            loss = loss * sampled["gt_confid"].reshape(-1)
        focal = loss.sum().reshape(1)
        fg = (cls >= 0) & (cls < K)
        prop, gtb = sampled["proposal_boxes"].reshape(-1, 4), sampled["gt_boxes"].reshape(-1, 4)
        zero = torch.zeros((), device=prop.device)
        if self.nbox == 1:
            pred = deltas
        else:
            col = (4 * cls.clamp(min=0, max=K - 1))[:, None] + torch.arange(4, device=cls.device)[None, :]
            pred = torch.gather(deltas, 1, col)
        # background / empty rows carry zero gt boxes (log(0), 0/0): neutral operands there keep the backward finite
        if self.box_reg_loss_type == "smooth_l1":
            tgt = torch.where(fg[:, None], rpn_get_deltas(prop, gtb, self.box_weights), zero)
            diff = (pred.float() - tgt).abs()
            if self.smooth_l1_beta >= 1e-5:                       # fvcore smooth_l1_loss
                diff = torch.where(diff < self.smooth_l1_beta, 0.5 * diff * diff / self.smooth_l1_beta, diff - 0.5 * self.smooth_l1_beta)
            box = torch.where(fg[:, None], diff, zero).sum().reshape(1)
        else:                                                     # "giou" (fast_rcnn.py:174-183)
            bx = rpn_apply_deltas(torch.where(fg[:, None], pred.float(), zero), prop, self.box_weights)
            box = torch.where(fg, self._giou_loss(bx, torch.where(fg[:, None], gtb, prop)), zero).sum().reshape(1)
        if raw:
            return {"focal": focal, "box": box, "tgt": cls.to(torch.int32).contiguous()}
        Rn = (cls >= 0).sum().clamp(min=1).float()
        return {"loss_cls": focal[0] / Rn, "loss_box_reg": box[0] / Rn}

    @torch.no_grad()
    def inference(self, predictions, proposals, max_cand=8192):
        """Detectron2 FastRCNNOutputLayers.inference [D2-recall]: predict_boxes (per-class apply_deltas), softmax, clip, score threshold,
        class-aware NMS, top-k per image; returns (padded detections, kept proposal rows)"""
        scores, deltas = predictions[0], predictions[1]
        N, P = proposals["valid"].shape
        K, nbx = self.K, self.nbox
        pb = proposals["boxes"]
        boxes = rpn_apply_deltas(deltas.float().view(N, P, nbx, 4), pb[:, :, None, :], self.box_weights)      # [N, P, nbx, 4]
        hwt = torch.tensor([[s[1], s[0], s[1], s[0]] for s in proposals.image_sizes], dtype=torch.float32, device=pb.device)[:, None, None, :]
        probs = F.softmax(scores.float(), dim=-1).view(N, P, K + 1)
        ok = proposals["valid"].bool() & torch.isfinite(boxes).all(dim=3).all(dim=2) & torch.isfinite(probs).all(dim=2)
        probs = probs[:, :, :K]
        boxes = torch.minimum(boxes.clamp(min=0), hwt)
        cand = (probs > self.test_score_thresh) & ok[:, :, None]
        flat = torch.where(cand, probs, torch.full_like(probs, -1.0)).reshape(N, P * K)
        k = min(max_cand, P * K)
        top = torch.topk(float_order_key(flat), k, dim=1, sorted=True).values
        idx = 4294967295 - (top & 4294967295)
        sc = torch.gather(flat, 1, idx)
        r, c = idx // K, (idx % K).to(torch.int32)
        bsel = r if nbx == 1 else idx                                   # the box of (row, class)
        cb = torch.gather(boxes.reshape(N, P * nbx, 4), 1, bsel[:, :, None].expand(-1, -1, 4)).contiguous()
        valid = (sc > self.test_score_thresh).to(torch.uint8)
        D = self.test_topk_per_image
        kidx, cnt = hip.nms_batched(cb, sc.contiguous(), c.contiguous(), valid.contiguous(), self.test_nms_thresh,
                                    class_aware=True, post_topk=-1, max_out=D)
        ix = kidx.clamp(min=0).long()
        keep_rows = torch.gather(r, 1, ix)
        out = PaddedBoxes(proposals.image_sizes,
                          boxes=torch.gather(cb, 1, ix[:, :, None].expand(-1, -1, 4)).contiguous(),
                          scores=torch.gather(sc, 1, ix).contiguous(), classes=torch.gather(c, 1, ix).contiguous(),
                          valid=(torch.arange(D, device=pb.device)[None, :] < cnt[:, None]).to(torch.uint8), count=cnt)
        return out, keep_rows


class FastRCNNCrossEntropyBoundaryVarOutputLayers(FastRCNNFocaltLossBoundaryVarOutputLayers):
    """MODEL.ROI_HEADS.LOSS "CrossEntropy_BoundaryVar" (reference fast_rcnn.py:214-712): the same predictor, box losses and inference;
    loss_cls is the mean softmax cross-entropy (:389,:400,:412) = the focal form (1-p)^gamma * CE at gamma 0, same kernel."""
    focal_gamma = 0.0


@ROI_HEADS_REGISTRY.register()
class StandardROIHeadsPseudoLab:
    def __init__(self, cfg, store, in_channels, prefix="roi_heads"):
        rh, bh = cfg.MODEL.ROI_HEADS, cfg.MODEL.ROI_BOX_HEAD
        self.in_features = list(rh.IN_FEATURES)
        self.scales = [1.0 / (2 ** int(f[1:])) for f in self.in_features]
        self.min_level = int(self.in_features[0][1:])
        self.res = bh.POOLER_RESOLUTION
        assert bh.POOLER_TYPE == "ROIAlignV2" and bh.POOLER_SAMPLING_RATIO == 0 and bh.NUM_CONV == 0
        self.num_classes = rh.NUM_CLASSES
        self.batch_size_per_image = rh.BATCH_SIZE_PER_IMAGE
        self.positive_fraction = rh.POSITIVE_FRACTION
        self.iou_threshold = rh.IOU_THRESHOLDS[0]
        self.proposal_append_gt = rh.PROPOSAL_APPEND_GT
        C, S, FC = in_channels, self.res, bh.FC_DIM
        self.fcs = []
        dim_in = C * S * S
        for i in range(bh.NUM_FC):
            p = "%s.box_head.fc%d" % (prefix, i + 1)
            if i == 0:  # weight stored [FC][S][S][C]: consumes the NHWC RoIAlign output as is
                w = store.new((FC, dim_in), "decay", _xavier_init(FC, dim_in, 1))
                w.export(p + ".weight", lambda t, FC=FC, S=S, C=C: t.view(FC, S, S, C).permute(0, 3, 1, 2).reshape(FC, -1),
                         load=lambda t, src, FC=FC, S=S, C=C: t.view(FC, S, S, C).copy_(src.view(FC, C, S, S).permute(0, 2, 3, 1)))
            else:
                w = store.new((FC, dim_in), "decay", _xavier_init(FC, dim_in, 1)).export(p + ".weight")
            b = store.new((FC,), "decay", lambda t: t.zero_()).export(p + ".bias")
            self.fcs.append(ops.Conv(w, dim_in, FC, 1, 1, 0, bias=b, relu=True))
            dim_in = FC
        if rh.LOSS == "FocalLoss_BoundaryVar":  # roi_heads.py:52-66
            self.box_predictor = FastRCNNFocaltLossBoundaryVarOutputLayers(cfg, store, dim_in, prefix + ".box_predictor")
        elif rh.LOSS == "CrossEntropy_BoundaryVar":
            self.box_predictor = FastRCNNCrossEntropyBoundaryVarOutputLayers(cfg, store, dim_in, prefix + ".box_predictor")
        elif rh.LOSS == "FocalLoss":      # the Unbiased-Teacher-v1 predictor (no boundary-variance head)
            self.box_predictor = FastRCNNFocaltLossOutputLayers(cfg, store, dim_in, prefix + ".box_predictor")
        elif rh.LOSS == "CrossEntropy":
            # Detectron2's plain FastRCNNOutputLayers: the reference cannot train it either - roi_heads.py:124 calls
            # `losses(predictions, proposals, branch)` on Detectron2's two-argument method (TypeError on the first step)
            raise NotImplementedError("MODEL.ROI_HEADS.LOSS 'CrossEntropy' selects Detectron2's FastRCNNOutputLayers, whose losses() the "
                                      "reference's ROI heads call with a third argument (roi_heads.py:124): it cannot train there either; "
                                      "use 'FocalLoss' (UTv1), 'FocalLoss_BoundaryVar' or 'CrossEntropy_BoundaryVar'")
        else:
            raise ValueError("Unknown ROI head loss.")
        self.training = True
        self.sample_keys = None

    def train(self, mode=True):
        self.training = mode

    @torch.no_grad()
    def label_and_sample_proposals(self, proposals, gt, branch=""):
        """roi_heads.py:141-270 (+ D2 add_ground_truth_to_proposals, Matcher(0.5), subsample 512 @ 25 % fg)."""
        pb, pv = proposals["boxes"], proposals["valid"].bool()
        if self.proposal_append_gt:
            pb = torch.cat((pb, gt["boxes"]), dim=1).contiguous()
            pv = torch.cat((pv, gt["valid"].bool()), dim=1)
        N, P = pv.shape
        mx, arg, _ = hip.match_boxes(pb, gt["boxes"], gt["valid"])
        if P <= 4096 and os.environ.get("UTV2_FUSED_SAMPLERS", "1") != "0":
            # classes, foreground / background draws, valid-first packing and the gathers of the sampled slots in one launch
            keys = _keys(self.sample_keys, N, P, pb.device)
            return hip.roi_sample(pb, pv, mx, arg, keys, gt["boxes"], gt["classes"], gt["valid"], gt["scores"] if "scores" in gt else None,
                                  gt["pred_boxes_std"] if "pred_boxes_std" in gt else None, self.iou_threshold, self.num_classes,
                                  self.batch_size_per_image, int(self.batch_size_per_image * self.positive_fraction))
        arg = arg.long()
        has_gt = gt["valid"].bool().any(dim=1, keepdim=True)
        fgm = (mx >= self.iou_threshold) & has_gt
        cls = torch.where(fgm, torch.gather(gt["classes"].long(), 1, arg), torch.full_like(arg, self.num_classes))
        keys = _keys(self.sample_keys, N, P, pb.device)
        nfg_max = int(self.batch_size_per_image * self.positive_fraction)
        fidx, fval = sample_k_smallest(keys, pv & (cls != self.num_classes), nfg_max)
        nfg = fval.sum(1, keepdim=True)
        bidx, bval = sample_k_smallest(keys, pv & (cls == self.num_classes), self.batch_size_per_image)
        bval = bval & (torch.arange(bidx.shape[1], device=pb.device)[None, :] < (self.batch_size_per_image - nfg))
        idx = torch.cat((fidx, bidx), dim=1)
        val = torch.cat((fval, bval), dim=1)
        order = torch.argsort((~val).to(torch.int8), dim=1, stable=True)[:, :self.batch_size_per_image]
        idx, val = torch.gather(idx, 1, order), torch.gather(val, 1, order)
        g = torch.gather(arg, 1, idx)
        hg = has_gt.expand_as(val)
        out = dict(proposal_boxes=torch.gather(pb, 1, idx[:, :, None].expand(-1, -1, 4)).contiguous(),
                   gt_classes=torch.where(val, torch.gather(cls, 1, idx), torch.full_like(idx, -1)),
                   gt_boxes=torch.where((val & hg)[:, :, None], torch.gather(gt["boxes"], 1, g[:, :, None].expand(-1, -1, 4)),
                                        torch.zeros((), device=pb.device)),
                   valid=val.to(torch.uint8).contiguous(), sampled_idx=idx)
        if "scores" in gt:
            out["gt_confid"] = torch.where(val & hg, torch.gather(gt["scores"], 1, g), torch.zeros((), device=pb.device))
            if "pred_boxes_std" in gt:
                out["gt_loc_std"] = torch.where((val & hg)[:, :, None], torch.gather(gt["pred_boxes_std"], 1, g[:, :, None].expand(-1, -1, 4)),
                                                torch.zeros((), device=pb.device))
        return out

    def _box_features(self, feats, boxes, valid, fanin=None):
        N, P = valid.shape
        rois = boxes.reshape(-1, 4).contiguous()
        batch = torch.arange(N, device=rois.device, dtype=torch.int32)[:, None].expand(N, P).reshape(-1).contiguous()
        x = ops.roi_align(feats, self.scales, self.min_level, rois, batch, valid.reshape(-1).contiguous(), self.res, rois_per_image=P,
                          fanin=fanin)
        x = x.view(x.shape[0], 1, 1, -1)
        for fc in self.fcs:
            x = fc(x)
        return x.view(x.shape[0], -1)

    def forward(self, features, proposals, targets=None, compute_loss=True, branch=""):
        feats = [features[f] for f in self.in_features]
        if self.training and compute_loss:
            assert targets is not None
            sampled = self.label_and_sample_proposals(proposals, targets, branch)
            self._last_sampled = sampled
            x = self._box_features(feats, sampled["proposal_boxes"], sampled["valid"], fanin=features.get("_fanin"))
            predictions = self.box_predictor(x)
            return sampled, self.box_predictor.losses(predictions, sampled, branch)
        x = self._box_features(feats, proposals["boxes"], proposals["valid"])
        predictions = self.box_predictor(x)
        pred_instances, _ = self.box_predictor.inference(predictions, proposals)
        return pred_instances, predictions

    __call__ = forward

    def forward_joint(self, features, proposals, n_labeled, gt_labeled, gt_unlabeled, raw=False):
        """forward(..., branch="supervised") on images [0, n_labeled) and forward(..., branch="unsup_data_train") on the rest as one
        RoIAlign / box head / predictor pass over all sampled ROIs; sampling and the losses stay per branch (the predictor output's rows
        are image-major, so each branch owns a contiguous row range).  Returns the two loss dicts."""
        assert self.training
        feats = [features[f] for f in self.in_features]
        nl = n_labeled
        s_l = self.label_and_sample_proposals(proposals.images(0, nl), gt_labeled, "supervised")
        s_u = self.label_and_sample_proposals(proposals.images(nl, proposals.n), gt_unlabeled, "unsup_data_train")
        self._last_sampled = s_u
        boxes = torch.cat((s_l["proposal_boxes"], s_u["proposal_boxes"]), dim=0)
        valid = torch.cat((s_l["valid"], s_u["valid"]), dim=0)
        scores, deltas, std = self.box_predictor(self._box_features(feats, boxes, valid, fanin=features.get("_fanin")))
        r = nl * boxes.shape[1]
        l_l = self.box_predictor.losses((scores[:r], deltas[:r], None if std is None else std[:r]), s_l, "supervised", raw=raw)
        l_u = self.box_predictor.losses((scores[r:], deltas[r:], None if std is None else std[r:]), s_u, "unsup_data_train", raw=raw)
        return l_l, l_u


@META_ARCH_REGISTRY.register()
class TwoStagePseudoLabGeneralizedRCNN(ArenaModel):
    """reference meta_arch/rcnn.py:6-72: forward(batched_inputs, branch, given_proposals, val_mode) -> 4-tuple."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, self.store, self.folder)
        self.proposal_generator = PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(
            cfg, self.store, self.backbone.out_channels)
        self.roi_heads = ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, self.store, self.backbone.out_channels)
        # D2 GeneralizedRCNN registers pixel_mean/std as NON-persistent buffers: not in state_dict
        self._mean_host, self._std_host = list(cfg.MODEL.PIXEL_MEAN), list(cfg.MODEL.PIXEL_STD)
        self._finalize()

    def _children(self):
        return [self.proposal_generator, self.roi_heads]

    def _gt(self, batched_inputs):
        first = batched_inputs[0]["instances"]
        if isinstance(first, PaddedBoxes):
            return first
        insts = [x["instances"] for x in batched_inputs]
        gt = PaddedBoxes.from_instances(insts, self.device)
        if len(insts) and insts[0].has("scores"):
            M = gt["valid"].shape[1]
            sc = torch.zeros((len(insts), M)); st = torch.zeros((len(insts), M, 4))
            for i, x in enumerate(insts):
                sc[i, :len(x)] = x.scores.detach().float().cpu()
                if x.has("pred_boxes_std"):
                    st[i, :len(x)] = x.pred_boxes_std.detach().float().cpu()
            gt.f["scores"] = sc.to(self.device)
            if insts[0].has("pred_boxes_std"):
                gt.f["pred_boxes_std"] = st.to(self.device)
        return gt

    def _fan_in(self, features):
        """one gradient hand-over per training pass: RoIAlign's level gradients go into the RPN conv's dgrad epilogue (ops.FanIn)"""
        lf = features.get("_levelfirst")
        rpn, roi = self.proposal_generator, self.roi_heads
        if (lf is None or not ops.fanin_enabled() or not torch.is_grad_enabled() or not lf[0].requires_grad
                or roi.in_features != rpn.in_features[:len(roi.in_features)]
                or lf[1].level_hw != [tuple(features[f].shape[1:3]) for f in rpn.in_features]):
            return
        fan = ops.FanIn(lf[1], len(roi.in_features))
        lf[0]._utv2_fanin = fan
        features["_fanin"] = fan

    def padded_canvas(self, batched_inputs):
        """(H, W) the batch is zero-padded to by preprocess_image (ImageList.from_tensors with size_divisibility)."""
        d = self.backbone.size_divisibility
        h = max(int(x["image"].shape[1]) for x in batched_inputs)
        w = max(int(x["image"].shape[2]) for x in batched_inputs)
        return ((h + d - 1) // d * d, (w + d - 1) // d * d) if d > 1 else (h, w)

    def forward_joint_begin(self, labeled_inputs, unlabeled_inputs, loss_weights=None):
        """model(labeled, branch="supervised") and model(unlabeled, branch="unsup_data_train") of one iteration (reference
        trainer.py:838-866) as ONE pass over the concatenated batch - every layer is per-image (FrozenBN; RPN / ROI heads work per image),
        so when both lists pad to the same canvas (the caller checks) the results are those of the two passes, with larger GEMMs, half
        the launches and one weight gradient per layer.  This first half runs everything that does not need the pseudo labels (backbone,
        RPN head, the proposals of all images, the labeled images' RPN losses) - the trainer runs it next to the teacher.
        loss_weights (optional, key -> the trainer's weight of that loss, `_pseudo` keys included): forward_joint_finish then runs the
        scalar tail of all eight losses fused and returns their weighted total as l_sup["weighted_total"]."""
        assert self.training
        both = list(labeled_inputs) + list(unlabeled_inputs)
        images = [x["image"].to(self.device) for x in both]
        x4, image_sizes = hip.preprocess_images(images, self._mean_host, self._std_host, self.backbone.size_divisibility,
                                                 bf16_stem=ops.amp() and not self.backbone.bottom_up.stem.trainable)
        gt_l = self._gt(labeled_inputs)
        self.folder.fold()
        features = self.backbone(x4)
        self._fan_in(features)
        raw = loss_weights is not None
        rctx, proposals, rpn_l = self.proposal_generator.forward_joint_begin(image_sizes, features, len(labeled_inputs), gt_l, raw=raw)
        return dict(features=features, rpn=rctx, proposals=proposals, rpn_l=rpn_l, gt_l=gt_l, n_labeled=len(labeled_inputs),
                    loss_weights=loss_weights)

    LOSS_KEYS = ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc")

    def forward_joint_finish(self, ctx, gt_unlabeled):
        """second half: the pseudo-labeled images' RPN losses and the ROI heads of both branches, given the pseudo ground truth
        (a PaddedBoxes of the unlabeled images).  Returns (supervised loss dict, pseudo loss dict).  When forward_joint_begin was given
        the trainer's loss weights (key -> weight) the scalar tail runs fused - ONE launch maps the raw kernel sums to the eight losses,
        their weighted total (l_sup["weighted_total"]) and the backward coefficients (ops.rcnn_loss_combine) instead of ~60 single-element
        ATen launches, each waiting for the host with nothing else queued."""
        lw = ctx.get("loss_weights")
        if lw is not None:
            pg, rh = self.proposal_generator, self.roi_heads
            rpn_l = ctx["rpn_l"]
            rpn_u = pg.forward_joint_finish(ctx["rpn"], gt_unlabeled, raw=True)
            roi_l, roi_u = rh.forward_joint(ctx["features"], ctx["proposals"], ctx["n_labeled"], ctx["gt_l"], gt_unlabeled, raw=True)
            wt = [float(lw[k]) for k in self.LOSS_KEYS] + [float(lw[k + "_pseudo"]) for k in self.LOSS_KEYS]
            consts = (rpn_l["norm"], rpn_u["norm"], pg.loss_weight.get("loss_rpn_cls", 1.0) ** 2, pg.loss_weight.get("loss_rpn_loc", 1.0) ** 2,
                      rh.box_predictor.loss_weight.get("loss_box_reg", 1.0), wt)
            total, rec = ops.rcnn_loss_combine(rpn_l["raw_sums"], rpn_u["raw_sums"], roi_l["focal"], roi_u["focal"], roi_l["box"], roi_u["box"],
                                               roi_l["tgt"], roi_u["tgt"], consts)
            l_l = {k: rec[i] for i, k in enumerate(self.LOSS_KEYS)}
            l_u = {k: rec[4 + i] for i, k in enumerate(self.LOSS_KEYS)}
            l_l["weighted_total"] = total
            return l_l, l_u
        rpn_u = self.proposal_generator.forward_joint_finish(ctx["rpn"], gt_unlabeled)
        roi_l, roi_u = self.roi_heads.forward_joint(ctx["features"], ctx["proposals"], ctx["n_labeled"], ctx["gt_l"], gt_unlabeled)
        l_l, l_u = {}, {}
        l_l.update(roi_l); l_l.update(ctx["rpn_l"])
        l_u.update(roi_u); l_u.update(rpn_u)
        return l_l, l_u

    def forward(self, batched_inputs, branch="supervised", given_proposals=None, val_mode=False):
        if (not self.training) and (not val_mode):
            return self.inference(batched_inputs)
        images = [x["image"].to(self.device) for x in batched_inputs]
        x4, image_sizes = hip.preprocess_images(images, self._mean_host, self._std_host, self.backbone.size_divisibility,
                                                 bf16_stem=ops.amp() and not self.backbone.bottom_up.stem.trainable)
        gt = self._gt(batched_inputs) if "instances" in batched_inputs[0] else None
        self.folder.fold()
        features = self.backbone(x4)
        if branch in ("supervised", "unsup_data_train"):
            self._fan_in(features)
            proposals_rpn, proposal_losses = self.proposal_generator(image_sizes, features, gt)
            _, detector_losses = self.roi_heads(features, proposals_rpn, gt, branch=branch)
            losses = {}
            losses.update(detector_losses)
            losses.update(proposal_losses)
            return losses, [], [], None
        elif branch == "unsup_data_weak":
            proposals_rpn, _ = self.proposal_generator(image_sizes, features, None, compute_loss=False)
            proposals_roih, roi_predictions = self.roi_heads(features, proposals_rpn, targets=None, compute_loss=False, branch=branch)
            return {}, proposals_rpn, proposals_roih, roi_predictions
        raise ValueError("Unknown branch: {}".format(branch))

    __call__ = forward

    @torch.no_grad()
    def inference(self, batched_inputs):
        images = [x["image"].to(self.device) for x in batched_inputs]
        x4, image_sizes = hip.preprocess_images(images, self._mean_host, self._std_host, self.backbone.size_divisibility,
                                                 bf16_stem=ops.amp() and not self.backbone.bottom_up.stem.trainable)
        self.folder.fold()
        features = self.backbone(x4)
        proposals, _ = self.proposal_generator(image_sizes, features, None, compute_loss=False)
        dets, _ = self.roi_heads(features, proposals, targets=None, compute_loss=False)
        from .one_stage_detector import detector_postprocess
        out = []
        for inst, inp, size in zip(_dets_to_instances(dets), batched_inputs, image_sizes):
            out.append({"instances": detector_postprocess(inst, inp.get("height", size[0]), inp.get("width", size[1]))})
        return out


def _dets_to_instances(dets):
    res = []
    for i in range(dets.n):
        m = dets["valid"][i].bool()
        inst = Instances(dets.image_sizes[i])
        inst.pred_boxes = Boxes(dets["boxes"][i][m])
        inst.scores = dets["scores"][i][m]
        inst.pred_classes = dets["classes"][i][m].long()
        if "pred_boxes_std" in dets:
            inst.pred_boxes_std = dets["pred_boxes_std"][i][m]
        res.append(inst)
    return res
