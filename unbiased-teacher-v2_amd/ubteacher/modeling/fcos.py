"""FCOS proposal generator (head + targets + losses + decode) on the HIP path.

Mirrors the reference's `FCOS` / `FCOSHead` / `FCOSOutputs` (ubteacher/modeling/fcos/fcos.py:44-376,
fcos_outputs.py:132-1320) behind the same registry name and the same branch / nms_method
arguments, re-laid-out for MI355X:

  * head outputs of all levels are written by the convs straight into two level-first
    [P, 80] buffers (cls logits; box = 68 reg bins | 4 std | 1 ctr | 7 pad): the reference's
    permute/reshape/cat (fcos_outputs.py:261-290) never exists;
  * bbox_pred / bbox_pred_std / ctrness share one fused 256->80 conv (their weights are adjacent
    rows of one arena matrix; state_dict still exposes the three reference tensors);
  * targets, focal, and the positive-location losses are dense kernels over all locations with
    device-side normalisers: no nonzero()/item() host syncs (the reference has ~10 per call);
  * pseudo-labels stay on the device as padded slots + validity masks.
"""
import math

import os

import torch

from .. import hip, ops
from ..d2.registry import PROPOSAL_GENERATOR_REGISTRY
from ..d2.structures import Boxes, Instances
from ..utils import comm
from .backbone import _nchw_view

INF = 100000000
METHODS = {"cls": 0, "cls_n_ctr": 1, "ctr": 2, "cls_n_loc": 3}
BOX_STRIDE = 80  # 4*(REG_MAX+1) + 4 + 1 = 73 -> padded to 80 (multiple of 16 for the dgrad fast path)


class PaddedBoxes:
    """Per-image padded detections / ground truth living on the device."""

    def __init__(self, image_sizes, **fields):
        self.image_sizes = image_sizes
        self.f = fields  # boxes [N,M,4], classes [N,M] int32, valid [N,M] uint8, scores, reg_pred_std, ...

    def __getitem__(self, k):
        """field name -> batched padded tensor; image index -> reference-style Instances of that image (the return type of the
        reference's predict_proposals / process_pseudo_label is list[Instances]: `out[0].gt_boxes` keeps working, at the price of a
        device sync - the trainers only use the batched form)"""
        if isinstance(k, int):
            return self.to_instances(as_gt=self.as_gt)[k]
        return self.f[k]

    def __contains__(self, k):
        return k in self.f

    def __len__(self):
        return self.n

    def __iter__(self):
        return iter(self.to_instances(as_gt=self.as_gt))

    as_gt = False   # True once thresholded into pseudo ground truth (gt_boxes / gt_classes instead of pred_boxes / pred_classes)

    @property
    def n(self):
        return self.f["valid"].shape[0]

    def images(self, a, b):
        """images [a, b) of the batch (views)"""
        out = PaddedBoxes(list(self.image_sizes[a:b]), **{k: v[a:b] for k, v in self.f.items()})
        out.as_gt = self.as_gt
        return out

    def pad_images(self, before, after):
        """The same boxes inside a longer batch: `before` / `after` empty images are added around them (device-side cat, no
        sync) - the ground truth of one loss branch of a fused student pass."""
        out = {}
        for k, v in self.f.items():
            if k == "count":
                continue
            out[k] = torch.nn.functional.pad(v, (0, 0) * (v.dim() - 1) + (before, after))   # one launch per field (was zeros, zeros, cat)
        sizes = list(self.image_sizes)
        pad = sizes[0] if sizes else (0, 0)
        return PaddedBoxes([pad] * before + sizes + [pad] * after, **out)

    def threshold(self, thr, ctr_thr=None):
        """pseudo_generator.py:62-131: keep scores > thr (or cls_confid > thr0 & centerness > thr1)."""
        if ctr_thr is None:
            keep = self.f["scores"] > thr
        else:
            keep = (self.f["cls_confid"] > thr) & (self.f["centerness"] > ctr_thr)
        out = dict(self.f)
        out["valid"] = (self.f["valid"].bool() & keep).to(torch.uint8)
        res = PaddedBoxes(self.image_sizes, **out)
        res.as_gt = True
        return res

    def to_instances(self, as_gt=False):
        """Materialise reference-style Instances (forces a device sync; API compatibility only)."""
        res = []
        for i in range(self.n):
            m = self.f["valid"][i].bool()
            inst = Instances(self.image_sizes[i])
            if as_gt:
                inst.gt_boxes = Boxes(self.f["boxes"][i][m])
                inst.gt_classes = self.f["classes"][i][m].long()
            else:
                inst.pred_boxes = Boxes(self.f["boxes"][i][m])
                inst.pred_classes = self.f["classes"][i][m].long()
            for k in ("scores", "centerness", "cls_confid", "reg_pred_std", "locations", "fpn_levels"):
                if k in self.f:
                    inst.set(k, self.f[k][i][m])
            res.append(inst)
        return res

    @staticmethod
    def from_instances(instances, device, min_slots=16):
        """list[Instances] with gt_boxes/gt_classes (+ optional scores, reg_pred_std) -> padded device arrays."""
        n = len(instances)
        g = max([len(x) for x in instances] + [1])
        M = max(min_slots, (g + 15) // 16 * 16)
        assert M <= 256, "more than 256 ground-truth boxes per image is not supported by utv2_fcos_targets"
        has_std = any(x.has("reg_pred_std") for x in instances if len(x))
        dev = torch.device(device)
        on_dev = dev.type == "cuda" and all(x.gt_boxes.tensor.device.type == "cuda" and x.gt_classes.device.type == "cuda" for x in instances if len(x))
        if on_dev:
            # ground truth that already lives on the device (GPU data pipeline, synthetic loaders) is padded there: the host round trip
            # below is a device-to-host sync per image plus pageable host-to-device copies, each of which drains the stream
            boxes = torch.zeros((n, M, 4), dtype=torch.float32, device=dev)
            classes = torch.zeros((n, M), dtype=torch.int32, device=dev)
            valid = torch.zeros((n, M), dtype=torch.uint8, device=dev)
            std = torch.zeros((n, M, 4), dtype=torch.float32, device=dev) if has_std else None
            for i, x in enumerate(instances):
                k = len(x)
                if k == 0:
                    continue
                boxes[i, :k] = x.gt_boxes.tensor.detach().float()
                classes[i, :k] = x.gt_classes.detach().to(torch.int32)
                valid[i, :k] = 1
                if has_std and x.has("reg_pred_std"):
                    std[i, :k] = x.reg_pred_std.detach().float().to(dev)
            f = dict(boxes=boxes, classes=classes, valid=valid)
            if std is not None:
                f["reg_pred_std"] = std
            return PaddedBoxes([x.image_size for x in instances], **f)
        if dev.type == "cuda":
            # host ground truth: all arrays packed into ONE pinned staging buffer and copied with non_blocking=True - a `.to(device)` of
            # pageable memory is followed by a stream synchronize (three of them drained the stream at the start of every step)
            nb, nc = n * M * 16, n * M * 4
            total = nb * (2 if has_std else 1) + nc + n * M
            stage = PaddedBoxes._stage(dev, total)
            boxes = stage[:nb].view(torch.float32).view(n, M, 4).zero_()
            std = stage[nb:2 * nb].view(torch.float32).view(n, M, 4).zero_() if has_std else None
            o = nb * (2 if has_std else 1)
            classes = stage[o:o + nc].view(torch.int32).view(n, M).zero_()
            valid = stage[o + nc:o + nc + n * M].view(n, M).zero_()
        else:
            boxes = torch.zeros((n, M, 4), dtype=torch.float32)
            classes = torch.zeros((n, M), dtype=torch.int32)
            valid = torch.zeros((n, M), dtype=torch.uint8)
            std = torch.zeros((n, M, 4), dtype=torch.float32) if has_std else None
        for i, x in enumerate(instances):
            k = len(x)
            if k == 0:
                continue
            boxes[i, :k] = x.gt_boxes.tensor.detach().float().cpu()
            classes[i, :k] = x.gt_classes.detach().cpu().to(torch.int32)
            valid[i, :k] = 1
            if has_std and x.has("reg_pred_std"):
                std[i, :k] = x.reg_pred_std.detach().float().cpu()
        if dev.type == "cuda":
            d = stage[:total].to(dev, non_blocking=True)
            PaddedBoxes._stage_done(dev)
            f = dict(boxes=d[:nb].view(torch.float32).view(n, M, 4), classes=d[o:o + nc].view(torch.int32).view(n, M),
                     valid=d[o + nc:o + nc + n * M].view(n, M))
            if has_std:
                f["reg_pred_std"] = d[nb:2 * nb].view(torch.float32).view(n, M, 4)
            return PaddedBoxes([x.image_size for x in instances], **f)
        f = dict(boxes=boxes.to(device), classes=classes.to(device), valid=valid.to(device))
        if std is not None:
            f["reg_pred_std"] = std.to(device)
        return PaddedBoxes([x.image_size for x in instances], **f)

    _STAGE = {}

    @staticmethod
    def _stage(dev, nbytes):
        """two pinned uint8 staging buffers per device, used alternately; a buffer is reused only after the copy issued from it has finished"""
        st = PaddedBoxes._STAGE.setdefault(str(dev), {"bufs": [None, None], "events": [None, None], "next": 0})
        i = st["next"]
        if st["events"][i] is not None:
            st["events"][i].synchronize()
        if st["bufs"][i] is None or st["bufs"][i].numel() < nbytes:
            st["bufs"][i] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
        return st["bufs"][i]

    @staticmethod
    def _stage_done(dev):
        st = PaddedBoxes._STAGE[str(dev)]
        i = st["next"]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        st["events"][i] = ev
        st["next"] = i ^ 1


def compute_locations(h, w, stride, device):
    """utils/comm.py:34-45 of the reference: (x*s + s//2, y*s + s//2), row-major."""
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32, device=device)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32, device=device)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    return torch.stack((xx.reshape(-1), yy.reshape(-1)), dim=1) + stride // 2


class RawOutput(dict):
    """The reference's `raw_output` dict (modeling/fcos/fcos.py:110-138): `logits_pred`, `reg_pred`, `reg_pred_std`, `ctrness_pred` as
    per-level NCHW tensors, `top_feats`, `bbox_towers`, `locations`, `image_sizes`.  The NCHW tensors are zero-copy strided views of
    the fused level-first NHWC buffers, built on first access (the trainers never touch them: PseudoGenerator.nms_from_dense reads
    `head_out` / `level_hw`, which ride along as extra keys; `box_pred` = the fused bbox | std | ctrness buffer per level, NHWC)."""

    _LAZY = ("logits_pred", "reg_pred", "reg_pred_std", "ctrness_pred", "top_feats", "bbox_towers", "locations", "box_pred")

    def __init__(self, head_out, level_hw, image_sizes, strides, reg_max):
        super().__init__(head_out=head_out, level_hw=level_hw, image_sizes=image_sizes)
        self._strides, self._nreg = list(strides), 4 * (reg_max + 1)

    def __missing__(self, key):
        if key not in self._LAZY:
            raise KeyError(key)
        ho, hw = dict.__getitem__(self, "head_out"), dict.__getitem__(self, "level_hw")
        meta, L, R = ho["meta"], len(hw), self._nreg

        def nchw(buf, c0, c1):
            return [meta.level_view(buf, l)[..., c0:c1].permute(0, 3, 1, 2) for l in range(L)]
        if key == "logits_pred":
            v = nchw(ho["logits"], 0, ho["logits"].shape[1])
        elif key == "reg_pred":
            v = nchw(ho["box"], 0, R)
        elif key == "reg_pred_std":
            v = nchw(ho["box"], R, R + 4)
        elif key == "ctrness_pred":
            v = nchw(ho["box"], R + 4, R + 5)
        elif key == "box_pred":
            v = [meta.level_view(ho["box"], l) for l in range(L)]
        elif key == "locations":
            v = [compute_locations(h, w, s, ho["logits"].device) for (h, w), s in zip(hw, self._strides)]
        elif key == "bbox_towers" and "bbox_tower" in ho:      # MODEL.FCOS.YIELD_PROPOSAL (fcos.py:135,338-350), NCHW views per level
            v = nchw(ho["bbox_tower"], 0, ho["bbox_tower"].shape[1])
        else:  # top_feats (no top module), bbox_towers without YIELD_PROPOSAL: empty lists there too
            v = []
        self[key] = v
        return v

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._LAZY

    def keys(self):
        return list(dict.keys(self)) + [k for k in self._LAZY if not dict.__contains__(self, k)]


class FCOSHead:
    yield_bbox_towers = False

    def __init__(self, cfg, store, in_channels, prefix):
        fc = cfg.MODEL.FCOS
        assert fc.NORM == "GN", "only the GN towers of the shipped configs are built"
        assert fc.REG_DISCRETE and fc.REG_MAX == 16, "UTv2 FCOS configs: REG_DISCRETE, REG_MAX 16"
        # KL_LOSS False (config-reachable, fcos.py:300-307 then builds no bbox_pred_std): the fused box conv keeps its 4 std channels so
        # the [reg | std | ctr] row layout is one layout; they then receive zero gradient and no consumer reads them.
        assert not fc.USE_DEFORMABLE
        self.num_classes = fc.NUM_CLASSES
        self.reg_max = fc.REG_MAX
        self.num_levels = len(fc.FPN_STRIDES)
        C = in_channels
        self.C = C
        self.towers = {}
        # The cls and the bbox tower are two independent chains of identical 3x3 256 -> 256 convs + GroupNorm(32) + ReLU
        # (fcos/fcos.py:252-304).  Equal depth, no shared tower (every shipped config): they are built PAIRED - depth i of both is one
        # [2C, 9C] weight (rows [0, C) = cls_tower.{3i}, rows [C, 2C) = bbox_tower.{3i}; state_dict() exposes the reference's tensors),
        # one conv launch over a [P, 2C] activation (depth 0: a plain C -> 2C conv of the shared FPN feature, deeper: a grouped conv,
        # groups = 2) and one GroupNorm(64) launch set.  Against two chains: half the launches, ONE tile-quantisation remainder per
        # depth instead of two (the 256-tile kernel runs whole rounds of 256 tiles; 2 x 1050 tiles = 8 rounds + 52 instead of
        # 2 x (4 rounds + 26)), half the split-K slabs of the weight gradients (18 tiles x 14 pixel splits instead of 2 x 9 x 28), and the
        # two gradients of the FPN feature are summed inside depth 0's dgrad K loop instead of by an add pass.
        self.paired = (fc.NUM_CLS_CONVS == fc.NUM_BOX_CONVS and fc.NUM_SHARE_CONVS == 0 and fc.NUM_CLS_CONVS > 0
                       and os.environ.get("UTV2_PAIR_TOWERS", "1") != "0")      # 0: two separate chains (A/B knob; another arena layout)
        if self.paired:
            n = fc.NUM_CLS_CONVS
            ws = []
            layers = []
            for i in range(n):
                pc, pb = "%s.cls_tower.%d" % (prefix, 3 * i), "%s.bbox_tower.%d" % (prefix, 3 * i)
                w = store.new((2 * C, 9 * C), "decay", None)
                w.export(pc + ".weight", lambda t: t[:C].view(C, 3, 3, C).permute(0, 3, 1, 2))
                w.export(pb + ".weight", lambda t: t[C:].view(C, 3, 3, C).permute(0, 3, 1, 2))
                ws.append(w)
                b = store.new((2 * C,), "decay", lambda t: t.zero_())
                b.export(pc + ".bias", lambda t: t[:C]).export(pb + ".bias", lambda t: t[C:])
                conv = ops.Conv(w, C, 2 * C, 3, 1, 1, bias=b, groups=1 if i == 0 else 2)
                conv.defer_wgrad = True     # see ops.defer_tower_wgrads (UTV2_DEFER_TOWER_WGRAD)
                gc, gb = "%s.cls_tower.%d" % (prefix, 3 * i + 1), "%s.bbox_tower.%d" % (prefix, 3 * i + 1)
                ga = store.new((2 * C,), "nodecay", lambda t: t.fill_(1.0))
                ga.export(gc + ".weight", lambda t: t[:C]).export(gb + ".weight", lambda t: t[C:])
                be = store.new((2 * C,), "nodecay", lambda t: t.zero_())
                be.export(gc + ".bias", lambda t: t[:C]).export(gb + ".bias", lambda t: t[C:])
                layers.append(ops.pair_conv_gn(conv, ops.GroupNormReLU(ga, be, 64, 1e-5, True)))
                if i > 0:
                    ops.chain_gn_conv(layers[i - 1][1], conv)   # layer i - 1's GroupNorm output feeds this conv only

            def init_all(_t):
                # the reference's module order (cls tower first, then the bbox tower): the random stream of the initialisation - and with
                # it every seeded golden - is the one of two separate towers
                for half in (0, 1):
                    for w in ws:
                        w.t[half * C:(half + 1) * C].normal_(0.0, 0.01)
            ws[0].init = init_all
            self.towers["pair"] = layers
            order = []
            for name in ("cls", "bbox"):
                for i in range(n):
                    order += ["%s.%s_tower.%d.weight" % (prefix, name, 3 * i), "%s.%s_tower.%d.bias" % (prefix, name, 3 * i),
                              "%s.%s_tower.%d.weight" % (prefix, name, 3 * i + 1), "%s.%s_tower.%d.bias" % (prefix, name, 3 * i + 1)]
            store.order_keys(order)
        for name, nconv in (("cls", fc.NUM_CLS_CONVS), ("bbox", fc.NUM_BOX_CONVS), ("share", fc.NUM_SHARE_CONVS)):
            if self.paired:
                self.towers[name] = []
                continue

            layers = []
            for i in range(nconv):
                p = "%s.%s_tower.%d" % (prefix, name, 3 * i)
                w = store.new((C, 9 * C), "decay", lambda t: t.normal_(0.0, 0.01)).export(p + ".weight", _nchw_view(C, C, 3))
                b = store.new((C,), "decay", lambda t: t.zero_()).export(p + ".bias")
                conv = ops.Conv(w, C, C, 3, 1, 1, bias=b)
                pg = "%s.%s_tower.%d" % (prefix, name, 3 * i + 1)
                ga = store.new((C,), "nodecay", lambda t: t.fill_(1.0)).export(pg + ".weight")
                be = store.new((C,), "nodecay", lambda t: t.zero_()).export(pg + ".bias")
                layers.append(ops.pair_conv_gn(conv, ops.GroupNormReLU(ga, be, 32, 1e-5, True)))
                if i > 0:
                    ops.chain_gn_conv(layers[i - 1][1], conv)
            self.towers[name] = layers
        nc = self.num_classes
        prior = fc.PRIOR_PROB
        bias_value = -math.log((1 - prior) / prior)
        w = store.new((nc, 9 * C), "decay", lambda t: t.normal_(0.0, 0.01)).export(prefix + ".cls_logits.weight", _nchw_view(nc, C, 3))
        b = store.new((nc,), "decay", lambda t: t.fill_(bias_value)).export(prefix + ".cls_logits.bias")
        self.cls_logits = ops.Conv(w, C, nc, 3, 1, 1, bias=b, out_fp32=True)
        # fused box head: rows [0:R4) bbox_pred, [R4:R4+4) bbox_pred_std, [R4+4] ctrness, rest zero padding
        R4 = 4 * (self.reg_max + 1)
        self.R4 = R4

        def init_box(t):
            t.zero_()
            t[:R4].normal_(0.0, 0.01)
            t[R4:R4 + 4].normal_(0.0, 0.0001)
            t[R4 + 4:R4 + 5].normal_(0.0, 0.01)

        def rows(r0, r1):
            def fn(t):
                return t[r0:r1].view(r1 - r0, 3, 3, C).permute(0, 3, 1, 2)
            return fn

        wb = store.new((BOX_STRIDE, 9 * C), "decay", init_box)
        wb.export(prefix + ".bbox_pred.weight", rows(0, R4))
        wb.export(prefix + ".bbox_pred_std.weight", rows(R4, R4 + 4))
        wb.export(prefix + ".ctrness.weight", rows(R4 + 4, R4 + 5))
        bb = store.new((BOX_STRIDE,), "decay", lambda t: t.zero_())
        bb.export(prefix + ".bbox_pred.bias", lambda t: t[0:R4])
        bb.export(prefix + ".bbox_pred_std.bias", lambda t: t[R4:R4 + 4])
        bb.export(prefix + ".ctrness.bias", lambda t: t[R4 + 4:R4 + 5])
        self.box_head = ops.Conv(wb, C, BOX_STRIDE, 3, 1, 1, bias=bb, colscale=R4, out_fp32=True)
        self.scales = None
        if fc.USE_SCALE:
            self.scales = [store.new((1,), "decay", lambda t: t.fill_(1.0)).export("%s.scales.%d.scale" % (prefix, l))
                           for l in range(self.num_levels)]

    def __call__(self, big, meta):
        """big: level-first [P, C] features of all levels; ONE launch per conv for all levels.
        Returns dict(logits [P,80], box [P,80], meta)."""
        t = big
        if self.paired:
            for conv, gn in self.towers["pair"]:
                t = gn(conv(t, meta=meta), meta)          # [P, 2C]: cls | bbox
            tc, tb = ops.split_cols(t)                    # column-half views (row pitch 2C): the prediction convs read them in place
        else:
            for conv, gn in self.towers["share"]:
                t = gn(conv(t, meta=meta), meta)
            tc = t
            for conv, gn in self.towers["cls"]:
                tc = gn(conv(tc, meta=meta), meta)
            tb = t
            for conv, gn in self.towers["bbox"]:
                tb = gn(conv(tb, meta=meta), meta)
        logits = self.cls_logits(tc, meta=meta)
        box = self.box_head(tb, meta=meta, colscale_handle=self.scales)
        out = {"logits": logits, "box": box, "meta": meta}
        if self.yield_bbox_towers:       # fcos.py:338-350 (MODEL.FCOS.YIELD_PROPOSAL): the bbox tower's output, level-first [P, C] (a view)
            out["bbox_tower"] = tb
        return out


class FCOSOutputs:
    """Targets, losses and proposal decode (reference fcos_outputs.py:132-1320) on dense kernels."""

    def __init__(self, cfg):
        fc = cfg.MODEL.FCOS
        self.focal_loss_alpha = fc.LOSS_ALPHA
        self.focal_loss_gamma = fc.LOSS_GAMMA
        self.center_radius = float(fc.POS_RADIUS) if fc.CENTER_SAMPLE else 0.0   # get_sample_region (fcos_outputs.py:700-770)
        self.pre_nms_thresh_train = fc.INFERENCE_TH_TRAIN
        self.pre_nms_topk_train = fc.PRE_NMS_TOPK_TRAIN
        self.post_nms_topk_train = fc.POST_NMS_TOPK_TRAIN
        self.pre_nms_thresh_test = fc.INFERENCE_TH_TEST
        self.pre_nms_topk_test = fc.PRE_NMS_TOPK_TEST
        self.post_nms_topk_test = fc.POST_NMS_TOPK_TEST
        self.nms_thresh = fc.NMS_TH
        assert not fc.THRESH_WITH_CTR
        self.num_classes = fc.NUM_CLASSES
        self.strides = list(fc.FPN_STRIDES)
        assert not cfg.SEMISUPNET.SOFT_CLS_LABEL
        assert cfg.SEMISUPNET.CLS_LOSS_METHOD == "focal"
        self.reg_max = fc.REG_MAX
        self.unify_ctrcls = fc.UNIFY_CTRCLS
        # config-reachable variants (fcos_outputs.py:165-186): the positive-location kernels take them as flags
        self.kl_loss, self.kl_loss_type = fc.KL_LOSS, fc.KL_LOSS_TYPE
        if self.kl_loss and self.kl_loss_type not in ("klloss", "nlloss"):
            raise ValueError("MODEL.FCOS.KL_LOSS_TYPE must be 'klloss' or 'nlloss'")
        if fc.LOC_LOSS_TYPE not in ("giou", "iou", "linear_iou"):
            raise NotImplementedError("MODEL.FCOS.LOC_LOSS_TYPE %r" % (fc.LOC_LOSS_TYPE,))  # iou_loss.py:70-71
        if fc.QUALITY_EST not in ("centerness", "iou"):
            raise ValueError("MODEL.FCOS.QUALITY_EST must be 'centerness' or 'iou'")
        # LOC_FUN_ALL is the `method` of the KL-type term (fcos_outputs.py:388,407,583): NLLoss ignores it (kl_loss.py:86-105), KLLoss
        # reduces by it (kl_loss.py:48-64)
        if fc.LOC_FUN_ALL not in ("mean", "sum", "weight_ctr_sum", "weight_ctr_mean"):
            raise ValueError("No defined regression loss method")   # kl_loss.py:63-64
        self.loc_fun_all = fc.LOC_FUN_ALL
        self.loc_flags = {"giou": 0, "iou": hip.LT_LOC_IOU, "linear_iou": hip.LT_LOC_LINEAR_IOU}[fc.LOC_LOSS_TYPE]
        if self.kl_loss and self.kl_loss_type == "klloss":
            self.loc_flags |= hip.LT_KLLOSS
            if self.loc_fun_all in ("weight_ctr_sum", "weight_ctr_mean"):
                self.loc_flags |= hip.LT_KL_WCTR
        self.quality_iou = fc.QUALITY_EST == "iou"
        self.kl_loss_weight = fc.KLLOSS_WEIGHT
        self.reg_unsup_loss = cfg.SEMISUPNET.CONSIST_REG_LOSS
        self.tsbetter_reg = cfg.SEMISUPNET.TS_BETTER
        self.tsbetter_reg_cert = cfg.SEMISUPNET.TS_BETTER_CERT
        soi, prev = [], -1
        for s in fc.SIZES_OF_INTEREST:
            soi.append([prev, s])
            prev = s
        soi.append([prev, INF])
        self.sizes_of_interest = soi
        self.training = True

    # -- targets --------------------------------------------------------------------------------
    def _targets(self, level_hw, gt, drop_empty, active=None, batch=None, img0=0):
        std = gt["reg_pred_std"] if "reg_pred_std" in gt else None
        return hip.fcos_targets(level_hw, self.strides, self.sizes_of_interest, gt["boxes"], gt["classes"],
                                gt["valid"], std, self.num_classes, drop_empty, active, center_radius=self.center_radius, batch=batch, img0=img0)

    @staticmethod
    def _normalisers(sums):
        """num_pos_avg = max(allreduce(n_pos)/world, 1), loss_denorm = max(allreduce(sum ctr)/world, 1e-6)
        (fcos_outputs.py:317-321,361-362) as ONE fused 2-float all-reduce kept on the device."""
        ws = comm.get_world_size()
        pair = sums[0:2].detach().clone()
        pair = comm.reduce_sum(pair)
        npa = (pair[0] / ws).clamp(min=1.0)
        den = (pair[1] / ws).clamp(min=1e-6)
        return npa, den

    @staticmethod
    def _joint_normaliser_sums(sums_s, sums_c, sums_r):
        """world sums of the (n_pos, sum ctrness) pairs of the three target sets of a fused student pass -> [6] (None in a world of 1).
        Three 2-float all-reduces in the order losses() / pseudo_losses() issue theirs through _normalisers(): a rank whose two student
        passes could not be fused (its lists pad to different canvases) runs those instead, and every rank must issue the same sequence
        of collectives."""
        if comm.get_world_size() == 1:
            return None
        return torch.cat([comm.reduce_sum(x[0:2].detach().clone()) for x in (sums_s, sums_c, sums_r)])

    def _kl_mean(self, sums, den=None):
        """The KL-type term reduced by LOC_FUN_ALL: NLLoss always averages over positives (kl_loss.py:93-105); KLLoss "mean" averages over
        positives x 4 boundaries (kl_loss.py:59-60), "sum" / "weight_ctr_sum" sum (the latter with the centerness-target weight the
        kernel applied, LT_KL_WCTR), "weight_ctr_mean" divides that by the loss normaliser (`den`, fcos_outputs.py:362)."""
        n = sums[0].detach().clamp(min=1.0)
        if self.kl_loss_type != "klloss":
            return sums[4] / n
        if self.loc_fun_all == "mean":
            return sums[4] / (4.0 * n)
        if self.loc_fun_all == "weight_ctr_mean":
            return sums[4] / den
        return sums[4]

    # -- supervised branch (fcos_outputs.py:212-444) -------------------------------------------------
    def losses(self, head_out, level_hw, gt, branch="labeled", active=None, ignore_near=False):
        """active (optional, uint8 [N]): images of the batch this branch owns (fused student pass); the others are ignored.
        ignore_near (fcos_outputs.py:841-851, with CENTER_SAMPLE): locations inside a box but in no box's sampling region are dropped."""
        if branch != "labeled":
            raise ValueError("Incorrect branch name")
        logits_all, box_all = head_out["logits"], head_out["box"]
        labels, reg_t, bvars, gt_inds = self._targets(level_hw, gt, drop_empty=3 if ignore_near else 1, active=active)
        focal = ops.focal_loss_sum(logits_all, labels, self.focal_loss_alpha, self.focal_loss_gamma)
        flags = self.loc_flags | (hip.LT_QUALITY_IOU if self.quality_iou else 0)  # QUALITY_EST acts on this branch only (:353-359)
        sums = ops.fcos_loc_terms(box_all, labels, reg_t, None, (self.num_classes, self.reg_max, 0.0, 0.0, flags))
        npa, den = self._normalisers(sums)
        loc = sums[3] / den
        if self.kl_loss:
            w = self.kl_loss_weight
            loc = w * (w * self._kl_mean(sums, den)) + loc  # weight applied twice (:381/:397, :400/:416)
        losses = {"loss_fcos_cls": focal[0] / npa, "loss_fcos_loc": loc, "loss_fcos_ctr": sums[2] / npa}
        extras = {"labels": labels, "reg_targets": reg_t, "gt_inds": gt_inds, "loss_denorm": den, "sums": sums}
        return extras, losses

    # -- pseudo branch (fcos_outputs.py:447-631) -----------------------------------------------------
    def pseudo_losses(self, head_out, level_hw, gt_dict, branch="unlabeled", active=None):
        assert branch == "unlabeled"
        logits_all, box_all = head_out["logits"], head_out["box"]
        losses, extras = {}, {}
        for labeltype, gt in gt_dict.items():
            labels, reg_t, bvars, gt_inds = self._targets(level_hw, gt, drop_empty=0, active=active)
            if labeltype == "cls":
                focal = ops.focal_loss_sum(logits_all, labels, self.focal_loss_alpha, self.focal_loss_gamma)
                sums = ops.fcos_loc_terms(box_all, labels, reg_t, None, (self.num_classes, self.reg_max, 0.0, 0.0, self.loc_flags))
                npa, den = self._normalisers(sums)
                losses["loss_fcos_cls"] = focal[0] / npa
                ctr = sums[2] / npa
                losses["loss_fcos_ctr"] = ctr * 0 if self.unify_ctrcls else ctr
            elif labeltype == "reg":
                if not self.kl_loss:
                    raise ValueError("pseudo regression loss needs MODEL.FCOS.KL_LOSS")  # fcos_outputs.py:587-588
                tsbetter = self.reg_unsup_loss == "ts_locvar_better_nms_nll_l1"
                sums = ops.fcos_loc_terms(box_all, labels, reg_t, bvars if tsbetter else None,
                                          (self.num_classes, self.reg_max, self.tsbetter_reg, self.tsbetter_reg_cert, self.loc_flags))
                _, den_r = self._normalisers(sums)  # the reference issues the same two reductions here (:504,:521)
                if tsbetter:
                    losses["loss_fcos_loc"] = sums[6] / sums[5].detach().clamp(min=1.0)
                    losses["teacher_better_student"] = sums[5].detach()
                else:  # any other CONSIST_REG_LOSS: KLLOSS_WEIGHT * (NLL | KL) term on the pseudo set (:571-585)
                    losses["loss_fcos_loc"] = self.kl_loss_weight * self._kl_mean(sums, den_r)
            else:
                raise ValueError(labeltype)
            extras["labels_" + labeltype] = labels
            extras["sums_" + labeltype] = sums
        return extras, losses

    # -- both branches of a fused student pass, scalar tail in one launch -------------------------------
    LOSS_KEYS = ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr", "loss_fcos_cls_pseudo", "loss_fcos_ctr_pseudo", "loss_fcos_loc_pseudo")

    def joint_losses(self, head_out, level_hw, gt_labeled, gt_unlabeled, n_labeled, N, loss_weights):
        """losses(labeled) + pseudo_losses(unlabeled) + the trainer's loss weighting: the same target / focal / positive-location kernels,
        but everything between their raw sums and the weighted total - normalisers, KL means, weights (about 60 scalar launches forward
        and 70 backward, the GPU idle in between) - is ONE utv2_fcos_loss_combine launch whose backward hands the kernels their
        coefficients.  loss_weights: key -> (mul, div), the loss enters the total as value * mul / div.  Returns (supervised dict,
        pseudo dict (keys without the suffix), weighted total); the dict entries are detached (metrics)."""
        logits_all, box_all = head_out["logits"], head_out["box"]
        nc, rm = self.num_classes, self.reg_max
        if os.environ.get("UTV2_JOINT_LOSS_NODE", "1") != "0":
            # round 4: the three branches as ONE autograd node whose backward writes one gradient tensor per head output (ops._FcosJointLossFn)
            lab_s, reg_s, _, _ = self._targets(level_hw, gt_labeled, drop_empty=1, batch=N, img0=0)
            lab_c, reg_c, _, _ = self._targets(level_hw, gt_unlabeled["cls"], drop_empty=0, batch=N, img0=n_labeled)
            lab_r, reg_r, bv_r, _ = self._targets(level_hw, gt_unlabeled["reg"], drop_empty=0, batch=N, img0=n_labeled)
            tsbetter = self.reg_unsup_loss == "ts_locvar_better_nms_nll_l1"
            flags = (1 if self.kl_loss else 0) | (2 if self.kl_loss_type == "klloss" else 0) | (4 if self.unify_ctrcls else 0) | (8 if tsbetter else 0)
            consts = (self.focal_loss_alpha, self.focal_loss_gamma, nc, rm, self.loc_flags | (hip.LT_QUALITY_IOU if self.quality_iou else 0),
                      self.loc_flags, self.tsbetter_reg, self.tsbetter_reg_cert, float(comm.get_world_size()), flags, self.kl_loss_weight,
                      [loss_weights[k][0] for k in self.LOSS_KEYS], [loss_weights[k][1] for k in self.LOSS_KEYS])
            total, rec = ops.fcos_joint_loss(logits_all, box_all, ((lab_s, reg_s), (lab_c, reg_c), (lab_r, reg_r, bv_r if tsbetter else None)),
                                             consts, self._joint_normaliser_sums)
            l_sup = {"loss_fcos_cls": rec[0], "loss_fcos_loc": rec[1], "loss_fcos_ctr": rec[2]}
            l_uns = {"loss_fcos_cls": rec[3], "loss_fcos_ctr": rec[4], "loss_fcos_loc": rec[5]}
            if tsbetter:
                l_uns["teacher_better_student"] = rec[6]
            return l_sup, l_uns, total
        # images [0, n_labeled) of the batch carry ground truth, [n_labeled, N) the pseudo labels: the target kernel takes the range
        labels, reg_t, _, _ = self._targets(level_hw, gt_labeled, drop_empty=1, batch=N, img0=0)
        focal_s = ops.focal_loss_sum(logits_all, labels, self.focal_loss_alpha, self.focal_loss_gamma)
        flags_s = self.loc_flags | (hip.LT_QUALITY_IOU if self.quality_iou else 0)
        sums_s = ops.fcos_loc_terms(box_all, labels, reg_t, None, (nc, rm, 0.0, 0.0, flags_s))
        labels, reg_t, _, _ = self._targets(level_hw, gt_unlabeled["cls"], drop_empty=0, batch=N, img0=n_labeled)
        focal_c = ops.focal_loss_sum(logits_all, labels, self.focal_loss_alpha, self.focal_loss_gamma)
        sums_c = ops.fcos_loc_terms(box_all, labels, reg_t, None, (nc, rm, 0.0, 0.0, self.loc_flags))
        tsbetter = self.reg_unsup_loss == "ts_locvar_better_nms_nll_l1"
        labels, reg_t, bvars, _ = self._targets(level_hw, gt_unlabeled["reg"], drop_empty=0, batch=N, img0=n_labeled)
        sums_r = ops.fcos_loc_terms(box_all, labels, reg_t, bvars if tsbetter else None,
                                    (nc, rm, self.tsbetter_reg, self.tsbetter_reg_cert, self.loc_flags))
        ws = comm.get_world_size()
        norm = self._joint_normaliser_sums(sums_s, sums_c, sums_r)
        flags = (1 if self.kl_loss else 0) | (2 if self.kl_loss_type == "klloss" else 0) | (4 if self.unify_ctrcls else 0) | (8 if tsbetter else 0)
        wmul = [loss_weights[k][0] for k in self.LOSS_KEYS]
        wdiv = [loss_weights[k][1] for k in self.LOSS_KEYS]
        total, rec = ops.fcos_loss_combine(focal_s, sums_s, focal_c, sums_c, sums_r, norm, float(ws), flags, self.kl_loss_weight, wmul, wdiv)
        l_sup = {"loss_fcos_cls": rec[0], "loss_fcos_loc": rec[1], "loss_fcos_ctr": rec[2]}
        l_uns = {"loss_fcos_cls": rec[3], "loss_fcos_ctr": rec[4], "loss_fcos_loc": rec[5]}
        if tsbetter:
            l_uns["teacher_better_student"] = rec[6]
        return l_sup, l_uns, total

    # -- decode + NMS (fcos_outputs.py:1046-1320) ---------------------------------------------------
    def predict_proposals(self, head_out, level_hw, image_sizes, nms_method="cls_n_ctr", max_det=None):
        """nms_method: one ranking criterion -> PaddedBoxes; a tuple / list of criteria -> a list of PaddedBoxes in that order, computed
        by ONE set of launches (ranking keys, exact top-k, decode, class-aware NMS) over (criterion, image) pairs - the UTv2 trainer needs
        the teacher's detections under two criteria every iteration (trainer.py:232-251) and these kernels are latency-bound."""
        if self.training:
            th, pre, post = self.pre_nms_thresh_train, self.pre_nms_topk_train, self.post_nms_topk_train
        else:
            th, pre, post = self.pre_nms_thresh_test, self.pre_nms_topk_test, self.post_nms_topk_test
        if max_det is None:
            # detection slots per image: POST_NMS_TOPK plus headroom for the kthvalue ties the reference keeps beyond it
            # (fcos_outputs.py:1300-1318); 128 for the shipped 100
            max_det = 128 if post <= 100 else (post + post // 4 + 63) // 64 * 64
        if post > max_det:
            raise ValueError("POST_NMS_TOPK %d exceeds the %d detection slots" % (post, max_det))
        single = isinstance(nms_method, str)
        names = [nms_method] if single else list(nms_method)
        for nm in names:
            if nm not in METHODS:
                raise ValueError("Undefined nms criteria")
        methods = [METHODS[nm] for nm in names]
        M = len(methods)
        meta = head_out["meta"]
        logits_all, box_all = head_out["logits"].detach(), head_out["box"].detach()
        N = meta.N
        NV = M * N  # virtual images: criterion-major
        dev = logits_all.device
        ks = [min(pre, h * w * self.num_classes) for (h, w) in level_hw]
        MAXC = sum(ks)
        outs = dict(
            boxes=torch.empty((NV, MAXC, 4), dtype=torch.float32, device=dev),
            scores=torch.empty((NV, MAXC), dtype=torch.float32, device=dev),
            classes=torch.empty((NV, MAXC), dtype=torch.int32, device=dev),
            locations=torch.empty((NV, MAXC, 2), dtype=torch.float32, device=dev),
            centerness=torch.empty((NV, MAXC), dtype=torch.float32, device=dev),
            cls_confid=torch.empty((NV, MAXC), dtype=torch.float32, device=dev),
            reg_pred_std=torch.empty((NV, MAXC, 4), dtype=torch.float32, device=dev),
            fpn_levels=torch.empty((NV, MAXC), dtype=torch.int32, device=dev),
            valid=torch.empty((NV, MAXC), dtype=torch.uint8, device=dev),
        )
        # per-level top-k (fcos_outputs.py:1238-1241): the ranking keys of every (level, criterion, image) triple are the ragged rows of
        # ONE flat buffer and one exact radix select (utv2_topk_rows_i64) serves them all - no padding, no per-level launches
        L = len(level_hw)
        C = self.num_classes
        widths = [h * w * C for h, w in level_hw]
        kmax = max(ks)
        if kmax <= 2048:
            ck = (NV, tuple(level_hw), C, str(dev))
            cached = getattr(self, "_topk_rows", None)
            if cached is None or cached[0] != ck:
                offs, o = [], 0
                for wd in widths:
                    for n in range(NV):
                        offs.append(o + n * wd)
                    o += NV * wd
                offs.append(o)
                cached = (ck, torch.tensor(offs, dtype=torch.int64, device=dev), o)
                self._topk_rows = cached
            row_off, total = cached[1], cached[2]
            keys = torch.empty(total, dtype=torch.int64, device=dev)
            o = 0
            for l, (h, w) in enumerate(level_hw):
                r0, r1 = meta.rows[l]
                for method in methods:
                    hip.fcos_rank_keys(logits_all[r0:r1], box_all[r0:r1], self.reg_max, N, h * w, th, method,
                                       out=keys[o:o + N * widths[l]].view(N, widths[l]), row_stride=widths[l])
                    o += N * widths[l]
            top_all = hip.topk_rows(keys, row_off, L * NV, max(widths), kmax)
            tops = {(l, m): top_all[(l * M + m) * N:(l * M + m + 1) * N, :ks[l]].contiguous() for l in range(L) for m in range(M)}
        else:  # very large PRE_NMS_TOPK: torch's radix select, one padded matrix per group of levels
            groups = [[0], list(range(1, L))] if L > 1 else [[0]]
            tops = {}
            for m, method in enumerate(methods):
                for grp in groups:
                    width = max(widths[l] for l in grp)
                    keys = torch.full((len(grp) * N, width), -1, dtype=torch.int64, device=dev)
                    for i, l in enumerate(grp):
                        h, w = level_hw[l]
                        r0, r1 = meta.rows[l]
                        hip.fcos_rank_keys(logits_all[r0:r1], box_all[r0:r1], self.reg_max, N, h * w, th, method,
                                           out=keys[i * N:(i + 1) * N], row_stride=width)
                    top_all = torch.topk(keys, max(ks[l] for l in grp), dim=1, sorted=True).values
                    for i, l in enumerate(grp):
                        tops[(l, m)] = top_all[i * N:(i + 1) * N, :ks[l]].contiguous()
        for m, method in enumerate(methods):
            outs_m = {k: v[m * N:(m + 1) * N] for k, v in outs.items()}
            slot0 = 0
            for l, (h, w) in enumerate(level_hw):
                r0, r1 = meta.rows[l]
                hip.fcos_decode(tops[(l, m)], logits_all[r0:r1], box_all[r0:r1], self.reg_max, N, h * w, w, self.strides[l], l, method, slot0,
                                outs_m)
                slot0 += ks[l]
        keep, cnt = hip.nms_batched(outs["boxes"], outs["scores"], outs["classes"], outs["valid"], self.nms_thresh,
                                    class_aware=True, post_topk=post, max_out=max_det)
        idx = keep.clamp(min=0).long()
        valid = (torch.arange(max_det, device=dev)[None, :] < cnt[:, None]).to(torch.uint8)
        f = {}
        for k, v in outs.items():
            if k == "valid":
                continue
            ix = idx if v.dim() == 2 else idx[:, :, None].expand(-1, -1, v.shape[2])
            f[k] = torch.gather(v, 1, ix).contiguous()
        f["valid"] = valid
        f["count"] = cnt
        res = [PaddedBoxes(image_sizes, **{k: v[m * N:(m + 1) * N].contiguous() for k, v in f.items()}) for m in range(M)]
        return res[0] if single else res


class _LazyProposals(dict):
    """the `results` dict of a training forward under MODEL.FCOS.YIELD_PROPOSAL: "proposals" is decoded + NMSed on first access"""

    def __init__(self, base, make):
        super().__init__(base)
        self._make = make

    def __missing__(self, key):
        if key != "proposals":
            raise KeyError(key)
        with torch.no_grad():
            v = self._make()
        self[key] = v
        return v

    def __contains__(self, key):
        return key == "proposals" or dict.__contains__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return list(dict.keys(self)) + ([] if dict.__contains__(self, "proposals") else ["proposals"])


@PROPOSAL_GENERATOR_REGISTRY.register()
class FCOS:
    def __init__(self, cfg, store, in_channels, prefix="proposal_generator"):
        fc = cfg.MODEL.FCOS
        self.in_features = list(fc.IN_FEATURES)
        self.fpn_strides = list(fc.FPN_STRIDES)
        self.yield_proposal = fc.YIELD_PROPOSAL
        self.fcos_head = FCOSHead(cfg, store, in_channels, prefix + ".fcos_head")
        self.fcos_head.yield_bbox_towers = self.yield_proposal
        self.fcos_outputs = FCOSOutputs(cfg)
        # Integral.project is a persistent buffer in the reference (fcos_outputs.py:61-63): keep the key.
        self.project = store.new((fc.REG_MAX + 1,), "buffer",
                                 lambda t: t.copy_(torch.linspace(0, fc.REG_MAX, fc.REG_MAX + 1))).export(
            prefix + ".fcos_outputs.integral.project")
        self.training = True

    def train(self, mode=True):
        self.training = mode
        self.fcos_outputs.training = mode

    def forward(self, image_sizes, features, gt_instances=None, output_raw=False, nms_method="cls_n_ctr",
                ignore_near=False, branch="labeled"):
        feats = [features[f] for f in self.in_features]
        level_hw = [(f.shape[1], f.shape[2]) for f in feats]
        lf = features.get("_levelfirst")
        if lf is not None and lf[1].level_hw == level_hw:
            big, meta = lf
        else:  # features that do not share a level-first buffer: build one (copy)
            meta = ops.LevelMeta(feats[0].shape[0], level_hw)
            big = torch.cat([f.reshape(-1, f.shape[-1]) for f in feats], dim=0)
        head_out = self.fcos_head(big, meta)
        raw_output = RawOutput(head_out, level_hw, image_sizes, self.fpn_strides, self.fcos_outputs.reg_max)
        results = {}
        if self.training:
            if branch == "labeled":
                results, losses = self.fcos_outputs.losses(head_out, level_hw, gt_instances, branch=branch, ignore_near=ignore_near)
            elif branch == "unlabeled":
                # ignore_near (SEMISUPNET.PSEUDO_CLS_IGNORE_NEAR, trainer.py:340,347) only fills keep_locations in the reference
                # (fcos_outputs.py:474,841-851), which its pseudo losses never read (:487-631): accepted, no effect - as there
                results, losses = self.fcos_outputs.pseudo_losses(head_out, level_hw, gt_instances, branch=branch)
            elif branch == "raw":
                results, losses = {}, {}
            else:
                raise ValueError("Unknown branch")
            if self.yield_proposal:
                # fcos.py:176-187 (every shipped FCOS YAML sets YIELD_PROPOSAL): the student's own detections under the *_TRAIN thresholds /
                # top-k, no gradient.  The reference computes them on every training forward and its trainer drops them
                # (one_stage_detector.py:213-217 returns the losses); here they are decoded when somebody reads results["proposals"].
                fo = self.fcos_outputs
                results = _LazyProposals(results, lambda: fo.predict_proposals(head_out, level_hw, image_sizes, nms_method))
            if output_raw:
                return results, losses, raw_output
            return results, losses
        with torch.no_grad():
            results = self.fcos_outputs.predict_proposals(head_out, level_hw, image_sizes, nms_method)
        if output_raw:
            return results, {}, raw_output
        return results, {}

    def forward_joint(self, image_sizes, features, n_labeled, gt_labeled, gt_unlabeled, loss_weights=None):
        """Both student passes of one UTv2 iteration on ONE batch: images [0, n_labeled) carry ground truth (branch
        "labeled"), the rest pseudo labels (branch "unlabeled").  Every op of the network is per-image (FrozenBN,
        per-image GroupNorm), so this equals the two separate forwards whenever both groups share one padded canvas."""
        return self.forward_joint_finish(self.forward_joint_begin(image_sizes, features, n_labeled, gt_labeled), gt_unlabeled, loss_weights)

    def forward_joint_begin(self, image_sizes, features, n_labeled, gt_labeled):
        assert self.training
        feats = [features[f] for f in self.in_features]
        level_hw = [(f.shape[1], f.shape[2]) for f in feats]
        big, meta = features["_levelfirst"]
        head_out = self.fcos_head(big, meta)
        return dict(head_out=head_out, level_hw=level_hw, n_labeled=n_labeled, gt_labeled=gt_labeled, N=meta.N, device=big.device)

    def forward_joint_finish(self, ctx, gt_unlabeled, loss_weights=None):
        """loss_weights (key -> (mul, div), the trainer's weighting of the loss dict): when given and both pseudo label sets are there,
        the scalar tail runs fused (FCOSOutputs.joint_losses) and the weighted total rides along as l_sup["weighted_total"]."""
        head_out, level_hw, n_labeled, N = ctx["head_out"], ctx["level_hw"], ctx["n_labeled"], ctx["N"]
        fo = self.fcos_outputs
        if (loss_weights is not None and set(gt_unlabeled) == {"cls", "reg"} and fo.kl_loss and all(k in loss_weights for k in fo.LOSS_KEYS)
                and (fo.kl_loss_type != "klloss" or fo.loc_fun_all == "mean")   # the fused scalar tail knows the "mean" reduction only
                and ctx["gt_labeled"].n == n_labeled and all(v.n == N - n_labeled for v in gt_unlabeled.values())):
            l_sup, l_uns, total = fo.joint_losses(head_out, level_hw, ctx["gt_labeled"], gt_unlabeled, n_labeled, N, loss_weights)
            l_sup["weighted_total"] = total
            return l_sup, l_uns
        act = torch.zeros(N, dtype=torch.uint8, device=ctx["device"])
        act[:n_labeled] = 1
        gtl = ctx["gt_labeled"].pad_images(0, N - n_labeled)
        gtu = {k: v.pad_images(n_labeled, 0) for k, v in gt_unlabeled.items()}
        _, l_sup = fo.losses(head_out, level_hw, gtl, active=act)
        _, l_uns = fo.pseudo_losses(head_out, level_hw, gtu, active=(1 - act))
        return l_sup, l_uns

    __call__ = forward
