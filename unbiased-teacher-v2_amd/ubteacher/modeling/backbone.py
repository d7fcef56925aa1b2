"""ResNet-50 + FPN on the HIP conv path.

Behavioural spec: Detectron2 v0.6 ResNet (STRIDE_IN_1X1, FrozenBN, FREEZE_AT=2) and FPN
[D2-recall, SURVEY.md appendix C]; FCOS top block: reference ubteacher/modeling/backbone/fpn.py:11-78
(P6 = conv3x3 s2 (P5), P7 = conv3x3 s2 (relu(P6))).  State-dict keys follow Detectron2's
(`bottom_up.stem.conv1.weight`, `bottom_up.res3.0.conv1.norm.running_mean`, `fpn_lateral3.weight`,
`top_block.p6.weight`, ...) so checkpoints stay interchangeable.

MI355X mapping: every conv is one implicit-GEMM launch with FrozenBN/ReLU/residual fused in
the epilogue; stem+res2 (frozen) run outside autograd; the image batch enters as NHWC4.
"""
import math
import os

import torch

from .. import hip, ops
from ..d2.registry import BACKBONE_REGISTRY


def _nchw_view(cout, cin, k):
    """[cout, k*k*cin] arena matrix -> [cout, cin, k, k]-shaped strided view (state_dict surface)."""
    def fn(t):
        return t.view(cout, k, k, cin).permute(0, 3, 1, 2)
    return fn


def _msra_init(cout, cin, k):
    def fn(t):  # kaiming_normal_(mode="fan_out", nonlinearity="relu")
        std = math.sqrt(2.0 / (cout * k * k))
        t.normal_(0.0, std)
    return fn


def _xavier_init(cout, cin, k):
    def fn(t):  # c2_xavier_fill: kaiming_uniform_(a=1)  -> bound = sqrt(3 / fan_in)
        bound = math.sqrt(3.0 / (cin * k * k))
        t.uniform_(-bound, bound)
    return fn


class FrozenBN:
    """Four per-channel buffers; folded to scale/shift by one launch for all layers (utv2_frozenbn_fold)."""

    def __init__(self, store, prefix, c):
        self.w = store.new((c,), "bn_w", lambda t: t.fill_(1.0)).export(prefix + ".weight")
        self.b = store.new((c,), "bn_b", lambda t: t.zero_()).export(prefix + ".bias")
        self.m = store.new((c,), "bn_m", lambda t: t.zero_()).export(prefix + ".running_mean")
        # D2 FrozenBatchNorm2d initialises running_var to 1 - eps so that an affine-only checkpoint (R-50.pkl has no running
        # statistics) yields scale == weight exactly [D2-recall]
        self.v = store.new((c,), "bn_v", lambda t: t.fill_(1.0 - 1e-5)).export(prefix + ".running_var")
        self.c = c
        self.scale = None
        self.shift = None


class BNFolder:
    def __init__(self, store):
        self.store = store
        self.layers = []
        self.scale_all = None

    def add(self, bn):
        self.layers.append(bn)
        return bn

    def materialize(self):
        st = self.store
        s0, e0 = st.ranges["bn_w"]
        n = e0 - s0
        self.n = n
        self.scale_all = torch.empty(n, dtype=torch.float32, device=st.flat.device)
        self.shift_all = torch.empty(n, dtype=torch.float32, device=st.flat.device)
        for bn in self.layers:
            o = bn.w.offset - s0
            bn.scale = self.scale_all[o: o + bn.c]
            bn.shift = self.shift_all[o: o + bn.c]

    def fold(self, eps=1e-5):
        if self.n == 0:
            return
        st = self.store
        hip.frozenbn_fold(st.region("bn_w"), st.region("bn_b"), st.region("bn_m"), st.region("bn_v"),
                          self.scale_all, self.shift_all, eps)


def _conv_bn(store, folder, prefix, cin, cout, k, stride, pad, relu, trainable):
    kind = "decay" if trainable else "frozen"
    w = store.new((cout, k * k * cin), kind, _msra_init(cout, cin, k)).export(prefix + ".weight", _nchw_view(cout, cin, k))
    bn = folder.add(FrozenBN(store, prefix + ".norm", cout))
    return ops.Conv(w, cin, cout, k, stride, pad, bias=None, bn=bn, relu=relu, trainable=trainable)


class Bottleneck:
    def __init__(self, store, folder, prefix, cin, cout, mid, stride, trainable):
        # STRIDE_IN_1X1: the stride sits on conv1
        self.conv1 = _conv_bn(store, folder, prefix + ".conv1", cin, mid, 1, stride, 0, True, trainable)
        self.conv2 = _conv_bn(store, folder, prefix + ".conv2", mid, mid, 3, 1, 1, True, trainable)
        self.conv3 = _conv_bn(store, folder, prefix + ".conv3", mid, cout, 1, 1, 0, True, trainable)  # relu after add
        self.shortcut = None
        if cin != cout:
            self.shortcut = _conv_bn(store, folder, prefix + ".shortcut", cin, cout, 1, stride, 0, False, trainable)
        # backward of the fused AMP node (ops._BottleneckFn): input_relu - the gradient returned for the input is masked by input > 0
        # in the dgrad epilogue; grad_premasked - every consumer of this block's output does the same, so no mask pass of its own.
        # Both are switched on by the FPN builder, which knows all consumers of the stage outputs.
        self.input_relu = False
        self.grad_premasked = False

    def __call__(self, x):
        return ops.bottleneck(self, x)


class ResNet50:
    def __init__(self, store, folder, prefix, out_features, freeze_at=2):
        self.out_features = out_features
        # stem on the NHWC4 image: [64][7*7*4 -> padded to 208]
        def stem_init(t):
            t.zero_()
            w = torch.empty(64, 7, 7, 3, device=t.device).normal_(0.0, math.sqrt(2.0 / (64 * 49)))
            t.view(64, 208)[:, :196].view(64, 7, 7, 4)[..., :3].copy_(w)
        stem_train = freeze_at < 1
        self.stem_w = store.new((64, 208), "decay" if stem_train else "frozen", stem_init).export(
            prefix + ".stem.conv1.weight", lambda t: torch.as_strided(t, (64, 3, 7, 7), (208, 1, 28, 4), t.storage_offset()))
        self.stem_bn = folder.add(FrozenBN(store, prefix + ".stem.conv1.norm", 64))
        self.stem = ops.Conv(self.stem_w, 4, 64, 7, 2, 3, bn=self.stem_bn, relu=True, trainable=stem_train, kred=208)
        self.stages = []
        cin = 64
        for si, (name, nblocks, mid, cout) in enumerate([("res2", 3, 64, 256), ("res3", 4, 128, 512),
                                                          ("res4", 6, 256, 1024), ("res5", 3, 512, 2048)]):
            trainable = freeze_at < si + 2
            blocks = []
            for b in range(nblocks):
                stride = 2 if (b == 0 and name != "res2") else 1
                blocks.append(Bottleneck(store, folder, "%s.%s.%d" % (prefix, name, b), cin, cout, mid, stride, trainable))
                cin = cout
            self.stages.append((name, blocks, trainable))
        self.channels = {"res2": 256, "res3": 512, "res4": 1024, "res5": 2048}
        self.strides = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}

    def __call__(self, x4):
        outs = {}
        if self.stem.trainable and torch.is_grad_enabled():
            # MODEL.BACKBONE.FREEZE_AT < 1: the stem trains (ops._StemFn; the image comes as fp32 NHWC4 in every precision mode)
            assert x4.dtype == torch.float32 and x4.shape[-1] == 4, "a trainable stem takes the fp32 NHWC4 image"
            x = ops.stem(self.stem, x4)
            for name, blocks, trainable in self.stages:
                for b in blocks:
                    x = b(x)
                if name in self.out_features:
                    outs[name] = x
            return outs
        # frozen stem + pool run outside autograd; under AMP the bf16 activation pipeline starts at the stem's output
        sc, sh = self.stem.scale_shift()
        if x4.dtype == hip.h16_dtype():  # AMP: 16-bit MFMA stem on the zero-bordered 16-bit image
            v = (self.stem_w.store.version, hip.H16[0])
            if getattr(self, "_w16s_version", None) != v:
                self._w16s = hip.stem_weight_image(self.stem_w.t)
                self._w16s_version = v
            if os.environ.get("UTV2_FUSED_STEM_POOL", "1") != "0" and self._w16s.shape[0] == 64:
                x = hip.stem_pool_fwd_bf16(x4, self._w16s, sc, sh)      # conv + FrozenBN + ReLU + max pool, the conv output stays in LDS
            else:
                x = hip.maxpool3x3s2(hip.conv2d_stem_fwd_bf16(x4, self._w16s, sc, sh, True, hip.h16_dtype()))
        else:
            x = hip.maxpool3x3s2(hip.conv2d_stem_fwd(x4, self.stem_w.t, sc, sh, 2, 3, 7, 7, True, ops.act_dtype()))
        for name, blocks, trainable in self.stages:
            if trainable:
                for b in blocks:
                    x = b(x)
            else:
                with torch.no_grad():
                    for b in blocks:
                        x = b(x)
            if name in self.out_features:
                outs[name] = x
        return outs


def _conv_bias(store, prefix, cin, cout, k, stride, pad, init, relu=False):
    w = store.new((cout, k * k * cin), "decay", init(cout, cin, k)).export(prefix + ".weight", _nchw_view(cout, cin, k))
    b = store.new((cout,), "decay", lambda t: t.zero_()).export(prefix + ".bias")
    return ops.Conv(w, cin, cout, k, stride, pad, bias=b, relu=relu, trainable=True)


class FPN:
    """lateral 1x1 + top-down nearest x2 sum + output 3x3 (FUSE_TYPE sum, NORM '')."""

    def __init__(self, store, folder, cfg, top_block_kind):
        self.bottom_up = ResNet50(store, folder, "backbone.bottom_up", cfg.MODEL.RESNETS.OUT_FEATURES,
                                  cfg.MODEL.BACKBONE.FREEZE_AT)
        self.in_features = list(cfg.MODEL.FPN.IN_FEATURES)
        oc = cfg.MODEL.FPN.OUT_CHANNELS
        self.out_channels = oc
        self.lateral, self.output = {}, {}
        for f in self.in_features:
            stage = int(math.log2(self.bottom_up.strides[f]))
            self.lateral[f] = _conv_bias(store, "backbone.fpn_lateral%d" % stage, self.bottom_up.channels[f], oc, 1, 1, 0, _xavier_init)
            self.output[f] = _conv_bias(store, "backbone.fpn_output%d" % stage, oc, oc, 3, 1, 1, _xavier_init)
        # Gradient pre-masking (ops.premask_on): a stage output is consumed by the next block / next stage's first block and by this
        # stage's lateral only, and all of them mask the gradient they return by (output > 0) in their dgrad epilogue - so the fused
        # bottlenecks of the trainable stages skip the mask pass at the top of their backward.
        for name, blocks, trainable in self.bottom_up.stages:
            if trainable:
                for blk in blocks:
                    blk.input_relu = True
                    blk.grad_premasked = True
                if name in self.lateral:
                    self.lateral[name].premask_input = True
        self.top_block_kind = top_block_kind
        self.levels_fp32 = False  # every consumer of the level buffer (tower / RPN convs, RoIAlign) takes bf16 under AMP
        self.top = []
        if top_block_kind == "p6p7":
            self.top = [_conv_bias(store, "backbone.top_block.p6", oc, oc, 3, 2, 1, _xavier_init),
                        _conv_bias(store, "backbone.top_block.p7", oc, oc, 3, 2, 1, _xavier_init)]
        elif top_block_kind == "p6":
            self.top = [_conv_bias(store, "backbone.top_block.p6", oc, oc, 3, 2, 1, _xavier_init)]
        stages = [int(math.log2(self.bottom_up.strides[f])) for f in self.in_features]
        self.out_names = ["p%d" % s for s in stages]
        last = stages[-1]
        ntop = {"p6p7": 2, "p6": 1, "maxpool": 1, "none": 0}[top_block_kind]
        self.out_names += ["p%d" % (last + 1 + i) for i in range(ntop)]
        self.size_divisibility = self.bottom_up.strides[self.in_features[-1]]

    def __call__(self, x4):
        """Returns {p_l: NHWC level tensor}.  All levels are rows of ONE level-first buffer
        (`results["_levelfirst"] = (big [P,C], LevelMeta)`) so shared heads run one launch for all levels."""
        feats = self.bottom_up(x4)
        N = x4.shape[0]
        C = self.out_channels
        hw = {}
        for f in self.in_features:
            hw["p%d" % int(math.log2(self.bottom_up.strides[f]))] = (feats[f].shape[1], feats[f].shape[2])
        last = "p%d" % int(math.log2(self.bottom_up.strides[self.in_features[-1]]))
        nlast = int(last[1:])
        h, w = hw[last]
        for i in range(len(self.out_names) - len(self.in_features)):
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1   # conv3x3 s2 p1 == max_pool(k1,s2) output size
            hw["p%d" % (nlast + 1 + i)] = (h, w)
        meta = ops.LevelMeta(N, [hw[k] for k in self.out_names])
        big = torch.empty((meta.P, C), dtype=torch.float32 if self.levels_fp32 else ops.act_dtype(), device=x4.device)
        slot = {k: meta.alias_view(big, i) for i, k in enumerate(self.out_names)}
        results = {}
        prev = None
        for f in reversed(self.in_features):
            lat = self.lateral[f](feats[f])
            if prev is not None:
                lat = ops.upsample2x_add(lat, prev)
            prev = lat
            name = "p%d" % int(math.log2(self.bottom_up.strides[f]))
            results[name] = self.output[f](lat, out=slot[name])
        if self.top_block_kind == "p6p7":
            p6 = self.top[0](results[last], out=slot["p%d" % (nlast + 1)])
            p7 = self.top[1](ops.relu(p6), out=slot["p%d" % (nlast + 2)])
            results["p%d" % (nlast + 1)] = p6
            results["p%d" % (nlast + 2)] = p7
        elif self.top_block_kind == "p6":
            results["p%d" % (nlast + 1)] = self.top[0](results[last], out=slot["p%d" % (nlast + 1)])
        elif self.top_block_kind == "maxpool":
            # LastLevelMaxPool: max_pool2d(kernel 1, stride 2) == strided subsample
            results["p%d" % (nlast + 1)] = ops.subsample2_into(results[last], slot["p%d" % (nlast + 1)])
        out = {k: results[k] for k in self.out_names}
        levels = [out[k] for k in self.out_names]
        if torch.is_grad_enabled() and any(t.requires_grad for t in levels):
            bigt = ops.assemble(big, [(r0, r1, tuple(t.shape)) for (r0, r1), t in zip(meta.rows, levels)], levels)
        else:
            bigt = big
        out["_levelfirst"] = (bigt, meta)
        return out


@BACKBONE_REGISTRY.register()
def build_fcos_resnet_fpn_backbone(cfg, store, folder):
    tl = cfg.MODEL.FCOS.TOP_LEVELS
    return FPN(store, folder, cfg, {2: "p6p7", 1: "p6", 0: "none"}[tl])


@BACKBONE_REGISTRY.register()
def build_resnet_fpn_backbone(cfg, store, folder):
    return FPN(store, folder, cfg, "maxpool")
