"""Autograd wrappers around the C-ABI HIP kernels (ubteacher.hip).

Design notes (MI355X-first, not a translation of torch.nn):
  * activations are NHWC tensors - fp32, or bf16 under AMP (`act_dtype()`); weights live in the flat fp32 ParamStore arena;
  * parameter gradients are NOT returned to autograd: every backward accumulates straight into
    the flat gradient arena (`handle.g`), so one flat RCCL all-reduce + one SGD launch follow;
  * a per-device `hook` tensor (requires_grad) is threaded through every Function so backward
    runs even when the activation input comes from the frozen stem/res2;
  * FrozenBN is folded into the conv epilogue (scale/shift), ReLU and the residual add too.
"""
import os

import torch

from . import hip

_HOOKS = {}
# "fp32" (exact-f32 MFMA; the parity path), "bf16" (AMP: bf16 MFMA operands, fp32 accumulate) or "fp16" (AMP with the reference's own
# autocast element type: IEEE fp16 operands / activations / activation gradients on v_mfma_f32_32x32x16_f16 - same rate -, dynamic
# loss scaling with torch.cuda.amp.GradScaler's semantics, engine/trainer.py:195,207,424-426; the second build of the kernel library)
PRECISION = ["fp32"]


def set_precision(p):
    assert p in ("fp32", "bf16", "fp16")
    PRECISION[0] = p
    hip.set_h16("fp16" if p == "fp16" else "bf16")


def amp():
    """mixed precision: 16-bit MFMA operands and activations (either 16-bit type)"""
    return PRECISION[0] != "fp32"


def act_dtype():
    """Element type of activations (and their gradients) in HBM: 16-bit under AMP, like autocast's conv outputs."""
    return hip.h16_dtype() if amp() else torch.float32


_VERSION = [0]  # bumped by the optimizer step: invalidates cached dgrad weight images
STEP_GRAPH = [False]  # a trainer runs its step as a captured hipGraph (engine.trainer.run_step_graph): no work may cross the step boundary on a side stream
GRAD_SYNC = [None]  # utils.grad_sync.GradBuckets when data parallel: backward-overlapped all-reduce of the gradient arena


# Weight gradients are off the backward's critical path (nothing consumes them before the optimizer step): while the trainer's
# backward runs (`wgrad_side_stream(True)`), every wgrad launch goes to a side stream and overlaps the dgrad chain of the main one -
# the backbone's res4 / res5 layers fill 40-80 % of the chip per launch in either direction.  The arguments are recorded on the side
# stream (the caching allocator must not hand their blocks out again while it still reads them); `join_wgrad_stream()` makes the main
# stream wait (before the optimizer step / before a gradient bucket is all-reduced).
_WGRAD = {"on": False, "streams": {}, "pending": set(), "lane_of": {}, "next": 0, "deferred": []}


def defer_tower_wgrads():
    """UTV2_DEFER_TOWER_WGRAD=1 (experiment, round 4): the weight gradients of layers marked `defer_wgrad` (the paired FCOS towers -
    MFMA-bound, one 256-workgroup launch per depth that holds every CU) are not launched next to the tower's own dgrad chain - where
    they compete with another MFMA-bound full-chip kernel and starve the HBM-bound GroupNorm backward - but held back until the first
    weight gradient of the backbone / FPN backward, whose dgrad chain is mostly HBM-bound 1x1 layers.  Off in a data-parallel world
    (the gradient buckets are reported ready when a layer's backward returns)."""
    return os.environ.get("UTV2_DEFER_TOWER_WGRAD", "0") == "1" and GRAD_SYNC[0] is None


def wgrad_lanes():
    """Side streams the weight gradients are spread over (UTV2_WGRAD_LANES, default 2).  One lane leaves the END of the backward to a
    serial chain of weight-gradient launches that each fill a fraction of the chip (the res3 layers: 9 tiles x splits of a 4-wave
    kernel) - 0.75 ms per step with the main stream already done; a layer always uses the SAME lane (its launches accumulate into one
    gradient: order kept, results bit-identical), consecutive layers of the backward alternate."""
    try:
        return max(1, min(4, int(os.environ.get("UTV2_WGRAD_LANES", "2"))))
    except ValueError:
        return 2


def wgrad_side_stream(on):
    _WGRAD["on"] = bool(on) and os.environ.get("UTV2_WGRAD_STREAM", "1") != "0"


def _lanes(dev):
    lanes = _WGRAD["streams"].get(dev)
    if lanes is None:
        lanes = _WGRAD["streams"][dev] = [torch.cuda.Stream(dev) for _ in range(wgrad_lanes())]
    return lanes


def _wgrad_launch(fn, *tensors, key=None):
    """key: the layer whose gradient `fn` accumulates into (lane choice; None: lane 0)"""
    dev = tensors[0].device
    if not _WGRAD["on"] or dev.type != "cuda":
        return fn()
    if key is not None and getattr(key, "defer_wgrad", False) and defer_tower_wgrads():
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))            # the operands are ready from here on
        _WGRAD["deferred"].append((fn, tensors, key, ev))
        _WGRAD["ndeferred"] = _WGRAD.get("ndeferred", 0) + 1
        _WGRAD["pending"].add(dev)
        return None
    flush_deferred_wgrads()
    _wgrad_issue(fn, tensors, key, None)


def flush_deferred_wgrads():
    held, _WGRAD["deferred"] = _WGRAD["deferred"], []
    for fn, tensors, key, ev in held:
        _wgrad_issue(fn, tensors, key, ev)


def _wgrad_issue(fn, tensors, key, ready):
    dev = tensors[0].device
    lanes = _lanes(dev)
    lane = 0
    if key is not None and len(lanes) > 1:
        lane = _WGRAD["lane_of"].get(id(key))
        if lane is None:
            lane = _WGRAD["lane_of"][id(key)] = _WGRAD["next"] % len(lanes)
            _WGRAD["next"] += 1
    side = lanes[lane]
    if ready is not None:
        side.wait_event(ready)                             # a deferred launch: only what was enqueued when its layer's backward ran
    else:
        side.wait_stream(torch.cuda.current_stream(dev))   # the operands (and the zeroed gradient arena) are ready
    with torch.cuda.stream(side):
        if _fold_on():
            # the split-K tails (slab / bias reductions) of this lane's launches are recorded and run as one launch per <= 8 launches
            folds = _WGRAD.setdefault("folds", {})
            fl = folds.get(side)
            if fl is None:
                fl = folds[side] = hip.WgradFoldLane()
            prev, hip.WGRAD_FOLD[0] = hip.WGRAD_FOLD[0], fl
            try:
                fn()
            finally:
                hip.WGRAD_FOLD[0] = prev
        else:
            fn()
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    _WGRAD["pending"].add(dev)


def _fold_on():
    """UTV2_WGRAD_FOLD=1 (opt-in).  Measured (round 6, profiles/r06_fold_ab.txt, r06_fold_cap_ab.txt; interleaved runs on one box): 472 ->
    ~415 dispatches per FCOS step and NO change of the step time - FCOS 2+2 298.7-299.6 against 298.7 img/s, 4+4 347.7-347.9 against
    347.1-350.1, Faster-RCNN 2+2 276-279 against 280.8 (the last flush of a lane sits between the end of backward and the optimizer),
    at any flush period (2 / 4 / 8 launches): the per-layer tails were already hidden on the weight-gradient lanes."""
    return os.environ.get("UTV2_WGRAD_FOLD", "0") == "1"


def flush_wgrad_folds(dev=None):
    """run the recorded split-K tails of every weight-gradient lane (on its own stream): after this the gradient arena holds every
    launch issued so far - called before anything reads gradients (the join in front of the optimizer, a gradient bucket's all-reduce)"""
    for side, fl in _WGRAD.get("folds", {}).items():
        if fl.n and (dev is None or side.device == dev):
            with torch.cuda.stream(side):
                fl.flush()


def wgrad_stream_behind_main(dev):
    """the first wgrad side stream of `dev`, made to wait for everything the main stream and the other lanes have enqueued so far - or
    None when the weight gradients are not on side streams.  A consumer of parameter gradients that runs on its own stream (a bucket
    all-reduce) is issued from this stream: it then waits for the gradient kernels of ALL streams while the main stream's dgrad chain
    keeps running."""
    if not _WGRAD["on"] or dev.type != "cuda":
        return None
    lanes = _WGRAD["streams"].get(dev)
    if lanes is None:
        return None
    flush_wgrad_folds(dev)
    side = lanes[0]
    side.wait_stream(torch.cuda.current_stream(dev))
    for other in lanes[1:]:
        side.wait_stream(other)
    return side


def join_wgrad_stream():
    flush_deferred_wgrads()
    flush_wgrad_folds()
    for dev in list(_WGRAD["pending"]):
        for side in _WGRAD["streams"][dev]:
            torch.cuda.current_stream(dev).wait_stream(side)
    _WGRAD["pending"].clear()


def _sync_handles(layer, cs=None):
    from .utils.grad_sync import param_handles
    hs = param_handles(layer)
    if cs is not None:
        hs = hs + (list(cs) if isinstance(cs, (list, tuple)) else [cs])
    return hs


def hook(device):
    key = str(device)
    h = _HOOKS.get(key)
    if h is None:
        h = torch.zeros(1, device=device, requires_grad=True)
        _HOOKS[key] = h
    return h


def bump_version():
    _VERSION[0] += 1


class LevelMeta:
    """Geometry of a level-first [P, C] activation: N images on every FPN level."""

    def __init__(self, N, level_hw):
        self.N = N
        self.level_hw = [tuple(x) for x in level_hw]
        self.rows = []
        r = 0
        for h, w in self.level_hw:
            self.rows.append((r, r + N * h * w))
            r += N * h * w
        self.P = r
        self.seg_rows = [h * w for (h, w) in self.level_hw for _ in range(N)]  # (level, image) segments in memory order

    def level_view(self, t2d, l):
        r0, r1 = self.rows[l]
        h, w = self.level_hw[l]
        return t2d[r0:r1].view(self.N, h, w, t2d.shape[1])

    def alias_view(self, t2d, l):
        """Same memory as level_view() but NOT an autograd view of `t2d` (separate version counter): used
        as the destination of per-level producers so autograd's view+inplace bookkeeping stays out of it."""
        r0, r1 = self.rows[l]
        h, w = self.level_hw[l]
        C = t2d.shape[1]
        return torch.empty((0,), dtype=t2d.dtype, device=t2d.device).set_(
            t2d.untyped_storage(), t2d.storage_offset() + r0 * C, (self.N, h, w, C), (h * w * C, w * C, C, 1))


def _scale_ml_on():
    return os.environ.get("UTV2_SCALE_ML", "1") != "0"


class FlipBank:
    """dgrad weight images ([C][KH][KW][K] bf16, optionally times the FrozenBN multiplier) of every registered layer of one
    ParamStore, refreshed by ONE launch whenever the arena changes (utv2_weight_flip_transpose_bf16_batched)."""

    def __init__(self, store):
        self.store = store
        self.layers = []      # (layer, scale tensor or None)
        self.slot = {}        # id(layer) -> index
        self.dirty = False    # registrations since the tables were built
        self.version = None
        self.bank = None
        self.views = []
        self.single = {}      # id(layer) -> (version, tensor): per-layer images served before the layer is in the tables

    def _current(self):
        return (_VERSION[0], self.store.version, hip.H16[0])

    def _rebuild(self):
        import struct
        self.h16 = hip.H16[0]
        dev = self.store.flat.device
        offs, total = [], 0
        for layer, _ in self.layers:
            offs.append(total)
            total += (layer.dgrad_cout() * layer.k * layer.k * layer.cin + 7) // 8 * 8
        self.bank = torch.empty(total, dtype=hip.h16_dtype(), device=dev)
        self.scales = None
        rec = bytearray()
        self.nrec = 0
        for (layer, sc), o in zip(self.layers, offs):
            so = -1
            if sc is not None:
                if self.scales is None:
                    self.scales = torch.empty(0, dtype=torch.float32, device=dev).set_(sc.untyped_storage())
                assert sc.untyped_storage().data_ptr() == self.scales.untyped_storage().data_ptr()
                so = sc.storage_offset()
            G = layer.groups
            if G == 1:
                rec += struct.pack("<qqqiiiiii", layer.w.offset, o, so, layer.cout, layer.k, layer.k, layer.cin, layer.dgrad_cout(), 0)
                self.nrec += 1
            else:
                # a grouped layer is `groups` independent [Kg][k][k][cin] blocks of consecutive weight rows; their images are stacked:
                # row (g * cin + ci) of the [groups * cin][k * k * Kg] image serves input channel ci of group g
                assert sc is None and layer.dgrad_cout() == layer.cout
                Kg = layer.cout // G
                for gi in range(G):
                    rec += struct.pack("<qqqiiiiii", layer.w.offset + gi * Kg * layer.kred, o + gi * layer.cin * layer.k * layer.k * Kg, -1,
                                       Kg, layer.k, layer.k, layer.cin, Kg, 0)
                    self.nrec += 1
        self.table = torch.frombuffer(rec, dtype=torch.uint8).clone().to(dev)
        self.views = [self.bank[o:o + l.dgrad_cout() * l.k * l.k * l.cin].view(l.groups * l.cin, l.k * l.k * l.dgrad_cout() // l.groups)
                      for (l, _), o in zip(self.layers, offs)]
        self.dirty = False
        self.single.clear()

    def refresh_ahead(self):
        """Right after the optimizer step: rebuild every registered image for the NEW weights on a side stream, off the next backward's
        critical path (the batched launch used to be the first thing a backward waited for: 67 us with the chip idle at the forward /
        backward seam).  The first get() of the new version waits for the side stream's event instead of launching."""
        cur = self._current()
        if (self.bank is None or self.dirty or self.version == cur or self.h16 != hip.H16[0] or os.environ.get("UTV2_FLIP_AHEAD", "1") == "0"
                or STEP_GRAPH[0]):   # a one-step graph capture cannot hold work that the NEXT step joins: the first dgrad rebuilds the images
            return
        dev = self.store.flat.device
        if dev.type != "cuda":
            return
        side = self.__dict__.get("_side")
        if side is None:
            side = self._side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))      # the optimizer's writes
        with torch.cuda.stream(side):
            hip.weight_flip_transpose_bf16_batched(self.store.flat, self.scales, self.bank, self.table, self.nrec)
            self._ahead = torch.cuda.Event()
            self._ahead.record(side)
        self.version = cur

    def _join_ahead(self):
        ev = self.__dict__.pop("_ahead", None)
        if ev is not None:
            torch.cuda.current_stream(self.store.flat.device).wait_event(ev)

    def get(self, layer, scale):
        cur = self._current()
        if "_ahead" in self.__dict__:
            self._join_ahead()
        i = self.slot.get(id(layer))
        if i is None:
            self.slot[id(layer)] = len(self.layers)
            self.layers.append((layer, scale))
            self.dirty = True
        elif self.version == cur and not self.dirty:
            return self.views[i]
        elif self.version != cur and (i is not None):
            # first request after the arena changed: refresh every registered layer in one launch
            if self.dirty or self.bank is None or self.h16 != hip.H16[0]:
                self._rebuild()
            hip.weight_flip_transpose_bf16_batched(self.store.flat, self.scales, self.bank, self.table, self.nrec)
            self.version = cur
            return self.views[self.slot[id(layer)]]
        ent = self.single.get(id(layer))
        if ent is None or ent[0] != cur:
            if layer.groups > 1:
                Kg = layer.cout // layer.groups
                img = torch.cat([hip.weight_flip_transpose_bf16(layer.w.t[gi * Kg:(gi + 1) * Kg], Kg, layer.k, layer.k, layer.cin, None)
                                 for gi in range(layer.groups)])
            else:
                img = hip.weight_flip_transpose_bf16(layer.w.t, layer.cout, layer.k, layer.k, layer.cin, scale)
            if layer.dgrad_cout() != layer.cout:   # first request of a padded layer only: later ones come from the batched launch
                img = torch.nn.functional.pad(img.view(layer.cin, layer.k * layer.k, layer.cout), (0, layer.dgrad_cout() - layer.cout)) \
                    .reshape(layer.cin, -1).contiguous()
            ent = (cur, img)
            self.single[id(layer)] = ent
        return ent[1]


class Conv:
    """One convolution (+ optional folded FrozenBN or bias, ReLU, residual) bound to arena handles."""

    def __init__(self, w, cin, cout, k, stride=1, pad=0, bias=None, bn=None, relu=False, trainable=True,
                 kred=None, colscale=None, out_fp32=False, groups=1):
        self.w = w  # Handle, shape [cout, kred]
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        # groups > 1 (level-first k x k convs only): output channels [g*cout/groups, ...) read input channels [g*cin, (g+1)*cin) of a
        # [P, groups*cin] matrix; cin / kred are PER GROUP.  Two independent same-shape chains (the FCOS cls / bbox towers) run as ONE
        # launch per depth this way - half the launches, tile-quantisation remainders and split-K slabs of two separate convs.
        self.groups = groups
        assert groups == 1 or (bn is None and colscale is None and cout % groups == 0)
        self.bias = bias  # Handle [cout] or None
        self.bn = bn  # FrozenBN or None
        self.relu = relu
        self.trainable = trainable
        self.kred = kred if kred is not None else k * k * cin
        self.colscale = colscale  # (Handle scalar, ncols): Scale layer on the first ncols output channels
        self.out_fp32 = out_fp32  # AMP: keep this layer's output fp32 (loss-side head outputs, RoIAlign inputs)
        self.premask_input = False  # set by the model builder: the input is a fused bottleneck's ReLU output (see premask_on)
        self.bias_by_gn = False     # set by pair_conv_gn(): the GroupNorm that consumes this conv's output produces its bias gradient
        self.gn_cpg = 0             # set by pair_conv_gn(): channels per group of that GroupNorm
        self._gn_part = None        # (data_ptr of the last output, its statistics partials): picked up by that GroupNorm
        self.gnb_src = None         # set by chain_gn_conv(): the GroupNorm whose output is this conv's only input
        self.grad_premasked = False  # set by the model builder: every consumer of this conv's ReLU output has premask_input, i.e. applies
        #                              the mask (output > 0) in its own dgrad epilogue - no mask pass at the top of this conv's backward
        self._wt = None
        self._wt_version = -1

    def scale_shift(self):
        if self.bn is not None:
            return self.bn.scale, self.bn.shift
        return None, (self.bias.t if self.bias is not None else None)

    def wt(self):
        v = (_VERSION[0], self.w.store.version)
        if self._wt is None or self._wt_version != v:
            self._wt = hip.weight_flip_transpose(self.w.t, self.cout, self.k, self.k, self.cin)
            self._wt_version = v
        return self._wt

    def wt16(self, scale=None):
        """bf16 dgrad weight image; `scale` (the folded FrozenBN multiplier) is baked in per output channel.  Images of all
        layers of a model come from ONE batched launch per optimizer step (FlipBank); a layer's first request registers it
        and is served by a per-layer launch."""
        bank = getattr(self.w.store, "_flipbank", None)
        if bank is None:
            bank = self.w.store._flipbank = FlipBank(self.w.store)
        return bank.get(self, scale)

    def dgrad_cout(self):
        """output channels of the bf16 dgrad weight image: the multi-level 3x3 layers whose cout is no multiple of 32 (the 80-channel
        prediction convs) run their dgrad on a zero-padded bf16 copy of the gradient - the LDS-DMA kernel stages 32-channel chunks; the
        generic kernel took 318 us per launch on them against ~100"""
        if self.k == 3 and self.stride == 1 and self.cout % 32 and self.cout % 8 == 0 and self.cout > 32:
            return (self.cout + 31) // 32 * 32
        return self.cout

    def use_bf16(self):
        return amp() and self.cin % 8 == 0 and self.kred == self.k * self.k * self.cin

    def use_bf16_wgrad(self):
        return (amp() and self.cin % 8 == 0 and self.cout % 8 == 0 and self.k * self.k <= 16
                and self.kred == self.k * self.k * self.cin)

    def gn_stats_fused(self):
        """the GroupNorm (8 channels per group) that consumes this conv's bf16 output takes its statistics from partial sums this conv's
        epilogue leaves behind (utv2_conv2d_ml_fwd_bf16_g gn_part) instead of a pass over the tensor; UTV2_GN_STATS_FUSED=0: the pass"""
        return self.gn_cpg == 8 and self.cout % 8 == 0 and not self.relu and os.environ.get("UTV2_GN_STATS_FUSED", "1") != "0"

    def bias_grad_from_gn(self):
        """the bias gradient = column sums of this conv's dY, and dY is the dx the following GroupNorm's backward writes: its last pass
        sums the columns while it has them in registers (utv2_groupnorm_relu_seg_bwd_colsum) instead of a separate pass over dY next
        to the weight-gradient kernel"""
        return self.bias_by_gn and self.bias is not None and self.trainable and self.use_bf16_wgrad() and self.use_bf16_dgrad() \
            and os.environ.get("UTV2_GN_BIAS_GRAD", "1") != "0"

    def use_bf16_dgrad(self):
        return amp() and self.cout % 8 == 0

    def __call__(self, x, residual=None, out=None, colscale_handle=None, meta=None):
        """x: NHWC tensor, or a level-first [P, C] matrix with `meta` (one launch for all levels; k x k
        'same' convs only).  colscale_handle: a Handle, or one Handle per level when `meta` is given."""
        if torch.is_grad_enabled() and self.trainable:
            return _ConvFn.apply(x, residual, hook(x.device), self, (out, getattr(x, "_utv2_fanin", None), getattr(x, "_utv2_pair", None)),
                                 colscale_handle, meta)
        return self._forward(x, residual, out, colscale_handle, meta)

    def _forward(self, x, residual, out, cs, meta, relu_bits=None):
        sc, sh = self.scale_shift()
        b16 = self.use_bf16()
        w = self.w.store.bf16(self.w) if b16 else self.w.t
        kw = {}
        if b16:  # output element type: the destination's if one is given, else bf16 unless this layer feeds fp32 consumers
            kw["out_dtype"] = out.dtype if out is not None else (torch.float32 if self.out_fp32 else hip.h16_dtype())
        else:
            assert x.dtype == torch.float32, "the fp32 conv kernels take fp32 activations (layer cin=%d)" % self.cin
        if not b16 and not x.is_contiguous():
            x = x.contiguous()      # a column slice of a paired tower's output: the fp32 kernels take dense matrices
        if self.groups > 1:
            assert meta is not None and self.k > 1 and residual is None and out is None and cs is None
        if meta is not None and self.k > 1:
            assert self.stride == 1 and self.pad == (self.k - 1) // 2
            if b16:
                part = None
                if self.gn_stats_fused() and out is None and residual is None and kw["out_dtype"] == hip.h16_dtype() and self.cin % 32 == 0:
                    part = hip.gn_part_buffer(x.shape[0], self.cout, x.device)
                y = hip.conv2d_ml_fwd_bf16(x, w, meta.level_hw, meta.N, scale=sc, bias=sh, residual=residual, k=self.k, pad=self.pad,
                                           relu=self.relu, out=out, groups=self.groups, gn_part=part, **kw)
                self._gn_part = (y.data_ptr(), part) if part is not None else None
            elif self.groups > 1:   # exact-f32 mode: one dense conv per group (the fp32 kernel has no grouped form)
                Kg = self.cout // self.groups
                y = torch.cat([hip.conv2d_ml_fwd(x[:, gi * self.cin:(gi + 1) * self.cin].contiguous(), w[gi * Kg:(gi + 1) * Kg], meta.level_hw,
                                                 meta.N, scale=None, bias=None if sh is None else sh[gi * Kg:(gi + 1) * Kg], k=self.k,
                                                 pad=self.pad, relu=self.relu) for gi in range(self.groups)], dim=1)
            else:
                y = hip.conv2d_ml_fwd(x, w, meta.level_hw, meta.N, scale=sc, bias=sh, residual=residual, k=self.k, pad=self.pad,
                                      relu=self.relu, out=out, **kw)
        elif meta is not None:  # 1x1 on a level-first matrix: plain GEMM rows
            fn = hip.conv2d_fwd_bf16 if b16 else hip.conv2d_fwd
            y = fn(x.view(1, x.shape[0], 1, x.shape[1]), w, scale=sc, bias=sh,
                   residual=None if residual is None else residual.view(1, x.shape[0], 1, -1), relu=self.relu,
                   out=None if out is None else out.view(1, x.shape[0], 1, -1), **kw).view(x.shape[0], self.cout)
        else:
            fn = hip.conv2d_fwd_bf16 if b16 else hip.conv2d_fwd
            if relu_bits is not None:
                kw["relu_bits"] = relu_bits
            y = fn(x, w, scale=sc, bias=sh, residual=residual, stride=self.stride, pad=self.pad, relu=self.relu, kh=self.k,
                   kw=self.k, out=out, **kw)
        if cs is not None:
            assert y.dtype == torch.float32
            if meta is not None and _scale_ml_on() and y.is_contiguous():
                hip.scale_cols_ml(y, meta.rows, self.colscale, [h.t for h in cs])      # all levels in one launch
            elif meta is not None:
                for l, h in enumerate(cs):
                    r0, r1 = meta.rows[l]
                    hip.scale_cols(y[r0:r1], self.colscale, h.t)
            else:
                hip.scale_cols(y.view(-1, self.cout), self.colscale, cs.t)
        return y


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, hk, layer, out_holder, cs, meta):
        out = out_holder[0]  # optional destination view (kept out of autograd's sight on purpose)
        y = layer._forward(x, residual, out, cs, meta)
        ctx.layer = layer
        ctx.cs = cs
        ctx.meta = meta
        ctx.has_res = residual is not None
        ctx.x_bits = _bits_of(x) if (layer.premask_input and meta is None) else None   # x: a fused bottleneck's ReLU output
        ctx.handoff = getattr(x, "_utv2_handoff", None) if (layer.premask_input and meta is None and grad_handoff_on()) else None
        if GRAD_SYNC[0] is not None:
            GRAD_SYNC[0].on_forward(_sync_handles(layer, cs))
        ctx.save_for_backward(x, y if (layer.relu or cs is not None) else None)
        ctx.xshape = tuple(x.shape)
        ctx.gnb = None      # (GroupNorm layer, its input, the ReLU bit plane of x): x is that GroupNorm's output (chain_gn_conv, gn_bwd_fuse_on)
        src = layer.gnb_src
        if src is not None and src._gnb_fwd is not None:
            gf, src._gnb_fwd = src._gnb_fwd, None
            if gf[0] == x.data_ptr() and x.is_contiguous() and tuple(gf[1].shape) == tuple(x.shape) and residual is None:
                ctx.gnb = (src, gf[1], gf[2])
        ctx.fanin = out_holder[1] if len(out_holder) > 1 else None   # FanIn: another consumer's gradient of x, added in this dgrad's epilogue
        ctx.pair = out_holder[2] if len(out_holder) > 2 else None    # (ColPair, half): x is a column half of a paired tower's output
        return y

    @staticmethod
    def backward(ctx, dy):
        layer = ctx.layer
        x, y = ctx.saved_tensors
        dy = dy.contiguous()
        if not x.is_contiguous() and not (layer.use_bf16_wgrad() and layer.use_bf16_dgrad()):
            x = x.contiguous()   # a column slice (paired towers) in exact-f32 mode: the fp32 kernels take dense matrices
        sc, _ = layer.scale_shift()
        meta = ctx.meta
        gp_ready = None
        if (ctx.cs is not None and meta is not None and _scale_ml_on() and y.is_contiguous() and dy.dtype == torch.float32
                and layer.k > 1 and ctx.needs_input_grad[0] and layer.use_bf16_dgrad() and layer.use_bf16_wgrad()
                and layer.dgrad_cout() != layer.cout and not layer.relu and sc is None and not ctx.has_res
                and os.environ.get("UTV2_SCALE_BWD_PAD16", "1") != "0"):
            # the bbox prediction conv (fcos.py:338-364): Scale backward, conversion and zero padding of the gradient in ONE pass, out of
            # place - no clone of the incoming gradient, no in-place scale, no separate pad pass (568 -> 224 MB on the student batch)
            gp_ready = hip.scale_cols_bwd_ml_pad16(dy, y, meta.rows, layer.colscale, [h.t for h in ctx.cs], [h.g for h in ctx.cs],
                                                   layer.dgrad_cout())
        elif ctx.cs is not None:
            g = dy.clone()
            if meta is not None and _scale_ml_on() and y.is_contiguous():
                # all levels in two launches (it was four per level, each waiting for the one before at the forward / backward seam)
                hip.scale_cols_bwd_ml(g, y, meta.rows, layer.colscale, [h.t for h in ctx.cs], [h.g for h in ctx.cs])
            elif meta is not None:
                for l, h in enumerate(ctx.cs):
                    r0, r1 = meta.rows[l]
                    dsum = hip.scale_cols_bwd(g[r0:r1], y[r0:r1], layer.colscale, h.t)
                    h.g.add_(dsum / h.t.view(-1))
            else:
                dsum = hip.scale_cols_bwd(g.view(-1, layer.cout), y.view(-1, layer.cout), layer.colscale, ctx.cs.t)
                ctx.cs.g.add_(dsum / ctx.cs.t.view(-1))
            dy = g
        gres = None
        d16 = layer.use_bf16_dgrad()
        # AMP + FrozenBN: the BN multiplier is folded into the dgrad weight image and the wgrad slab reduction, so the
        # only elementwise backward pass left is the ReLU mask (none at all for the ReLU-less shortcut convs)
        fold = sc is not None and d16 and layer.use_bf16_wgrad() and layer.bias is None
        wsc = sc if fold else None
        relu = layer.relu and not (layer.grad_premasked and premask_on())   # premasked: dy arrives with the ReLU mask applied
        if fold:
            g = hip.relu_bwd_scale(dy, y, None) if relu else dy
            gres = g if ctx.has_res else None
        elif ctx.has_res:
            gm = hip.relu_bwd_scale(dy, y if relu else None, None) if relu else dy
            gres = gm
            g = hip.relu_bwd_scale(gm, None, sc) if sc is not None else gm
        else:
            if relu or sc is not None:
                g = hip.relu_bwd_scale(dy, y if relu else None, sc)
            else:
                g = dy
        dx = None
        bias_done = False
        if meta is not None and layer.k > 1:
            G = layer.groups
            Kg = layer.cout // G
            if ctx.needs_input_grad[0]:
                if d16:
                    gp = gp_ready if gp_ready is not None else (g if layer.dgrad_cout() == layer.cout else hip.pad_cols_bf16(g, layer.dgrad_cout()))
                    other = ctx.fanin.take() if ctx.fanin is not None else None
                    if other is not None and (other.dtype != x.dtype or tuple(other.shape) != tuple(x.shape)):
                        raise RuntimeError("FanIn: stored gradient does not match the conv input")
                    # x a column half of a paired tower's output: the gradient is written straight into its half of ONE [P, 2C] buffer
                    # (row pitch 2C), which ColPair's backward hands on as the gradient of the whole matrix - no cat pass
                    dest = ctx.pair[0].grad_half(ctx.pair[1], x) if (ctx.pair is not None and other is None) else None
                    wt = layer.wt16(wsc)
                    if (ctx.gnb is not None and other is None and dest is None and x.dtype == hip.h16_dtype()
                            and hip.gnb_eligible(gp, wt, layer.k, layer.k - 1 - layer.pad, G)):
                        src, gx, bits = ctx.gnb
                        part = hip.gnb_part_buffer(x.shape[0], x.shape[1], x.device)
                        dx = hip.conv2d_ml_fwd_bf16(gp, wt, meta.level_hw, meta.N, k=layer.k, pad=layer.k - 1 - layer.pad, out_dtype=x.dtype,
                                                    groups=G, gnb=(bits, gx, part))
                        src._gnb_bwd = (dx.data_ptr(), part)
                    else:
                        dx = hip.conv2d_ml_fwd_bf16(gp, wt, meta.level_hw, meta.N, k=layer.k, pad=layer.k - 1 - layer.pad,
                                                    residual=other, out_dtype=x.dtype, out=dest, groups=G)
                elif G > 1:
                    dx = torch.cat([hip.conv2d_ml_dgrad(g[:, gi * Kg:(gi + 1) * Kg].contiguous(),
                                                        hip.weight_flip_transpose(layer.w.t[gi * Kg:(gi + 1) * Kg], Kg, layer.k, layer.k, layer.cin),
                                                        meta.level_hw, meta.N, layer.k, layer.pad) for gi in range(G)], dim=1)
                else:
                    dx = hip.conv2d_ml_dgrad(g, layer.wt(), meta.level_hw, meta.N, layer.k, layer.pad)
            if layer.use_bf16_wgrad():
                # the 80-channel prediction convs: the weight gradient reads the zero-padded bf16 copy of the gradient the dgrad was given
                # (its leading `cout` columns; the same RNE-rounded values the kernel would make of the fp32 gradient while staging it -
                # identical results from 60 % of the bytes and no conversion work in the staging loop)
                gw = gp[:, :layer.cout] if (d16 and ctx.needs_input_grad[0] and gp is not g) else g
                _wgrad_launch(lambda: hip.conv2d_wgrad_bf16(
                    x, gw, layer.w.g, hip.rowinfo_ml(meta.N, meta.level_hw, layer.pad, layer.k, x.device), layer.cin, layer.k, layer.k,
                    accumulate=True, db=layer.bias.g if (layer.bias is not None and not layer.bias_grad_from_gn()) else None, rowscale=wsc,
                    groups=G, x_pitch=x.stride(0)), x, gw, key=layer)
                bias_done = True
            elif G > 1:
                for gi in range(G):
                    xg, gg = x[:, gi * layer.cin:(gi + 1) * layer.cin].contiguous(), g[:, gi * Kg:(gi + 1) * Kg].contiguous()
                    _wgrad_launch(lambda xg=xg, gg=gg, gi=gi: hip.conv2d_ml_wgrad(xg, gg, layer.w.g[gi * Kg:(gi + 1) * Kg], meta.level_hw, meta.N,
                                                                                   layer.k, layer.pad, accumulate=True), xg, gg, key=layer)
            else:
                _wgrad_launch(lambda: hip.conv2d_ml_wgrad(x, g, layer.w.g, meta.level_hw, meta.N, layer.k, layer.pad, accumulate=True), x, g, key=layer)
        else:
            x4 = x.view(1, x.shape[0], 1, x.shape[1]) if meta is not None else x
            g4 = g.view(1, g.shape[0], 1, g.shape[1]) if meta is not None else g
            if ctx.needs_input_grad[0]:
                # premask_input: x is the ReLU output of a fused bottleneck that expects its incoming gradient already masked by x > 0
                pm = x4 if (layer.premask_input and premask_on()) else None
                if d16 and layer.k == 1 and layer.stride == 2 and layer.pad == 0:
                    # only the even input pixels receive gradient: a plain GEMM on the compact grid + a zero-interleave,
                    # instead of a 4x larger dilated gather that is 3/4 masked
                    dxc = hip.conv2d_fwd_bf16(g4, layer.wt16(wsc), out_dtype=x.dtype)
                    dx = hip.zero_interleave2x(dxc, x4.shape[1], x4.shape[2], mask=pm)
                elif (d16 and layer.k == 3 and layer.stride == 2 and layer.pad == 1 and g4.shape[-1] % 32 == 0
                      and x4.shape[0] * x4.shape[1] * x4.shape[2] <= 65536):
                    # small stride-2 3x3 layers (FPN p6 / p7): the dilated-gather dgrad only exists in the generic kernel, whose long K loop
                    # on a 52-workgroup grid sat on the backward's critical path (78 us alone, ~250 us in the step; -0.12 ms / step); the gradient
                    # zero-interleaved to the input grid is a plain stride-1 conv for the LDS-DMA kernel (4x the MACs of a tiny layer)
                    gd = hip.zero_interleave2x(g4.contiguous(), x4.shape[1], x4.shape[2])
                    dx = hip.conv2d_fwd_bf16(gd, layer.wt16(wsc), pad=1, kh=3, kw=3, out_dtype=x.dtype, mask=pm)
                elif d16:
                    gd = g4
                    if layer.dgrad_cout() != layer.cout:
                        gd = hip.pad_cols_bf16(g4.reshape(-1, layer.cout), layer.dgrad_cout()).view(g4.shape[:-1] + (layer.dgrad_cout(),))
                    pb = ctx.x_bits if (pm is not None and relu_bits_on() and x.dtype == hip.h16_dtype()) else None
                    BITS_STATS["reads"] += pb is not None
                    dx = hip.conv2d_dgrad_bf16(gd, layer.wt16(wsc), tuple(x4.shape), layer.stride, layer.pad, layer.k, layer.k,
                                               out_dtype=x.dtype, mask=None if pb is not None else pm, mask_bits=pb)
                else:
                    dx = hip.conv2d_dgrad(g4, layer.wt(), tuple(x4.shape), layer.stride, layer.pad, layer.k, layer.k)
                    if pm is not None:
                        dx = hip.relu_bwd_scale(dx, x4, None)
                if meta is not None:
                    dx = dx.view(x.shape)
                elif ctx.handoff is not None and pm is not None and dx.dtype == hip.h16_dtype() and ctx.handoff.park(dx):
                    dx = None    # the next stage's first block adds it where it makes its own gradient of x (GradHandoff)
            if layer.use_bf16_wgrad():
                n_, h_, w_, _ = x4.shape
                ri = hip.rowinfo_nhwc(n_, h_, w_, g4.shape[1], g4.shape[2], layer.stride, layer.pad, layer.k, layer.k, x.device)
                _wgrad_launch(lambda: hip.conv2d_wgrad_bf16(
                    x4, g4.reshape(-1, layer.cout), layer.w.g, ri, layer.cin, layer.k, layer.k, accumulate=True,
                    db=layer.bias.g if (layer.bias is not None and not layer.bias_grad_from_gn()) else None, rowscale=wsc), x4, g4, key=layer)
                bias_done = True
            else:
                _wgrad_launch(lambda: hip.conv2d_wgrad(x4, g4, layer.w.g, layer.stride, layer.pad, layer.k, layer.k, accumulate=True), x4, g4, key=layer)
        if layer.bias is not None and not bias_done:
            _wgrad_launch(lambda: hip.colsum(g.view(-1, layer.cout), layer.bias.g, accumulate=True), g, key=layer)
        if GRAD_SYNC[0] is not None:
            GRAD_SYNC[0].on_backward_done(_sync_handles(layer, ctx.cs))
        return dx, gres, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# AMP ResNet bottleneck as ONE autograd node: the backward chains the three (four) convs explicitly so that the ReLU
# backward of conv1 / conv2 rides in the epilogue of the dgrad that produces their gradient (mask), the identity / shortcut
# gradient is added in the epilogue of conv1's dgrad (residual) and a stride-2 block interleaves zeros once for both
# branches.  Left as elementwise work: one ReLU-mask pass over the block output's gradient.
def _wgrad16(layer, x4, g4):
    n_, h_, w_, _ = x4.shape
    ri = hip.rowinfo_nhwc(n_, h_, w_, g4.shape[1], g4.shape[2], layer.stride, layer.pad, layer.k, layer.k, x4.device)
    _wgrad_launch(lambda: hip.conv2d_wgrad_bf16(x4, g4.reshape(-1, layer.cout), layer.w.g, ri, layer.cin, layer.k, layer.k, accumulate=True,
                                                rowscale=layer.bn.scale), x4, g4, key=layer)


def _dgrad16(layer, g4, in_shape, mask=None, residual=None, post_mask=None, mask_bits=None, post_mask_bits=None):
    if mask_bits is not None:
        mask = None
    if post_mask_bits is not None:
        post_mask = None
    BITS_STATS["reads"] += (mask_bits is not None) + (post_mask_bits is not None)
    return hip.conv2d_dgrad_bf16(g4, layer.wt16(layer.bn.scale), tuple(in_shape), layer.stride, layer.pad, layer.k, layer.k,
                                 out_dtype=hip.h16_dtype(), mask=mask, residual=residual, post_mask=post_mask, mask_bits=mask_bits,
                                 post_mask_bits=post_mask_bits)


class GradHandoff:
    """Gradient fan-in of a backbone stage output without autograd's add pass: the activation feeds the next stage's first block and the
    FPN lateral; the lateral was built later, so its backward runs first - it parks its (already masked) input gradient here and reports
    none, and the block adds it where it makes its own input gradient (the zero-interleave of a stride-2 block, a dgrad epilogue's
    residual otherwise).  Either side falls back to the plain path when the other has not / has already run; FanIn.check() after
    backward() fails loudly if a parked gradient was never picked up.  UTV2_GRAD_HANDOFF=0: autograd sums (A/B; the sum is the same
    one rounding later: the parked 16-bit gradient is added in fp32 before the block's gradient is rounded)."""
    _live = []

    def __init__(self):
        self.grad = None
        self.closed = False
        if len(GradHandoff._live) >= 64:
            del GradHandoff._live[:32]
        GradHandoff._live.append(self)

    def park(self, g):
        if self.closed or self.grad is not None:
            return False
        self.grad = g
        return True

    def take(self):
        self.closed = True
        g, self.grad = self.grad, None
        return g


def grad_handoff_on():
    return premask_on() and os.environ.get("UTV2_GRAD_HANDOFF", "1") != "0"


def relu_bits_on():
    """The fused bottlenecks keep the sign of their three ReLU outputs as bit planes (written by the forward conv epilogues, a sixteenth of
    the activation's bytes) and their dgrads read those instead of the 16-bit activations (22 C -> 16.4 C bytes per pixel of a block's
    dgrad chain).  UTV2_RELU_BITS=0: the dgrads re-read the activations (A/B runs; identical results)."""
    return os.environ.get("UTV2_RELU_BITS", "1") != "0"


BITS_STATS = {"planes": 0, "reads": 0}     # bit planes written / read in place of a 16-bit tensor since import (tests)


def _bits_of(t):
    """the bit plane `t > 0` its producer attached to a ReLU output (None: unknown, read the tensor)"""
    b = getattr(t, "_utv2_relu_bits", None)
    if b is not None and (b[0] != t.data_ptr() or b[1] != t._version):
        return None     # another tensor's plane, or the tensor was written since
    return None if b is None else b[2]


def premask_on():
    """Gradients that flow into a bottleneck's ReLU output are masked where they are produced (the next block's conv1 dgrad epilogue,
    the FPN lateral's dgrad epilogue) instead of by a separate pass at the top of the block's backward.  UTV2_PREMASK=0: the separate
    pass (A/B runs; identical results - a 0/1 mask distributes over the sum of the contributions)."""
    return os.environ.get("UTV2_PREMASK", "1") != "0"


class _BottleneckFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, hk, block):
        c1, c2, c3, cs = block.conv1, block.conv2, block.conv3, block.shortcut
        res = cs._forward(x, None, None, None, None) if cs is not None else x
        b1 = b2 = b3 = None
        if relu_bits_on() and all(c.relu and c.cout % 8 == 0 for c in (c1, c2, c3)):
            planes, (n_, h_, w_, _) = [], x.shape
            for c in (c1, c2, c3):
                h_, w_ = hip.conv_out_size(h_, c.k, c.stride, c.pad), hip.conv_out_size(w_, c.k, c.stride, c.pad)
                planes.append(hip.relu_bits_buffer((n_, h_, w_, c.cout), x.device))
            b1, b2, b3 = planes
            BITS_STATS["planes"] += 3
        y1 = c1._forward(x, None, None, None, None, relu_bits=b1)
        y2 = c2._forward(y1, None, None, None, None, relu_bits=b2)
        y3 = c3._forward(y2, res, None, None, None, relu_bits=b3)
        ctx.block = block
        ctx.handoff = None
        if ctx.needs_input_grad[0] and grad_handoff_on() and getattr(x, "_utv2_handoff", None) is None:
            ctx.handoff = x._utv2_handoff = GradHandoff()   # a premasked consumer of x built later (FPN lateral) parks its gradient here
        ctx.bits = (_bits_of(x), b1, b2)
        if b3 is not None:
            y3._utv2_relu_bits = (y3.data_ptr(), y3._version, b3)    # for whoever back-propagates into this ReLU output
        ctx.save_for_backward(x, y1, y2, y3)
        if GRAD_SYNC[0] is not None:
            for l in (c1, c2, c3, cs):
                if l is not None:
                    GRAD_SYNC[0].on_forward(_sync_handles(l))
        return y3

    @staticmethod
    def backward(ctx, dy):
        block = ctx.block
        c1, c2, c3, cs = block.conv1, block.conv2, block.conv3, block.shortcut
        x, y1, y2, y3 = ctx.saved_tensors
        pre = premask_on()
        if pre and block.grad_premasked:    # every producer of dy applied this block's output ReLU mask (y3 > 0) already
            gm = dy.contiguous()
        else:
            gm = hip.relu_bwd_scale(dy.contiguous(), y3, None)     # gradient at conv3's BN output == at the residual input
        pm = x if (pre and block.input_relu) else None              # x is a ReLU output whose producer expects a masked gradient
        bx, b1, b2 = ctx.bits
        if pm is None:
            bx = None
        _wgrad16(c3, y2, gm)
        g2 = _dgrad16(c3, gm, y2.shape, mask=y2, mask_bits=b2)      # ... through ReLU(conv2): at conv2's BN output
        _wgrad16(c2, y1, g2)
        g1 = _dgrad16(c2, g2, y1.shape, mask=y1, mask_bits=b1)
        _wgrad16(c1, x, g1)
        if cs is not None:
            _wgrad16(cs, x, gm)
        dx = None
        if ctx.needs_input_grad[0]:
            parked = ctx.handoff.take() if ctx.handoff is not None else None   # the FPN lateral's gradient of x (masked by x > 0 there)
            if parked is not None and (parked.dtype != hip.h16_dtype() or tuple(parked.shape) != tuple(x.shape)):
                raise RuntimeError("GradHandoff: parked gradient of another shape / type")
            if cs is None:
                dx = _dgrad16(c1, g1, x.shape, residual=gm, post_mask=pm, post_mask_bits=bx)   # identity branch added in the epilogue
                if parked is not None:
                    dx = dx + parked
            elif c1.stride == 2:
                c = hip.conv2d_fwd_bf16(gm, cs.wt16(cs.bn.scale), out_dtype=hip.h16_dtype())       # compact grids: only the
                c = hip.conv2d_fwd_bf16(g1, c1.wt16(c1.bn.scale), residual=c, out_dtype=hip.h16_dtype())   # even pixels get gradient
                dx = hip.zero_interleave2x(c, x.shape[1], x.shape[2], mask=pm if bx is None else None, mask_bits=bx if pm is not None else None,
                                           add=parked)
                BITS_STATS["reads"] += (bx is not None and pm is not None)
            else:
                d = _dgrad16(cs, gm, x.shape, residual=parked)
                dx = _dgrad16(c1, g1, x.shape, residual=d, post_mask=pm, post_mask_bits=bx)
        if GRAD_SYNC[0] is not None:
            for l in (c3, c2, c1, cs):
                if l is not None:
                    GRAD_SYNC[0].on_backward_done(_sync_handles(l))
        return dx, None, None


def _fused_frozen_block_on():
    return os.environ.get("UTV2_FUSED_FROZEN_BLOCK", "1") != "0"


def bottleneck(block, x):
    """one fused autograd node when every conv of the block runs on the bf16 kernels, else the per-conv graph"""
    convs = [c for c in (block.conv1, block.conv2, block.conv3, block.shortcut) if c is not None]
    fused = (amp() and torch.is_grad_enabled() and x.dtype == hip.h16_dtype()
             and all(c.trainable and c.bn is not None and c.bias is None and c.use_bf16() and c.use_bf16_wgrad() and c.use_bf16_dgrad()
                     for c in convs)
             and block.conv1.stride in (1, 2) and block.conv1.k == 1 and block.conv3.k == 1)
    if fused:
        return _BottleneckFn.apply(x, hook(x.device), block)
    if (amp() and x.dtype == hip.h16_dtype() and x.is_contiguous() and _fused_frozen_block_on()
            and (not torch.is_grad_enabled() or not any(c.trainable for c in convs)) and not x.requires_grad
            and all(c.stride == 1 for c in convs) and block.conv1.k == 1 and block.conv2.k == 3 and block.conv3.k == 1
            and (block.shortcut is None or block.shortcut.k == 1) and block.conv3.cout == 256
            and all(c.bn is not None and c.bias is None and c.use_bf16() for c in convs)
            and x.shape[1] * x.shape[2] * 256 < 2 ** 31           # the kernel's 32-bit element offsets inside one image
            and hip.bottleneck_supported(block.conv1.cin, block.conv1.cout, block.shortcut is not None)):
        # a frozen stride-1 block (res2 under FREEZE_AT 2; the teacher's too): nothing is kept for a backward, so its convs run as one
        # kernel with the 64-channel intermediates in LDS (csrc/bottleneck.hip)
        st = block.conv1.w.store
        (s1, b1), (s2, b2), (s3, b3) = block.conv1.scale_shift(), block.conv2.scale_shift(), block.conv3.scale_shift()
        kw = {}
        if block.shortcut is not None:
            ssc, bsc = block.shortcut.scale_shift()
            kw = dict(wsc=st.bf16(block.shortcut.w), ssc=ssc, bsc=bsc)
        return hip.bottleneck_fwd_bf16(x, st.bf16(block.conv1.w), st.bf16(block.conv2.w), st.bf16(block.conv3.w), s1, b1, s2, b2, s3, b3, **kw)
    sc = block.shortcut(x) if block.shortcut is not None else x
    out = block.conv1(x)
    out = block.conv2(out)
    return block.conv3(out, residual=sc)


class _StemFn(torch.autograd.Function):
    """A TRAINABLE ResNet stem (MODEL.BACKBONE.FREEZE_AT < 1; D2 BasicStem: 7x7 stride-2 conv + FrozenBN + ReLU + 3x3 stride-2 max pool)
    on the fp32 NHWC4 image as one autograd node: the backward routes the pooled gradient to the first maximum of every window and
    through the ReLU in one gather kernel (utv2_maxpool3x3s2_bwd_nhwc), then takes the weight gradient with the exact-f32 kernel (K
    reduction 7 * 7 * 4 = 196: a small layer; its 16-bit activations are widened once).  The image needs no gradient."""

    @staticmethod
    def forward(ctx, x4, hk, layer):
        sc, sh = layer.scale_shift()
        y = hip.conv2d_stem_fwd(x4, layer.w.t, sc, sh, layer.stride, layer.pad, layer.k, layer.k, True, act_dtype())
        ctx.layer = layer
        ctx.save_for_backward(x4, y)
        if GRAD_SYNC[0] is not None:
            GRAD_SYNC[0].on_forward(_sync_handles(layer))
        return hip.maxpool3x3s2(y)

    @staticmethod
    def backward(ctx, dpool):
        layer = ctx.layer
        x4, y = ctx.saved_tensors
        g = hip.maxpool3x3s2_bwd(y, dpool.contiguous(), relu=True)                 # d(conv * scale + shift), ReLU and pool undone
        g = g.float() * layer.bn.scale                                              # through the frozen BN's scale
        K, kred = layer.cout, layer.k * layer.k * layer.cin

        def wgrad():
            dw = torch.empty((K, kred), dtype=torch.float32, device=g.device)
            hip.conv2d_wgrad(x4, g, dw, layer.stride, layer.pad, layer.k, layer.k, accumulate=False)
            layer.w.g.view(K, -1)[:, :kred] += dw                                   # rows are padded to 208 in the arena
        _wgrad_launch(wgrad, x4, g, key=layer)
        if GRAD_SYNC[0] is not None:
            GRAD_SYNC[0].on_backward_done(_sync_handles(layer))
        return None, None, None


def stem(layer, x4):
    return _StemFn.apply(x4, hook(x4.device), layer)


class ColPair:
    """The two column halves of a [P, 2C] matrix (the paired towers' output: cls | bbox) as separate autograd tensors for their two
    consumers, without a cat pass in the backward: each consumer's dgrad writes its half of ONE [P, 2C] gradient buffer (row pitch 2C,
    `Conv` asks grad_half() for the destination) and the split's backward recognises the two halves and returns the buffer itself."""

    def __init__(self):
        self.buf = None

    def grad_half(self, idx, x):
        if self.buf is None:
            self.buf = torch.empty((x.shape[0], 2 * x.shape[1]), dtype=x.dtype, device=x.device)
        C = x.shape[1]
        return self.buf[:, idx * C:(idx + 1) * C]

    def take(self, g0, g1):
        b, self.buf = self.buf, None
        if (b is not None and g0 is not None and g1 is not None and g0.dtype == b.dtype and g0.data_ptr() == b.data_ptr()
                and g1.data_ptr() == b.data_ptr() + g0.shape[1] * b.element_size() and g0.stride(0) == b.stride(0) == g1.stride(0)):
            return b
        return None


class _SplitColsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, pair):
        C = t.shape[1] // 2
        ctx.pair, ctx.C = pair, C
        ctx.meta_ = (t.dtype, t.device, t.shape[0])
        return t[:, :C], t[:, C:]

    @staticmethod
    def backward(ctx, g0, g1):
        b = ctx.pair.take(g0, g1)
        if b is not None:
            return b, None
        dt, dev, P = ctx.meta_
        z = None
        if g0 is None or g1 is None:
            z = torch.zeros((P, ctx.C), dtype=dt, device=dev)
        return torch.cat((g0 if g0 is not None else z, g1 if g1 is not None else z), dim=1), None


def split_cols(t):
    """(left half, right half) column views of t [P, 2C]; under autograd the halves carry `_utv2_pair` so that the level-first convs that
    consume them write their input gradients into one shared buffer (ColPair)"""
    C = t.shape[1] // 2
    if not (torch.is_grad_enabled() and t.requires_grad):
        return t[:, :C], t[:, C:]
    pair = ColPair()
    a, b = _SplitColsFn.apply(t, pair)
    a._utv2_pair, b._utv2_pair = (pair, 0), (pair, 1)
    return a, b


def pair_conv_gn(conv, gn):
    """declare that `gn` normalises exactly the output of `conv` (conv -> GN -> ReLU, fcos.py:263-264): the GroupNorm's backward then
    also produces the conv's bias gradient (Conv.bias_grad_from_gn)"""
    conv.bias_by_gn = True
    conv.gn_cpg = conv.cout // gn.groups
    gn.bias_conv = conv
    return conv, gn


def gn_bwd_fuse_on():
    """(UTV2_GN_BWD_FUSE=0 switches it off.)  Inside a conv -> GN -> ReLU -> conv chain (the FCOS towers) the SECOND conv's dgrad applies the ReLU
    mask (a bit plane the GroupNorm's apply pass writes) and leaves GroupNorm backward's per-channel partial sums while its rows are in
    registers (hip.conv2d_ml_fwd_bf16 gnb) - the backward's gn_bwd_partial pass (dy and x read again from HBM) disappears.  The sums
    are the same quantities added in another order: gradients agree with the unfused path to fp32 rounding, not bit for bit."""
    return os.environ.get("UTV2_GN_BWD_FUSE", "1") != "0"


def chain_gn_conv(gn, conv):
    """declare that `conv` (level-first, 3x3) consumes exactly the output of `gn` and nothing else does (tower layer i -> i + 1)"""
    gn.next_conv = conv
    conv.gnb_src = gn
    return conv


class GroupNormReLU:
    def __init__(self, gamma, beta, groups=32, eps=1e-5, relu=True):
        self.gamma, self.beta, self.groups, self.eps, self.relu = gamma, beta, groups, eps, relu
        self.bias_conv = None   # see pair_conv_gn
        self.next_conv = None   # see chain_gn_conv
        self._gnb_fwd = None    # (data_ptr of the last output, its input, its ReLU bit plane): picked up by next_conv's forward
        self._gnb_bwd = None    # (data_ptr of the gradient next_conv's dgrad wrote, the partial sums it left): picked up by this layer's backward

    def __call__(self, x, meta=None):
        if torch.is_grad_enabled():
            return _GNFn.apply(x, hook(x.device), self, meta)
        return self._fwd(x, meta)[0]

    def _fwd(self, x, meta, for_backward=False):
        if meta is None:  # one NHWC tensor: a segment per image
            N, H, W, C = x.shape
            y, mean, rstd = hip.groupnorm_relu_seg_fwd(x.view(-1, C), [H * W] * N, self.gamma.t, self.beta.t, self.groups, self.eps,
                                                       self.relu)
            return y.view(x.shape), mean, rstd
        # statistics are per (image, level, group): ONE launch over all (level, image) segments
        conv = self.bias_conv
        gp = conv._gn_part if conv is not None else None
        if gp is not None:
            conv._gn_part = None
            if gp[0] == x.data_ptr() and x.dtype == hip.h16_dtype():   # x is the output that conv just wrote: its epilogue left the statistics
                bits = None
                if (for_backward and self.next_conv is not None and self.relu and gn_bwd_fuse_on() and x.shape[1] % 32 == 0
                        and x.is_contiguous()):
                    bits = torch.empty((x.shape[0] * x.shape[1] // 8,), dtype=torch.uint8, device=x.device)
                out = hip.groupnorm_relu_seg_fwd_p32(x, meta.seg_rows, self.gamma.t, self.beta.t, gp[1], self.groups, self.eps, self.relu,
                                                     relu_bits=bits)
                self._gnb_fwd = (out[0].data_ptr(), x, bits) if bits is not None else None
                return out
        return hip.groupnorm_relu_seg_fwd(x, meta.seg_rows, self.gamma.t, self.beta.t, self.groups, self.eps, self.relu)


GNB_STATS = {"fused": 0}     # GroupNorm backwards that ran from a dgrad epilogue's partial sums since import (tests)


class _GNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, hk, layer, meta):
        y, mean, rstd = layer._fwd(x, meta, for_backward=ctx.needs_input_grad[0])
        ctx.layer = layer
        ctx.meta = meta
        if GRAD_SYNC[0] is not None:
            GRAD_SYNC[0].on_forward(_sync_handles(layer))
        ctx.save_for_backward(x, y, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer, meta = ctx.layer, ctx.meta
        x, y, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        conv = layer.bias_conv
        cs = conv is not None and conv.bias_grad_from_gn()   # the SAME predicate makes the conv's backward skip its own bias sum
        if meta is None:
            N, H, W, C = x.shape
            dx = hip.groupnorm_relu_seg_bwd(dy.view(-1, C), y.view(-1, C), x.view(-1, C), [H * W] * N, mean, rstd, layer.gamma.t,
                                            layer.gamma.g, layer.beta.g, layer.groups, layer.relu, beta=layer.beta.t, want_colsum=cs)
        else:
            gb, layer._gnb_bwd = layer._gnb_bwd, None
            if gb is not None and gb[0] == dy.data_ptr() and dy.dtype == hip.h16_dtype():
                # dy came from next_conv's dgrad with the ReLU mask applied and the first reduction made (gn_bwd_fuse_on)
                dx = hip.groupnorm_seg_bwd_p64(dy, x, meta.seg_rows, mean, rstd, layer.gamma.t, layer.gamma.g, layer.beta.g, layer.groups,
                                               gb[1], want_colsum=cs)
                GNB_STATS["fused"] += 1
            else:
                dx = hip.groupnorm_relu_seg_bwd(dy, y, x, meta.seg_rows, mean, rstd, layer.gamma.t, layer.gamma.g, layer.beta.g,
                                                layer.groups, layer.relu, beta=layer.beta.t, want_colsum=cs)
        if cs:
            dx, part = dx
            # the few-hundred-row reduction of the per-chunk sums is nobody's dependency before the optimizer: weight-gradient stream
            _wgrad_launch(lambda: hip.colsum_partials(part, conv.bias.g, accumulate=True), part, key=conv)
        if meta is None:
            dx = dx.view(x.shape)
        if GRAD_SYNC[0] is not None:
            GRAD_SYNC[0].on_backward_done(_sync_handles(layer))
        return dx, None, None, None


class _UpAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lateral, top):
        return hip.upsample2x_add(lateral, top)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return dy, hip.downsample2x_sum(dy)


def upsample2x_add(lateral, top):
    if torch.is_grad_enabled() and (lateral.requires_grad or top.requires_grad):
        return _UpAddFn.apply(lateral, top)
    return hip.upsample2x_add(lateral, top)


class _ReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.relu(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return hip.relu_bwd_scale(dy.contiguous(), y, None)


def relu(x):
    return _ReLUFn.apply(x) if (torch.is_grad_enabled() and x.requires_grad) else torch.relu(x)


# ------------------------------------------------------------------------------------------------
# dense FCOS losses: inputs are the per-level head outputs, which alias row ranges of one
# level-first buffer (`big`), so no cat/permute is materialised.
class _FocalSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, alpha, gamma):
        ctx.alpha, ctx.gamma = alpha, gamma
        ctx.save_for_backward(logits, labels)
        return hip.sigmoid_focal_fwd(logits, labels, alpha, gamma)

    @staticmethod
    def backward(ctx, gout):
        logits, labels = ctx.saved_tensors
        return hip.sigmoid_focal_bwd(logits, labels, ctx.alpha, ctx.gamma, gout.reshape(1).contiguous().float()), None, None, None


def focal_loss_sum(logits, labels, alpha, gamma):
    return _FocalSumFn.apply(logits, labels, alpha, gamma)


class _LocTermsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, box, labels, reg_targets, bvars, args):
        ctx.args = args
        ctx.has_bv = bvars is not None
        ctx.save_for_backward(box, labels, reg_targets, bvars if bvars is not None else labels)
        nc, reg_max, tsb, tsc = args[:4]
        return hip.fcos_loc_terms_fwd(labels, box, reg_targets, bvars, nc, reg_max, tsb, tsc, flags=args[4] if len(args) > 4 else 0)

    @staticmethod
    def backward(ctx, gsums):
        box, labels, reg_targets, bv = ctx.saved_tensors
        nc, reg_max, tsb, tsc = ctx.args[:4]
        coef = torch.stack((gsums[2], gsums[3], gsums[4], gsums[6])).contiguous().float()
        d = hip.fcos_loc_terms_bwd(labels, box, reg_targets, bv if ctx.has_bv else None, nc, reg_max, tsb, tsc, coef,
                                   flags=ctx.args[4] if len(ctx.args) > 4 else 0)
        return d, None, None, None, None


class _FcosCombineFn(torch.autograd.Function):
    """The scalar tail of the FCOS losses of a fused student pass as one node (utv2_fcos_loss_combine): raw kernel sums -> (weighted
    total, the individual losses for the metrics); backward = the stored d total / d sums, scaled."""

    @staticmethod
    def forward(ctx, focal_sup, sums_sup, focal_cls, sums_cls, sums_reg, norm, world, flags, kl_weight, wmul, wdiv):
        rec, coef = hip.fcos_loss_combine(focal_sup.detach(), sums_sup.detach(), focal_cls.detach(), sums_cls.detach(), sums_reg.detach(),
                                          norm, world, flags, kl_weight, wmul, wdiv)
        ctx.save_for_backward(coef)
        ctx.mark_non_differentiable(rec)
        return rec[7].clone(), rec

    @staticmethod
    def backward(ctx, g, _):
        c = ctx.saved_tensors[0] * g
        return (c[0:1], c[1:9], c[9:10], c[10:18], c[18:26]) + (None,) * 6


def fcos_loss_combine(focal_sup, sums_sup, focal_cls, sums_cls, sums_reg, norm, world, flags, kl_weight, wmul, wdiv):
    return _FcosCombineFn.apply(focal_sup, sums_sup, focal_cls, sums_cls, sums_reg, norm, world, flags, kl_weight, wmul, wdiv)


class _FcosJointLossFn(torch.autograd.Function):
    """The whole loss tail of a fused FCOS student pass as ONE autograd node (round 4): the focal / positive-location kernels of the
    supervised, pseudo-classification and pseudo-regression target sets, the optional normaliser all-reduce and the scalar combine
    forward; backward = five launches that write / add into ONE gradient tensor per head output (utv2_*_bwd_acc).  As separate nodes
    every branch returned a dense [P, 80] gradient of its own and autograd folded them with three elementwise add passes, a zero fill
    and the stack / cat glue around the coefficient vectors (about 20 launches and 0.25 ms with the chip idle between the forward and the
    backward of the step).  Same kernels, same arithmetic per branch; the sum of the branches is formed in the order sup, cls, reg."""

    @staticmethod
    def forward(ctx, logits, box, targets, consts, norm_fn):
        (lab_s, reg_s), (lab_c, reg_c), (lab_r, reg_r, bv_r) = targets
        alpha, gamma, nc, rm, flags_s, flags_p, tsb, tsc, world, cflags, klw, wmul, wdiv = consts
        lg, bx = logits.detach(), box.detach()
        focal_s = hip.sigmoid_focal_fwd(lg, lab_s, alpha, gamma)
        sums_s = hip.fcos_loc_terms_fwd(lab_s, bx, reg_s, None, nc, rm, 0.0, 0.0, flags=flags_s)
        focal_c = hip.sigmoid_focal_fwd(lg, lab_c, alpha, gamma)
        sums_c = hip.fcos_loc_terms_fwd(lab_c, bx, reg_c, None, nc, rm, 0.0, 0.0, flags=flags_p)
        sums_r = hip.fcos_loc_terms_fwd(lab_r, bx, reg_r, bv_r, nc, rm, tsb, tsc, flags=flags_p)
        norm = norm_fn(sums_s, sums_c, sums_r)
        rec, coef = hip.fcos_loss_combine(focal_s, sums_s, focal_c, sums_c, sums_r, norm, world, cflags, klw, wmul, wdiv)
        ctx.consts = consts
        ctx.has_bv = bv_r is not None
        ctx.save_for_backward(lg, bx, lab_s, reg_s, lab_c, reg_c, lab_r, reg_r, bv_r if bv_r is not None else lab_r, coef)
        ctx.mark_non_differentiable(rec)
        return rec[7].clone(), rec

    @staticmethod
    def backward(ctx, g, _):
        lg, bx, lab_s, reg_s, lab_c, reg_c, lab_r, reg_r, bv_r, coef = ctx.saved_tensors
        alpha, gamma, nc, rm, flags_s, flags_p, tsb, tsc = ctx.consts[:8]
        gs = g.reshape(1).contiguous().float()
        dlg, dbx = torch.empty_like(lg), torch.empty_like(bx)
        hip.sigmoid_focal_bwd_acc(lg, lab_s, alpha, gamma, coef[0:1], gs, dlg, False)
        hip.sigmoid_focal_bwd_acc(lg, lab_c, alpha, gamma, coef[9:10], gs, dlg, True)
        hip.fcos_loc_terms_bwd_acc(lab_s, bx, reg_s, None, nc, rm, 0.0, 0.0, coef[1:9], gs, dbx, False, flags=flags_s)
        hip.fcos_loc_terms_bwd_acc(lab_c, bx, reg_c, None, nc, rm, 0.0, 0.0, coef[10:18], gs, dbx, True, flags=flags_p)
        hip.fcos_loc_terms_bwd_acc(lab_r, bx, reg_r, bv_r if ctx.has_bv else None, nc, rm, tsb, tsc, coef[18:26], gs, dbx, True, flags=flags_p)
        return dlg, dbx, None, None, None


def fcos_joint_loss(logits, box, targets, consts, norm_fn):
    return _FcosJointLossFn.apply(logits, box, targets, consts, norm_fn)


class _RcnnCombineFn(torch.autograd.Function):
    """The scalar tail of the Faster-RCNN losses of a fused student pass as one node (utv2_rcnn_loss_combine): raw kernel sums of both
    branches -> (weighted total, the eight losses for the metrics); backward = the stored d total / d sums, scaled."""

    @staticmethod
    def forward(ctx, rpn_sup, rpn_uns, focal_sup, focal_uns, box_sup, box_uns, tgt_sup, tgt_uns, consts):
        rec, coef = hip.rcnn_loss_combine(rpn_sup.detach(), rpn_uns.detach(), focal_sup.detach(), focal_uns.detach(), box_sup.detach(),
                                          box_uns.detach(), tgt_sup, tgt_uns, *consts)
        ctx.save_for_backward(coef)
        ctx.mark_non_differentiable(rec)
        return rec[8].clone(), rec

    @staticmethod
    def backward(ctx, g, _):
        c = ctx.saved_tensors[0] * g      # order: {cls, box, rpn_cls, rpn_loc} x {sup, uns}
        return (torch.stack((c[2], c[3])), torch.stack((c[6], c[7])), c[0:1], c[4:5], c[1:2], c[5:6], None, None, None)


def rcnn_loss_combine(rpn_sup, rpn_uns, focal_sup, focal_uns, box_sup, box_uns, tgt_sup, tgt_uns, consts):
    return _RcnnCombineFn.apply(rpn_sup, rpn_uns, focal_sup, focal_uns, box_sup, box_uns, tgt_sup, tgt_uns, consts)


def fcos_loc_terms(box, labels, reg_targets, bvars, args):
    return _LocTermsFn.apply(box, labels, reg_targets, bvars, args)


# ------------------------------------------------------------------------------------------------
# Faster-RCNN helpers
class FanIn:
    """Gradient fan-in of the FPN levels of a Faster-RCNN pass without the add kernels: the levels feed the RPN head (as rows of the
    level-first buffer) and RoIAlign (as level tensors), and autograd would sum the two gradients of every level with one elementwise
    pass each (3 tensor passes over [N, H, W, 256]; 0.67 ms per step).  The ROI heads were built last, so their backward runs first:
    RoIAlign's backward writes its level gradients into the row ranges of ONE level-first buffer kept here and reports no gradient;
    the RPN 3x3 conv's dgrad - whose output is the gradient of that same level-first buffer - adds it in its epilogue (`residual`).
    Either side falls back to the plain path when the other has not / has already run; check() after backward() fails loudly if a stored
    gradient was never picked up."""
    _live = []

    def __init__(self, meta, levels):
        self.meta, self.levels = meta, levels     # levels: how many leading levels of meta RoIAlign reads
        self.buf = None
        self.closed = False                       # the consumer (RPN dgrad) has run
        if len(FanIn._live) >= 64:                # forwards whose backward never ran through a trainer (tests): keep the list short
            del FanIn._live[:32]
        FanIn._live.append(self)

    def store(self, C, dtype, device):
        """-> (level-first buffer, its per-level destination views), or None when the consumer already ran"""
        if self.closed or self.buf is not None:
            return None
        m = self.meta
        buf = torch.empty((m.P, C), dtype=dtype, device=device)
        outs = [m.alias_view(buf, l) for l in range(self.levels)]
        r0 = m.rows[self.levels][0] if self.levels < len(m.rows) else m.P
        if r0 < m.P:
            buf[r0:].zero_()                      # levels RoIAlign does not read (p6)
        self.buf = buf
        return buf, outs

    def take(self):
        self.closed = True
        b, self.buf = self.buf, None
        return b

    @staticmethod
    def check():
        live, FanIn._live = FanIn._live, []
        for f in live:
            if f.buf is not None:
                raise RuntimeError("FanIn: a RoIAlign gradient was stored but the RPN dgrad that adds it never ran")
        parked, GradHandoff._live = GradHandoff._live, []
        for h in parked:
            if h.grad is not None:
                raise RuntimeError("GradHandoff: a lateral's input gradient was parked but the block that adds it never ran")


def fanin_enabled():
    return amp() and os.environ.get("UTV2_FANIN", "1") != "0"


class _AssembleFn(torch.autograd.Function):
    """Makes the level-first buffer the convs wrote into (`big`) visible to autograd as one tensor:
    output aliases `big`; backward hands each level its row range of the incoming gradient."""

    @staticmethod
    def forward(ctx, big_holder, rows, *levels):
        ctx.rows = rows
        return big_holder[0].view(big_holder[0].shape)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return (None, None) + tuple(g[r0:r1].view(shape) for (r0, r1, shape) in ctx.rows)


def assemble(big, rows, levels):
    return _AssembleFn.apply((big,), rows, *levels)


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, roi_batch, roi_valid, cfg, *feats):
        scales, min_level, out_size, per_image, fanin = cfg
        ctx.cfg = cfg[:3]
        ctx.per_image = per_image
        ctx.fanin = fanin
        ctx.shapes = [tuple(f.shape) for f in feats]
        ctx.fdtype = feats[0].dtype
        ctx.save_for_backward(rois, roi_batch, roi_valid)
        return hip.roi_align_fwd([f.detach() for f in feats], scales, min_level, rois, roi_batch, roi_valid, out_size)

    @staticmethod
    def backward(ctx, dy):
        rois, roi_batch, roi_valid = ctx.saved_tensors
        scales, min_level, out_size = ctx.cfg
        N, C = ctx.shapes[0][0], ctx.shapes[0][3]
        R = rois.shape[0]
        if ctx.per_image > 0 and R == N * ctx.per_image and C <= 256 and out_size <= 7:
            # ROIs laid out image by image (the ROI heads' [N, P] slots): deterministic gather, final dtype written directly
            st = ctx.fanin.store(C, ctx.fdtype, dy.device) if (ctx.fanin is not None and ctx.fdtype == hip.h16_dtype()) else None
            if st is not None and [tuple(o.shape) for o in st[1]] == [tuple(x) for x in ctx.shapes]:
                # the RPN conv's dgrad adds these level gradients in its epilogue (FanIn): no gradient reported from here
                hip.roi_align_bwd_tiled(ctx.shapes, ctx.fdtype, scales, min_level, rois, roi_valid, dy.contiguous(), ctx.per_image, outs=st[1])
                return (None, None, None, None) + (None,) * len(ctx.shapes)
            if st is not None:
                ctx.fanin.take(); ctx.fanin.closed = False
            dfeats = hip.roi_align_bwd_tiled(ctx.shapes, ctx.fdtype, scales, min_level, rois, roi_valid, dy.contiguous(), ctx.per_image)
            return (None, None, None, None) + tuple(dfeats)
        dfeats = [torch.zeros(s, dtype=torch.float32, device=dy.device) for s in ctx.shapes]   # fp32: atomics
        hip.roi_align_bwd(dfeats, scales, min_level, rois, roi_batch, roi_valid, dy.contiguous())
        if ctx.fdtype != torch.float32:
            dfeats = [d.to(ctx.fdtype) for d in dfeats]
        return (None, None, None, None) + tuple(dfeats)


def roi_align(feats, scales, min_level, rois, roi_batch, roi_valid, out_size, rois_per_image=0, fanin=None):
    """rois_per_image > 0: the caller guarantees roi_batch == repeat_interleave(arange(N), rois_per_image) (the backward then runs as
    the deterministic tiled gather instead of the atomic scatter); fanin: see FanIn"""
    if torch.is_grad_enabled() and any(f.requires_grad for f in feats):
        return _RoIAlignFn.apply(rois, roi_batch, roi_valid, (tuple(scales), min_level, out_size, int(rois_per_image), fanin), *feats)
    return hip.roi_align_fwd(list(feats), scales, min_level, rois, roi_batch, roi_valid, out_size)


class _SoftmaxFocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, gamma):
        ctx.gamma = gamma
        ctx.save_for_backward(logits, target)
        return hip.softmax_focal_fwd(logits.contiguous(), target, gamma)

    @staticmethod
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        return hip.softmax_focal_bwd(logits.contiguous(), target, ctx.gamma, g.reshape(1).contiguous().float()), None, None


def softmax_focal_sum(logits, target, gamma):
    return _SoftmaxFocalFn.apply(logits, target, gamma)


class _Subsample2Fn(torch.autograd.Function):
    """x[:, ::2, ::2, :] written into a given destination view (LastLevelMaxPool: max_pool2d(k=1, s=2))."""

    @staticmethod
    def forward(ctx, x, out_holder):
        ctx.shape = tuple(x.shape)
        out = out_holder[0]
        out.copy_(x[:, ::2, ::2, :])
        return out

    @staticmethod
    def backward(ctx, g):
        dx = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)
        dx[:, ::2, ::2, :] = g
        return dx, None


def subsample2_into(x, out):
    if torch.is_grad_enabled() and x.requires_grad:
        return _Subsample2Fn.apply(x, (out,))
    out.copy_(x[:, ::2, ::2, :])
    return out
