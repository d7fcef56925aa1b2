"""Flat parameter arena: all model state lives in ONE contiguous fp32 buffer laid out for the
MI355X step (288 GB HBM: no reason to scatter ~300 small tensors):

    [ decay | nodecay | frozen | buffer ]
      ^^^^^^^^^^^^^^^  trainable prefix: one SGD launch per sub-range, one flat RCCL all-reduce
      ^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^^  whole arena: one EMA launch (the reference EMAs params
                                           AND buffers, engine/trainer.py:477-486)

Modules declare `Handle`s while they are constructed; `finalize()` assigns offsets, allocates the
arena (+ grad and momentum arenas for the trainable prefix) and materialises views.  Conv weights
are stored [Cout][KH][KW][Cin]; state_dict() exposes them as [Cout,Cin,KH,KW]-shaped strided
views so checkpoints keep the reference's key names and shapes (modelStudent.* / modelTeacher.*).
"""
from collections import OrderedDict

import os

import torch

KINDS = ("decay", "nodecay", "frozen", "buffer")
# sub-kinds that must each be contiguous across all BN layers (single-launch FrozenBN fold)
BN_KINDS = ("bn_w", "bn_b", "bn_m", "bn_v")
_ORDER = ("decay", "nodecay", "frozen", "bn_w", "bn_b", "buffer", "bn_m", "bn_v")


class Handle:
    __slots__ = ("shape", "kind", "init", "numel", "offset", "t", "g", "exports", "loaders", "store")

    def __init__(self, shape, kind, init):
        self.shape = tuple(int(s) for s in shape)
        self.kind = kind
        self.init = init
        n = 1
        for s in self.shape:
            n *= s
        self.numel = n
        self.offset = -1
        self.t = None  # view into the state arena
        self.g = None  # view into the grad arena (trainable only)
        self.exports = []  # (state_dict key, fn(view)->tensor view)
        self.loaders = {}

    def export(self, key, fn=None, load=None):
        """fn(view) -> tensor exposed under `key`; load(view, src) copies a checkpoint tensor back when the
        exposed tensor is not a writable view (e.g. a re-ordered copy)."""
        self.exports.append((key, fn))
        if load is not None:
            self.loaders[key] = load
        return self


class ParamStore:
    def __init__(self):
        self.handles = []
        self.finalized = False
        self.key_orders = []   # lists of state_dict keys whose RELATIVE order is fixed (see order_keys)

    def order_keys(self, keys):
        """state_dict() lists keys in handle-creation order; handles that fuse tensors of several reference modules (the paired FCOS
        towers: cls_tower.* and bbox_tower.* rows of one matrix) would interleave their keys.  `keys` = those keys in the reference's
        module order: state_dict() keeps the slots they occupy and fills them in this order (checkpoints and the step goldens'
        fingerprint tables list keys in the reference's order)."""
        self.key_orders.append(list(keys))

    def new(self, shape, kind, init=None):
        assert not self.finalized
        assert kind in _ORDER, kind
        h = Handle(shape, kind, init)
        h.store = self
        self.handles.append(h)
        return h

    @staticmethod
    def _align(n, a=4):
        return (n + a - 1) // a * a

    def finalize(self, device):
        off = 0
        self.ranges = {}
        for kind in _ORDER:
            start = off
            for h in self.handles:
                if h.kind == kind:
                    h.offset = off
                    off += self._align(h.numel)
            self.ranges[kind] = (start, off)
        self.total = off
        self.n_train = self.ranges["nodecay"][1]
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.n_train, dtype=torch.float32, device=device)
        self.mom = None  # allocated by the optimizer
        for h in self.handles:
            h.t = self.flat[h.offset: h.offset + h.numel].view(h.shape)
            if h.kind in ("decay", "nodecay"):
                h.g = self.grad[h.offset: h.offset + h.numel].view(h.shape)
        self.finalized = True
        self.version = 0      # bumped whenever the arena content changes (SGD step, EMA, load_state_dict)
        self._flat16 = None   # bf16 mirror for the mixed-precision conv kernels
        self._v16 = -1
        for h in self.handles:
            if h.init is not None:
                h.init(h.t)
        return self

    def touch(self):
        self.version += 1

    def bf16(self, h):
        """bf16 view of handle `h` from the arena's bf16 mirror (ONE conversion launch per arena version)."""
        from . import hip
        if self._flat16 is None or self._flat16.dtype != hip.h16_dtype():
            self._flat16 = torch.empty(self.total, dtype=hip.h16_dtype(), device=self.flat.device)
            self._v16 = -1
        if self._v16 != self.version:
            hip.f32_to_bf16(self.flat, self._flat16)
            self._v16 = self.version
        return self._flat16[h.offset: h.offset + h.numel].view(h.shape)

    def mirror16(self, need_fresh):
        """the 16-bit mirror of the arena as the target of a fused update (SGD: a range, so it must be FRESH - the untouched ranges stay
        valid; EMA: the whole arena, so existence and element type suffice), or None; UTV2_FUSED_MIRROR=0: never"""
        from . import hip
        if os.environ.get("UTV2_FUSED_MIRROR", "1") == "0" or self._flat16 is None or self._flat16.dtype != hip.h16_dtype():
            return None
        if need_fresh and self._v16 != self.version:
            return None
        return self._flat16

    def mirror16_written(self):
        """the update that was handed mirror16() has run and touch() has been called: the mirror is the new version's"""
        self._v16 = self.version

    def region(self, kind):
        s, e = self.ranges[kind]
        return self.flat[s:e]

    def grad_region(self, kind):
        s, e = self.ranges[kind]
        return self.grad[s:e]

    # ---- state_dict surface ------------------------------------------------------------
    def state_dict(self):
        items = []
        for h in self.handles:
            for key, fn in h.exports:
                items.append((key, h.t if fn is None else fn(h.t)))
        for order in self.key_orders:
            pos = {k: i for i, (k, _) in enumerate(items)}
            slots = sorted(pos[k] for k in order)
            vals = [items[pos[k]] for k in order]
            for i, v in zip(slots, vals):
                items[i] = v
        return OrderedDict(items)

    def trainable_named(self):
        out = OrderedDict()
        for h in self.handles:
            if h.kind in ("decay", "nodecay"):
                for key, fn in h.exports:
                    out[key] = (h.t if fn is None else fn(h.t), h.g if fn is None else fn(h.g))
        return out

    def load_state_dict(self, sd, strict=True):
        mine = self.state_dict()
        missing = [k for k in mine if k not in sd]
        unexpected = [k for k in sd if k not in mine]
        if strict and (missing or unexpected):
            raise RuntimeError("load_state_dict: missing %s unexpected %s" % (missing[:5], unexpected[:5]))
        loaders = {}
        for h in self.handles:
            for k, fn in h.loaders.items():
                loaders[k] = (h, fn)
        with torch.no_grad():
            for k, v in mine.items():
                if k in sd:
                    src = sd[k]
                    if tuple(src.shape) != tuple(v.shape):
                        raise RuntimeError("size mismatch for %s: %s vs %s" % (k, tuple(src.shape), tuple(v.shape)))
                    src = src.to(device=v.device, dtype=v.dtype)
                    if k in loaders:
                        loaders[k][1](loaders[k][0].t, src)
                    else:
                        v.copy_(src)
        self.touch()
        return missing, unexpected
