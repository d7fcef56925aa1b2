"""Autograd wrappers around the C-ABI HIP kernels (ubteacher.hip).

Design notes (MI355X-first, not a translation of torch.nn):
  * activations are NHWC fp32 tensors; weights live in the flat ParamStore arena;
  * parameter gradients are NOT returned to autograd: every backward accumulates straight into
    the flat gradient arena (`handle.g`), so one flat RCCL all-reduce + one SGD launch follow;
  * a per-device `hook` tensor (requires_grad) is threaded through every Function so backward
    runs even when the activation input comes from the frozen stem/res2;
  * FrozenBN is folded into the conv epilogue (scale/shift), ReLU and the residual add too.
"""
import torch

from . import hip

_HOOKS = {}
_VERSION = [0]  # bumped by the optimizer step: invalidates cached dgrad weight images


def hook(device):
    key = str(device)
    h = _HOOKS.get(key)
    if h is None:
        h = torch.zeros(1, device=device, requires_grad=True)
        _HOOKS[key] = h
    return h


def bump_version():
    _VERSION[0] += 1


class Conv:
    """One convolution (+ optional folded FrozenBN or bias, ReLU, residual) bound to arena handles."""

    def __init__(self, w, cin, cout, k, stride=1, pad=0, bias=None, bn=None, relu=False, trainable=True,
                 kred=None, colscale=None):
        self.w = w  # Handle, shape [cout, kred]
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        self.bias = bias  # Handle [cout] or None
        self.bn = bn  # FrozenBN or None
        self.relu = relu
        self.trainable = trainable
        self.kred = kred if kred is not None else k * k * cin
        self.colscale = colscale  # (Handle scalar, ncols): Scale layer on the first ncols output channels
        self._wt = None
        self._wt_version = -1

    def scale_shift(self):
        if self.bn is not None:
            return self.bn.scale, self.bn.shift
        return None, (self.bias.t if self.bias is not None else None)

    def wt(self):
        if self._wt is None or self._wt_version != _VERSION[0]:
            self._wt = hip.weight_flip_transpose(self.w.t, self.cout, self.k, self.k, self.cin)
            self._wt_version = _VERSION[0]
        return self._wt

    def __call__(self, x, residual=None, out=None, colscale_handle=None):
        if torch.is_grad_enabled() and self.trainable:
            return _ConvFn.apply(x, residual, hook(x.device), self, (out,), colscale_handle)
        sc, sh = self.scale_shift()
        y = hip.conv2d_fwd(x, self.w.t, scale=sc, bias=sh, residual=residual, stride=self.stride, pad=self.pad,
                           relu=self.relu, kh=self.k, kw=self.k, out=out)
        if colscale_handle is not None:
            hip.scale_cols(y.view(-1, self.cout), self.colscale, colscale_handle.t)
        return y


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, hk, layer, out_holder, cs):
        out = out_holder[0]  # optional destination view (kept out of autograd's sight on purpose)
        sc, sh = layer.scale_shift()
        y = hip.conv2d_fwd(x, layer.w.t, scale=sc, bias=sh, residual=residual, stride=layer.stride, pad=layer.pad,
                           relu=layer.relu, kh=layer.k, kw=layer.k, out=out)
        if cs is not None:
            hip.scale_cols(y.view(-1, layer.cout), layer.colscale, cs.t)
        ctx.layer = layer
        ctx.cs = cs
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, y if (layer.relu or cs is not None) else None)
        ctx.xshape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer = ctx.layer
        x, y = ctx.saved_tensors
        dy = dy.contiguous()
        sc, _ = layer.scale_shift()
        if ctx.cs is not None:
            g = dy.clone()
            dsum = hip.scale_cols_bwd(g.view(-1, layer.cout), y.view(-1, layer.cout), layer.colscale, ctx.cs.t)
            ctx.cs.g.add_(dsum / ctx.cs.t.view(-1))
            dy = g
        gres = None
        if ctx.has_res:
            gm = hip.relu_bwd_scale(dy, y if layer.relu else None, None) if layer.relu else dy
            gres = gm
            g = hip.relu_bwd_scale(gm, None, sc) if sc is not None else gm
        else:
            if layer.relu or sc is not None:
                g = hip.relu_bwd_scale(dy, y if layer.relu else None, sc)
            else:
                g = dy
        dx = None
        if ctx.needs_input_grad[0]:
            dx = hip.conv2d_dgrad(g, layer.wt(), ctx.xshape, layer.stride, layer.pad, layer.k, layer.k)
        hip.conv2d_wgrad(x, g, layer.w.g, layer.stride, layer.pad, layer.k, layer.k, accumulate=True)
        if layer.bias is not None:
            hip.colsum(g.view(-1, layer.cout), layer.bias.g, accumulate=True)
        return dx, gres, None, None, None, None


class GroupNormReLU:
    def __init__(self, gamma, beta, groups=32, eps=1e-5, relu=True):
        self.gamma, self.beta, self.groups, self.eps, self.relu = gamma, beta, groups, eps, relu

    def __call__(self, x):
        if torch.is_grad_enabled():
            return _GNFn.apply(x, hook(x.device), self)
        y, _, _ = hip.groupnorm_relu_fwd(x, self.gamma.t, self.beta.t, self.groups, self.eps, self.relu)
        return y


class _GNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, hk, layer):
        y, mean, rstd = hip.groupnorm_relu_fwd(x, layer.gamma.t, layer.beta.t, layer.groups, layer.eps, layer.relu)
        ctx.layer = layer
        ctx.save_for_backward(x, y, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer = ctx.layer
        x, y, mean, rstd = ctx.saved_tensors
        dx = hip.groupnorm_relu_bwd(dy.contiguous(), y, x, mean, rstd, layer.gamma.t, layer.gamma.g, layer.beta.g,
                                    layer.groups, layer.relu)
        return dx, None, None


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
    def forward(ctx, big, labels, alpha, gamma, rows, *levels):
        ctx.big, ctx.labels, ctx.alpha, ctx.gamma, ctx.rows = big, labels, alpha, gamma, rows
        return hip.sigmoid_focal_fwd(big, labels, alpha, gamma)

    @staticmethod
    def backward(ctx, gout):
        coef = gout.reshape(1).contiguous().float()
        d = hip.sigmoid_focal_bwd(ctx.big, ctx.labels, ctx.alpha, ctx.gamma, coef)
        grads = []
        for (r0, r1, shape) in ctx.rows:
            grads.append(d[r0:r1].view(shape))
        return (None, None, None, None, None) + tuple(grads)


def focal_loss_sum(big, labels, alpha, gamma, rows, levels):
    return _FocalSumFn.apply(big, labels, alpha, gamma, rows, *levels)


class _LocTermsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, big, labels, reg_targets, bvars, args, rows, *levels):
        ctx.big, ctx.labels, ctx.reg_targets, ctx.bvars, ctx.args, ctx.rows = big, labels, reg_targets, bvars, args, rows
        nc, reg_max, tsb, tsc = args
        return hip.fcos_loc_terms_fwd(labels, big, reg_targets, bvars, nc, reg_max, tsb, tsc)

    @staticmethod
    def backward(ctx, gsums):
        nc, reg_max, tsb, tsc = ctx.args
        coef = torch.stack((gsums[2], gsums[3], gsums[4], gsums[6])).contiguous().float()
        d = hip.fcos_loc_terms_bwd(ctx.labels, ctx.big, ctx.reg_targets, ctx.bvars, nc, reg_max, tsb, tsc, coef)
        grads = []
        for (r0, r1, shape) in ctx.rows:
            grads.append(d[r0:r1].view(shape))
        return (None, None, None, None, None, None) + tuple(grads)


def fcos_loc_terms(big, labels, reg_targets, bvars, args, rows, levels):
    return _LocTermsFn.apply(big, labels, reg_targets, bvars, args, rows, *levels)


# ------------------------------------------------------------------------------------------------
# Faster-RCNN helpers
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
        scales, min_level, out_size = cfg
        ctx.cfg = cfg
        ctx.shapes = [tuple(f.shape) for f in feats]
        ctx.save_for_backward(rois, roi_batch, roi_valid)
        return hip.roi_align_fwd([f.detach() for f in feats], scales, min_level, rois, roi_batch, roi_valid, out_size)

    @staticmethod
    def backward(ctx, dy):
        rois, roi_batch, roi_valid = ctx.saved_tensors
        scales, min_level, out_size = ctx.cfg
        dfeats = [torch.zeros(s, dtype=torch.float32, device=dy.device) for s in ctx.shapes]
        hip.roi_align_bwd(dfeats, scales, min_level, rois, roi_batch, roi_valid, dy.contiguous())
        return (None, None, None, None) + tuple(dfeats)


def roi_align(feats, scales, min_level, rois, roi_batch, roi_valid, out_size):
    if torch.is_grad_enabled() and any(f.requires_grad for f in feats):
        return _RoIAlignFn.apply(rois, roi_batch, roi_valid, (tuple(scales), min_level, out_size), *feats)
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
