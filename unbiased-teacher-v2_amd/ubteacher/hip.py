"""ctypes binding of the C-ABI HIP library (include/utv2.h).

Every wrapper takes torch CUDA tensors, passes raw device pointers + the current HIP stream,
and raises RuntimeError on a non-zero return code.  There is NO fallback: if the shared
library is missing the import of any product op fails loudly (build it with
`python __graft_entry__.py`).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_DIR = os.environ.get("UTV2_LIB_DIR") or os.path.join(os.path.dirname(_HERE), "lib")   # UTV2_LIB_DIR: A/B of two builds on one box
LIB_PATH = os.path.join(_LIB_DIR, "libutv2_hip.so")
# two builds of the same sources (csrc/common.h h16_t): the 16-bit float type of the mixed-precision kernels is bfloat16 in the first and
# IEEE fp16 (the reference's own autocast element type) in the second; the dtype code UTV2_BF16 means "the library's 16-bit type"
LIB_PATHS = {"bf16": LIB_PATH, "fp16": os.path.join(_LIB_DIR, "libutv2_hip_f16.so")}
H16 = ["bf16"]          # the build every call goes to (ops.set_precision selects it)


_H16_DT = [torch.bfloat16]


def set_h16(kind):
    assert kind in LIB_PATHS
    H16[0] = kind
    _H16_DT[0] = torch.bfloat16 if kind == "bf16" else torch.float16


def h16_dtype():
    """torch dtype of the selected library's 16-bit activations"""
    return _H16_DT[0]

_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_f = ctypes.c_float

HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "utv2.h")


def _parse_header(path):
    """Build ctypes signatures from the prototypes in include/utv2.h (single source of truth)."""
    import re

    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    sigs = {}
    for m in re.finditer(r"\b(int64_t|int)\s+(utv2_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argt = []
        for a in [x.strip() for x in args.replace("\n", " ").split(",") if x.strip()]:
            if "*" in a or "utv2_stream_t" in a:
                argt.append(c_p)
            elif a.startswith("int64_t"):
                argt.append(c_i64)
            elif a.startswith("double"):
                argt.append(ctypes.c_double)
            elif a.startswith("float"):
                argt.append(c_f)
            elif a.startswith("int"):
                argt.append(c_i)
            elif a == "void":
                pass
            else:
                raise RuntimeError("utv2.h: cannot map argument %r of %s" % (a, name))
        sigs[name] = (c_i64 if ret == "int64_t" else c_i, argt)
    return sigs


_SIGS = _parse_header(HEADER_PATH)
_PLAIN = {"utv2_aug_resize_workspace_bytes", "utv2_topk_rows_workspace_bytes", "utv2_groupnorm_seg_workspace_floats", "utv2_groupnorm_seg_chunks", "utv2_conv2d_wgrad_bf16_splits", "utv2_conv2d_wgrad_bf16_workspace_floats", "utv2_conv2d_bf16_supported", "utv2_conv2d_wgrad_splits", "utv2_conv2d_wgrad_workspace_floats", "utv2_groupnorm_workspace_floats",
          "utv2_nms_mpad", "utv2_nms_workspace_bytes", "utv2_bottleneck_supported", "utv2_wgrad_fold_table_bytes", "utv2_wgrad_fold_pending"}  # return a value, not a status


_libs = {}


def load(kind=None):
    """dlopen the library (of the selected 16-bit type, or of `kind`) and bind every declared symbol (works without a GPU)."""
    kind = kind or H16[0]
    lib = _libs.get(kind)
    if lib is not None:
        return lib
    path = LIB_PATHS[kind]
    if not os.path.exists(path):
        raise RuntimeError(
            "HIP extension %s is missing - run `python __graft_entry__.py` (hipcc, gfx950)" % path
        )
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _libs[kind] = lib
    return lib


def symbols():
    return sorted(_SIGS)


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


# The wrappers below run ~300 times per training step with ~1400 pointer arguments, and a 2 + 2-image step is within 10 % of being
# host-bound: pointers and the stream are handed to ctypes as plain ints (the declared argtypes convert them; no c_void_p objects), the
# device / stream come from the raw C getters (torch.cuda.current_device() walks the lazy-init checks: 0.17 ms per step), the bound
# function objects are cached per library.
def _stream():
    """the current HIP stream of the current device as an integer handle (the raw getters: ~0.3 us against ~9 us for
    torch.cuda.current_stream())"""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "HIP ops need contiguous CUDA tensors"
    return t.data_ptr()


def _dt(t):
    """element-type code of an activation tensor (UTV2_F32 / UTV2_BF16 in include/utv2.h)"""
    d = t.dtype
    if d == torch.float32:
        return 0
    if d == _H16_DT[0]:
        return 1
    raise TypeError("activation tensors are fp32 or %s (the selected library's 16-bit type), got %s" % (h16_dtype(), t.dtype))


def _same_dt(*ts):
    ts = [t for t in ts if t is not None]
    assert all(t.dtype == ts[0].dtype for t in ts), [t.dtype for t in ts]
    return _dt(ts[0])


def _p_any(t):
    return c_p(t.data_ptr())


def _check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (name, rc))


_FN = {}


def call(name, *args):
    fn = _FN.get((H16[0], name))
    if fn is None:
        fn = _FN[(H16[0], name)] = getattr(load(), name)
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (name, rc))


# --------------------------------------------------------------------------------------------
_ws_cache = {}


def workspace(nfloats, device, tag="default"):
    """A reusable fp32 scratch buffer per (device, stream, tag); grows monotonically.  Per stream: the teacher's forward may run on a
    side stream next to the student's (engine/trainer.py), and kernels of two streams must not share scratch."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0, tag)
    t = _ws_cache.get(key)
    if t is None or t.numel() < nfloats:
        t = torch.empty(max(int(nfloats), 1024), dtype=torch.float32, device=device)
        _ws_cache[key] = t
    return t


def conv_out_size(h, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1


def conv2d_fwd(x, w, scale=None, bias=None, residual=None, stride=1, pad=0, relu=False, kh=1, kw=1,
               out=None, in_dil=1, out_hw=None, accumulate=False):
    """x [N,H,W,C] fp32 NHWC; w [K, Kred] (row = one output channel, k = (kh,kw,ci))."""
    N, H, W, C = x.shape
    K, Kred = w.shape
    if out_hw is None:
        OH, OW = conv_out_size(H, kh, stride, pad), conv_out_size(W, kw, stride, pad)
    else:
        OH, OW = out_hw
    if out is None:
        out = torch.empty((N, OH, OW, K), dtype=torch.float32, device=x.device)
    call("utv2_conv2d_nhwc_fwd", _p(x), _p(w), _p(out), _p(scale), _p(bias), _p(residual),
         N, H, W, C, K, kh, kw, stride, pad, in_dil, OH, OW, int(relu), int(accumulate), Kred, _stream())
    return out


def conv2d_stem_fwd(x4, w, scale, bias, stride, pad, kh, kw, relu, out_dtype):
    """x4 [N,H,W,4] fp32 image; w [K, Kred] (16-padded rows); output fp32 or bf16."""
    N, H, W, C = x4.shape
    assert C == 4
    K, Kred = w.shape
    OH, OW = conv_out_size(H, kh, stride, pad), conv_out_size(W, kw, stride, pad)
    out = torch.empty((N, OH, OW, K), dtype=out_dtype, device=x4.device)
    call("utv2_conv2d_stem_fwd", _p(x4), _p(w), _p(out), _dt(out), _p(scale), _p(bias), N, H, W, K, kh, kw, stride, pad, OH, OW,
         int(relu), Kred, _stream())
    return out


def conv2d_dgrad(dy, w_t, in_shape, stride, pad, kh, kw, out=None, accumulate=False):
    """dy [N,OH,OW,K]; w_t [C, kh*kw*K] (flipped/transposed image of w); returns dx [N,H,W,C]."""
    N, H, W, C = in_shape
    _, OH, OW, K = dy.shape
    if out is None:
        out = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
    call("utv2_conv2d_nhwc_fwd", _p(dy), _p(w_t), _p(out), c_p(0), c_p(0), c_p(0),
         N, OH, OW, K, C, kh, kw, 1, kh - 1 - pad, stride, H, W, 0, int(accumulate), kh * kw * K, _stream())
    return out


def weight_flip_transpose(w, K, kh, kw, C):
    wt = torch.empty((C, kh * kw * K), dtype=torch.float32, device=w.device)
    call("utv2_weight_flip_transpose", _p(w), _p(wt), K, kh, kw, C, _stream())
    return wt


def conv2d_wgrad(x, dy, dw, stride, pad, kh, kw, accumulate=True):
    """dw [K, kh*kw*C] (+)= wgrad."""
    N, H, W, C = x.shape
    _, OH, OW, K = dy.shape
    lib = load()
    nws = lib.utv2_conv2d_wgrad_workspace_floats(N, OH, OW, K, kh * kw * C)
    ws = workspace(nws, x.device, "wgrad")
    call("utv2_conv2d_nhwc_wgrad", _p(x), _p(dy), _p(dw), _p(ws), N, H, W, C, K, kh, kw, stride, pad, OH, OW,
         int(accumulate), _stream())
    return dw


def colsum_partials(part, db, accumulate=True):
    """db (+)= column sums of a small fp32 [nb, K] matrix of per-block partial sums"""
    nb, K = part.shape
    call("utv2_colsum_partials", _p(part), _p(db), nb, K, int(accumulate), _stream())
    return db


def colsum(g2d, db, accumulate=True):
    M, C = g2d.shape
    ws = workspace(1024 * C, g2d.device, "colsum")
    call("utv2_colsum", _p(g2d), _p(db), _p(ws), M, C, int(accumulate), _stream())
    return db


# --------------------------------------------------------------------------------------------
# elementwise / optimiser
def conv_clock_probe():
    """(GHz, microseconds): shader clock and lifetime of workgroup 0 of the last persistent 256-tile conv launch (utv2_conv_clock_probe;
    a measurement aid - it synchronises the device)"""
    ghz, us = ctypes.c_double(0.0), ctypes.c_double(0.0)
    call("utv2_conv_clock_probe", ctypes.byref(ghz), ctypes.byref(us))
    return ghz.value, us.value


def ema_axpby(teacher_flat, student_flat, keep_rate, mirror16=None):
    """mirror16 (optional, the library's 16-bit type, same length): also receives the 16-bit rounding of the new teacher values"""
    assert teacher_flat.numel() == student_flat.numel()
    if mirror16 is not None:
        assert mirror16.dtype == h16_dtype() and mirror16.numel() == teacher_flat.numel()
        call("utv2_ema_axpby_m16", _p(teacher_flat), _p(student_flat), _p(mirror16), teacher_flat.numel(), float(keep_rate), _stream())
        return
    call("utv2_ema_axpby", _p(teacher_flat), _p(student_flat), teacher_flat.numel(), float(keep_rate), _stream())


def sgd_momentum(param, grad, mom, lr, momentum, weight_decay, grad_scale=1.0, zero_grad=True, mirror16=None):
    if mirror16 is not None:
        assert mirror16.dtype == h16_dtype() and mirror16.numel() == param.numel()
        call("utv2_sgd_momentum_m16", _p(param), _p(grad), _p(mom), _p(mirror16), param.numel(), float(lr), float(momentum),
             float(weight_decay), float(grad_scale), int(zero_grad), _stream())
        return
    call("utv2_sgd_momentum", _p(param), _p(grad), _p(mom), param.numel(), float(lr), float(momentum),
         float(weight_decay), float(grad_scale), int(zero_grad), _stream())


def amp_found_inf(grad, state):
    call("utv2_amp_found_inf", _p(grad), grad.numel(), _p(state), _stream())


def sgd_momentum_amp(param, grad, mom, lr, momentum, weight_decay, grad_scale, state, mirror16=None):
    if mirror16 is not None:
        assert mirror16.dtype == h16_dtype() and mirror16.numel() == param.numel()
        call("utv2_sgd_momentum_amp_m16", _p(param), _p(grad), _p(mom), _p(mirror16), param.numel(), float(lr), float(momentum),
             float(weight_decay), float(grad_scale), _p(state), _stream())
        return
    call("utv2_sgd_momentum_amp", _p(param), _p(grad), _p(mom), param.numel(), float(lr), float(momentum), float(weight_decay),
         float(grad_scale), _p(state), _stream())


def amp_update_scale(state, growth=2.0, backoff=0.5, interval=2000):
    call("utv2_amp_update_scale", _p(state), float(growth), float(backoff), int(interval), _stream())


def relu_bwd_scale(dy, y=None, scale=None, out=None):
    C = dy.shape[-1]
    M = dy.numel() // C
    if out is None:
        out = torch.empty_like(dy)
    call("utv2_relu_bwd_scale", _p(dy), _p(y), _p(scale), _p(out), M, C, _same_dt(dy, y, out), _stream())
    return out


def add(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    call("utv2_add", _p(a), _p(b), _p(out), a.numel(), _stream())
    return out


def maxpool3x3s2_bwd(x, dpool, relu=True):
    """gradient of maxpool3x3s2 w.r.t. its input x [N,H,W,C] given dpool [N,OH,OW,C] (first-maximum rule), times (x > 0) when relu"""
    N, H, W, C = x.shape
    _, OH, OW, _ = dpool.shape
    dpool = dpool.contiguous()
    dx = torch.empty((N, H, W, C), dtype=dpool.dtype, device=x.device)
    call("utv2_maxpool3x3s2_bwd_nhwc", _p(x), _dt(x), _p(dpool), _p(dx), _dt(dpool), N, H, W, C, OH, OW, int(relu), _stream())
    return dx


def maxpool3x3s2(x, out_dtype=None):
    N, H, W, C = x.shape
    OH, OW = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, OH, OW, C), dtype=out_dtype or x.dtype, device=x.device)
    call("utv2_maxpool3x3s2_nhwc", _p(x), _dt(x), _p(y), _dt(y), N, H, W, C, OH, OW, _stream())
    return y


def upsample2x_add(lateral, top):
    N, H, W, C = lateral.shape
    assert top.shape == (N, H // 2, W // 2, C), (lateral.shape, top.shape)
    out = torch.empty_like(lateral)
    call("utv2_upsample2x_add_nhwc", _p(lateral), _p(top), _p(out), N, H, W, C, _same_dt(lateral, top), _stream())
    return out


def downsample2x_sum(g, out=None, accumulate=False):
    N, H, W, C = g.shape
    if out is None:
        out = torch.empty((N, H // 2, W // 2, C), dtype=g.dtype, device=g.device)
    call("utv2_downsample2x_sum_nhwc", _p(g), _p(out), N, H // 2, W // 2, C, int(accumulate), _same_dt(g, out), _stream())
    return out


def zero_interleave2x(src, H, W, mask=None, mask_bits=None, add=None):
    """[N,(H+1)//2,(W+1)//2,C] -> [N,H,W,C] with src on the even pixels and zeros elsewhere; mask (optional, [N,H,W,C], same dtype):
    the values are zeroed where mask <= 0 (mask_bits: the same mask as a bit plane); add (optional, [N,H,W,C]): summed in, unmasked"""
    N, TH, TW, C = src.shape
    assert TH == (H + 1) // 2 and TW == (W + 1) // 2
    out = torch.empty((N, H, W, C), dtype=src.dtype, device=src.device)
    if mask is not None:
        assert mask.dtype == src.dtype and tuple(mask.shape) == (N, H, W, C)
    if src.dtype == h16_dtype() and C % 8 == 0 and (mask_bits is not None or add is not None or mask is not None):
        if add is not None:
            assert add.dtype == src.dtype and tuple(add.shape) == (N, H, W, C) and add.is_contiguous()
        if mask_bits is not None:
            mask = None
            _bits_ok(mask_bits, out.shape)
        call("utv2_zero_interleave2x_add_nhwc", _p(src), _p(mask), _p(mask_bits), _p(add), _p(out), N, H, W, C, _stream())
        return out
    assert mask_bits is None and add is None, "bit-plane mask / fused add: 16-bit tensors with C % 8 == 0"
    call("utv2_zero_interleave2x_nhwc", _p(src), _p(mask), _p(out), N, H, W, C, _dt(src), _stream())
    return out


_stem_in_cache = {}


def preprocess_images(images, mean, std, size_divisibility, bf16_stem=False):
    """list of [3,H,W] uint8/float CUDA tensors -> ([N,Hp,Wp,4] fp32 NHWC4, image_sizes).
    bf16_stem: instead the input image of the bf16-MFMA stem, bf16 [N,Hp+6,Wp+8,4] with the image at pixel offset (3,3)
    inside a zero border (`.canvas` = (Hp, Wp)); the buffer is cached per shape (its border is zeroed once, the interior
    is rewritten in full by every call)."""
    sizes = [(int(im.shape[1]), int(im.shape[2])) for im in images]
    Hm, Wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
    d = size_divisibility
    if d > 1:
        Hm, Wm = (Hm + d - 1) // d * d, (Wm + d - 1) // d * d
    dev = images[0].device
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    if bf16_stem and Wm % 2 == 0:
        # keyed by stream too (as workspace() is): the teacher's pass on its side stream and a student pass on the main stream may
        # preprocess batches of the same shape concurrently
        key = (len(images), Hm, Wm, str(dev), int(torch.cuda.current_stream(dev).cuda_stream), H16[0])
        out = _stem_in_cache.get(key)
        if out is None:
            if len(_stem_in_cache) > 16:
                _stem_in_cache.clear()
            out = torch.zeros((len(images), Hm + 6, Wm + 8, 4), dtype=h16_dtype(), device=dev)
            _stem_in_cache[key] = out
        n = len(images)
        dt = images[0].dtype
        if n <= 32 and all(im.dtype == dt for im in images):      # the whole batch in one launch
            for im in images:
                assert im.is_cuda and im.is_contiguous() and im.dtype in (torch.uint8, torch.float32)
            ptrs = (ctypes.c_void_p * n)(*[im.data_ptr() for im in images])
            call("utv2_preprocess_images_bf16pad", ctypes.cast(ptrs, c_p), int(dt == torch.uint8), _p(out), ctypes.cast(_iarr([a for a, _ in sizes]), c_p),
                 ctypes.cast(_iarr([b for _, b in sizes]), c_p), n, Hm, Wm, ctypes.cast(m, c_p), ctypes.cast(s, c_p), _stream())
        else:
            for i, im in enumerate(images):
                assert im.is_contiguous() and im.dtype in (torch.uint8, torch.float32)
                call("utv2_preprocess_image_bf16pad", _p(im), int(im.dtype == torch.uint8), c_p(out[i].data_ptr()), sizes[i][0],
                     sizes[i][1], Hm, Wm, ctypes.cast(m, c_p), ctypes.cast(s, c_p), _stream())
        out.canvas = (Hm, Wm)
        return out, sizes
    out = torch.empty((len(images), Hm, Wm, 4), dtype=torch.float32, device=dev)
    for i, im in enumerate(images):
        assert im.is_contiguous() and im.dtype in (torch.uint8, torch.float32)
        call("utv2_preprocess_image", _p(im), int(im.dtype == torch.uint8), c_p(out[i].data_ptr()), sizes[i][0],
             sizes[i][1], Hm, Wm, ctypes.cast(m, c_p), ctypes.cast(s, c_p), _stream())
    return out, sizes


def conv2d_stem_fwd_bf16(xpad16, w16s, scale, bias, relu, out_dtype):
    """xpad16 from preprocess_images(bf16_stem=True); w16s bf16 [K, 7*32] (stem_weight_image); 7x7 stride 2 pad 3."""
    N = xpad16.shape[0]
    H, W = xpad16.canvas
    K = w16s.shape[0]
    OH, OW = conv_out_size(H, 7, 2, 3), conv_out_size(W, 7, 2, 3)
    out = torch.empty((N, OH, OW, K), dtype=out_dtype, device=xpad16.device)
    call("utv2_conv2d_stem_fwd_bf16", _p(xpad16), _p(w16s), _p(out), _dt(out), _p(scale), _p(bias), N, H, W, K, OH, OW,
         int(relu), _stream())
    return out


def stem_pool_fwd_bf16(xpad16, w16s, scale, shift):
    """frozen BasicStem conv + FrozenBN + ReLU + max_pool2d(3, 2, 1) in one launch (utv2_stem_pool_fwd_bf16); inputs as conv2d_stem_fwd_bf16"""
    N = xpad16.shape[0]
    H, W = xpad16.canvas
    K = w16s.shape[0]
    OH, OW = conv_out_size(H, 7, 2, 3), conv_out_size(W, 7, 2, 3)
    PH, PW = (OH + 2 - 3) // 2 + 1, (OW + 2 - 3) // 2 + 1
    out = torch.empty((N, PH, PW, K), dtype=h16_dtype(), device=xpad16.device)
    call("utv2_stem_pool_fwd_bf16", _p(xpad16), _p(w16s), _p(out), _p(scale), _p(shift), N, H, W, K, _stream())
    return out


def stem_weight_image(w208):
    """fp32 [K, 208] (7x7x4 taps, 16-padded rows) -> bf16 [K, 7*32]: per kernel row 7 taps x 4 channels + 4 zeros"""
    K = w208.shape[0]
    w = w208[:, :196].reshape(K, 7, 28)
    return torch.nn.functional.pad(w, (0, 4)).reshape(K, 224).to(h16_dtype()).contiguous()


def frozenbn_fold(w, b, mean, var, scale, shift, eps=1e-5):
    call("utv2_frozenbn_fold", _p(w), _p(b), _p(mean), _p(var), _p(scale), _p(shift), w.numel(), float(eps), _stream())


def groupnorm_relu_fwd(x, gamma, beta, G=32, eps=1e-5, relu=True, out=None):
    N, H, W, C = x.shape
    HW = H * W
    y = torch.empty_like(x) if out is None else out
    mean = torch.empty((N, G), dtype=torch.float32, device=x.device)
    rstd = torch.empty((N, G), dtype=torch.float32, device=x.device)
    ws = workspace(load().utv2_groupnorm_workspace_floats(N, HW, C), x.device, "gn")
    call("utv2_groupnorm_relu_fwd", _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(ws), N, HW, C, G,
         float(eps), int(relu), _stream())
    return y, mean, rstd


def groupnorm_relu_bwd(dy, y, x, mean, rstd, gamma, dgamma, dbeta, G=32, relu=True, out=None):
    N, H, W, C = x.shape
    HW = H * W
    dx = torch.empty_like(x) if out is None else out
    ws = workspace(load().utv2_groupnorm_workspace_floats(N, HW, C), x.device, "gn")
    call("utv2_groupnorm_relu_bwd", _p(dy), _p(y), _p(x), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dgamma), _p(dbeta),
         _p(ws), N, HW, C, G, int(relu), _stream())
    return dx


# --------------------------------------------------------------------------------------------
# FCOS
def _iarr(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _farr(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def fcos_targets(level_hw, strides, soi, gt_boxes, gt_classes, gt_valid, gt_std, num_classes, drop_empty, active=None,
                 center_radius=0.0, batch=None, img0=0):
    """gt_* padded [n,MAXG,...]; returns labels[int32 P], reg_targets[P,4], bvars[P,4], gt_inds[P].
    batch / img0: the gt arrays belong to images [img0, img0 + n) of a batch of `batch` images (default: n = the whole batch)."""
    n_gt, MAXG = gt_classes.shape
    N = n_gt if batch is None else int(batch)
    L = sum(h * w for h, w in level_hw)
    P = N * L
    dev = gt_boxes.device
    labels = torch.empty(P, dtype=torch.int32, device=dev)
    reg = torch.empty((P, 4), dtype=torch.float32, device=dev)
    bv = torch.empty((P, 4), dtype=torch.float32, device=dev)
    gi = torch.empty(P, dtype=torch.int32, device=dev)
    H = _iarr([h for h, _ in level_hw])
    W = _iarr([w for _, w in level_hw])
    S = _iarr(strides)
    flat = []
    for lo, hi in soi:
        flat += [lo, hi]
    so = _farr(flat)
    call("utv2_fcos_targets_range", len(level_hw), ctypes.cast(H, c_p), ctypes.cast(W, c_p), ctypes.cast(S, c_p),
         ctypes.cast(so, c_p), N, MAXG, _p(gt_boxes), _p(gt_classes), _p(gt_valid), _p(gt_std), int(img0), n_gt, num_classes,
         int(drop_empty), float(center_radius), _p(active), _p(labels), _p(reg), _p(bv), _p(gi), _stream())
    return labels, reg, bv, gi


def sigmoid_focal_fwd(logits, labels, alpha, gamma):
    P, C = logits.shape
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    ws = workspace(4096, logits.device, "loss")
    call("utv2_sigmoid_focal_fwd", _p(logits), _p(labels), P, C, float(alpha), float(gamma), _p(out), _p(ws), _stream())
    return out


def sigmoid_focal_bwd(logits, labels, alpha, gamma, coef, out=None):
    P, C = logits.shape
    if out is None:
        out = torch.empty_like(logits)
    call("utv2_sigmoid_focal_bwd", _p(logits), _p(labels), P, C, float(alpha), float(gamma), _p(coef), _p(out), _stream())
    return out


def sigmoid_focal_bwd_acc(logits, labels, alpha, gamma, coef, gscale, out, accumulate):
    """one branch's focal gradient written (accumulate False) or added (True) into `out` - see utv2_sigmoid_focal_bwd_acc"""
    P, C = logits.shape
    call("utv2_sigmoid_focal_bwd_acc", _p(logits), _p(labels), P, C, float(alpha), float(gamma), _p(coef), _p(gscale), _p(out),
         int(bool(accumulate)), _stream())
    return out


def fcos_loc_terms_bwd_acc(labels, box, reg_targets, bvars, num_classes, reg_max, ts_better, ts_cert, coef8, gscale, out, accumulate, flags=0):
    P, BS = box.shape
    call("utv2_fcos_loc_terms_bwd_acc", _p(labels), _p(box), BS, _p(reg_targets), _p(bvars), P, num_classes, reg_max, float(ts_better),
         float(ts_cert), int(flags), _p(coef8), _p(gscale), _p(out), int(bool(accumulate)), _stream())
    return out


LT_QUALITY_IOU, LT_KLLOSS, LT_LOC_IOU, LT_LOC_LINEAR_IOU, LT_KL_WCTR = 1, 2, 1 << 2, 2 << 2, 16  # variant flags of the fcos_loc_terms kernels


def fcos_loc_terms_fwd(labels, box, reg_targets, bvars, num_classes, reg_max, ts_better, ts_cert, flags=0):
    P, BS = box.shape
    sums = torch.empty(8, dtype=torch.float32, device=box.device)
    ws = workspace(4096, box.device, "loss")
    call("utv2_fcos_loc_terms_fwd", _p(labels), _p(box), BS, _p(reg_targets), _p(bvars), P, num_classes, reg_max,
         float(ts_better), float(ts_cert), int(flags), _p(sums), _p(ws), _stream())
    return sums


def fcos_loss_combine(focal_sup, sums_sup, focal_cls, sums_cls, sums_reg, norm, world, flags, kl_weight, wmul, wdiv):
    """(rec [8], coef [26]) - see utv2_fcos_loss_combine"""
    dev = sums_sup.device
    rec = torch.empty(8, dtype=torch.float32, device=dev)
    coef = torch.empty(26, dtype=torch.float32, device=dev)
    call("utv2_fcos_loss_combine", _p(focal_sup), _p(sums_sup), _p(focal_cls), _p(sums_cls), _p(sums_reg), _p(norm), float(world), int(flags),
         float(kl_weight), ctypes.cast(_farr(wmul), c_p), ctypes.cast(_farr(wdiv), c_p), _p(rec), _p(coef), _stream())
    return rec, coef


def fcos_loc_terms_bwd(labels, box, reg_targets, bvars, num_classes, reg_max, ts_better, ts_cert, coef, out=None, flags=0):
    P, BS = box.shape
    if out is None:
        out = torch.empty_like(box)
    call("utv2_fcos_loc_terms_bwd", _p(labels), _p(box), BS, _p(reg_targets), _p(bvars), P, num_classes, reg_max,
         float(ts_better), float(ts_cert), int(flags), _p(coef), _p(out), _stream())
    return out


def fcos_rank_keys(logits, box, reg_max, N, HW, thr, method, out=None, row_stride=None):
    """keys [N][HW*C] int64 (or written into the first HW*C columns of `out`, whose rows are row_stride apart)"""
    C = logits.shape[-1]
    BS = box.shape[-1]
    if out is None:
        out = torch.empty((N, HW * C), dtype=torch.int64, device=logits.device)
        row_stride = HW * C
    assert out.stride(1) == 1 and out.stride(0) == row_stride and out.shape[0] == N and out.shape[1] >= HW * C
    call("utv2_fcos_rank_keys", _p_any(logits), _p_any(box), BS, reg_max, N, HW, C, float(thr), method, c_p(out.data_ptr()),
         int(row_stride), _stream())
    return out


def topk_rows(keys_flat, row_off, rows, max_width, k):
    """exact descending top-k of every ragged row of int64 keys (-1 = empty); returns [rows, k] int64, -1 padded"""
    out = torch.empty((rows, k), dtype=torch.int64, device=keys_flat.device)
    nbytes = load().utv2_topk_rows_workspace_bytes(rows, int(k))
    ws = workspace((nbytes + 3) // 4, keys_flat.device, "topk")
    call("utv2_topk_rows_i64", _p(keys_flat), _p(row_off), rows, int(max_width), int(k), _p(out), _p(ws), _stream())
    return out


def fcos_decode(topkeys, logits, box, reg_max, N, HW, Wl, stride, level, method, slot0, outs):
    K = topkeys.shape[1]
    C = logits.shape[-1]
    BS = box.shape[-1]
    MAXC = outs["scores"].shape[1]
    call("utv2_fcos_decode", _p(topkeys), K, _p(logits), _p(box), BS, reg_max, N, HW, Wl, C, stride, level, method, MAXC,
         slot0, _p(outs["boxes"]), _p(outs["scores"]), _p(outs["classes"]), _p(outs["locations"]), _p(outs["centerness"]),
         _p(outs["cls_confid"]), _p(outs["reg_pred_std"]), _p(outs["fpn_levels"]), _p(outs["valid"]), _stream())


def scale_cols(y2d, ncols, s):
    rows, BS = y2d.shape
    call("utv2_scale_cols", _p(y2d), rows, BS, ncols, _p(s), _stream())


def _i64arr(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])


def scale_cols_ml(y2d, rows, ncols, scales):
    """the Scale layers of all FPN levels of a level-first matrix in one launch: rows = [(r0, r1)] per level (contiguous, ascending),
    scales = the per-level 1-element tensors"""
    assert len(rows) == len(scales) <= 8 and all(rows[i][1] == rows[i + 1][0] for i in range(len(rows) - 1))
    r0 = _i64arr([r[0] for r in rows] + [rows[-1][1]])
    sp = _ptr_array(scales)
    call("utv2_scale_cols_ml", _p(y2d), len(rows), ctypes.cast(r0, c_p), y2d.shape[1], ncols, ctypes.cast(sp, c_p), _stream())


def scale_cols_bwd_ml(g2d, ypost2d, rows, ncols, scales, sgrads):
    """backward of scale_cols_ml: g2d[:, :ncols] *= s_l in place, sgrads[l] += sum(g_in * ypost) / s_l (two launches for all levels)"""
    assert len(rows) == len(scales) == len(sgrads) <= 8 and all(rows[i][1] == rows[i + 1][0] for i in range(len(rows) - 1))
    r0 = _i64arr([r[0] for r in rows] + [rows[-1][1]])
    sp, gp = _ptr_array(scales), _ptr_array(sgrads)
    ws = workspace(8 * 1024, g2d.device, "loss")
    call("utv2_scale_cols_bwd_ml", _p(g2d), _p(ypost2d), len(rows), ctypes.cast(r0, c_p), g2d.shape[1], ncols, ctypes.cast(sp, c_p),
         ctypes.cast(gp, c_p), _p(ws), _stream())


def scale_cols_bwd_ml_pad16(g2d, ypost2d, rows, ncols, scales, sgrads, cpad):
    """scale_cols_bwd_ml out of place: returns the 16-bit [P, cpad] zero-padded scaled gradient, g2d untouched (utv2_scale_cols_bwd_ml_pad16)"""
    assert len(rows) == len(scales) == len(sgrads) <= 8 and all(rows[i][1] == rows[i + 1][0] for i in range(len(rows) - 1))
    r0 = _i64arr([r[0] for r in rows] + [rows[-1][1]])
    sp, gp = _ptr_array(scales), _ptr_array(sgrads)
    ws = workspace(8 * 1024, g2d.device, "loss")
    out = torch.empty((g2d.shape[0], cpad), dtype=h16_dtype(), device=g2d.device)
    call("utv2_scale_cols_bwd_ml_pad16", _p(g2d), _p(ypost2d), len(rows), ctypes.cast(r0, c_p), g2d.shape[1], ncols, ctypes.cast(sp, c_p),
         ctypes.cast(gp, c_p), _p(ws), _p(out), int(cpad), _stream())
    return out


def scale_cols_bwd(g2d, ypost2d, ncols, s):
    rows, BS = g2d.shape
    dsum = torch.empty(1, dtype=torch.float32, device=g2d.device)
    ws = workspace(4096, g2d.device, "loss")
    call("utv2_scale_cols_bwd", _p(g2d), _p(ypost2d), rows, BS, ncols, _p(s), _p(dsum), _p(ws), _stream())
    return dsum


# --------------------------------------------------------------------------------------------
# NMS / IoU
def nms_batched(boxes, scores, classes, valid, iou_thr, class_aware=True, post_topk=-1, max_out=128):
    """boxes [N,M,4], scores [N,M], classes [N,M] int32, valid [N,M] uint8 ->
    keep [N,max_out] int32 (slot ids, -1 padded, descending score), count [N] int32."""
    N, M = scores.shape
    keep = torch.empty((N, max_out), dtype=torch.int32, device=boxes.device)
    cnt = torch.empty(N, dtype=torch.int32, device=boxes.device)
    nbytes = load().utv2_nms_workspace_bytes(N, M)
    ws = workspace((nbytes + 3) // 4, boxes.device, "nms")
    call("utv2_nms_batched", _p(boxes), _p(scores), _p(classes), _p(valid), N, M, float(iou_thr), int(class_aware),
         int(post_topk), max_out, _p(keep), _p(cnt), _p(ws), _stream())
    return keep, cnt


def box_iou(a, b):
    A, B = a.shape[0], b.shape[0]
    out = torch.empty((A, B), dtype=torch.float32, device=a.device)
    call("utv2_box_iou", _p(a), _p(b), A, B, _p(out), _stream())
    return out


# --------------------------------------------------------------------------------------------
# Faster-RCNN
def match_boxes(boxes, gt_boxes, gt_valid, want_gt_max=False):
    """boxes [P,4] (shared anchors) or [N,P,4]; gt_boxes [N,G,4]; gt_valid [N,G] uint8.
    -> max_iou [N,P] (-1 if the image has no gt), argmax [N,P] int32, gt_max_bits [N,G] (or None)."""
    N, G = gt_valid.shape
    shared = boxes.dim() == 2
    P = boxes.shape[-2]
    dev = gt_boxes.device
    mx = torch.empty((N, P), dtype=torch.float32, device=dev)
    arg = torch.empty((N, P), dtype=torch.int32, device=dev)
    gmax = torch.zeros((N, G), dtype=torch.int32, device=dev) if want_gt_max else None
    call("utv2_match_boxes", _p(boxes), 0 if shared else P * 4, N, P, _p(gt_boxes), _p(gt_valid), G, _p(mx), _p(arg),
         _p(gmax), _stream())
    return mx, arg, gmax


def match_lowq(boxes, gt_boxes, gt_valid, gmax):
    N, G = gt_valid.shape
    shared = boxes.dim() == 2
    P = boxes.shape[-2]
    out = torch.empty((N, P), dtype=torch.uint8, device=gt_boxes.device)
    call("utv2_match_lowq", _p(boxes), 0 if shared else P * 4, N, P, _p(gt_boxes), _p(gt_valid), G, _p(gmax), _p(out),
         _stream())
    return out


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


def roi_align_fwd(feats, scales, min_level, rois, roi_batch, roi_valid, out_size):
    """feats: list of NHWC level tensors; rois [R,4]; roi_batch [R] int32 -> [R, PH, PW, C]."""
    R = rois.shape[0]
    C = feats[0].shape[-1]
    out = torch.empty((R, out_size, out_size, C), dtype=feats[0].dtype, device=rois.device)
    fp = _ptr_array(feats)
    H = _iarr([f.shape[1] for f in feats]); W = _iarr([f.shape[2] for f in feats]); S = _farr(scales)
    call("utv2_roi_align_fwd", len(feats), min_level, ctypes.cast(fp, c_p), ctypes.cast(H, c_p), ctypes.cast(W, c_p),
         ctypes.cast(S, c_p), _p(rois), _p(roi_batch), _p(roi_valid), R, C, out_size, out_size, _p(out), _same_dt(*feats), _stream())
    return out


def roi_align_bwd(dfeats, scales, min_level, rois, roi_batch, roi_valid, dy):
    R, PH, PW, C = dy.shape
    fp = _ptr_array(dfeats)
    H = _iarr([f.shape[1] for f in dfeats]); W = _iarr([f.shape[2] for f in dfeats]); S = _farr(scales)
    call("utv2_roi_align_bwd", len(dfeats), min_level, ctypes.cast(fp, c_p), ctypes.cast(H, c_p), ctypes.cast(W, c_p),
         ctypes.cast(S, c_p), _p(rois), _p(roi_batch), _p(roi_valid), R, C, PH, PW, _p(dy), _dt(dy), _stream())


def roi_align_bwd_tiled(shapes, out_dtype, scales, min_level, rois, roi_valid, dy, rois_per_image, outs=None):
    """deterministic gather form: returns the per-level gradient maps (every element written by the kernel); outs: preallocated
    contiguous destinations of these shapes / dtype (e.g. the levels' row ranges of one level-first buffer)"""
    R, PH, PW, C = dy.shape
    N = shapes[0][0]
    assert R == N * rois_per_image
    if outs is None:
        dfeats = [torch.empty(s, dtype=out_dtype, device=dy.device) for s in shapes]
    else:
        dfeats = list(outs)
        assert all(tuple(d.shape) == tuple(s) and d.dtype == out_dtype and d.is_contiguous() for d, s in zip(dfeats, shapes))
    fp = _ptr_array(dfeats)
    H = _iarr([s[1] for s in shapes]); W = _iarr([s[2] for s in shapes]); S = _farr(scales)
    call("utv2_roi_align_bwd_tiled", len(dfeats), min_level, ctypes.cast(fp, c_p), ctypes.cast(H, c_p), ctypes.cast(W, c_p),
         ctypes.cast(S, c_p), _p(rois), _p(roi_valid), N, rois_per_image, C, PH, PW, _p(dy), _dt(dy), _dt(dfeats[0]), _stream())
    return dfeats


def rpn_rank_keys(head, hw, N, A):
    """sortable int64 keys of every objectness logit of the level-first RPN head output, in memory order (the rows of utv2_topk_rows_i64)"""
    rows, ch = head.shape
    assert head.dtype == torch.float32 and rows == N * sum(hw)
    keys = torch.empty(rows * A, dtype=torch.int64, device=head.device)
    call("utv2_rpn_rank_keys", _p(head), len(hw), ctypes.cast(_iarr(hw), c_p), N, A, ch, _p(keys), _stream())
    return keys


def rpn_decode(top, head, anchors, image_hw, hw, ks, N, A, weights, scale_clamp, min_size):
    """selected keys [L*N, maxk] -> per image the sum(ks) candidates in level order: boxes [N,K,4], scores [N,K], lvls int32, keep uint8"""
    K = int(sum(ks))
    dev = head.device
    boxes = torch.empty((N, K, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((N, K), dtype=torch.float32, device=dev)
    lvls = torch.empty((N, K), dtype=torch.int32, device=dev)
    keep = torch.empty((N, K), dtype=torch.uint8, device=dev)
    assert top.dtype == torch.int64 and top.shape[0] == len(hw) * N and anchors.dtype == torch.float32 and image_hw.dtype == torch.float32
    call("utv2_rpn_decode", _p(top), top.shape[1], _p(head), _p(anchors), _p(image_hw), len(hw), ctypes.cast(_iarr(hw), c_p),
         ctypes.cast(_iarr(ks), c_p), N, A, head.shape[1], ctypes.cast(_farr(weights), c_p), float(scale_clamp),
         float(min_size), _p(boxes), _p(scores), _p(lvls), _p(keep), _stream())
    return boxes, scores, lvls, keep


def _u8(t):
    """bool / uint8 tensor as a contiguous uint8 tensor (same bytes)"""
    t = t.contiguous()
    return t.view(torch.uint8) if t.dtype == torch.bool else t


def rpn_loss_fwd(obj, deltas, head_hw, N, A, R, anchors, s, gt_boxes, gt_scores, weights, batch=None, img0=0):
    """(sums [2], gobj [N, npos+nneg], gdl [N, npos, 4]); head_hw = None: dense obj [N,R] / deltas [N,R,4]; else obj is deltas is the
    level-first head output and head_hw the pixels per level.  s: the sampler's dict (pos_idx, pos_valid, neg_idx, neg_valid, matched32, has_gt)."""
    dev = obj.device
    npos, nneg = s["pos_idx"].shape[1], s["neg_idx"].shape[1]
    sums = torch.empty(2, dtype=torch.float32, device=dev)
    gobj = torch.empty((N, npos + nneg), dtype=torch.float32, device=dev)
    gdl = torch.empty((N, npos, 4), dtype=torch.float32, device=dev)
    head = head_hw is not None
    hw = _iarr(head_hw if head else [1])
    G = gt_boxes.shape[1]
    call("utv2_rpn_loss_fwd_range", _p(obj), _p(deltas), int(head), len(head_hw) if head else 1, ctypes.cast(hw, c_p), N,
         N if batch is None else batch, img0, A, obj.shape[-1] if head else 0, R,
         _p(anchors), _p(s["pos_idx"]), _p(_u8(s["pos_valid"])), npos, _p(s["neg_idx"]), _p(_u8(s["neg_valid"])), nneg, _p(s["matched32"]),
         _p(_u8(s["has_gt"])), _p(gt_boxes), _p(gt_scores), G, ctypes.cast(_farr(weights), c_p), _p(sums), _p(gobj), _p(gdl), _stream())
    return sums, gobj, gdl


def rpn_loss_bwd(gobj, gdl, gout_cls, gout_loc, head_hw, N, A, ch, R, s, grad_obj, grad_deltas, batch=None, img0=0):
    head = head_hw is not None
    hw = _iarr(head_hw if head else [1])
    npos, nneg = s["pos_idx"].shape[1], s["neg_idx"].shape[1]
    call("utv2_rpn_loss_bwd_range", _p(gobj), _p(gdl), _p(gout_cls), _p(gout_loc), int(head), len(head_hw) if head else 1, ctypes.cast(hw, c_p), N,
         N if batch is None else batch, img0, A, ch, R,
         _p(s["pos_idx"]), _p(_u8(s["pos_valid"])), npos, _p(s["neg_idx"]), _p(_u8(s["neg_valid"])), nneg, _p(grad_obj), _p(grad_deltas), _stream())


def rpn_sample(mx, lowq, gt_valid, keys, lo, hi, npos_max, batch):
    """PseudoLabRPN.label_and_sample after the matching: labels + "k smallest keys" subsampling of positives / negatives as three
    launches (sortable keys, exact radix select, unpack).  -> pos_idx [N,npos_max] int64, pos_valid uint8, neg_idx [N,batch], neg_valid,
    has_gt [N,1] uint8."""
    N, R = mx.shape
    dev = mx.device
    G = gt_valid.shape[1]
    k = max(npos_max, batch)
    assert k <= 2048 and keys.dtype == torch.float32 and tuple(keys.shape) == (N, R)
    skeys = torch.empty((2 * N, R), dtype=torch.int64, device=dev)
    call("utv2_rpn_sample_keys", _p(mx), _p(_u8(lowq)), _p(gt_valid), G, _p(keys.contiguous()), N, R, float(lo), float(hi), _p(skeys), _stream())
    ck = (N, R, str(dev))
    row_off = _rpn_sample_rows.get(ck)
    if row_off is None:
        if len(_rpn_sample_rows) >= 16:
            _rpn_sample_rows.clear()
        row_off = _rpn_sample_rows[ck] = torch.arange(2 * N + 1, dtype=torch.int64, device=dev) * R
    top = topk_rows(skeys, row_off, 2 * N, R, k)
    pos_idx = torch.empty((N, npos_max), dtype=torch.int64, device=dev)
    pos_valid = torch.empty((N, npos_max), dtype=torch.uint8, device=dev)
    neg_idx = torch.empty((N, batch), dtype=torch.int64, device=dev)
    neg_valid = torch.empty((N, batch), dtype=torch.uint8, device=dev)
    has_gt = torch.empty((N, 1), dtype=torch.uint8, device=dev)
    call("utv2_rpn_sample_unpack", _p(top), k, N, npos_max, batch, _p(gt_valid), G, _p(pos_idx), _p(pos_valid), _p(neg_idx), _p(neg_valid),
         _p(has_gt), _stream())
    return pos_idx, pos_valid, neg_idx, neg_valid, has_gt


_rpn_sample_rows = {}


def roi_sample(boxes, valid, mx, arg, keys, gt_boxes, gt_classes, gt_valid, gt_scores, gt_std, iou_thr, num_classes, batch, nfg_max):
    """StandardROIHeadsPseudoLab.label_and_sample_proposals after the matching, one launch (one workgroup per image)"""
    N, P = mx.shape
    dev = mx.device
    G = gt_valid.shape[1]
    out = dict(proposal_boxes=torch.empty((N, batch, 4), dtype=torch.float32, device=dev),
               gt_classes=torch.empty((N, batch), dtype=torch.int64, device=dev),
               gt_boxes=torch.empty((N, batch, 4), dtype=torch.float32, device=dev),
               valid=torch.empty((N, batch), dtype=torch.uint8, device=dev),
               sampled_idx=torch.empty((N, batch), dtype=torch.int64, device=dev))
    if gt_scores is not None:
        out["gt_confid"] = torch.empty((N, batch), dtype=torch.float32, device=dev)
        if gt_std is not None:
            out["gt_loc_std"] = torch.empty((N, batch, 4), dtype=torch.float32, device=dev)
    call("utv2_roi_sample", _p(boxes), _p(_u8(valid)), _p(mx), _p(arg), _p(keys.contiguous()), N, P, _p(gt_boxes), _p(gt_classes), _p(gt_valid),
         _p(gt_scores), _p(gt_std if gt_scores is not None else None), G, float(iou_thr), int(num_classes), int(batch), int(nfg_max),
         _p(out["proposal_boxes"]), _p(out["gt_classes"]), _p(out["gt_boxes"]), _p(out["valid"]), _p(out["sampled_idx"]),
         _p(out.get("gt_confid")), _p(out.get("gt_loc_std")), _stream())
    return out


def nms_pack(kidx, cnt, boxes, scores):
    N, D = kidx.shape
    M = scores.shape[1]
    ob = torch.empty((N, D, 4), dtype=torch.float32, device=boxes.device)
    osc = torch.empty((N, D), dtype=torch.float32, device=boxes.device)
    ov = torch.empty((N, D), dtype=torch.uint8, device=boxes.device)
    call("utv2_nms_pack", _p(kidx), _p(cnt), _p(boxes), _p(scores), N, M, D, _p(ob), _p(osc), _p(ov), _stream())
    return ob, osc, ov


def roi_infer_keys(probs, deltas, prop, valid, whwh, K, wx, wy, scale_clamp, thr):
    """-> (decoded + clipped boxes [N, P, 4], sortable keys [N, P*K]) - see utv2_roi_infer_keys"""
    N, P = valid.shape
    boxes = torch.empty((N, P, 4), dtype=torch.float32, device=probs.device)
    keys = torch.empty((N, P * K), dtype=torch.int64, device=probs.device)
    call("utv2_roi_infer_keys", _p(probs), _p(deltas), _p(prop), _p(valid), _p(whwh), N, P, K, float(wx), float(wy), float(scale_clamp), float(thr),
         _p(boxes), _p(keys), _stream())
    return boxes, keys


def roi_infer_gather(top, boxes, K, thr):
    N, k = top.shape
    P = boxes.shape[1]
    dev = top.device
    sc = torch.empty((N, k), dtype=torch.float32, device=dev)
    rows = torch.empty((N, k), dtype=torch.int64, device=dev)
    cls = torch.empty((N, k), dtype=torch.int32, device=dev)
    cb = torch.empty((N, k, 4), dtype=torch.float32, device=dev)
    valid = torch.empty((N, k), dtype=torch.uint8, device=dev)
    call("utv2_roi_infer_gather", _p(top), _p(boxes), N, P, K, k, float(thr), _p(sc), _p(rows), _p(cls), _p(cb), _p(valid), _stream())
    return sc, rows, cls, cb, valid


def roi_infer_pack(kidx, cnt, cb, sc, cls, rows, std, P, D):
    N, k = sc.shape
    dev = sc.device
    ob = torch.empty((N, D, 4), dtype=torch.float32, device=dev)
    osc = torch.empty((N, D), dtype=torch.float32, device=dev)
    oc = torch.empty((N, D), dtype=torch.int32, device=dev)
    ostd = torch.empty((N, D, 4), dtype=torch.float32, device=dev)
    orows = torch.empty((N, D), dtype=torch.int64, device=dev)
    ov = torch.empty((N, D), dtype=torch.uint8, device=dev)
    call("utv2_roi_infer_pack", _p(kidx), _p(cnt), _p(cb), _p(sc), _p(cls), _p(rows), _p(std), N, P, k, D, _p(ob), _p(osc), _p(oc), _p(ostd),
         _p(orows), _p(ov), _stream())
    return ob, osc, oc, ostd, orows, ov


def roi_box_loss(deltas, std, cls, prop, gtb, gstd, num_classes, mode, wx, wy, scale_clamp, ts_better, t_cert):
    """(sum [1], d sum / d deltas [R,4], d sum / d std [R,4]); deltas / std may be column slices of the predictor output (row pitch = stride(0))"""
    R = deltas.shape[0]
    assert deltas.dtype == torch.float32 and std.dtype == torch.float32 and deltas.stride(1) == 1 and std.stride(1) == 1
    assert deltas.stride(0) == std.stride(0) and cls.dtype == torch.int64
    dev = deltas.device
    out = torch.empty(1, dtype=torch.float32, device=dev)
    gd = torch.empty((R, 4), dtype=torch.float32, device=dev)
    gs = torch.empty((R, 4), dtype=torch.float32, device=dev)
    call("utv2_roi_box_loss", c_p(deltas.data_ptr()), c_p(std.data_ptr()), deltas.stride(0), _p(cls), _p(prop), _p(gtb), _p(gstd), R, num_classes,
         mode, float(wx), float(wy), float(scale_clamp), float(ts_better), float(t_cert), _p(out), _p(gd), _p(gs), _stream())
    return out, gd, gs


def softmax_focal_fwd(logits, target, gamma):
    R, C = logits.shape
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    ws = workspace(4096, logits.device, "loss")
    call("utv2_softmax_focal_fwd", _p(logits), _p(target), R, C, float(gamma), _p(out), _p(ws), _stream())
    return out


def rcnn_loss_combine(rpn_sup, rpn_uns, focal_sup, focal_uns, box_sup, box_uns, tgt_sup, tgt_uns, norm_sup, norm_uns, w_rpn_cls, w_rpn_loc,
                      w_box, wt):
    """(rec [9], coef [8]) - see utv2_rcnn_loss_combine"""
    dev = rpn_sup.device
    rec = torch.empty(9, dtype=torch.float32, device=dev)
    coef = torch.empty(8, dtype=torch.float32, device=dev)
    call("utv2_rcnn_loss_combine", _p(rpn_sup), _p(rpn_uns), _p(focal_sup), _p(focal_uns), _p(box_sup), _p(box_uns), _p(tgt_sup),
         tgt_sup.numel(), _p(tgt_uns), tgt_uns.numel(), float(norm_sup), float(norm_uns), float(w_rpn_cls), float(w_rpn_loc), float(w_box),
         ctypes.cast(_farr(wt), c_p), _p(rec), _p(coef), _stream())
    return rec, coef


def softmax_focal_bwd(logits, target, gamma, coef):
    R, C = logits.shape
    out = torch.empty_like(logits)
    call("utv2_softmax_focal_bwd", _p(logits), _p(target), R, C, float(gamma), _p(coef), _p(out), _stream())
    return out


# --------------------------------------------------------------------------------------------
# multi-level ("level-first") convs: one launch for all FPN levels of a shared head
def conv2d_ml_fwd(x2d, w, level_hw, N, scale=None, bias=None, residual=None, k=3, pad=1, relu=False, out=None, accumulate=False):
    P, C = x2d.shape
    K = w.shape[0]
    assert P == N * sum(h * w_ for h, w_ in level_hw)
    if out is None:
        out = torch.empty((P, K), dtype=torch.float32, device=x2d.device)
    H = _iarr([h for h, _ in level_hw]); W = _iarr([w_ for _, w_ in level_hw])
    call("utv2_conv2d_ml_fwd", _p(x2d), _p(w), _p(out), _p(scale), _p(bias), _p(residual), len(level_hw), ctypes.cast(H, c_p),
         ctypes.cast(W, c_p), N, C, K, k, k, pad, int(relu), int(accumulate), _stream())
    return out


def conv2d_ml_dgrad(dy2d, w_t, level_hw, N, k, pad, out=None):
    """w_t [C, k*k*K] flipped/transposed image; returns dx [P, C]."""
    return conv2d_ml_fwd(dy2d, w_t, level_hw, N, k=k, pad=k - 1 - pad, out=out)


def conv2d_ml_wgrad(x2d, dy2d, dw, level_hw, N, k, pad, accumulate=True):
    P, C = x2d.shape
    K = dy2d.shape[1]
    nws = load().utv2_conv2d_wgrad_workspace_floats(1, 1, P, K, k * k * C)
    ws = workspace(nws, x2d.device, "wgrad")
    H = _iarr([h for h, _ in level_hw]); W = _iarr([w_ for _, w_ in level_hw])
    call("utv2_conv2d_ml_wgrad", _p(x2d), _p(dy2d), _p(dw), _p(ws), len(level_hw), ctypes.cast(H, c_p), ctypes.cast(W, c_p), N, C, K,
         k, k, pad, int(accumulate), _stream())
    return dw


# --------------------------------------------------------------------------------------------
# mixed precision (bf16 MFMA operands, fp32 accumulate / activations)
def f32_to_bf16(src, dst16):
    call("utv2_f32_to_bf16", _p(src), _p(dst16), src.numel(), _stream())


def weight_flip_transpose_bf16(w, K, kh, kw, C, scale=None):
    """dgrad weight image; scale [K] (optional) = folded FrozenBN multiplier applied per output channel"""
    wt = torch.empty((C, kh * kw * K), dtype=h16_dtype(), device=w.device)
    call("utv2_weight_flip_transpose_bf16", _p(w), _p(wt), _p(scale), K, kh, kw, C, _stream())
    return wt


def _act_dtype(x, out_dtype):
    return out_dtype if out_dtype is not None else x.dtype


def pad_cols_bf16(src2d, cpad):
    """[rows, c] fp32 / bf16 -> bf16 [rows, cpad], zero padded columns"""
    rows, c = src2d.shape
    out = torch.empty((rows, cpad), dtype=h16_dtype(), device=src2d.device)
    call("utv2_pad_cols_bf16", _p(src2d), _dt(src2d), _p(out), rows, c, cpad, _stream())
    return out


def weight_flip_transpose_bf16_batched(arena, scales, bank, table, nlayers):
    call("utv2_weight_flip_transpose_bf16_batched", _p(arena), c_p(scales.data_ptr()) if scales is not None else c_p(0), _p(bank),
         _p(table), int(nlayers), _stream())


def relu_bits_buffer(shape, device):
    """uint8 bit plane of an NHWC / [P, K] 16-bit activation (K % 8 == 0): one bit per element, `value > 0` (utv2_conv2d_nhwc_fwd_bf16_bits)"""
    assert shape[-1] % 8 == 0
    return torch.empty(tuple(shape[:-1]) + (shape[-1] // 8,), dtype=torch.uint8, device=device)


def _bits_ok(bits, shape):
    n = 1
    for d in shape:
        n *= int(d)
    assert bits.dtype == torch.uint8 and bits.is_contiguous() and bits.numel() * 8 == n, "bit plane of another tensor"
    return bits


def conv2d_fwd_bf16(x, w16, scale=None, bias=None, residual=None, stride=1, pad=0, relu=False, kh=1, kw=1, out=None,
                    in_dil=1, out_hw=None, accumulate=False, out_dtype=None, mask=None, post_mask=None, relu_bits=None):
    """x: fp32 or bf16 NHWC; the output (and `residual`) element type is out_dtype (default: x's).  relu_bits (optional uint8
    [N, OH, OW, K / 8], written): bit plane `output > 0` of a 16-bit output."""
    N, H, W, C = x.shape
    K = w16.shape[0]
    assert w16.dtype == h16_dtype() and C % 8 == 0
    if out_hw is None:
        OH, OW = conv_out_size(H, kh, stride, pad), conv_out_size(W, kw, stride, pad)
    else:
        OH, OW = out_hw
    if out is None:
        out = torch.empty((N, OH, OW, K), dtype=_act_dtype(x, out_dtype), device=x.device)
    ri = None
    if _CONV_ROWINFO and kh * kw > 1 and in_dil == 1 and x.dtype == h16_dtype() and C % 32 == 0 and kh * kw <= 16 and W < 32768:
        ri = rowinfo_nhwc(N, H, W, OH, OW, stride, pad, kh, kw, x.device)   # the table the weight gradient of this conv reads
    if relu_bits is not None:
        assert out.dtype == h16_dtype()
        call("utv2_conv2d_nhwc_fwd_bf16_bits", _p(x), _dt(x), _p(w16), _p(out), _same_dt(out, residual, mask, post_mask), _p(scale), _p(bias),
             _p(residual), _p(mask), _p(post_mask), N, H, W, C, K, kh, kw, stride, pad, in_dil, OH, OW, int(relu), int(accumulate), _p(ri),
             _p(_bits_ok(relu_bits, out.shape)), c_p(0), c_p(0), _stream())
        return out
    call("utv2_conv2d_nhwc_fwd_bf16_ri", _p(x), _dt(x), _p(w16), _p(out), _same_dt(out, residual, mask, post_mask), _p(scale), _p(bias),
         _p(residual), _p(mask), _p(post_mask), N, H, W, C, K, kh, kw, stride, pad, in_dil, OH, OW, int(relu), int(accumulate), _p(ri), _stream())
    return out


def bottleneck_fwd_bf16(x, w1, w2, w3, s1, b1, s2, b2, s3, b3, wsc=None, ssc=None, bsc=None, out=None):
    """one frozen bottleneck (conv1 1x1 -> conv2 3x3 -> conv3 1x1 + x, or + the 1x1 shortcut conv `wsc` of x; FrozenBN as scale /
    shift; ReLUs) in one launch (utv2_bottleneck_fwd_bf16): x [N, H, W, C] 16-bit NHWC -> y [N, H, W, 256]"""
    N, H, W, C = x.shape
    MID, K = w1.shape[0], w3.shape[0]
    assert x.dtype == h16_dtype() and w1.dtype == w2.dtype == w3.dtype == h16_dtype() and K == 256
    assert tuple(w1.shape) == (MID, C) and tuple(w2.shape) == (MID, 9 * MID) and tuple(w3.shape) == (K, MID)
    assert wsc is None or (tuple(wsc.shape) == (K, C) and wsc.dtype == h16_dtype())
    if out is None:
        out = torch.empty((N, H, W, K), dtype=x.dtype, device=x.device)
    call("utv2_bottleneck_fwd_bf16", _p(x), _p(out), _p(w1), _p(w2), _p(w3), _p(wsc), _p(s1), _p(b1), _p(s2), _p(b2), _p(s3), _p(b3),
         _p(ssc), _p(bsc), N, H, W, C, MID, _stream())
    return out


def bottleneck_supported(C, MID, has_shortcut):
    return bool(load().utv2_bottleneck_supported(C, MID, int(has_shortcut)))


def conv2d_dgrad_bf16(dy, wt16, in_shape, stride, pad, kh, kw, out=None, out_dtype=None, mask=None, residual=None, post_mask=None,
                      mask_bits=None, post_mask_bits=None):
    """dx = dgrad(dy); mask (the forward activation dx is the gradient of): dx = mask > 0 ? dx : 0; residual: dx += residual;
    post_mask: dx = post_mask > 0 ? dx : 0 after the residual add.  mask_bits / post_mask_bits: the same masks as the bit planes the
    forward convs wrote (relu_bits) - a sixteenth of the bytes."""
    N, H, W, C = in_shape
    _, OH, OW, K = dy.shape
    if out is None:
        out = torch.empty((N, H, W, C), dtype=_act_dtype(dy, out_dtype), device=dy.device)
    ri = None
    if _CONV_ROWINFO and kh * kw > 1 and stride == 1 and dy.dtype == h16_dtype() and K % 32 == 0 and kh * kw <= 16 and OW < 32768:
        ri = rowinfo_nhwc(N, OH, OW, H, W, 1, kh - 1 - pad, kh, kw, dy.device)   # dgrad = a stride-1 conv over dy with pad k-1-pad
    if mask_bits is not None or post_mask_bits is not None:
        assert out.dtype == h16_dtype() and (mask is None or mask_bits is None) and (post_mask is None or post_mask_bits is None)
        call("utv2_conv2d_nhwc_fwd_bf16_bits", _p(dy), _dt(dy), _p(wt16), _p(out), _same_dt(out, residual, mask, post_mask), c_p(0), c_p(0),
             _p(residual), _p(mask), _p(post_mask), N, OH, OW, K, C, kh, kw, 1, kh - 1 - pad, stride, H, W, 0, 0, _p(ri), c_p(0),
             _p(_bits_ok(mask_bits, out.shape)) if mask_bits is not None else c_p(0),
             _p(_bits_ok(post_mask_bits, out.shape)) if post_mask_bits is not None else c_p(0), _stream())
        return out
    call("utv2_conv2d_nhwc_fwd_bf16_ri", _p(dy), _dt(dy), _p(wt16), _p(out), _same_dt(out, residual, mask, post_mask), c_p(0), c_p(0),
         _p(residual), _p(mask), _p(post_mask), N, OH, OW, K, C, kh, kw, 1, kh - 1 - pad, stride, H, W, 0, 0, _p(ri), _stream())
    return out


def _rows_ptr(t):
    """pointer + row pitch of a [rows, cols] matrix whose rows are contiguous runs (a column slice of a wider matrix is fine)"""
    assert t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1], "row-major matrix (or a column slice of one) expected"
    return c_p(t.data_ptr()), int(t.stride(0))


def gnb_eligible(x2d, w16, k, pad, groups):
    """can conv2d_ml_fwd_bf16(..., gnb=...) take this dgrad?  (utv2_conv2d_ml_fwd_bf16_gnb's argument rules)"""
    C = x2d.shape[1] // groups
    K = w16.shape[0]
    return (x2d.dtype == h16_dtype() and C % 64 == 0 and k * k * C >= 1024 and K % groups == 0 and (K // groups) % 128 == 0
            and x2d.stride(1) == 1 and x2d.stride(0) % 8 == 0 and x2d.shape[0] * x2d.stride(0) < (1 << 31))


def gnb_part_buffer(P, K, device):
    """destination of the dgrad epilogue's GroupNorm-backward partials: fp32 [ceil(P / 64), K, 2]"""
    return torch.empty(((P + 63) // 64, K, 2), dtype=torch.float32, device=device)


def conv2d_ml_fwd_bf16(x2d, w16, level_hw, N, scale=None, bias=None, residual=None, k=3, pad=1, relu=False, out=None,
                       out_dtype=None, groups=1, gn_part=None, gnb=None):
    """x2d [P, groups*C] (may be a column slice of a wider matrix); groups > 1: grouped conv, w16 [K, k*k*C] with K/groups output
    channels per group; out (optional) may be a column slice too (then residual must be None).
    gnb (optional) = (mask_bits, gn_x, part64): this conv is the dgrad that produces the gradient of a GroupNorm + ReLU output - the ReLU
    bit plane of that output, the GroupNorm's input [P, K] and the partial-sum buffer (gnb_part_buffer); 16-bit dense output."""
    P = x2d.shape[0]
    C = x2d.shape[1] // groups
    K = w16.shape[0]
    assert w16.shape[1] == k * k * C, (tuple(w16.shape), k, C)
    if out is None:
        out = torch.empty((P, K), dtype=_act_dtype(x2d, out_dtype), device=x2d.device)
    H = _iarr([h for h, _ in level_hw]); W = _iarr([w_ for _, w_ in level_hw])
    xp, xpitch = _rows_ptr(x2d)
    yp, ypitch = _rows_ptr(out)
    if gnb is not None:
        bits, gx, part = gnb
        assert (scale is None and bias is None and residual is None and not relu and gn_part is None and out.dtype == h16_dtype()
                and ypitch == K and gx.dtype == h16_dtype() and gx.is_contiguous() and tuple(gx.shape) == (P, K)
                and bits.numel() * bits.element_size() * 8 == P * K and tuple(part.shape) == ((P + 63) // 64, K, 2))
        ri = rowinfo_ml(N, level_hw, pad, k, x2d.device) if (_CONV_ROWINFO and k > 1) else None
        call("utv2_conv2d_ml_fwd_bf16_gnb", xp, xpitch, _p(w16), yp, len(level_hw), ctypes.cast(H, c_p), ctypes.cast(W, c_p), N, C, K, k, k,
             pad, int(groups), _p(ri), _p(bits), _p(gx), _p(part), _stream())
        return out
    # the geometry table the weight gradients read (cached per geometry): the conv's tile prologues load it instead of decoding it
    ri = None
    if (_CONV_ROWINFO and k > 1 and x2d.dtype == h16_dtype() and C % 32 == 0 and K % 4 == 0 and xpitch % 8 == 0 and ypitch % 8 == 0
            and (groups == 1 or (K // groups) % 128 == 0) and P * xpitch < (1 << 31)):
        ri = rowinfo_ml(N, level_hw, pad, k, x2d.device)
    if groups == 1 and xpitch == C and ypitch == K and gn_part is None and ri is None:
        call("utv2_conv2d_ml_fwd_bf16", xp, _dt(x2d), _p(w16), yp, _same_dt(out, residual), _p(scale), _p(bias), _p(residual),
             len(level_hw), ctypes.cast(H, c_p), ctypes.cast(W, c_p), N, C, K, k, k, pad, int(relu), 0, _stream())
    else:
        assert residual is None or (residual.stride(0) == ypitch and residual.stride(1) == 1)
        call("utv2_conv2d_ml_fwd_bf16_g", xp, _dt(x2d), xpitch, _p(w16), yp, _same_dt(out, residual), ypitch, _p(scale), _p(bias),
             c_p(residual.data_ptr()) if residual is not None else c_p(0), len(level_hw), ctypes.cast(H, c_p), ctypes.cast(W, c_p), N, C, K,
             k, k, pad, int(relu), 0, int(groups), _p(gn_part), _p(ri), _stream())
    return out


def gn_part_buffer(P, K, device):
    """destination of the conv epilogue's GroupNorm statistics partials: fp32 [ceil(P / 32), K / 8, 2]"""
    return torch.empty(((P + 31) // 32, K // 8, 2), dtype=torch.float32, device=device)


def groupnorm_relu_seg_fwd_p32(x2d, seg_rows, gamma, beta, part32, G, eps=1e-5, relu=True, relu_bits=None):
    """groupnorm_relu_seg_fwd for bf16 x2d with 8 channels per group whose producer left the statistics partials in part32.
    relu_bits (optional, uint8 [rows * C / 8], C % 32 == 0): also write the ReLU mask of y as a bit plane"""
    rows, C = x2d.shape
    assert sum(seg_rows) == rows and x2d.dtype == h16_dtype() and C == 8 * G
    y = torch.empty_like(x2d)
    S = len(seg_rows)
    mean = torch.empty((S, G), dtype=torch.float32, device=x2d.device)
    rstd = torch.empty((S, G), dtype=torch.float32, device=x2d.device)
    sr = _iarr(seg_rows)
    if relu_bits is not None:
        assert relu_bits.dtype == torch.uint8 and relu_bits.numel() * 8 == rows * C and C % 32 == 0
        call("utv2_groupnorm_relu_seg_fwd_p32b", _p(x2d), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(part32), S, ctypes.cast(sr, c_p),
             C, G, float(eps), int(relu), _p(relu_bits), _stream())
    else:
        call("utv2_groupnorm_relu_seg_fwd_p32", _p(x2d), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(part32), S, ctypes.cast(sr, c_p),
             C, G, float(eps), int(relu), _stream())
    return y, mean, rstd


_rowinfo_cache = {}
_CONV_ROWINFO = os.environ.get("UTV2_CONV_ROWINFO", "1") != "0"   # A/B: conv tile prologues decode their geometry themselves


def _rowinfo_build(parts, device):
    """[sum N*OH*OW, 2] int32 geometry table: {input pixel index of tap (0,0), (W << 16) | mask of the taps inside the image} per output
    pixel, built by utv2_rowinfo_nhwc on the device (one launch per part = (N, H, W, OH, OW, stride, pad, kh, kw, first pixel index)).
    A table is built once per geometry and then read by launches of EVERY stream (teacher side stream, weight-gradient lanes): the
    building stream is drained before the table enters the cache - once per new canvas, no host arithmetic, no copy."""
    total = sum(p[0] * p[3] * p[4] for p in parts)
    t = torch.empty((total, 2), dtype=torch.int32, device=device)
    off = 0
    for (N, H, W, OH, OW, stride, pad, kh, kw, start) in parts:
        call("utv2_rowinfo_nhwc", c_p(t.data_ptr() + 8 * off), N, H, W, OH, OW, stride, pad, kh, kw, start, _stream())
        off += N * OH * OW
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("a conv geometry table was built inside a hipGraph capture: run the step eagerly once for every canvas first")
    torch.cuda.current_stream(t.device).synchronize()
    return t


def rowinfo_nhwc(N, H, W, OH, OW, stride, pad, kh, kw, device):
    """per-output-pixel geometry table for the bf16 wgrad kernel (cached per geometry)."""
    assert kh * kw <= 16 and W < 32768
    key = ("nhwc", N, H, W, OH, OW, stride, pad, kh, kw, str(device))
    t = _rowinfo_cache.get(key)
    if t is None:
        t = _rowinfo_cache[key] = _rowinfo_build([(N, H, W, OH, OW, stride, pad, kh, kw, 0)], device)
    return t


def rowinfo_ml(N, level_hw, pad, k, device):
    key = ("ml", N, tuple(level_hw), pad, k, str(device))
    t = _rowinfo_cache.get(key)
    if t is None:
        parts, start = [], 0
        for (h, w) in level_hw:
            parts.append((N, h, w, h, w, 1, pad, k, k, start))
            start += N * h * w
        t = _rowinfo_cache[key] = _rowinfo_build(parts, device)
    return t


class WgradFoldLane:
    """The recorded split-K tails of one weight-gradient stream (include/utv2.h utv2_conv2d_wgrad_bf16_d): up to CAP launches leave their
    slabs in consecutive pieces of one arena and their {slabs, gradient, splits, scale} in a host table; flush() reduces all of them with
    ONE launch (bit-identical to the per-launch kernels) instead of 1-2 launches behind every weight gradient.  Everything here is issued
    on the lane's stream, the caller's current stream."""
    CAP = max(1, min(8, int(os.environ.get("UTV2_WGRAD_FOLD_CAP", "8"))))

    def __init__(self):
        self.table = ctypes.create_string_buffer(int(load().utv2_wgrad_fold_table_bytes()))
        self.arena = None
        self.off = 0
        self.n = 0
        self.dsts = set()

    def take(self, nfloats, device, dsts):
        """a piece of the arena for one launch whose tails write `dsts` (data pointers); flushes first when the table is full, the arena
        is, or an already recorded tail writes the same gradient (two tails of one flush run concurrently)"""
        nfl = (int(nfloats) + 63) // 64 * 64
        if self.n >= self.CAP or any(d in self.dsts for d in dsts) or (self.arena is not None and self.off + nfl > self.arena.numel()):
            self.flush()
        if self.arena is None or nfl > self.arena.numel():
            self.flush()
            self.arena = torch.empty(max(4 * nfl, 1 << 24), dtype=torch.float32, device=device)
            self.off = 0
        v = self.arena[self.off:self.off + nfl]
        self.off += nfl
        self.n += 1
        self.dsts.update(dsts)
        return v

    def flush(self):
        if self.n:
            call("utv2_wgrad_fold_flush", ctypes.cast(self.table, c_p), _stream())
        self.n = 0
        self.off = 0
        self.dsts.clear()


WGRAD_FOLD = [None]     # the WgradFoldLane the weight gradients of the current stream record their tails in (ops._wgrad_issue), or None


def conv2d_wgrad_bf16(x, dy2d, dw, rowinfo, C, kh, kw, accumulate=True, db=None, rowscale=None, groups=1, x_pitch=None):
    """x: fp32/bf16 activations (any layout consistent with rowinfo; x_pitch = elements between consecutive pixels when x is a channel
    slice of a wider matrix), dy2d [M,K] fp32/bf16 (may be the leading K columns of a wider, zero-padded matrix); dw [K, kh*kw*C] (+)=
    wgrad (C = input channels per group); db [K] (optional) (+)= column sums of dy (bias gradient, fused into the dY staging)."""
    M, K = dy2d.shape
    assert dy2d.stride(1) == 1
    dy_pitch = int(dy2d.stride(0))
    nws = load().utv2_conv2d_wgrad_bf16_workspace_floats(M, K, kh * kw * C)
    lane = WGRAD_FOLD[0]
    if lane is not None:
        ws = lane.take(nws, dy2d.device, [dw.data_ptr()] + ([db.data_ptr()] if db is not None else []))
        pitch = int(x_pitch) if x_pitch is not None else groups * C
        call("utv2_conv2d_wgrad_bf16_d", c_p(x.data_ptr()), _dt(x), pitch, c_p(dy2d.data_ptr()), _dt(dy2d), dy_pitch, _p(dw), _p(db), _p(ws),
             _p(rowinfo), _p(rowscale), M, C, K, kh, kw, int(accumulate), int(groups), ctypes.cast(lane.table, c_p), _stream())
        return dw
    ws = workspace(nws, dy2d.device, "wgrad")
    if groups == 1 and (x_pitch is None or x_pitch == C) and dy_pitch == K:
        call("utv2_conv2d_wgrad_bf16", _p(x), _dt(x), _p(dy2d), _dt(dy2d), _p(dw), _p(db), _p(ws), _p(rowinfo), _p(rowscale), M, C, K, kh,
             kw, int(accumulate), _stream())
    else:
        xp = c_p(x.data_ptr())
        pitch = int(x_pitch) if x_pitch is not None else groups * C
        call("utv2_conv2d_wgrad_bf16_g", xp, _dt(x), pitch, c_p(dy2d.data_ptr()), _dt(dy2d), dy_pitch, _p(dw), _p(db), _p(ws), _p(rowinfo),
             _p(rowscale), M, C, K, kh, kw, int(accumulate), int(groups), _stream())
    return dw


def groupnorm_relu_seg_fwd(x2d, seg_rows, gamma, beta, G=32, eps=1e-5, relu=True):
    """x2d [rows, C]; seg_rows: rows of each consecutive (image, level) segment."""
    rows, C = x2d.shape
    assert sum(seg_rows) == rows
    y = torch.empty_like(x2d)
    S = len(seg_rows)
    mean = torch.empty((S, G), dtype=torch.float32, device=x2d.device)
    rstd = torch.empty((S, G), dtype=torch.float32, device=x2d.device)
    sr = _iarr(seg_rows)
    ws = workspace(load().utv2_groupnorm_seg_workspace_floats(S, ctypes.cast(sr, c_p), C), x2d.device, "gn")
    call("utv2_groupnorm_relu_seg_fwd", _p(x2d), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(ws), S, ctypes.cast(sr, c_p), C, G,
         float(eps), int(relu), _dt(x2d), _stream())
    return y, mean, rstd


def groupnorm_relu_seg_bwd(dy, y, x2d, seg_rows, mean, rstd, gamma, dgamma, dbeta, G=32, relu=True, beta=None, want_colsum=False):
    """-> dx, or (dx, colsum_part [chunks, C] fp32: per-chunk column sums of dx as stored) with want_colsum"""
    rows, C = x2d.shape
    S = len(seg_rows)
    dx = torch.empty_like(x2d)
    sr = _iarr(seg_rows)
    ws = workspace(load().utv2_groupnorm_seg_workspace_floats(S, ctypes.cast(sr, c_p), C), x2d.device, "gn")
    part = None
    if want_colsum:
        part = torch.empty((load().utv2_groupnorm_seg_chunks(S, ctypes.cast(sr, c_p)), C), dtype=torch.float32, device=x2d.device)
    call("utv2_groupnorm_relu_seg_bwd_colsum", _p(dy), _p(y), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(beta), _p(dx), _p(dgamma), _p(dbeta),
         _p(ws), S, ctypes.cast(sr, c_p), C, G, int(relu), _same_dt(dy, y, x2d), _p(part), _stream())
    return (dx, part) if want_colsum else dx


def groupnorm_seg_bwd_p64(g, x2d, seg_rows, mean, rstd, gamma, dgamma, dbeta, G, part64, want_colsum=False):
    """GroupNorm (+ ReLU) backward from the partial sums the producing dgrad left (conv2d_ml_fwd_bf16 gnb): g = the incoming gradient with
    the ReLU mask applied.  -> dx, or (dx, colsum_part) as groupnorm_relu_seg_bwd"""
    rows, C = x2d.shape
    assert g.dtype == h16_dtype() and x2d.dtype == h16_dtype() and g.is_contiguous() and x2d.is_contiguous() and tuple(g.shape) == (rows, C)
    assert tuple(part64.shape) == ((rows + 63) // 64, C, 2) and sum(seg_rows) == rows
    S = len(seg_rows)
    dx = torch.empty_like(x2d)
    sr = _iarr(seg_rows)
    ws = workspace(load().utv2_groupnorm_seg_workspace_floats(S, ctypes.cast(sr, c_p), C), x2d.device, "gn")
    part = None
    if want_colsum:
        part = torch.empty((load().utv2_groupnorm_seg_chunks(S, ctypes.cast(sr, c_p)), C), dtype=torch.float32, device=x2d.device)
    call("utv2_groupnorm_seg_bwd_p64", _p(g), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(dx), _p(dgamma), _p(dbeta), _p(ws), S,
         ctypes.cast(sr, c_p), C, G, _p(part64), _p(part), _stream())
    return (dx, part) if want_colsum else dx


# ---- two-crop data path: Pillow-exact image arithmetic on uint8 [H][W][3] device tensors (csrc/augment.hip) ----------------------------
def _u8_hwc(img):
    assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3 and img.is_cuda and img.is_contiguous(), \
        "augmentation ops take contiguous uint8 [H][W][3] CUDA tensors"
    return img


def aug_resize(img, out_h, out_w, flip=False):
    """PIL.Image.resize((out_w, out_h), BILINEAR) (+ horizontal flip) -> new uint8 [out_h][out_w][3]"""
    _u8_hwc(img)
    H, W = int(img.shape[0]), int(img.shape[1])
    nbytes = load().utv2_aug_resize_workspace_bytes(H, W, int(out_h), int(out_w))
    if nbytes < 0:
        raise RuntimeError("utv2_aug_resize_workspace_bytes: bad argument")
    ws = workspace((nbytes + 3) // 4, img.device, "aug_resize")
    out = torch.empty((int(out_h), int(out_w), 3), dtype=torch.uint8, device=img.device)
    call("utv2_aug_resize_bilinear_u8", _p(img), H, W, _p(out), int(out_h), int(out_w), int(bool(flip)), _p(ws), _stream())
    return out


def aug_brightness(img, factor):
    """ImageEnhance.Brightness(img).enhance(factor), in place"""
    _u8_hwc(img)
    call("utv2_aug_blend_u8", _p(img), img.shape[0] * img.shape[1], 0, float(factor), c_p(0), _stream())
    return img


def aug_contrast(img, factor):
    """ImageEnhance.Contrast(img).enhance(factor), in place (the grey mean is reduced on the device, no host sync)"""
    _u8_hwc(img)
    ws = workspace(4, img.device, "aug_mean")  # [0:2] 64-bit sum, [2] mean
    wi = ws.view(torch.int32)
    call("utv2_aug_gray_mean_u8", _p(img), img.shape[0] * img.shape[1], _p(ws), c_p(wi.data_ptr() + 8), _stream())
    call("utv2_aug_blend_u8", _p(img), img.shape[0] * img.shape[1], 1, float(factor), c_p(wi.data_ptr() + 8), _stream())
    return img


def aug_saturation(img, factor):
    """ImageEnhance.Color(img).enhance(factor), in place"""
    _u8_hwc(img)
    call("utv2_aug_blend_u8", _p(img), img.shape[0] * img.shape[1], 2, float(factor), c_p(0), _stream())
    return img


def aug_hue(img, hue_factor):
    """torchvision F_pil.adjust_hue(img, hue_factor), in place"""
    import math
    _u8_hwc(img)
    if not -0.5 <= hue_factor <= 0.5:
        raise ValueError("hue_factor ({}) is not in [-0.5, 0.5].".format(hue_factor))
    call("utv2_aug_hue_u8", _p(img), img.shape[0] * img.shape[1], int(math.trunc(hue_factor * 255)) & 255, _stream())
    return img


def aug_grayscale(img):
    """img.convert("L") replicated to three channels, in place"""
    _u8_hwc(img)
    call("utv2_aug_grayscale_u8", _p(img), img.shape[0] * img.shape[1], _stream())
    return img


def _box_blur_constants(radius, passes=3):
    """BoxBlur.c _gaussian_blur_radius + the (radius, ww, fw) of ImagingHorizontalBoxBlur, in the float32 arithmetic of the C code"""
    import math
    import numpy as np
    f32 = np.float32
    radius = f32(radius)
    sigma2 = f32(radius * radius / f32(passes))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32(f32(2 * l + 1) * f32(f32(l * f32(l + 1)) - f32(3 * sigma2)))
    a = f32(a / f32(6 * f32(sigma2 - f32(f32(l + 1) * f32(l + 1)))))
    fr = f32(l + a)
    r = int(fr)
    ww = int(f32(f32(1 << 24) / f32(fr * 2 + 1)))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw


def aug_gaussian_blur(img, radius, passes=3):
    """img.filter(PIL.ImageFilter.GaussianBlur(radius)) -> uint8 [H][W][3] (may alias a scratch buffer; `img` is clobbered)"""
    _u8_hwc(img)
    H, W = int(img.shape[0]), int(img.shape[1])
    r, ww, fw = _box_blur_constants(radius, passes)
    a, b = img, torch.empty_like(img)
    for vertical in (0, 1):
        for _ in range(passes):
            call("utv2_aug_box_blur_u8", _p(a), _p(b), H, W, vertical, r, ww, fw, _stream())
            a, b = b, a
    return a


def aug_erase(img, i, j, h, w, noise):
    """ToTensor -> img[..., i:i+h, j:j+w] = noise -> ToPILImage, in place; noise float32 [3][h][w]"""
    _u8_hwc(img)
    assert noise.dtype == torch.float32 and tuple(noise.shape) == (3, h, w)
    call("utv2_aug_erase_u8", _p(img), int(img.shape[0]), int(img.shape[1]), int(i), int(j), int(h), int(w), _p(noise.contiguous()), _stream())
    return img


def aug_to_chw(img):
    _u8_hwc(img)
    out = torch.empty((3, img.shape[0], img.shape[1]), dtype=torch.uint8, device=img.device)
    call("utv2_aug_hwc_to_chw_u8", _p(img), _p(out), img.shape[0] * img.shape[1], _stream())
    return out
