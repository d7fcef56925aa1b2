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
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libutv2_hip.so")

_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_f = ctypes.c_float

# name -> (restype, argtypes).  Must list every symbol declared in include/utv2.h.
_SIGS = {
    "utv2_conv2d_nhwc_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p] + [c_i] * 15 + [c_p]),
    "utv2_conv2d_wgrad_splits": (c_i, [c_i] * 5),
    "utv2_conv2d_wgrad_workspace_floats": (c_i64, [c_i] * 5),
    "utv2_conv2d_nhwc_wgrad": (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 12 + [c_p]),
    "utv2_colsum": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "utv2_weight_flip_transpose": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
}


def load():
    """dlopen the library and bind every declared symbol (works without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "HIP extension %s is missing - run `python __graft_entry__.py` (hipcc, gfx950)" % LIB_PATH
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def symbols():
    return sorted(_SIGS)


def _stream():
    return c_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return c_p(0)
    assert t.is_cuda and t.is_contiguous(), "HIP ops need contiguous CUDA tensors"
    return c_p(t.data_ptr())


def _check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (name, rc))


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    _check(rc, name)


# --------------------------------------------------------------------------------------------
_ws_cache = {}


def workspace(nfloats, device, tag="default"):
    """A reusable fp32 scratch buffer per (device, tag); grows monotonically."""
    key = (str(device), tag)
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


def colsum(g2d, db, accumulate=True):
    M, C = g2d.shape
    ws = workspace(64 * C, g2d.device, "colsum")
    call("utv2_colsum", _p(g2d), _p(db), _p(ws), M, C, int(accumulate), _stream())
    return db
