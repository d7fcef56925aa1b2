"""Per-shape table of every convolution launch (forward, dgrad, weight gradient) of real UTv2 steps.

Hooks `ubteacher.hip.call` during the timed region of `bench.py`, keeps one (entry point, argument tuple) per distinct shape, then
replays each distinct launch alone and back to back (the in-step event pairs would include queueing behind the other streams).
Argument meaning comes from the prototypes of include/utv2.h, parsed by name, so the table follows the C-ABI as it evolves.

    python tools/profile_shapes.py [--model rcnn] [--dtype bf16] > profiles/rNN_conv_shapes.txt

Columns: launches per step, isolated microseconds per launch, ms per step (= launches x isolated time), TFLOP/s on the algorithmic
FLOPs `2*M*K*KH*KW*C/groups`, GB/s on the algorithmic bytes (activations in + out + residual / mask operands + weights), and which of the
two floors (1000 TFLOP/s = what the tuned 256-tile kernel reaches alone; 5 TB/s) is the higher one for that shape - `floor_us`."""
import collections
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
import torch  # noqa: E402
from ubteacher import hip  # noqa: E402
import bench  # noqa: E402


def arg_names():
    txt = open(hip.HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int64_t|int)\s+(utv2_\w+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        names = []
        for a in [x.strip() for x in m.group(2).replace("\n", " ").split(",") if x.strip()]:
            names.append(re.sub(r"\[.*\]", "", a).split()[-1].lstrip("*"))
        out[m.group(1)] = names
    return out


NAMES = arg_names()
CONV = [n for n in NAMES if (n.startswith("utv2_conv2d") or n == "utv2_bottleneck_fwd_bf16") and not n.endswith(("_splits", "_workspace_floats", "_supported"))]


def val(a):
    if isinstance(a, ctypes.Array):
        return tuple(a)
    if isinstance(a, (ctypes.c_void_p,)):
        return a.value or 0
    if hasattr(a, "value"):
        return a.value
    return a


def describe(name, args, levels_hw):
    """(kind, shape key, flop, bytes) of one launch"""
    A = dict(zip(NAMES[name], [val(a) for a in args]))
    eb = lambda k: 2 if A.get(k, 0) == 1 else 4          # *_dtype: 1 = the 16-bit type
    if name == "utv2_bottleneck_fwd_bf16":               # a whole frozen bottleneck: conv1 + conv2 + conv3 (+ shortcut); bytes = x + y
        N, H, W, C, MID = (A[k] for k in ("N", "H", "W", "C", "MID"))
        sc = bool(A.get("wsc"))
        fl = 2.0 * N * H * W * (C * MID + 9 * MID * MID + MID * 256 + (C * 256 if sc else 0))
        return "block" + ("+sc" if sc else ""), (N * H * W, C, 256, 3, 1, 1), fl, 2.0 * N * H * W * (C + 256)
    if "wgrad" in name:
        if "M" in A:                                       # 16-bit wgrad over M output pixels
            M, C, K, KH, KW = A["M"], A["C"], A["K"], A["KH"], A["KW"]
            groups = A.get("groups", 1) or 1
            Ct = C * groups
            fl = 2.0 * M * K * KH * KW * C
            by = M * (eb("x_dtype") * Ct + eb("dy_dtype") * K) + 4.0 * K * KH * KW * C
            return "wgrad", (M, Ct, K, KH, 1, groups), fl, by
        if "nlev" in A:
            return "wgrad32ml", (0, A.get("C", 0), A.get("K", 0), A.get("KH", 0), 1, 1), 0.0, 0.0
        N, H, W, C, K, KH, KW, OH, OW = (A[k] for k in ("N", "H", "W", "C", "K", "KH", "KW", "OH", "OW"))
        return "wgrad32", (N * OH * OW, C, K, KH, A.get("stride", 1), 1), 2.0 * N * OH * OW * K * KH * KW * C, 4.0 * N * (H * W * C + OH * OW * K)
    if "stem" in name:
        N, H, W, K, OH, OW = (A[k] for k in ("N", "H", "W", "K", "OH", "OW"))
        return "stem", (N * OH * OW, 3, K, 7, 2, 1), 2.0 * N * OH * OW * K * 147, 2.0 * N * ((H + 6) * (W + 8) * 4 + OH * OW * K)
    groups = A.get("groups", 1) or 1
    if "nlev" in A:                                        # multi-level (level-first buffers)
        N, C, K, KH, KW = A["N"], A["C"], A["K"], A["KH"], A["KW"]
        P = N * sum(h * w for h, w in zip(A["H_host"], A["W_host"]))
        Ct = C * groups
        fl = 2.0 * P * K * KH * KW * C
        xb = eb("x_dtype") if "x_dtype" in A else 4
        yb = eb("y_dtype") if "y_dtype" in A else 4
        by = P * (xb * Ct + yb * K * (1 + bool(A.get("residual")) + bool(A.get("accumulate")))) + 2.0 * K * KH * KW * C
        return "ml", (P, Ct, K, KH, 1, groups), fl, by
    N, H, W, C, K, KH, KW, OH, OW = (A[k] for k in ("N", "H", "W", "C", "K", "KH", "KW", "OH", "OW"))
    dil = A.get("in_dil", 1) or 1
    stride = A.get("stride", 1)
    xb = eb("x_dtype") if "x_dtype" in A else 4
    yb = eb("y_dtype") if "y_dtype" in A else 4
    fl = 2.0 * N * OH * OW * K * KH * KW * C / (dil * dil)
    extra = bool(A.get("residual")) + bool(A.get("accumulate")) + bool(A.get("mask")) + bool(A.get("post_mask"))
    bits = (bool(A.get("relu_bits")) + bool(A.get("mask_bits")) + bool(A.get("post_mask_bits"))) * K / 8.0
    by = xb * N * H * W * C + N * OH * OW * (yb * K * (1 + extra) + bits) + 2.0 * K * KH * KW * C
    tag = "conv" + ("+res" if A.get("residual") else "") + ("+mask" if (A.get("mask") or A.get("mask_bits")) else "") + \
          ("+pmask" if (A.get("post_mask") or A.get("post_mask_bits")) else "")
    return tag, (N * OH * OW, C, K, KH, stride if dil == 1 else -dil, 1), fl, by


def main():
    argv = [a for a in sys.argv[1:]]
    steps = 2
    recs = []
    first = {}
    enabled = [False]
    orig_call = hip.call

    keep_alive = []

    def own_host_arrays(name, args):
        """host int arrays (`*_host`, nlev entries) are only valid during the call: copy them, so the replay can pass them again"""
        names = NAMES[name]
        if "nlev" not in names:
            return args, ()
        nlev = val(args[names.index("nlev")])
        args, dims = list(args), []
        for i, n in enumerate(names):
            if n.endswith("_host") and isinstance(args[i], ctypes.c_void_p) and args[i].value:
                src = ctypes.cast(args[i], ctypes.POINTER(ctypes.c_int))
                arr = (ctypes.c_int * nlev)(*[src[j] for j in range(nlev)])
                keep_alive.append(arr)
                args[i] = arr
                dims.append(tuple(arr))
        return tuple(args), tuple(dims)

    def call(name, *args):
        if enabled[0] and name in CONV:
            mine, dims = own_host_arrays(name, args)
            key = (name, dims) + tuple(val(a) if not isinstance(a, ctypes.c_void_p) else bool(a.value) for a in args[:-1]
                                       if not isinstance(a, ctypes.Array))
            recs.append(key)
            if key not in first:
                first[key] = (name, mine)
        orig_call(name, *args)

    hip.call = call

    class T(bench.ConvTimer):
        def __setattr__(self, k, v):
            object.__setattr__(self, k, v)
            if k == "enabled":
                enabled[0] = bool(v)
    bench.ConvTimer = T
    sys.argv = ["bench.py", "--steps", str(steps), "--warmup", "2", "--no-cpu-baseline", "--no-f32", "--no-rcnn", "--timed-only"] + argv
    out = sys.stdout
    sys.stdout = sys.stderr            # the bench's JSON line goes to stderr; the table alone to stdout
    bench.main()
    sys.stdout = out
    torch.cuda.synchronize()
    hip.call = orig_call
    # level-first pixel count per image of this run (5 FPN levels of the padded canvas)
    levels_hw = getattr(bench, "LEVELS_HW_SUM", None) or 22400
    REP = 20
    iso = {}
    stream0 = hip._stream()
    for key, (name, args) in first.items():
        a = list(args)
        a[-1] = stream0                 # replay everything on the current stream
        orig_call(name, *a)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            orig_call(name, *a)
        e1.record()
        torch.cuda.synchronize()
        iso[key] = e0.elapsed_time(e1) / REP * 1e3   # us
    agg = collections.OrderedDict()
    for key in recs:
        name, args = first[key]
        kind, shape, fl, by = describe(name, args, levels_hw)
        entry = name.replace("utv2_conv2d_", "").replace("utv2_", "")
        a = agg.setdefault((kind, entry) + shape, [0, 0.0, fl, by])
        a[0] += 1
        a[1] += iso[key]
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for v in agg.values()) / steps / 1e3
    floor_tot = 0.0
    print("# per-shape replay of every conv launch of %d timed steps; args: %s" % (steps, " ".join(argv) or "(FCOS 4+4 f16 default)"))
    print("# sum of isolated conv time: %.2f ms / step over %d launches / step" % (tot, len(recs) // steps))
    print("%-16s %-22s %8s %5s %5s %2s %2s %2s | %5s %8s %8s %7s %7s %8s %5s %5s" % (
        "kind", "entry", "M", "C", "K", "k", "s", "g", "n/stp", "us", "ms/step", "TF/s", "GB/s", "floor_us", "bound", "pct"))
    for key, (n, us, fl, by) in rows:
        kind, entry = key[0], key[1]
        M, C, K, k, s, g = key[2:]
        per = us / n
        f_m, f_h = fl / 1000e12 * 1e6, by / 5e12 * 1e6
        floor = max(f_m, f_h)
        floor_tot += floor * n / steps / 1e3
        print("%-16s %-22s %8d %5d %5d %2d %2d %2d | %5.1f %8.1f %8.3f %7.0f %7.0f %8.1f %5s %4.1f%%" % (
            kind, entry, M, C, K, k, s, g, n / steps, per, us / steps / 1e3, fl / per / 1e6 if per else 0, by / per / 1e3 if per else 0,
            floor, "mfma" if f_m >= f_h else "hbm", 100 * us / steps / 1e3 / tot))
    print("# sum of floors (max(FLOP / 1000 TF/s, bytes / 5 TB/s) per launch): %.2f ms / step" % floor_tot)


if __name__ == "__main__":
    main()
