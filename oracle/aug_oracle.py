"""TEST INFRASTRUCTURE ONLY - CPU restatement (numpy) of the image arithmetic on the reference's two-crop data path
(SURVEY 8f rank 1).  Nothing in the product imports this file.

The reference does its augmentation with third-party code that is NOT in /root/reference:
  * weak view  : Detectron2 `ResizeShortestEdge` + `RandomFlip` (`ubteacher/data/dataset_mapper.py:40,97-99`);
                 `ResizeTransform.apply_image` calls `PIL.Image.resize(..., BILINEAR)` on the uint8 array;
  * strong view: `ubteacher/data/detection_utils.py:8-46` = torchvision `ColorJitter(.4,.4,.4,.1)` p .8, `RandomGrayscale` p .2,
                 the reference's own `GaussianBlur` (`data/transforms/augmentation_impl.py:7-22`, `PIL.ImageFilter.GaussianBlur`)
                 p .5, `ToTensor`, 3 x `RandomErasing(value="random")`, `ToPILImage`; on PIL images torchvision's colour
                 ops are `PIL.ImageEnhance.{Brightness,Contrast,Color}`, an HSV round trip and `convert("L")`.
So the arithmetic that decides the pixels is Pillow's (this image: Pillow 12.2.0; libImaging Resample.c, Blend.c, Convert.c,
BoxBlur.c).  Each function below restates the published algorithm of one of those C routines; `tests/test_aug_oracle.py` pins
every one of them BIT-EXACTLY against Pillow itself (importable here and on the GPU box), the HSV pair exhaustively.
torchvision is not installed: the parameter SAMPLING of ColorJitter / RandomErasing is restated from its documented behaviour
(`sample_strong_params`) and is "parity unpinned"; the pixel arithmetic, given the parameters, is pinned."""
import math

import numpy as np

f32 = np.float32
PRECISION_BITS = 32 - 8 - 2  # Resample.c


# ---- Resample.c: precompute_coeffs + normalize_coeffs_8bpc + ImagingResample{Horizontal,Vertical}_8bpc, bilinear filter ----------
def resample_coeffs(in_size, out_size):
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale  # bilinear support 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0:
                a = -a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def resize_bilinear(img, out_h, out_w):
    """PIL.Image.resize((out_w, out_h), BILINEAR) on a uint8 [H][W][C] array: horizontal pass, then vertical pass on its uint8 result."""
    H, W, C = img.shape
    x = img.astype(np.int64)
    if out_w != W:
        kk, b = resample_coeffs(W, out_w)
        t = np.zeros((H, out_w, C), dtype=np.int64)
        for xx in range(out_w):
            x0, n = b[xx]
            acc = np.full((H, C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for k in range(n):
                acc += x[:, x0 + k, :] * kk[xx, k]
            t[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        x = t
    if out_h != H:
        kk, b = resample_coeffs(H, out_h)
        t = np.zeros((out_h, x.shape[1], C), dtype=np.int64)
        for yy in range(out_h):
            y0, n = b[yy]
            acc = np.full((x.shape[1], C), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for k in range(n):
                acc += x[y0 + k] * kk[yy, k]
            t[yy] = np.clip(acc >> PRECISION_BITS, 0, 255)
        x = t
    return x.astype(np.uint8)


# ---- Convert.c rgb2l, Blend.c ImagingBlend, ImageEnhance -------------------------------------------------------------------------
def to_l(img):
    r, g, b = (img[..., i].astype(np.int64) for i in range(3))
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(deg, img, alpha):
    """ImagingBlend(in1=deg, in2=img, float alpha): float32 arithmetic, truncation; clipping only outside [0, 1]."""
    a = f32(alpha)
    t = deg.astype(np.int32).astype(f32) + a * (img.astype(np.int32) - deg.astype(np.int32)).astype(f32)
    if 0.0 <= float(a) <= 1.0:
        return t.astype(np.int32).astype(np.uint8)
    return np.where(t <= 0, 0, np.where(t >= 255, 255, t.astype(np.int32))).astype(np.uint8)


def adjust_brightness(img, f):
    return blend(np.zeros_like(img), img, f)


def gray_mean(img):
    """int(ImageStat.Stat(img.convert("L")).mean[0] + 0.5)"""
    return int(to_l(img).astype(np.float64).sum() / (img.shape[0] * img.shape[1]) + 0.5)


def adjust_contrast(img, f):
    return blend(np.full_like(img, gray_mean(img)), img, f)


def adjust_saturation(img, f):
    return blend(np.repeat(to_l(img)[..., None], 3, axis=2), img, f)


def to_grayscale3(img):
    return np.repeat(to_l(img)[..., None], 3, axis=2)


# ---- Convert.c rgb2hsv / hsv2rgb (float divisions, double adds / fmod), torchvision F_pil.adjust_hue ---------------------------------
def rgb2hsv(img):
    r = img[..., 0].astype(np.int32); g = img[..., 1].astype(np.int32); b = img[..., 2].astype(np.int32)
    maxc = np.maximum(r, np.maximum(g, b)); minc = np.minimum(r, np.minimum(g, b))
    gray = maxc == minc
    cr = np.where(gray, 1, maxc - minc).astype(f32)
    s = (maxc - minc).astype(f32) / np.where(maxc == 0, 1, maxc).astype(f32)
    rc = (maxc - r).astype(f32) / cr; gc = (maxc - g).astype(f32) / cr; bc = (maxc - b).astype(f32) / cr
    D = np.float64
    h = np.where(r == maxc, bc.astype(D) - gc.astype(D),
                 np.where(g == maxc, 2.0 + rc.astype(D) - bc.astype(D), 4.0 + gc.astype(D) - rc.astype(D))).astype(f32)
    h = np.fmod(h.astype(D) / 6.0 + 1.0, 1.0).astype(f32)
    uh = np.clip((h.astype(D) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(D) * 255.0).astype(np.int32), 0, 255)
    return np.stack([np.where(gray, 0, uh), np.where(gray, 0, us), maxc], -1).astype(np.uint8)


def hsv2rgb(hsv):
    hh = hsv[..., 0].astype(f32); ss = hsv[..., 1]; vv = hsv[..., 2].astype(np.int32)
    x = hh * f32(6.0) / f32(255.0)
    i = np.floor(x)
    f = x - i
    fs = ss.astype(f32) / f32(255.0)
    vf = vv.astype(f32)

    def rnd(a):
        return np.clip(np.floor(a + f32(0.5)).astype(np.int32), 0, 255)
    p = rnd(vf * (f32(1.0) - fs)); q = rnd(vf * (f32(1.0) - fs * f)); t = rnd(vf * (f32(1.0) - fs * (f32(1.0) - f)))
    ii = i.astype(np.int32) % 6
    rr = np.choose(ii, [vv, q, p, p, t, vv]); gg = np.choose(ii, [t, vv, vv, q, p, p]); bb = np.choose(ii, [p, p, t, vv, vv, q])
    z = ss == 0
    return np.stack([np.where(z, vv, rr), np.where(z, vv, gg), np.where(z, vv, bb)], -1).astype(np.uint8)


def hue_shift_u8(hue_factor):
    """np.uint8(hue_factor * 255) of torchvision's F_pil.adjust_hue: C truncation, two's-complement wrap"""
    return int(math.trunc(hue_factor * 255)) & 255


def adjust_hue(img, hue_factor):
    hsv = rgb2hsv(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + hue_shift_u8(hue_factor)).astype(np.uint8)
    return hsv2rgb(hsv)


# ---- BoxBlur.c: _gaussian_blur_radius, ImagingLineBoxBlur8/32, ImagingGaussianBlur(passes = 3) -------------------------------------------
def box_radius(radius, passes=3):
    radius = f32(radius)
    sigma2 = f32(radius * radius / f32(passes))
    L = f32(math.sqrt(12.0 * float(sigma2) + 1.0))
    l = f32(math.floor((float(L) - 1.0) / 2.0))
    a = f32(f32(2 * l + 1) * f32(f32(l * f32(l + 1)) - f32(3 * sigma2)))
    a = f32(a / f32(6 * f32(sigma2 - f32(f32(l + 1) * f32(l + 1)))))
    return f32(l + a)


def box_weights(float_radius):
    fr = f32(float_radius)
    r = int(fr)
    ww = int(f32(f32(1 << 24) / f32(fr * 2 + 1)))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    return r, ww, fw


def _box_lines(x, float_radius):
    r, ww, fw = box_weights(float_radius)
    n = x.shape[1]
    idx = np.arange(n)
    acc = np.zeros_like(x)
    for k in range(-r, r + 1):
        acc += x[:, np.clip(idx + k, 0, n - 1)]
    far = x[:, np.clip(idx - r - 1, 0, n - 1)] + x[:, np.clip(idx + r + 1, 0, n - 1)]
    bulk = (acc * ww + far * fw) & 0xFFFFFFFF
    return ((bulk + (1 << 23)) & 0xFFFFFFFF) >> 24


def gaussian_blur(img, radius, passes=3):
    """PIL.ImageFilter.GaussianBlur(radius): `passes` box blurs along x, then `passes` along y, edge pixels replicated."""
    br = box_radius(radius, passes)
    H, W, C = img.shape
    t = img.astype(np.int64).transpose(0, 2, 1).reshape(H * C, W)
    for _ in range(passes):
        t = _box_lines(t, br)
    x = t.reshape(H, C, W).transpose(0, 2, 1)
    t = x.transpose(1, 2, 0).reshape(W * C, H)
    for _ in range(passes):
        t = _box_lines(t, br)
    return t.reshape(W, C, H).transpose(2, 0, 1).astype(np.uint8)


# ---- torchvision ToTensor -> RandomErasing(value="random") -> ToPILImage ---------------------------------------------------------------
def erase(img, i, j, h, w, noise):
    """img uint8 [H][W][3]; noise float32 [3][h][w] (the normal_() draw).  ToTensor / ToPILImage round-trip uint8 exactly
    ((x / 255) * 255 truncates back to x for all 256 values); erased pixels become `(noise * 255).byte()` = truncation with
    two's-complement wrap."""
    out = img.copy()
    v = (noise.astype(f32) * f32(255.0))
    out[i:i + h, j:j + w, :] = (np.trunc(v).astype(np.int64) & 255).astype(np.uint8).transpose(1, 2, 0)
    return out


# ---- parameter sampling (restated; torchvision / Detectron2 not installed: unpinned) ------------------------------------------------------
def sample_resize_flip(rng, h, w, min_size, max_size, sample_style="range", flip_prob=0.5):
    """Detectron2 ResizeShortestEdge.get_transform + RandomFlip: (new_h, new_w, do_flip)."""
    if sample_style == "range":
        size = int(rng.integers(min_size[0], min_size[1] + 1))
    else:
        size = int(rng.choice(list(min_size)))
    scale = size * 1.0 / min(h, w)
    newh, neww = (size, scale * w) if h < w else (scale * h, size)
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5), bool(rng.random() < flip_prob)


ERASERS = ((0.7, (0.05, 0.2), (0.3, 3.3)), (0.5, (0.02, 0.2), (0.1, 6.0)), (0.3, (0.02, 0.2), (0.05, 8.0)))  # detection_utils.py:29-37


def sample_strong_params(rng, h, w):
    """The random decisions of build_strong_augmentation (detection_utils.py:19-41) for one h x w image, drawn from `rng`
    (numpy Generator) in this fixed order: jitter gate, order, b, c, s, hue; gray gate; blur gate, sigma; per eraser: gate, up to
    10 (area, log-ratio) attempts, position."""
    p = {}
    p["jitter"] = bool(rng.random() < 0.8)
    p["order"] = [int(v) for v in rng.permutation(4)]
    p["brightness"] = float(rng.uniform(0.6, 1.4))
    p["contrast"] = float(rng.uniform(0.6, 1.4))
    p["saturation"] = float(rng.uniform(0.6, 1.4))
    p["hue"] = float(rng.uniform(-0.1, 0.1))
    p["gray"] = bool(rng.random() < 0.2)
    p["blur"] = bool(rng.random() < 0.5)
    p["sigma"] = float(rng.uniform(0.1, 2.0))
    p["erase"] = []
    for prob, scale, ratio in ERASERS:
        rect = None
        if rng.random() < prob:
            area = h * w
            for _ in range(10):
                ea = area * float(rng.uniform(scale[0], scale[1]))
                ar = math.exp(float(rng.uniform(math.log(ratio[0]), math.log(ratio[1]))))
                eh, ew = int(round(math.sqrt(ea * ar))), int(round(math.sqrt(ea / ar)))
                if not (eh < h and ew < w):
                    continue
                rect = (int(rng.integers(0, h - eh + 1)), int(rng.integers(0, w - ew + 1)), eh, ew)
                break
        p["erase"].append(rect)
    return p


def strong_augment(img, p, noises):
    """Apply build_strong_augmentation with the decisions `p`; noises[k]: float32 [3][h][w] for eraser k (None when not applied)."""
    x = img
    if p["jitter"]:
        for fn in p["order"]:
            if fn == 0:
                x = adjust_brightness(x, p["brightness"])
            elif fn == 1:
                x = adjust_contrast(x, p["contrast"])
            elif fn == 2:
                x = adjust_saturation(x, p["saturation"])
            else:
                x = adjust_hue(x, p["hue"])
    if p["gray"]:
        x = to_grayscale3(x)
    if p["blur"]:
        x = gaussian_blur(x, p["sigma"])
    for rect, nz in zip(p["erase"], noises):
        if rect is not None:
            x = erase(x, rect[0], rect[1], rect[2], rect[3], nz)
    return x
