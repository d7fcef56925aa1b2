"""Pins oracle/aug_oracle.py BIT-EXACTLY against Pillow itself - the third-party code whose arithmetic decides the pixels of the
reference's weak / strong views (Detectron2 ResizeTransform -> PIL resize; torchvision's PIL colour ops; the reference's own
GaussianBlur = PIL.ImageFilter.GaussianBlur, data/transforms/augmentation_impl.py:7-22)."""
import numpy as np
import pytest

from oracle import aug_oracle as A

PIL = pytest.importorskip("PIL")
from PIL import Image, ImageEnhance, ImageFilter  # noqa: E402


def rand_img(seed, h, w):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


@pytest.mark.parametrize("shape", [(37, 53, 20, 31), (37, 53, 80, 101), (64, 48, 64, 100), (50, 70, 33, 70), (240, 320, 200, 267),
                                   (100, 120, 299, 97), (5, 7, 1, 1), (1, 1, 4, 3)])
def test_resize_bilinear_bit_exact(shape):
    H, W, OH, OW = shape
    img = rand_img(1, H, W)
    ref = np.asarray(Image.fromarray(img).resize((OW, OH), Image.BILINEAR))
    assert np.array_equal(A.resize_bilinear(img, OH, OW), ref)


@pytest.mark.parametrize("f", [0.0, 0.6, 0.83, 1.0, 1.2, 1.3999, 1.4])
def test_enhance_ops_bit_exact(f):
    img = rand_img(2, 97, 131)
    pil = Image.fromarray(img)
    assert np.array_equal(A.to_l(img), np.asarray(pil.convert("L")))
    assert np.array_equal(A.adjust_brightness(img, f), np.asarray(ImageEnhance.Brightness(pil).enhance(f)))
    assert np.array_equal(A.adjust_contrast(img, f), np.asarray(ImageEnhance.Contrast(pil).enhance(f)))
    assert np.array_equal(A.adjust_saturation(img, f), np.asarray(ImageEnhance.Color(pil).enhance(f)))


def test_hsv_round_trip_exhaustive():
    r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(0, 256, 5), indexing="ij")
    img = np.stack([r, g, b], -1).reshape(256, -1, 3).astype(np.uint8)
    assert np.array_equal(A.rgb2hsv(img), np.asarray(Image.fromarray(img).convert("HSV")))
    h, s, v = np.meshgrid(np.arange(256), np.arange(256), np.arange(0, 256, 3), indexing="ij")
    hsv = np.stack([h, s, v], -1).reshape(256, -1, 3).astype(np.uint8)
    ref = np.asarray(Image.frombytes("HSV", (hsv.shape[1], hsv.shape[0]), hsv.tobytes()).convert("RGB"))
    assert np.array_equal(A.hsv2rgb(hsv), ref)


@pytest.mark.parametrize("hf", [-0.1, -0.0371, 0.0, 0.052, 0.1])
def test_adjust_hue_bit_exact(hf):
    """torchvision F_pil.adjust_hue restated with PIL itself: HSV split, uint8 wrap-add on H, merge, back to RGB."""
    img = rand_img(3, 64, 80)
    h, s, v = Image.fromarray(img).convert("HSV").split()
    nh = np.array(h, dtype=np.uint8)
    nh = (nh.astype(np.int32) + (int(np.trunc(hf * 255)) & 255)).astype(np.uint8)
    ref = np.asarray(Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB"))
    assert np.array_equal(A.adjust_hue(img, hf), ref)


@pytest.mark.parametrize("radius", [0.1, 0.35, 0.5, 0.77, 1.0, 1.3, 1.62, 1.99, 2.0, 3.7])
def test_gaussian_blur_bit_exact(radius):
    img = rand_img(4, 61, 83)
    ref = np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius=radius)))
    assert np.array_equal(A.gaussian_blur(img, radius), ref)
    tiny = rand_img(5, 3, 2)  # narrower than the window
    assert np.array_equal(A.gaussian_blur(tiny, radius), np.asarray(Image.fromarray(tiny).filter(ImageFilter.GaussianBlur(radius=radius))))


def test_erase_matches_torch_totensor_topil_semantics():
    import torch
    img = rand_img(6, 40, 50)
    noise = (np.random.default_rng(7).standard_normal((3, 11, 17)) * 1.5).astype(np.float32)
    t = torch.from_numpy(img).permute(2, 0, 1).to(torch.float32).div(255)       # ToTensor
    t[:, 5:16, 9:26] = torch.from_numpy(noise)                                   # RandomErasing: img[..., i:i+h, j:j+w] = v
    ref = t.mul(255).byte().permute(1, 2, 0).numpy()                             # ToPILImage
    assert np.array_equal(A.erase(img, 5, 9, 11, 17, noise), ref)


def test_strong_chain_runs_and_is_deterministic():
    rng = np.random.default_rng(11)
    img = rand_img(8, 90, 120)
    outs = []
    for _ in range(2):
        r = np.random.default_rng(12)
        p = A.sample_strong_params(r, 90, 120)
        nz = [None if q is None else np.random.default_rng(13 + k).standard_normal((3, q[2], q[3])).astype(np.float32) for k, q in enumerate(p["erase"])]
        outs.append(A.strong_augment(img, p, nz))
    assert np.array_equal(outs[0], outs[1]) and outs[0].shape == img.shape
    nh, nw, fl = A.sample_resize_flip(rng, 480, 640, (400, 1200), 1333)
    assert 400 <= min(nh, nw) <= 1200 and max(nh, nw) <= 1333 and abs(nw / nh - 640 / 480) < 0.01
