"""HIP augmentation kernels (csrc/augment.hip, through the C-ABI) vs the CPU oracle (oracle/aug_oracle.py) and vs Pillow itself,
BIT-EXACT (uint8 results, integer / float32 arithmetic restated from Pillow's C code): the weak view (resize + flip) and every op of the
strong view (ubteacher/data/detection_utils.py:8-46)."""
import numpy as np
import pytest
import torch

from oracle import aug_oracle as A

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rand_img(seed, h, w):
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


def dev(img):
    return torch.from_numpy(np.ascontiguousarray(img)).to(DEV)


@pytest.mark.parametrize("shape", [(37, 53, 20, 31), (37, 53, 80, 101), (64, 48, 64, 100), (50, 70, 33, 70), (50, 70, 50, 70),
                                   (480, 640, 800, 1067), (427, 640, 400, 600), (5, 7, 1, 1), (1, 1, 4, 3), (1200, 1600, 400, 533)])
@pytest.mark.parametrize("flip", [False, True])
def test_resize_flip_bit_exact(shape, flip):
    from ubteacher import hip
    from PIL import Image
    H, W, OH, OW = shape
    img = rand_img(1, H, W)
    ref = np.asarray(Image.fromarray(img).resize((OW, OH), Image.BILINEAR))   # what Detectron2's ResizeTransform.apply_image runs
    if H * W <= 500 * 700:
        assert np.array_equal(A.resize_bilinear(img, OH, OW), ref)
    if flip:
        ref = ref[:, ::-1]
    out = hip.aug_resize(dev(img), OH, OW, flip).cpu().numpy()
    assert out.shape == ref.shape and np.array_equal(out, ref)


@pytest.mark.parametrize("f", [0.0, 0.6, 0.83, 1.0, 1.2, 1.3999, 1.4])
def test_colour_ops_bit_exact(f):
    from ubteacher import hip
    img = rand_img(2, 97, 131)
    assert np.array_equal(hip.aug_brightness(dev(img), f).cpu().numpy(), A.adjust_brightness(img, f))
    assert np.array_equal(hip.aug_contrast(dev(img), f).cpu().numpy(), A.adjust_contrast(img, f))
    assert np.array_equal(hip.aug_saturation(dev(img), f).cpu().numpy(), A.adjust_saturation(img, f))
    assert np.array_equal(hip.aug_grayscale(dev(img)).cpu().numpy(), A.to_grayscale3(img))


@pytest.mark.parametrize("hf", [-0.1, -0.0371, 0.0, 0.052, 0.1])
def test_hue_bit_exact_exhaustive_colours(hf):
    from ubteacher import hip
    r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(0, 256, 5), indexing="ij")
    img = np.stack([r, g, b], -1).reshape(256, -1, 3).astype(np.uint8)
    assert np.array_equal(hip.aug_hue(dev(img), hf).cpu().numpy(), A.adjust_hue(img, hf))
    with pytest.raises(ValueError):
        hip.aug_hue(dev(img), 0.6)


@pytest.mark.parametrize("radius", [0.1, 0.35, 0.77, 1.0, 1.3, 1.62, 1.99, 2.0, 3.7])
def test_gaussian_blur_bit_exact(radius):
    from ubteacher import hip
    from PIL import Image, ImageFilter
    for seed, (h, w) in enumerate([(61, 83), (3, 2), (200, 301)]):
        img = rand_img(10 + seed, h, w)
        ref = np.asarray(Image.fromarray(img).filter(ImageFilter.GaussianBlur(radius=radius)))
        assert np.array_equal(hip.aug_gaussian_blur(dev(img), radius).cpu().numpy(), ref)


def test_erase_and_layout():
    from ubteacher import hip
    img = rand_img(6, 40, 50)
    noise = (np.random.default_rng(7).standard_normal((3, 11, 17)) * 1.5).astype(np.float32)
    out = hip.aug_erase(dev(img), 5, 9, 11, 17, torch.from_numpy(noise).to(DEV))
    ref = A.erase(img, 5, 9, 11, 17, noise)
    assert np.array_equal(out.cpu().numpy(), ref)
    assert np.array_equal(hip.aug_to_chw(out).cpu().numpy(), ref.transpose(2, 0, 1))
    with pytest.raises(RuntimeError):
        hip.aug_erase(dev(img), 35, 9, 11, 17, torch.from_numpy(noise).to(DEV))   # rectangle leaves the image


@pytest.mark.parametrize("seed", range(6))
def test_strong_view_chain_bit_exact(seed):
    """the whole strong view with the same random decisions: product (GPU) == oracle (Pillow arithmetic)"""
    from ubteacher.data.transforms import apply_strong
    rng = np.random.default_rng(100 + seed)
    h, w = int(rng.integers(60, 140)), int(rng.integers(60, 180))
    img = rand_img(200 + seed, h, w)
    p = A.sample_strong_params(np.random.default_rng(300 + seed), h, w)
    p["jitter"] = True if seed < 4 else p["jitter"]
    p["blur"] = True if seed % 2 == 0 else p["blur"]
    p["gray"] = seed == 3
    noises = [None if q is None else np.random.default_rng(400 + seed + k).standard_normal((3, q[2], q[3])).astype(np.float32)
              for k, q in enumerate(p["erase"])]
    ref = A.strong_augment(img, p, noises)
    out = apply_strong(dev(img), p, [None if n is None else torch.from_numpy(n).to(DEV) for n in noises])
    assert np.array_equal(out.cpu().numpy(), ref)
