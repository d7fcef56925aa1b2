"""Weak / strong views of the two-crop mapper on the GPU.

Random decisions are drawn on the host (a numpy Generator owned by the mapper), the pixels are produced by the HIP kernels of
csrc/augment.hip, bit-exact to the Pillow arithmetic the reference runs on the CPU:
  weak   = Detectron2 `utils.build_augmentation(cfg, True)` (dataset_mapper.py:40): ResizeShortestEdge + RandomFlip;
  strong = `build_strong_augmentation` (data/detection_utils.py:8-46) on the weak view."""
import math

import numpy as np
import torch

from .. import hip

ERASERS = ((0.7, (0.05, 0.2), (0.3, 3.3)), (0.5, (0.02, 0.2), (0.1, 6.0)), (0.3, (0.02, 0.2), (0.05, 8.0)))  # detection_utils.py:29-37
JITTER = dict(p=0.8, brightness=(0.6, 1.4), contrast=(0.6, 1.4), saturation=(0.6, 1.4), hue=(-0.1, 0.1))     # ColorJitter(.4, .4, .4, .1)


class ResizeShortestEdge:
    """Detectron2 ResizeShortestEdge: short edge sampled from `short_edge_length` ("range": [min, max]; "choice": one of), long edge
    capped at max_size, sizes rounded half up."""

    def __init__(self, short_edge_length, max_size, sample_style="range"):
        assert sample_style in ("range", "choice"), sample_style
        if isinstance(short_edge_length, int):
            short_edge_length = (short_edge_length, short_edge_length)
        if sample_style == "range":
            assert len(short_edge_length) == 2, "short_edge_length must be two values using 'range' sample style."
        self.short_edge_length, self.max_size, self.is_range = tuple(short_edge_length), max_size, sample_style == "range"

    def get_params(self, rng, h, w):
        if self.is_range:
            size = int(rng.integers(self.short_edge_length[0], self.short_edge_length[1] + 1))
        else:
            size = int(rng.choice(self.short_edge_length))
        if size == 0:
            return h, w
        scale = size * 1.0 / min(h, w)
        newh, neww = (size, scale * w) if h < w else (scale * h, size)
        if max(newh, neww) > self.max_size:
            scale = self.max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return int(newh + 0.5), int(neww + 0.5)


class RandomCrop:
    """Detectron2 `T.RandomCrop(crop_type, crop_size)` [D2-recall], which the reference's mapper puts in front of the weak augmentation
    when INPUT.CROP.ENABLED (data/dataset_mapper.py:38-41): crop size by type ("relative": fractions of (h, w); "relative_range":
    fractions drawn uniformly from [crop_size, 1]; "absolute": pixels, capped at the image; "absolute_range": both sides drawn from
    [crop_size[0], crop_size[1]] capped at the image), sizes rounded half up, the corner uniform over the positions that fit."""

    def __init__(self, crop_type, crop_size):
        assert crop_type in ("relative_range", "relative", "absolute", "absolute_range"), crop_type
        self.crop_type, self.crop_size = crop_type, tuple(crop_size)

    def get_crop_size(self, rng, h, w):
        if self.crop_type == "relative":
            ch, cw = self.crop_size
            return int(h * ch + 0.5), int(w * cw + 0.5)
        if self.crop_type == "relative_range":
            cs = np.asarray(self.crop_size, dtype=np.float32)
            ch, cw = cs + rng.random(2) * (1 - cs)
            return int(h * ch + 0.5), int(w * cw + 0.5)
        if self.crop_type == "absolute":
            return min(int(self.crop_size[0]), h), min(int(self.crop_size[1]), w)
        assert self.crop_size[0] <= self.crop_size[1]
        ch = int(rng.integers(min(h, int(self.crop_size[0])), min(h, int(self.crop_size[1])) + 1))
        cw = int(rng.integers(min(w, int(self.crop_size[0])), min(w, int(self.crop_size[1])) + 1))
        return ch, cw

    def get_params(self, rng, h, w):
        """(x0, y0, crop_w, crop_h) of the CropTransform"""
        ch, cw = self.get_crop_size(rng, h, w)
        assert h >= ch and w >= cw, "Shape computation in RandomCrop has bugs."
        y0 = int(rng.integers(h - ch + 1))
        x0 = int(rng.integers(w - cw + 1))
        return x0, y0, cw, ch


def build_weak_augmentation(cfg, is_train=True):
    """(ResizeShortestEdge, flip probability) of Detectron2's build_augmentation for this cfg"""
    if is_train:
        rs = ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TRAIN, cfg.INPUT.MAX_SIZE_TRAIN, cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING)
        flip = 0.5 if cfg.INPUT.RANDOM_FLIP == "horizontal" else 0.0
        if cfg.INPUT.RANDOM_FLIP not in ("horizontal", "none"):
            raise NotImplementedError("INPUT.RANDOM_FLIP %r" % (cfg.INPUT.RANDOM_FLIP,))
    else:
        rs = ResizeShortestEdge(cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, "choice")
        flip = 0.0
    return rs, flip


def transform_boxes(boxes, h, w, newh, neww, flip):
    """XYXY boxes of the h x w image through ResizeTransform (+ HFlipTransform): Detectron2 transform_instance_annotations semantics -
    apply to the corners, take min / max, clip to the new image.  float64 numpy in, float32 [G, 4] out."""
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4).copy()
    b[:, [0, 2]] *= neww * 1.0 / w
    b[:, [1, 3]] *= newh * 1.0 / h
    if flip:
        x1 = neww - b[:, 2]
        x2 = neww - b[:, 0]
        b[:, 0], b[:, 2] = x1, x2
    b = np.clip(b, 0, None)
    b = np.minimum(b, np.array([neww, newh, neww, newh], dtype=np.float64))
    return b.astype(np.float32)


def sample_strong_params(rng, h, w):
    """The random decisions of the strong view for one h x w image, drawn in a fixed order: jitter gate, op order, brightness, contrast,
    saturation, hue; grayscale gate; blur gate, sigma; then per eraser: gate, up to 10 (area, log-ratio) attempts, position
    (torchvision RandomApply / ColorJitter / RandomGrayscale / RandomErasing semantics)."""
    p = {"jitter": bool(rng.random() < JITTER["p"]), "order": [int(v) for v in rng.permutation(4)]}
    for k in ("brightness", "contrast", "saturation", "hue"):
        p[k] = float(rng.uniform(*JITTER[k]))
    p["gray"] = bool(rng.random() < 0.2)
    p["blur"] = bool(rng.random() < 0.5)
    p["sigma"] = float(rng.uniform(0.1, 2.0))  # GaussianBlur([0.1, 2.0]), augmentation_impl.py:19-21
    p["erase"] = []
    for prob, scale, ratio in ERASERS:
        rect = None
        if rng.random() < prob:
            for _ in range(10):
                ea = h * w * float(rng.uniform(scale[0], scale[1]))
                ar = math.exp(float(rng.uniform(math.log(ratio[0]), math.log(ratio[1]))))
                eh, ew = int(round(math.sqrt(ea * ar))), int(round(math.sqrt(ea / ar)))
                if eh < h and ew < w:
                    rect = (int(rng.integers(0, h - eh + 1)), int(rng.integers(0, w - ew + 1)), eh, ew)
                    break
        p["erase"].append(rect)
    return p


def apply_weak(img, newh, neww, flip):
    """uint8 [H][W][3] device tensor -> resized (+ flipped) uint8 [newh][neww][3]"""
    return hip.aug_resize(img, newh, neww, flip)


def apply_strong(img, p, noises=None, generator=None):
    """Strong view of a uint8 [H][W][3] device tensor under the decisions `p`; the input is not modified.  noises[k] (float32
    [3][h][w], the normal draw of eraser k) may be injected; otherwise it is drawn on the device from `generator`."""
    x = img.clone()
    if p["jitter"]:
        for fn in p["order"]:
            if fn == 0:
                hip.aug_brightness(x, p["brightness"])
            elif fn == 1:
                hip.aug_contrast(x, p["contrast"])
            elif fn == 2:
                hip.aug_saturation(x, p["saturation"])
            else:
                hip.aug_hue(x, p["hue"])
    if p["gray"]:
        hip.aug_grayscale(x)
    if p["blur"]:
        x = hip.aug_gaussian_blur(x, p["sigma"])
    for k, rect in enumerate(p["erase"]):
        if rect is None:
            continue
        i, j, h, w = rect
        nz = noises[k] if noises is not None and noises[k] is not None else torch.randn((3, h, w), device=x.device, generator=generator)
        hip.aug_erase(x, i, j, h, w, nz)
    return x
