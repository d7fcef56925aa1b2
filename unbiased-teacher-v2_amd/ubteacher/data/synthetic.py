"""Synthetic COCO-shaped two-crop batches (SURVEY.md 8d "Synthetic inputs").

Output contract = the reference loader's (ubteacher/data/common.py:158-163,
dataset_mapper.py:139-157): each iteration yields
    (label_strong, label_weak, unlabel_strong, unlabel_weak)
four lists of dicts {image: uint8 [3,H,W] BGR, height, width, instances: Instances(gt_boxes, gt_classes)}.
Images are generated once (seeded) and kept resident on the device; the strong view is the weak
view with three noise-filled erase rectangles (area 2-20 %).
"""
import math

import numpy as np
import torch

from ..d2.structures import Boxes, Instances
from ..utils import comm


def make_image(rng, h, w):
    return torch.from_numpy(rng.integers(0, 256, size=(3, h, w), dtype=np.uint8))


def strong_view(rng, img):
    out = img.clone()
    _, h, w = img.shape
    for _ in range(3):
        area = rng.uniform(0.02, 0.2) * h * w
        ar = math.exp(rng.uniform(math.log(0.3), math.log(3.3)))
        eh, ew = int(round(math.sqrt(area * ar))), int(round(math.sqrt(area / ar)))
        eh, ew = min(eh, h), min(ew, w)
        y0 = int(rng.integers(0, h - eh + 1))
        x0 = int(rng.integers(0, w - ew + 1))
        out[:, y0:y0 + eh, x0:x0 + ew] = torch.from_numpy(rng.integers(0, 256, size=(3, eh, ew), dtype=np.uint8))
    return out


def make_gt(rng, h, w, num_classes=80, max_boxes=15):
    g = int(rng.integers(1, max_boxes + 1))
    cx = rng.uniform(0, w, size=g)
    cy = rng.uniform(0, h, size=g)
    bw = np.exp(rng.uniform(math.log(16), math.log(600), size=g))
    bh = np.exp(rng.uniform(math.log(16), math.log(600), size=g))
    x1 = np.clip(cx - bw / 2, 0, w - 1)
    y1 = np.clip(cy - bh / 2, 0, h - 1)
    x2 = np.clip(cx + bw / 2, x1 + 2, w)
    y2 = np.clip(cy + bh / 2, y1 + 2, h)
    boxes = torch.tensor(np.stack([x1, y1, x2, y2], 1), dtype=torch.float32)
    classes = torch.from_numpy(rng.integers(0, num_classes, size=g)).long()
    inst = Instances((h, w))
    inst.gt_boxes = Boxes(boxes)
    inst.gt_classes = classes
    return inst


def resize_shortest_edge_size(rng, min_range, max_size, aspect=4.0 / 3.0):
    """(h, w) of a landscape image of the given aspect ratio after Detectron2's ResizeShortestEdge with MIN_SIZE_TRAIN_SAMPLING "range":
    the short side drawn uniformly from min_range (the UTv2 recipes: (400, 1200)), the long side capped at MAX_SIZE_TRAIN (1333)"""
    s = int(rng.integers(min_range[0], min_range[1] + 1))
    h, w = float(s), s * aspect
    if w > max_size:
        h, w = h * max_size / w, float(max_size)
    return int(h + 0.5), int(w + 0.5)


class SyntheticTwoCropLoader:
    def __init__(self, cfg, height=800, width=1333, seed=0, device=None, num_batches=1, ragged=None):
        """ragged = (min_lo, min_hi, max_size): every image gets its own ResizeShortestEdge size (the reference recipes train at
        INPUT.MIN_SIZE_TRAIN (400, 1200) "range", configs/utv2_*_r50.yaml): the labeled and the unlabeled lists then pad to different
        canvases and the student's two passes cannot be fused (engine/trainer.py) - use with num_batches > 1"""
        ws = comm.get_world_size()
        bl, bu = cfg.SOLVER.IMG_PER_BATCH_LABEL, cfg.SOLVER.IMG_PER_BATCH_UNLABEL
        assert bl % ws == 0 and bu % ws == 0, "batch must be divisible by world size (data/build.py:228-238)"
        self.bl, self.bu = bl // ws, bu // ws
        dev = torch.device(device if device is not None else cfg.MODEL.DEVICE)
        rng = np.random.default_rng(seed + 1000 * comm.get_rank())
        nc = cfg.MODEL.FCOS.NUM_CLASSES if "FCOS" in cfg.MODEL else 80
        self.batches = []
        for _ in range(num_batches):
            lq, lk, uq, uk = [], [], [], []
            for _ in range(self.bl):
                if ragged is not None:
                    height, width = resize_shortest_edge_size(rng, ragged[:2], ragged[2])
                weak = make_image(rng, height, width)
                strong = strong_view(rng, weak)
                gt = make_gt(rng, height, width, nc)
                lk.append({"image": weak.to(dev), "height": height, "width": width, "instances": gt})
                lq.append({"image": strong.to(dev), "height": height, "width": width, "instances": gt})
            for _ in range(self.bu):
                if ragged is not None:
                    height, width = resize_shortest_edge_size(rng, ragged[:2], ragged[2])
                weak = make_image(rng, height, width)
                strong = strong_view(rng, weak)
                uk.append({"image": weak.to(dev), "height": height, "width": width})
                uq.append({"image": strong.to(dev), "height": height, "width": width})
            self.batches.append((lq, lk, uq, uk))
        self._i = 0
        self.static_batches = num_batches == 1     # the same device tensors every iteration (engine.trainer.run_step_graph)

    def __iter__(self):
        return self

    def __next__(self):
        b = self.batches[self._i % len(self.batches)]
        self._i += 1
        # fresh dicts: the trainer deletes / adds keys in place (trainer.py:161-175)
        return tuple([dict(d) for d in part] for part in b)


class SyntheticTestLoader:
    """Fixed-length test loader (batch size 1 per iteration, like D2's build_detection_test_loader): dicts with `image`
    (already at the network input size), the ORIGINAL `height` / `width` the detections are rescaled to, `image_id` and
    the ground truth `instances` in original-image coordinates."""

    def __init__(self, cfg, num_images=8, height=800, width=1333, orig_scale=1.25, seed=123, device=None):
        dev = torch.device(device if device is not None else cfg.MODEL.DEVICE)
        rng = np.random.default_rng(seed)
        nc = cfg.MODEL.FCOS.NUM_CLASSES if "FCOS" in cfg.MODEL else 80
        self.items = []
        for i in range(num_images):
            oh, ow = int(round(height * orig_scale)), int(round(width * orig_scale))
            gt = make_gt(rng, oh, ow, nc)
            self.items.append([{"image": make_image(rng, height, width).to(dev), "height": oh, "width": ow, "image_id": i,
                                "instances": gt}])

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter([[dict(d) for d in b] for b in self.items])
