"""Minimal dataset catalog (Detectron2 DatasetCatalog contract: name -> callable returning a list of dataset dicts) and a synthetic
COCO-shaped dataset (no image files exist in this environment).  A dataset dict carries either `file_name` (decoded on the host with
Pillow) or an in-memory `image` (uint8 [H][W][3], RGB), plus `height`, `width`, `image_id` and `annotations` = list of
{bbox: [x1, y1, x2, y2], bbox_mode: "XYXY_ABS", category_id, iscrowd}."""
import numpy as np


class _Catalog(dict):
    def register(self, name, func):
        assert callable(func), "You must register a function with `DatasetCatalog.register`!"
        assert name not in self, "Dataset '{}' is already registered!".format(name)
        self[name] = func

    def get(self, name):
        try:
            f = self[name]
        except KeyError as e:
            raise KeyError("Dataset '{}' is not registered! Available datasets are: {}".format(name, ", ".join(self.keys()))) from e
        return f()

    def remove(self, name):
        self.pop(name)


DatasetCatalog = _Catalog()

COCO_SHAPES = ((480, 640), (427, 640), (640, 480), (640, 427), (375, 500), (500, 375), (480, 640), (426, 640))


def synthetic_coco_dicts(num_images, seed=0, num_classes=80, max_boxes=8, with_annotations=True):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(num_images):
        h, w = COCO_SHAPES[int(rng.integers(0, len(COCO_SHAPES)))]
        # smooth-ish content (a blocky upsample of noise) so that resize / blur act on structure, plus per-pixel noise
        base = rng.integers(0, 256, (h // 16 + 1, w // 16 + 1, 3), dtype=np.uint8)
        img = np.kron(base, np.ones((16, 16, 1), dtype=np.uint8))[:h, :w]
        img = (img.astype(np.int16) + rng.integers(-20, 21, (h, w, 3))).clip(0, 255).astype(np.uint8)
        d = {"image": img, "height": h, "width": w, "image_id": i}
        if with_annotations:
            g = int(rng.integers(1, max_boxes + 1))
            cx, cy = rng.uniform(0, w, g), rng.uniform(0, h, g)
            bw, bh = np.exp(rng.uniform(np.log(12), np.log(w * 0.8), g)), np.exp(rng.uniform(np.log(12), np.log(h * 0.8), g))
            x1, y1 = np.clip(cx - bw / 2, 0, w - 2), np.clip(cy - bh / 2, 0, h - 2)
            x2, y2 = np.clip(cx + bw / 2, x1 + 1, w), np.clip(cy + bh / 2, y1 + 1, h)
            d["annotations"] = [{"bbox": [float(a), float(b), float(c), float(e)], "bbox_mode": "XYXY_ABS",
                                 "category_id": int(rng.integers(0, num_classes)), "iscrowd": 0} for a, b, c, e in zip(x1, y1, x2, y2)]
        out.append(d)
    return out


def register_synthetic(name, num_images, seed=0, **kw):
    DatasetCatalog.register(name, lambda: synthetic_coco_dicts(num_images, seed, **kw))
