"""Minimal dataset catalog (Detectron2 DatasetCatalog contract: name -> callable returning a list of dataset dicts) and a synthetic
COCO-shaped dataset (no image files exist in this environment).  A dataset dict carries either `file_name` (decoded on the host with
Pillow) or an in-memory `image` (uint8 [H][W][3], RGB), plus `height`, `width`, `image_id` and `annotations` = list of
{bbox: [x1, y1, x2, y2], bbox_mode: "XYXY_ABS", category_id, iscrowd}."""
import json
import os

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
        MetadataCatalog.pop(name, None)

    def available(self, name):
        """registered AND loadable: a file-backed set (register_coco_instances / register_coco_unlabel_instances) needs its json on disk.
        The shipped configs name coco_2017_train / coco_2017_val, which are registered lazily (register_builtin_coco) whether or not the
        files exist; the trainers fall back to the synthetic loaders - with a warning - when they do not."""
        if name not in self:
            return False
        jf = MetadataCatalog.get(name).get("json_file")
        return jf is None or os.path.exists(jf)


class _Metadata(dict):
    """Detectron2 Metadata surface used here: attribute access + set(**kw)"""
    __getattr__ = dict.get

    def set(self, **kw):
        self.update(kw)
        return self


class _MetadataCatalog(dict):
    def get(self, name):
        if name not in self:
            self[name] = _Metadata(name=name)
        return self[name]


MetadataCatalog = _MetadataCatalog()
DatasetCatalog = _Catalog()


def load_coco_json(json_file, image_root, dataset_name=None):
    """A COCO-format instances json -> Detectron2 dataset dicts (restates detectron2/data/datasets/coco.py::load_coco_json [D2-recall]
    without pycocotools; box fields only): images in ascending id order; per image the annotations of that image in file order, each
    {bbox: [x, y, w, h], bbox_mode: XYWH_ABS, category_id: CONTIGUOUS id, iscrowd, area?}; annotations with `ignore` != 0 are skipped;
    category ids (sorted) map to 0..K-1 and the mapping / names are stored in MetadataCatalog[dataset_name]
    (thing_dataset_id_to_contiguous_id, thing_classes).  An annotation whose image_id is not its image's id is an error."""
    with open(json_file) as f:
        data = json.load(f)
    cats = sorted(data.get("categories", []), key=lambda c: c["id"])
    id_map = {c["id"]: i for i, c in enumerate(cats)}
    if dataset_name is not None:
        MetadataCatalog.get(dataset_name).set(thing_classes=[c.get("name", str(c["id"])) for c in cats],
                                              thing_dataset_id_to_contiguous_id=id_map)
    by_image = {}
    for a in data.get("annotations", []):
        by_image.setdefault(a["image_id"], []).append(a)
    ann_ids = [a["id"] for a in data.get("annotations", []) if "id" in a]
    assert len(set(ann_ids)) == len(ann_ids), "Annotation ids in '{}' are not unique!".format(json_file)
    out = []
    for img in sorted(data["images"], key=lambda d: d["id"]):
        rec = {"file_name": os.path.join(image_root, img["file_name"]), "height": img["height"], "width": img["width"], "image_id": img["id"]}
        objs = []
        for a in by_image.get(img["id"], []):
            assert a["image_id"] == img["id"]
            if a.get("ignore", 0) != 0:
                continue
            obj = {"bbox": [float(v) for v in a["bbox"]], "bbox_mode": "XYWH_ABS", "iscrowd": int(a.get("iscrowd", 0))}
            if id_map:
                if a["category_id"] not in id_map:
                    raise KeyError("Encountered category_id={} but this id does not exist in 'categories' of the json file."
                                   .format(a["category_id"]))
                obj["category_id"] = id_map[a["category_id"]]
            else:
                obj["category_id"] = a["category_id"]
            if "area" in a:
                obj["area"] = float(a["area"])
            objs.append(obj)
        rec["annotations"] = objs
        out.append(rec)
    return out


def register_coco_instances(name, metadata, json_file, image_root):
    """Detectron2 register_coco_instances [D2-recall]: lazy registration of a COCO-format detection set"""
    assert isinstance(name, str) and isinstance(json_file, (str, os.PathLike)) and isinstance(image_root, (str, os.PathLike))
    DatasetCatalog.register(name, lambda: load_coco_json(json_file, image_root, name))
    MetadataCatalog.get(name).set(json_file=str(json_file), image_root=str(image_root), evaluator_type="coco", **metadata)


def load_coco_unlabel_json(json_file, image_root, dataset_name=None):
    """reference ubteacher/data/datasets/builtin.py:60-96: image records only (file_name, height, width), ascending image id"""
    with open(json_file) as f:
        data = json.load(f)
    return [{"file_name": os.path.join(image_root, d["file_name"]), "height": d["height"], "width": d["width"]}
            for d in sorted(data["images"], key=lambda d: d["id"])]


def register_coco_unlabel_instances(name, metadata, json_file, image_root):
    """reference builtin.py:31-57"""
    assert isinstance(name, str) and isinstance(json_file, (str, os.PathLike)) and isinstance(image_root, (str, os.PathLike))
    DatasetCatalog.register(name, lambda: load_coco_unlabel_json(json_file, image_root, name))
    MetadataCatalog.get(name).set(json_file=str(json_file), image_root=str(image_root), evaluator_type="coco", **metadata)


def register_builtin_coco(root=None):
    """The dataset names the shipped configs use, under $DETECTRON2_DATASETS (default ./datasets) in Detectron2's layout [D2-recall]
    (coco/{train,val}2017 + coco/annotations/instances_*.json) and the reference's unlabeled split (builtin.py:13-28:
    coco/unlabeled2017 + coco/annotations/image_info_unlabeled2017.json).  Lazy: nothing is read until a loader asks for the set."""
    root = root if root is not None else os.environ.get("DETECTRON2_DATASETS", "datasets")
    for name, (img, ann) in {"coco_2017_train": ("coco/train2017", "coco/annotations/instances_train2017.json"),
                             "coco_2017_val": ("coco/val2017", "coco/annotations/instances_val2017.json")}.items():
        if name not in DatasetCatalog:
            register_coco_instances(name, {}, os.path.join(root, ann), os.path.join(root, img))
    if "coco_2017_unlabel" not in DatasetCatalog:
        register_coco_unlabel_instances("coco_2017_unlabel", {}, os.path.join(root, "coco/annotations/image_info_unlabeled2017.json"),
                                        os.path.join(root, "coco/unlabeled2017"))

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


register_builtin_coco()
