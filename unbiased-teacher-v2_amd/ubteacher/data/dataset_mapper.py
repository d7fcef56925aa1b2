"""Two-crop mapper (reference ubteacher/data/dataset_mapper.py:14-157) with the pixel work on the GPU.

One dataset dict -> (strong dict, weak dict): the weak view is the decoded image through ResizeShortestEdge + RandomFlip, the strong
view is the weak view through build_strong_augmentation; both dicts carry the SAME geometry, `image` uint8 [3][H][W] in cfg.INPUT.FORMAT
channel order (device resident), the original `height` / `width`, and `instances` (gt_boxes, gt_classes) in the resized frame.
Host work: JPEG decode (Pillow) when the dict has `file_name`, the random decisions, the box arithmetic.  Everything per-pixel runs in
csrc/augment.hip, bit-exact to the Pillow arithmetic of the reference pipeline (tests/test_aug_gpu.py, tests/test_data_pipeline_gpu.py).
Reference quirk kept: the strong augmentation is applied to the image in cfg.INPUT.FORMAT order (BGR) labelled "RGB"
(dataset_mapper.py:139), so the grey weights / hue act on swapped channels."""
import numbers
import copy

import numpy as np
import torch

from ..d2.structures import Boxes, Instances
from . import transforms as T


def read_image(dataset_dict, fmt="BGR"):
    """uint8 [H][W][3] numpy in `fmt` channel order: in-memory `image` (RGB) or Pillow decode of `file_name`"""
    if "image" in dataset_dict and dataset_dict["image"] is not None:
        img = np.asarray(dataset_dict["image"])
    else:
        from PIL import Image
        with Image.open(dataset_dict["file_name"]) as im:
            img = np.asarray(im.convert("RGB"))
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    if fmt == "BGR":
        img = img[:, :, ::-1]
    elif fmt != "RGB":
        raise NotImplementedError("INPUT.FORMAT %r" % (fmt,))
    return np.ascontiguousarray(img)


def check_image_size(dataset_dict, image):
    if "width" in dataset_dict or "height" in dataset_dict:
        wh, expected = (image.shape[1], image.shape[0]), (dataset_dict["width"], dataset_dict["height"])
        if wh != expected:
            raise ValueError("Mismatched image shape{}, got {}, expect {}.".format(
                " for image " + dataset_dict["file_name"] if "file_name" in dataset_dict else "", wh, expected))
    dataset_dict.setdefault("width", image.shape[1])
    dataset_dict.setdefault("height", image.shape[0])


# Detectron2 BoxMode values [D2-recall]: the enum members themselves are ints; dataset dicts may carry the int or the name
XYXY_ABS, XYWH_ABS = 0, 1
_WARNED_NO_MODE = False
_BOX_MODES = {0: XYXY_ABS, 1: XYWH_ABS, "XYXY_ABS": XYXY_ABS, "XYWH_ABS": XYWH_ABS, "BoxMode.XYXY_ABS": XYXY_ABS, "BoxMode.XYWH_ABS": XYWH_ABS}


def to_xyxy_abs(anno):
    """the annotation's box as XYXY_ABS (Detectron2 transform_instance_annotations converts through BoxMode before the transforms;
    load_coco_json - the format of the reference's builtin COCO sets - stores XYWH_ABS).  Detectron2 requires `bbox_mode`; a dict
    without it is read as XYXY_ABS with ONE warning (a COCO-style XYWH dict that lost the field would otherwise be silently wrong)."""
    if "bbox_mode" not in anno:
        global _WARNED_NO_MODE
        if not _WARNED_NO_MODE:
            _WARNED_NO_MODE = True
            import logging
            logging.getLogger(__name__).warning("annotation without `bbox_mode`: assuming XYXY_ABS (Detectron2 requires the field)")
    mode = anno.get("bbox_mode", XYXY_ABS)
    mode = getattr(mode, "value", mode)                      # an enum member
    if isinstance(mode, numbers.Integral) and not isinstance(mode, bool):
        key = int(mode)                                      # python int, numpy.int32 / int64 (dicts built from numpy / pandas)
    else:
        key = str(mode)
    if key not in _BOX_MODES:
        raise ValueError("unsupported bbox_mode {!r} (XYXY_ABS and XYWH_ABS are)".format(anno.get("bbox_mode")))
    x0, y0, a, b = (float(v) for v in anno["bbox"])
    return [x0, y0, x0 + a, y0 + b] if _BOX_MODES[key] == XYWH_ABS else [x0, y0, a, b]


class DatasetMapperTwoCropSeparate:
    def __init__(self, cfg, is_train=True, seed=None, device=None):
        # data/dataset_mapper.py:38-41: the crop goes in FRONT of the weak augmentation (both views see the cropped image)
        self.crop = T.RandomCrop(cfg.INPUT.CROP.TYPE, cfg.INPUT.CROP.SIZE) if (cfg.INPUT.CROP.ENABLED and is_train) else None
        self.resize, self.flip_prob = T.build_weak_augmentation(cfg, is_train)
        self.img_format = cfg.INPUT.FORMAT
        self.is_train = is_train
        self.device = torch.device(device if device is not None else cfg.MODEL.DEVICE)
        from ..utils import comm
        base = int(cfg.SEED) if seed is None and int(cfg.SEED) >= 0 else (0 if seed is None else int(seed))
        self.rng = np.random.default_rng([base, comm.get_rank()])
        self.generator = torch.Generator(device=self.device)
        self.generator.manual_seed(base * 1000003 + comm.get_rank())
        self.last_params = None  # the random decisions of the most recent call (tests / debugging)

    def __call__(self, dataset_dict):
        annos_in = dataset_dict.get("annotations")
        dataset_dict = {k: (v if k == "image" else copy.deepcopy(v)) for k, v in dataset_dict.items() if k != "annotations"}
        image = read_image(dataset_dict, self.img_format)
        check_image_size(dataset_dict, image)
        dataset_dict.pop("image", None)
        h, w = image.shape[:2]
        cx0 = cy0 = 0
        if self.crop is not None:
            cx0, cy0, w, h = self.crop.get_params(self.rng, h, w)
            image = np.ascontiguousarray(image[cy0:cy0 + h, cx0:cx0 + w])
        newh, neww = self.resize.get_params(self.rng, h, w)
        flip = bool(self.rng.random() < self.flip_prob) if self.flip_prob > 0 else False
        x = torch.from_numpy(image).to(self.device, non_blocking=True)
        weak = T.apply_weak(x, newh, neww, flip)
        if not self.is_train:
            dataset_dict["image"] = hip_to_chw(weak)
            return dataset_dict
        if annos_in is not None:
            keep = [a for a in annos_in if a.get("iscrowd", 0) == 0]
            raw = np.asarray([to_xyxy_abs(a) for a in keep], dtype=np.float64).reshape(-1, 4) - np.array([cx0, cy0, cx0, cy0], dtype=np.float64)
            boxes = T.transform_boxes(raw, h, w, newh, neww, flip) if keep else np.zeros((0, 4), np.float32)   # one clip, after all transforms
            classes = np.array([a["category_id"] for a in keep], dtype=np.int64)
            nonempty = ((boxes[:, 2] - boxes[:, 0]) > 1e-5) & ((boxes[:, 3] - boxes[:, 1]) > 1e-5)  # filter_empty_instances
            inst = Instances((newh, neww))
            inst.gt_boxes = Boxes(torch.from_numpy(boxes[nonempty]))
            inst.gt_classes = torch.from_numpy(classes[nonempty])
            dataset_dict["instances"] = inst
        p = T.sample_strong_params(self.rng, newh, neww)
        self.last_params = dict(p, newh=newh, neww=neww, flip=flip, crop=(cx0, cy0, w, h))
        strong = T.apply_strong(weak, p, generator=self.generator)
        strong_dict = dataset_dict
        weak_dict = dict(dataset_dict)
        strong_dict["image"] = hip_to_chw(strong)
        weak_dict["image"] = hip_to_chw(weak)
        return strong_dict, weak_dict


def hip_to_chw(img):
    from .. import hip
    return hip.aug_to_chw(img)
