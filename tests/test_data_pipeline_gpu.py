"""Two-crop mapper + loader end to end on the GPU: for real dataset dicts the product's (strong, weak) views
equal - BIT FOR BIT - what the reference pipeline's arithmetic gives for the same random decisions: Pillow's bilinear resize + flip
for the weak view, and the oracle's (Pillow-pinned) strong chain on top of it; the loader's 4-tuples drive a real training step."""
import os

import numpy as np
import pytest
import torch

from oracle import aug_oracle as A
from tests.utv2_testutil import small_fcos_cfg

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cfg_for_data(name, n, seed_table=None):
    from ubteacher.data import DatasetCatalog, register_synthetic
    cfg = small_fcos_cfg()
    if name in DatasetCatalog:
        DatasetCatalog.remove(name)
    register_synthetic(name, n, seed=3)
    cfg.DATASETS.TRAIN = (name,)
    cfg.DATASETS.CROSS_DATASET = False
    cfg.INPUT.MIN_SIZE_TRAIN = (96, 160)
    cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING = "range"
    cfg.INPUT.MAX_SIZE_TRAIN = 224
    cfg.SEED = 11
    return cfg


def test_mapper_views_bit_exact_vs_pillow_arithmetic():
    from PIL import Image
    from ubteacher.data import DatasetCatalog, DatasetMapperTwoCropSeparate
    cfg = cfg_for_data("syn_map", 8)
    dicts = DatasetCatalog.get("syn_map")
    mapper = DatasetMapperTwoCropSeparate(cfg, True)
    replay = torch.Generator(device="cuda")
    replay.manual_seed(11 * 1000003 + 0)
    flips = 0
    for d in dicts:
        strong, weak = mapper(d)
        p = mapper.last_params
        flips += p["flip"]
        bgr = np.ascontiguousarray(d["image"][:, :, ::-1])                     # cfg.INPUT.FORMAT = BGR
        ref_w = np.asarray(Image.fromarray(bgr).resize((p["neww"], p["newh"]), Image.BILINEAR))
        if p["flip"]:
            ref_w = ref_w[:, ::-1]
        assert weak["image"].dtype == torch.uint8 and tuple(weak["image"].shape) == (3, p["newh"], p["neww"])
        assert np.array_equal(weak["image"].cpu().numpy(), ref_w.transpose(2, 0, 1))
        noises = [None if r is None else torch.randn((3, r[2], r[3]), device="cuda", generator=replay).cpu().numpy() for r in p["erase"]]
        ref_s = A.strong_augment(np.ascontiguousarray(ref_w), p, noises)
        assert np.array_equal(strong["image"].cpu().numpy(), ref_s.transpose(2, 0, 1))
        # same geometry and labels in both dicts; original size kept; boxes inside the resized frame
        assert strong["height"] == d["height"] and strong["width"] == d["width"] and strong["instances"] is weak["instances"]
        b = weak["instances"].gt_boxes.tensor
        assert len(b) == len(weak["instances"].gt_classes) <= len(d["annotations"])
        assert float(b[:, 0::2].max()) <= p["neww"] and float(b[:, 1::2].max()) <= p["newh"] and float(b.min()) >= 0
        sx, sy = p["neww"] / d["width"], p["newh"] / d["height"]
        a0 = d["annotations"][0]["bbox"]
        want = [a0[0] * sx, a0[1] * sy, a0[2] * sx, a0[3] * sy]
        if p["flip"]:
            want = [p["neww"] - want[2], want[1], p["neww"] - want[0], want[3]]
        assert np.allclose(b[0].numpy(), np.clip(want, 0, [p["neww"], p["newh"]] * 2), atol=1e-3)
    assert "annotations" in dicts[0] and "image" in dicts[0]                   # the dataset dicts are not modified
    assert 0 < flips < len(dicts)


def test_loader_feeds_a_training_step():
    """registered dataset -> label / unlabel split by the seed table -> samplers -> GPU mapper -> aspect-ratio batcher -> trainer step"""
    import json
    from ubteacher.engine import UBTeacherTrainer
    cfg = cfg_for_data("syn_train", 50)
    cfg.DATALOADER.SUP_PERCENT = 10.0
    cfg.DATALOADER.RANDOM_DATA_SEED = 1
    cfg.DATALOADER.RANDOM_DATA_SEED_PATH = os.path.join(G, "supervision_small.json")
    cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS = False
    torch.manual_seed(0)
    tr = UBTeacherTrainer(cfg)
    from ubteacher.data import AspectRatioGroupedSemiSupDatasetTwoCrop
    assert isinstance(tr._data_loader, AspectRatioGroupedSemiSupDatasetTwoCrop)
    labeled = set(json.load(open(cfg.DATALOADER.RANDOM_DATA_SEED_PATH))["10.0"]["1"])
    it = iter(tr._data_loader)
    for _ in range(3):
        lq, lk, uq, uk = next(it)
        assert len(lq) == len(lk) == 2 and len(uq) == len(uk) == 2
        assert all(d["image_id"] in labeled for d in lq) and all(d["image_id"] not in labeled for d in uq)
        for q, k in zip(lq + uq, lk + uk):
            assert q["image"].shape == k["image"].shape and q["image"].is_cuda and q["image_id"] == k["image_id"]
        assert len({d["width"] > d["height"] for d in lq}) == 1 and len({d["width"] > d["height"] for d in uq}) == 1
    tr.iter = 1
    losses = tr.run_step_full_semisup()
    torch.cuda.synchronize()
    assert torch.isfinite(losses).all()
    rec = tr.flush_metrics()
    assert "loss_fcos_cls" in rec and "loss_fcos_cls_pseudo" in rec


def test_mapper_reads_files_handles_empty_annotations_and_eval_mode(tmp_path):
    """`file_name` dicts are decoded on the host with Pillow (PNG here: lossless, so the pixels are known), images without boxes give
    empty Instances, and the is_train=False mapper returns one dict with the resized image only (dataset_mapper.py:102-106)."""
    from PIL import Image
    from ubteacher.data import DatasetMapperTwoCropSeparate
    cfg = cfg_for_data("syn_file", 2)
    cfg.INPUT.MIN_SIZE_TRAIN = (96,)
    cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING = "choice"
    cfg.INPUT.RANDOM_FLIP = "none"
    rgb = np.random.default_rng(0).integers(0, 256, (60, 90, 3), dtype=np.uint8)
    path = str(tmp_path / "img.png")
    Image.fromarray(rgb).save(path)
    d = {"file_name": path, "height": 60, "width": 90, "image_id": 7, "annotations": []}
    mapper = DatasetMapperTwoCropSeparate(cfg, True)
    strong, weak = mapper(d)
    ref = np.asarray(Image.fromarray(np.ascontiguousarray(rgb[:, :, ::-1])).resize((144, 96), Image.BILINEAR))
    assert np.array_equal(weak["image"].cpu().numpy(), ref.transpose(2, 0, 1))
    assert len(weak["instances"]) == 0 and strong["image"].shape == weak["image"].shape and weak["image_id"] == 7
    with pytest.raises(ValueError):
        mapper({"file_name": path, "height": 61, "width": 90})                      # check_image_size
    ev = DatasetMapperTwoCropSeparate(cfg, False)
    cfg.INPUT.MIN_SIZE_TEST = 120
    ev = DatasetMapperTwoCropSeparate(cfg, False)
    out = ev({"file_name": path, "height": 60, "width": 90, "annotations": [{"bbox": [1, 2, 30, 40], "category_id": 3}]})
    assert isinstance(out, dict) and tuple(out["image"].shape) == (3, 120, 180) and "instances" not in out
    ref = np.asarray(Image.fromarray(np.ascontiguousarray(rgb[:, :, ::-1])).resize((180, 120), Image.BILINEAR))
    assert np.array_equal(out["image"].cpu().numpy(), ref.transpose(2, 0, 1))


def test_mapper_random_crop_in_front_of_the_weak_augmentation(tmp_path):
    """INPUT.CROP.ENABLED (data/dataset_mapper.py:38-41): the crop comes first - the weak view is the resized CROP, the boxes move with
    it (shift, then scale, one clip at the end) and the ones left outside are dropped (filter_empty_instances)."""
    from PIL import Image
    from ubteacher.data import DatasetMapperTwoCropSeparate
    cfg = cfg_for_data("syn_file", 2)
    cfg.INPUT.MIN_SIZE_TRAIN = (96,)
    cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING = "choice"
    cfg.INPUT.RANDOM_FLIP = "none"
    cfg.INPUT.CROP.ENABLED = True
    cfg.INPUT.CROP.TYPE = "absolute"
    cfg.INPUT.CROP.SIZE = [48, 72]
    rgb = np.random.default_rng(1).integers(0, 256, (60, 90, 3), dtype=np.uint8)
    path = str(tmp_path / "img.png")
    Image.fromarray(rgb).save(path)
    annos = [{"bbox": [0, 0, 90, 60], "category_id": 1}, {"bbox": [80, 50, 90, 60], "category_id": 2}, {"bbox": [0, 0, 4, 4], "category_id": 3}]
    mapper = DatasetMapperTwoCropSeparate(cfg, True)
    seen = set()
    for _ in range(12):
        strong, weak = mapper({"file_name": path, "height": 60, "width": 90, "annotations": annos})
        x0, y0, cw, ch = mapper.last_params["crop"]
        seen.add((x0, y0))
        assert (cw, ch) == (72, 48) and 0 <= x0 <= 18 and 0 <= y0 <= 12
        crop = np.ascontiguousarray(rgb[y0:y0 + ch, x0:x0 + cw, ::-1])
        ref = np.asarray(Image.fromarray(crop).resize((144, 96), Image.BILINEAR))
        assert np.array_equal(weak["image"].cpu().numpy(), ref.transpose(2, 0, 1))
        inst = weak["instances"]
        want = []
        for a in annos:
            b = np.array(a["bbox"], dtype=np.float64) - [x0, y0, x0, y0]
            b = np.minimum(np.clip(b * 2.0, 0, None), [144, 96, 144, 96])
            if b[2] - b[0] > 1e-5 and b[3] - b[1] > 1e-5:
                want.append((b, a["category_id"]))
        assert len(inst) == len(want)
        for i, (b, c) in enumerate(want):
            assert np.allclose(inst.gt_boxes.tensor[i].numpy(), b) and int(inst.gt_classes[i]) == c
    assert len(seen) > 3
    assert DatasetMapperTwoCropSeparate(cfg, False).crop is None          # is_train only
