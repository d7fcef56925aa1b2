"""Host logic of the two-crop loader vs goldens produced by executing the reference's own sources
(tests/golden/gen_golden.py::gen_data_pipeline): the aspect-ratio grouped 4-tuple batcher incl. its drop-while-waiting behaviour
(data/common.py:93-167) and the label / unlabel split (data/build.py:30-53); plus known-answer tests of the restated Detectron2
pieces (TrainingSampler, ResizeShortestEdge, box transforms) that are not in the reference tree."""
import itertools
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLD = json.load(open(os.path.join(G, "data_pipeline.json")))


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_batcher_matches_reference(case):
    from ubteacher.data import AspectRatioGroupedSemiSupDatasetTwoCrop
    g = GOLD["batcher_%d" % case]

    def stream(sizes, tag):
        for i, (w, h) in enumerate(sizes):
            yield ({"width": w, "height": h, "id": i, "view": tag + "s"}, {"width": w, "height": h, "id": i, "view": tag + "w"})
    ds = AspectRatioGroupedSemiSupDatasetTwoCrop((stream(g["label_wh"], "l"), stream(g["unlabel_wh"], "u")), (g["bl"], g["bu"]))
    got = []
    for ls, lw, us, uw in ds:
        assert [d["view"] for d in ls] == ["ls"] * g["bl"] and [d["view"] for d in lw] == ["lw"] * g["bl"]
        assert [d["view"] for d in us] == ["us"] * g["bu"] and [d["view"] for d in uw] == ["uw"] * g["bu"]
        assert [d["id"] for d in ls] == [d["id"] for d in lw] and [d["id"] for d in us] == [d["id"] for d in uw]
        # one aspect-ratio group per list
        assert len({d["width"] > d["height"] for d in ls}) == 1 and len({d["width"] > d["height"] for d in us}) == 1
        got.append([[d["id"] for d in ls], [d["id"] for d in us]])
    assert got == g["batches"] and len(got) > 5


def test_divide_label_unlabel_matches_reference():
    from ubteacher.data import divide_label_unlabel
    dicts = [{"image_id": i} for i in range(50)]
    path = os.path.join(G, "supervision_small.json")
    for key, want in GOLD["divide_small"].items():
        pct, seed = key.split("_")
        lab, unl = divide_label_unlabel(dicts, float(pct), int(seed), path)
        assert [d["image_id"] for d in lab] == want["label"] and [d["image_id"] for d in unl] == want["unlabel"]
    with pytest.raises(AssertionError):  # listed count must equal int(percent / 100 * len)
        divide_label_unlabel(dicts[:40], 10.0, 0, path)


@pytest.mark.skipif(not os.path.isfile("/root/reference/dataseed/COCO_supervision.txt"), reason="the shipped seed table is not in this tree")
def test_divide_label_unlabel_on_the_shipped_seed_table():
    from ubteacher.data import divide_label_unlabel
    dicts = list(range(117266))
    for key, want in GOLD["divide_coco"].items():
        pct, seed = key.split("_")
        lab, unl = divide_label_unlabel(dicts, float(pct), int(seed), "/root/reference/dataseed/COCO_supervision.txt")
        assert (len(lab), len(unl), int(np.sum(lab)), lab[:8], unl[:8]) == (want["n_label"], want["n_unlabel"], want["label_sum"],
                                                                            want["label_head"], want["unlabel_head"])


def test_training_sampler_stream_and_rank_sharding():
    import torch
    from ubteacher.data import TrainingSampler
    g = torch.Generator().manual_seed(7)
    want = torch.randperm(11, generator=g).tolist() + torch.randperm(11, generator=g).tolist() + torch.randperm(11, generator=g).tolist()
    full = list(itertools.islice(iter(TrainingSampler(11, seed=7, rank=0, world_size=1)), 33))
    assert full == want
    for ws in (2, 4):
        shards = [list(itertools.islice(iter(TrainingSampler(11, seed=7, rank=r, world_size=ws)), 8)) for r in range(ws)]
        inter = [shards[i % ws][i // ws] for i in range(8 * ws)]
        assert inter == want[:8 * ws]                      # the ranks partition ONE shared stream
    assert list(itertools.islice(iter(TrainingSampler(4, shuffle=False, seed=0, rank=1, world_size=2)), 6)) == [1, 3, 1, 3, 1, 3]


def test_resize_shortest_edge_and_boxes():
    from ubteacher.data.transforms import ResizeShortestEdge, transform_boxes
    rng = np.random.default_rng(0)
    rs = ResizeShortestEdge((400, 1200), 1333, "range")
    for (h, w) in [(480, 640), (640, 480), (375, 500), (300, 1200)]:
        for _ in range(20):
            nh, nw = rs.get_params(rng, h, w)
            assert max(nh, nw) <= 1333 and abs(nh / nw - h / w) < 2e-3
            assert 400 <= min(nh, nw) <= 1200 or max(nh, nw) == 1333
    assert ResizeShortestEdge((800,), 1333, "choice").get_params(rng, 480, 640) == (800, 1067)     # Detectron2's canonical 800 x 1067
    assert ResizeShortestEdge((800,), 1333, "choice").get_params(rng, 300, 1200) == (333, 1333)
    b = transform_boxes([[10, 20, 110, 220], [600, 400, 700, 500]], 480, 640, 960, 1280, False)
    assert np.allclose(b, [[20, 40, 220, 440], [1200, 800, 1280, 960]])                              # second box clipped to the image
    f = transform_boxes([[10, 20, 110, 220]], 480, 640, 960, 1280, True)
    assert np.allclose(f, [[1280 - 220, 40, 1280 - 20, 440]])


def test_random_crop_sizes_and_positions():
    """INPUT.CROP (data/dataset_mapper.py:38-41 -> Detectron2 T.RandomCrop): the four crop types' size rules and an in-image corner"""
    from ubteacher.data.transforms import RandomCrop
    rng = np.random.default_rng(3)
    assert RandomCrop("relative", (0.5, 0.25)).get_crop_size(rng, 481, 640) == (241, 160)          # int(x + 0.5)
    assert RandomCrop("absolute", (384, 600)).get_crop_size(rng, 300, 800) == (300, 600)          # capped at the image
    for _ in range(50):
        ch, cw = RandomCrop("relative_range", (0.9, 0.9)).get_crop_size(rng, 480, 640)
        assert int(0.9 * 480 + 0.5) <= ch <= 480 and int(0.9 * 640 + 0.5) <= cw <= 640
        ch, cw = RandomCrop("absolute_range", (384, 600)).get_crop_size(rng, 480, 640)
        assert 384 <= ch <= 480 and 384 <= cw <= 600
        x0, y0, w, h = RandomCrop("relative_range", (0.5, 0.5)).get_params(rng, 480, 640)
        assert 0 <= x0 and x0 + w <= 640 and 0 <= y0 and y0 + h <= 480 and w >= 320 and h >= 240
    xs = {RandomCrop("absolute", (100, 100)).get_params(rng, 120, 130)[:2] for _ in range(200)}
    assert len(xs) > 50 and max(x for x, _ in xs) == 30 and max(y for _, y in xs) == 20             # corners cover [0, w - cw] x [0, h - ch]


def test_strong_param_sampling_ranges():
    from ubteacher.data.transforms import sample_strong_params
    rng = np.random.default_rng(5)
    n, jit, gray, blur, er = 400, 0, 0, 0, [0, 0, 0]
    for _ in range(n):
        p = sample_strong_params(rng, 600, 800)
        jit += p["jitter"]; gray += p["gray"]; blur += p["blur"]
        assert sorted(p["order"]) == [0, 1, 2, 3] and 0.6 <= p["brightness"] <= 1.4 and -0.1 <= p["hue"] <= 0.1 and 0.1 <= p["sigma"] <= 2.0
        for k, r in enumerate(p["erase"]):
            if r is not None:
                er[k] += 1
                i, j, h, w = r
                assert 0 <= i and i + h <= 600 and 0 <= j and j + w <= 800 and 0.019 * 480000 <= h * w <= 0.21 * 480000
    assert abs(jit / n - 0.8) < 0.07 and abs(gray / n - 0.2) < 0.07 and abs(blur / n - 0.5) < 0.08
    assert abs(er[0] / n - 0.7) < 0.08 and abs(er[1] / n - 0.5) < 0.08 and abs(er[2] / n - 0.3) < 0.08


def test_annotation_bbox_mode_is_honoured():
    """Detectron2-format COCO dicts (load_coco_json: the reference's builtin datasets) carry XYWH_ABS boxes; the reference converts through
    BoxMode in transform_instance_annotations (data/dataset_mapper.py:118-131)."""
    import pytest
    from ubteacher.data.dataset_mapper import XYWH_ABS, XYXY_ABS, to_xyxy_abs
    assert to_xyxy_abs({"bbox": [10, 20, 30, 40], "bbox_mode": XYWH_ABS}) == [10.0, 20.0, 40.0, 60.0]
    assert to_xyxy_abs({"bbox": [10, 20, 30, 40], "bbox_mode": "XYWH_ABS"}) == [10.0, 20.0, 40.0, 60.0]
    assert to_xyxy_abs({"bbox": [10, 20, 30, 40], "bbox_mode": XYXY_ABS}) == [10.0, 20.0, 30.0, 40.0]
    assert to_xyxy_abs({"bbox": [10, 20, 30, 40]}) == [10.0, 20.0, 30.0, 40.0]

    class Mode:  # an enum member such as detectron2.structures.BoxMode.XYWH_ABS
        value = 1
    assert to_xyxy_abs({"bbox": [1, 2, 3, 4], "bbox_mode": Mode()}) == [1.0, 2.0, 4.0, 6.0]
    with pytest.raises(ValueError):
        to_xyxy_abs({"bbox": [0, 0, 1, 1], "bbox_mode": 4})   # XYWHA_ABS (rotated): not supported


def test_bbox_mode_accepts_numpy_integers_and_names():
    """dataset dicts built from numpy / pandas carry numpy integer bbox_mode values: any Integral is a BoxMode value"""
    import numpy as np
    from ubteacher.data.dataset_mapper import to_xyxy_abs
    for mode in (1, np.int64(1), np.int32(1), "XYWH_ABS", "BoxMode.XYWH_ABS"):
        assert to_xyxy_abs({"bbox": [10, 20, 5, 6], "bbox_mode": mode}) == [10.0, 20.0, 15.0, 26.0]
    for mode in (0, np.int64(0), "XYXY_ABS"):
        assert to_xyxy_abs({"bbox": [10, 20, 15, 26], "bbox_mode": mode}) == [10.0, 20.0, 15.0, 26.0]
    assert to_xyxy_abs({"bbox": [1, 2, 3, 4]}) == [1.0, 2.0, 3.0, 4.0]      # no field: XYXY_ABS (with one warning)
    import pytest
    with pytest.raises(ValueError):
        to_xyxy_abs({"bbox": [1, 2, 3, 4], "bbox_mode": np.int64(7)})
