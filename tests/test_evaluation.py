"""COCO box AP restatement: known-answer tests of the protocol."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))


def _box(x, y, w, h):
    return [x, y, x + w, y + h]


def test_perfect_detections_score_100():
    from ubteacher.evaluation import coco_box_ap
    gt = {0: dict(boxes=[_box(10, 10, 50, 60), _box(100, 40, 120, 130)], classes=[3, 7]),
          1: dict(boxes=[_box(5, 5, 20, 20)], classes=[3])}
    pred = {k: dict(boxes=v["boxes"], scores=[0.9] * len(v["classes"]), classes=v["classes"]) for k, v in gt.items()}
    r = coco_box_ap(pred, gt)
    assert r["AP"] == pytest.approx(100.0) and r["AP50"] == pytest.approx(100.0) and r["AP75"] == pytest.approx(100.0)
    assert r["APs"] == pytest.approx(100.0) and r["APm"] == pytest.approx(100.0) and r["APl"] == pytest.approx(100.0)


def test_false_positive_ranked_first_halves_precision():
    from ubteacher.evaluation import coco_box_ap
    gt = {0: dict(boxes=[_box(10, 10, 50, 50)], classes=[1])}
    pred = {0: dict(boxes=[_box(200, 200, 50, 50), _box(10, 10, 50, 50)], scores=[0.9, 0.8], classes=[1, 1])}
    r = coco_box_ap(pred, gt)
    assert r["AP"] == pytest.approx(50.0) and r["AP50"] == pytest.approx(50.0)   # precision 1/2 at every recall point


def test_iou_threshold_sweep():
    from ubteacher.evaluation import coco_box_ap
    # detection covers the left 0.6 of the box: IoU = 0.6 -> a hit for thresholds 0.50, 0.55, 0.60 only
    gt = {0: dict(boxes=[_box(0, 0, 100, 100)], classes=[0])}
    pred = {0: dict(boxes=[_box(0, 0, 60, 100)], scores=[0.5], classes=[0])}
    r = coco_box_ap(pred, gt)
    assert r["AP50"] == pytest.approx(100.0) and r["AP75"] == pytest.approx(0.0) and r["AP"] == pytest.approx(30.0)


def test_recall_interpolation_two_images():
    from ubteacher.evaluation import coco_box_ap
    # 2 ground truths; ranked detections: TP, FP, TP -> precision envelope: 1.0 up to recall 0.5, 2/3 up to recall 1.0
    gt = {0: dict(boxes=[_box(0, 0, 40, 40)], classes=[2]), 1: dict(boxes=[_box(0, 0, 40, 40)], classes=[2])}
    pred = {0: dict(boxes=[_box(0, 0, 40, 40), _box(300, 300, 40, 40)], scores=[0.9, 0.8], classes=[2, 2]),
            1: dict(boxes=[_box(0, 0, 40, 40)], scores=[0.7], classes=[2])}
    r = coco_box_ap(pred, gt)
    expect = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101 * 100
    assert r["AP50"] == pytest.approx(expect)


def test_crowd_and_area_ranges():
    from ubteacher.evaluation import coco_box_ap
    # a crowd region absorbs detections without penalty; a small object counts only in "all" and "small"
    gt = {0: dict(boxes=[_box(0, 0, 200, 200), _box(300, 300, 20, 20)], classes=[5, 5], iscrowd=[1, 0])}
    pred = {0: dict(boxes=[_box(10, 10, 50, 50), _box(60, 60, 50, 50), _box(300, 300, 20, 20)], scores=[0.9, 0.8, 0.7], classes=[5, 5, 5])}
    r = coco_box_ap(pred, gt)
    assert r["AP"] == pytest.approx(100.0) and r["APs"] == pytest.approx(100.0)
    assert r["APm"] == -1.0 and r["APl"] == -1.0        # no non-ignored ground truth of those sizes


def test_missed_object_and_duplicate_detection():
    from ubteacher.evaluation import coco_box_ap
    # two objects, one never detected; the detected one is reported twice (the duplicate is a false positive)
    gt = {0: dict(boxes=[_box(0, 0, 50, 50), _box(200, 0, 50, 50)], classes=[1, 1])}
    pred = {0: dict(boxes=[_box(0, 0, 50, 50), _box(1, 1, 50, 50)], scores=[0.9, 0.6], classes=[1, 1])}
    r = coco_box_ap(pred, gt)
    # recall tops out at 0.5 with precision 1.0: 51 of the 101 recall points are reached
    assert r["AP50"] == pytest.approx(51.0 / 101.0 * 100.0)
