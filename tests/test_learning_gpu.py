"""Does the whole thing learn?  (round 6; found missing by using the launcher like a user: every other test pins a step, a kernel or a file
format against the reference - none trains.)  A COCO-layout tree of coloured patches on noise (tools/make_tiny_coco.py: 64 train / 16 val
files, 4 categories), registered as a COCO-format dataset, half of it labeled by a seed table; the product's own data path (two-crop mapper
on the GPU, aspect-ratio batcher), trainer, LR schedule and AMP mode; 500 supervised (burn-in) iterations at the recipe's learning rate
from tools/make_synthetic_backbone.py's stand-in for R-50.pkl; then Trainer.test on the val FILES.  Measured (tools/r06_probes/learn_tiny.sh,
profiles/r06_learn_tiny.txt): FCOS student box AP 58.9 / AP50 88.5 at iteration 500, Faster-RCNN 10.9 / 30.4 at 500 and 60.8 / 89.9 at
1500.  The bounds here are far below that: the test asks whether detection is being learned at all, on held-out files, by both trainers
(the Faster-RCNN recipe in its YAML's fp32, 900 iterations)."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,iters,min_ap,min_ap50", [("fcos", 500, 12.0, 30.0), ("rcnn", 900, 8.0, 25.0)])
def test_student_learns_to_detect_on_held_out_files(kind, iters, min_ap, min_ap50, tmp_path, monkeypatch):
    import numpy as np
    import make_synthetic_backbone
    import make_tiny_coco
    from ubteacher.data import DatasetCatalog, register_coco_instances
    from ubteacher.engine import UBRCNNTeacherTrainer, UBTeacherTrainer
    from ubteacher.presets import get_config
    T = UBTeacherTrainer if kind == "fcos" else UBRCNNTeacherTrainer
    root = str(tmp_path / "ds")
    monkeypatch.setattr(sys, "argv", ["make_tiny_coco.py", root, "64", "16", "colour"])
    make_tiny_coco.main()
    weights = str(tmp_path / "backbone.pth")
    monkeypatch.setattr(sys, "argv", ["make_synthetic_backbone.py", kind, weights, "0"])
    make_synthetic_backbone.main()
    names = {}
    for split in ("train", "val"):
        names[split] = "tiny_learn_%s_%d" % (split, os.getpid())
        if names[split] in DatasetCatalog:
            DatasetCatalog.remove(names[split])
        register_coco_instances(names[split], {}, os.path.join(root, "coco", "annotations", "instances_%s2017.json" % split),
                                os.path.join(root, "coco", "%s2017" % split))
    try:
        cfg = get_config(kind, 1, [
            "MODEL.DEVICE", "cuda", "MODEL.WEIGHTS", weights, "SOLVER.MAX_ITER", iters, "SEMISUPNET.BURN_UP_STEP", 100000,
            "SOLVER.IMG_PER_BATCH_LABEL", 4, "SOLVER.IMG_PER_BATCH_UNLABEL", 4, "SOLVER.CHECKPOINT_PERIOD", 0, "TEST.EVAL_PERIOD", 0,
            "DATALOADER.SUP_PERCENT", 50.0, "DATALOADER.RANDOM_DATA_SEED_PATH", os.path.join(root, "seed.json"),
            "INPUT.MIN_SIZE_TRAIN", (160, 224), "INPUT.MAX_SIZE_TRAIN", 320, "INPUT.MIN_SIZE_TEST", 192, "INPUT.MAX_SIZE_TEST", 320,
            "OUTPUT_DIR", str(tmp_path / "out"), "SEED", 7])
        cfg.DATASETS.TRAIN = (names["train"],)
        cfg.DATASETS.TEST = (names["val"],)
        seed = int(os.environ.get("UTV2_LEARN_TEST_SEED", "7"))      # (the margin below was checked over seeds 1-6 as well)
        torch.manual_seed(seed); np.random.seed(seed)
        import random
        random.seed(seed)
        tr = T(cfg)
        assert type(tr._data_loader).__name__ != "SyntheticTwoCropLoader"          # the files, not the synthetic stand-in
        tr.resume_or_load(resume=False)
        tr.checkpointer.save = lambda *a, **k: None
        tr.train_loop(0, iters)
        lines = [json.loads(l) for l in open(os.path.join(cfg.OUTPUT_DIR, "metrics.json"))]
        first, last = lines[0], lines[-1]
        assert last["iteration"] == iters - 1 and all(v == v for v in last.values())
        assert last["total_loss"] < 0.75 * first["total_loss"], (first["total_loss"], last["total_loss"])
        res = T.test(cfg, tr.model)
        print("learned:", {k: round(v, 1) for k, v in res["bbox"].items()}, "total_loss %.3f -> %.3f" % (first["total_loss"], last["total_loss"]))
        assert res["bbox"]["AP50"] > min_ap50 and res["bbox"]["AP"] > min_ap, res["bbox"]
    finally:
        for n in names.values():
            if n in DatasetCatalog:
                DatasetCatalog.remove(n)
