"""Config surface: our flattened presets must resolve to exactly the values of the reference's
eight shipped YAMLs (checked in the build container, where /root/reference exists), and the
yacs-like node must behave (unknown keys rejected, tuple decoding, freeze)."""
import glob
import os

import pytest

from ubteacher import add_ubteacher_config
from ubteacher.d2 import get_cfg
from ubteacher.presets import get_config

REF = "/root/reference/configs"


def _flat(node, prefix=""):
    out = {}
    for k, v in node.items():
        if isinstance(v, dict):
            out.update(_flat(v, prefix + k + "."))
        else:
            out[prefix + k] = v
    return out


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("family,sub,pat", [("fcos", "FCOS", "fcos_R_50_ut2_sup%d_run0.yaml"),
                                             ("rcnn", "Faster-RCNN", "faster_rcnn_R_50_FPN_ut2_sup%d_run0.yaml")])
@pytest.mark.parametrize("sup", [1, 2, 5, 10])
def test_presets_equal_reference_yaml(family, sub, pat, sup):
    ref = get_cfg()
    add_ubteacher_config(ref)
    ref.merge_from_file(os.path.join(REF, sub, "coco-standard", pat % sup))
    mine = get_config(family, sup)
    a, b = _flat(ref), _flat(mine)
    assert set(a) == set(b)
    diff = {k: (a[k], b[k]) for k in a if a[k] != b[k]}
    assert not diff, diff


def test_cfgnode_behaviour():
    cfg = get_config("fcos", 1, ["SOLVER.IMG_PER_BATCH_LABEL", "4", "SOLVER.STEPS", "(10, 20)"])
    assert cfg.SOLVER.IMG_PER_BATCH_LABEL == 4 and cfg.SOLVER.STEPS == (10, 20)
    assert cfg.INPUT.MIN_SIZE_TRAIN == (400, 1200) and cfg.INPUT.MIN_SIZE_TRAIN_SAMPLING == "range"
    with pytest.raises(KeyError):
        cfg.merge_from_list(["SEMISUPNET.NOT_A_KEY", 1])
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.SOLVER.BASE_LR = 0.1
    c2 = cfg.clone()
    c2.defrost()
    c2.SOLVER.BASE_LR = 0.1
    assert cfg.SOLVER.BASE_LR == 0.01


def test_library_exports_every_declared_symbol():
    from ubteacher import hip
    lib = hip.load()
    assert len(hip.symbols()) >= 30
    for name in hip.symbols():
        assert hasattr(lib, name), name


def test_lr_schedulers():
    """reference solver/build.py:9-45 + lr_scheduler.py:9-57: the three scheduler names, warmup, stage factors."""
    import math
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unbiased-teacher-v2_amd"))
    import pytest
    from ubteacher.engine.trainer import build_lr_scheduler
    from ubteacher.presets import get_config

    class Opt:
        def __init__(self):
            self.param_groups = [{"lr": 0.0}]
    base = ["SOLVER.BASE_LR", 0.01, "SOLVER.WARMUP_ITERS", 100, "SOLVER.WARMUP_FACTOR", 0.001, "SOLVER.MAX_ITER", 1000]
    cfg = get_config("fcos", 1, base + ["SOLVER.LR_SCHEDULER_NAME", "WarmupTwoStageMultiStepLR", "SOLVER.STEPS", (300, 600),
                                        "SOLVER.FACTOR_LIST", (1, 0.5, 0.05)])
    o = Opt()
    sch = build_lr_scheduler(cfg, o)
    assert o.param_groups[0]["lr"] == pytest.approx(0.01 * 0.001)                 # it = 0: warmup factor
    assert sch.lr_at(50) == pytest.approx(0.01 * (0.001 * 0.5 + 0.5))             # linear warmup
    assert sch.lr_at(299) == pytest.approx(0.01) and sch.lr_at(300) == pytest.approx(0.005) and sch.lr_at(999) == pytest.approx(0.0005)
    for _ in range(300):
        sch.step()
    assert o.param_groups[0]["lr"] == pytest.approx(0.005)
    with pytest.raises(ValueError):
        build_lr_scheduler(get_config("fcos", 1, base + ["SOLVER.LR_SCHEDULER_NAME", "WarmupTwoStageMultiStepLR", "SOLVER.STEPS", (300, 600),
                                                         "SOLVER.FACTOR_LIST", (1, 0.5)]), Opt())
    cos = build_lr_scheduler(get_config("fcos", 1, base + ["SOLVER.LR_SCHEDULER_NAME", "WarmupCosineLR"]), Opt())
    assert cos.lr_at(500) == pytest.approx(0.01 * 0.5 * (1 + math.cos(math.pi * 0.5)))
    ms = build_lr_scheduler(get_config("fcos", 1, base + ["SOLVER.STEPS", (300,), "SOLVER.GAMMA", 0.1]), Opt())
    assert ms.lr_at(400) == pytest.approx(0.001)
    with pytest.raises(ValueError):
        build_lr_scheduler(get_config("fcos", 1, base + ["SOLVER.LR_SCHEDULER_NAME", "Nope"]), Opt())
