"""Config presets: the two flattened YAMLs under configs/ + the per-supervision-level deltas of the
reference's eight shipped configs (configs/{FCOS,Faster-RCNN}/coco-standard/*_sup{1,2,5,10}_run0.yaml)."""
import os

from .config import add_ubteacher_config
from .d2.config import get_cfg

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_FILES = {"fcos": "utv2_fcos_r50.yaml", "rcnn": "utv2_frcnn_r50.yaml"}
# differences from the sup1 file
_DELTAS = {
    ("fcos", 1): [],
    ("fcos", 2): ["DATALOADER.SUP_PERCENT", 2.0, "SEMISUPNET.BURN_UP_STEP", 15000, "SOLVER.STEPS", (179995, 179999)],
    ("fcos", 5): ["DATALOADER.SUP_PERCENT", 5.0, "SEMISUPNET.BURN_UP_STEP", 20000, "SEMISUPNET.UNSUP_LOSS_WEIGHT", 2.0],
    ("fcos", 10): ["DATALOADER.SUP_PERCENT", 10.0, "SEMISUPNET.BURN_UP_STEP", 30000, "SEMISUPNET.UNSUP_LOSS_WEIGHT", 2.0],
    ("rcnn", 1): [],
    ("rcnn", 2): ["DATALOADER.SUP_PERCENT", 2.0, "SEMISUPNET.UNSUP_LOSS_WEIGHT", 3.0],
    ("rcnn", 5): ["DATALOADER.SUP_PERCENT", 5.0, "SEMISUPNET.UNSUP_LOSS_WEIGHT", 2.0],
    ("rcnn", 10): ["DATALOADER.SUP_PERCENT", 10.0, "TEST.EVAL_PERIOD", 2000],
}


def get_config(family="fcos", sup=1, opts=()):
    cfg = get_cfg()
    add_ubteacher_config(cfg)
    cfg.merge_from_file(os.path.join(_ROOT, "configs", _FILES[family]))
    cfg.merge_from_list(list(_DELTAS[(family, sup)]))
    cfg.merge_from_list(list(opts))
    return cfg
