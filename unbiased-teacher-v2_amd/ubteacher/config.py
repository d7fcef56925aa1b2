"""`add_ubteacher_config(cfg)`: the semi-supervised / FCOS config surface, key-for-key with the
reference's ubteacher/config.py:7-219 (names and default values are the drop-in contract; they are
declared here as one table and applied onto the Detectron2-style default tree)."""
import os

from .d2.config import CfgNode as CN

_SEED_FILE = "dataseed/COCO_supervision.txt"
_SEED_FILE_FB = "manifold://mobile_vision_dataset/tree/unbiased_teacher/COCO_supervision.txt"

# dotted key -> default value.  A value of CN marks a new (empty) sub-node.
_UBTEACHER_KEYS = [
    ("TEST.VAL_LOSS", True),
    ("MODEL.RPN.UNSUP_LOSS_WEIGHT", 1.0),
    ("MODEL.RPN.LOSS", "CrossEntropy"),
    ("MODEL.ROI_HEADS.LOSS", "CrossEntropy"),
    ("SOLVER.IMG_PER_BATCH_LABEL", 1),
    ("SOLVER.IMG_PER_BATCH_UNLABEL", 1),
    ("SOLVER.FACTOR_LIST", (1,)),
    ("DATASETS.TRAIN_LABEL", ("coco_2017_train",)),
    ("DATASETS.TRAIN_UNLABEL", ("coco_2017_train",)),
    ("DATASETS.CROSS_DATASET", False),
    ("TEST.EVALUATOR", "COCOeval"),
    ("SEMISUPNET", CN),
    ("SEMISUPNET.MLP_DIM", 128),
    ("SEMISUPNET.Trainer", "ubteacher"),
    ("SEMISUPNET.TEACHER_UPDATE_ITER", 1),
    ("SEMISUPNET.BURN_UP_STEP", 12000),
    ("SEMISUPNET.UNSUP_LOSS_WEIGHT", 4.0),
    ("SEMISUPNET.UNSUP_REG_LOSS_WEIGHT", 0.0),
    ("SEMISUPNET.SUP_LOSS_WEIGHT", 0.5),
    ("SEMISUPNET.LOSS_WEIGHT_TYPE", "standard"),
    ("SEMISUPNET.PROBE", True),
    ("SEMISUPNET.PSEUDO_CTR_THRES", 0.5),
    ("SEMISUPNET.EMA_SCHEDULE", False),
    ("SEMISUPNET.PSEUDO_CLS_IGNORE_NEAR", False),
    ("SEMISUPNET.SOFT_CLS_LABEL", False),
    ("SEMISUPNET.CLS_LOSS_METHOD", "focal"),
    ("SEMISUPNET.CLS_LOSS_PSEUDO_METHOD", "focal"),
    ("SEMISUPNET.REG_FG_THRES", 0.5),
    ("DATALOADER.SUP_PERCENT", 100.0),
    ("DATALOADER.RANDOM_DATA_SEED", 0),
    ("DATALOADER.RANDOM_DATA_SEED_PATH", None),  # resolved below
    ("EMAMODEL", CN),
    ("EMAMODEL.SUP_CONSIST", True),
    # FCOS head (generalized-focal variant)
    ("MODEL.FCOS", CN),
    ("MODEL.FCOS.NUM_CLASSES", 80),
    ("MODEL.FCOS.IN_FEATURES", ["p3", "p4", "p5", "p6", "p7"]),
    ("MODEL.FCOS.FPN_STRIDES", [8, 16, 32, 64, 128]),
    ("MODEL.FCOS.PRIOR_PROB", 0.01),
    ("MODEL.FCOS.INFERENCE_TH_TRAIN", 0.05),
    ("MODEL.FCOS.INFERENCE_TH_TEST", 0.05),
    ("MODEL.FCOS.NMS_TH", 0.6),
    ("MODEL.FCOS.PRE_NMS_TOPK_TRAIN", 1000),
    ("MODEL.FCOS.PRE_NMS_TOPK_TEST", 1000),
    ("MODEL.FCOS.POST_NMS_TOPK_TRAIN", 100),
    ("MODEL.FCOS.POST_NMS_TOPK_TEST", 100),
    ("MODEL.FCOS.TOP_LEVELS", 2),
    ("MODEL.FCOS.NORM", "GN"),
    ("MODEL.FCOS.USE_SCALE", True),
    ("MODEL.FCOS.THRESH_WITH_CTR", False),
    ("MODEL.FCOS.LOSS_ALPHA", 0.25),
    ("MODEL.FCOS.LOSS_GAMMA", 2.0),
    ("MODEL.FCOS.SIZES_OF_INTEREST", [64, 128, 256, 512]),
    ("MODEL.FCOS.USE_RELU", True),
    ("MODEL.FCOS.USE_DEFORMABLE", False),
    ("MODEL.FCOS.NUM_CLS_CONVS", 4),
    ("MODEL.FCOS.NUM_BOX_CONVS", 4),
    ("MODEL.FCOS.NUM_SHARE_CONVS", 0),
    ("MODEL.FCOS.CENTER_SAMPLE", True),
    ("MODEL.FCOS.POS_RADIUS", 1.5),
    ("MODEL.FCOS.LOC_LOSS_TYPE", "giou"),
    ("MODEL.FCOS.YIELD_PROPOSAL", False),
    ("MODEL.FCOS.NMS_CRITERIA_TRAIN", "cls"),
    ("MODEL.FCOS.NMS_CRITERIA_TEST", "cls_n_ctr"),
    ("MODEL.FCOS.NMS_CRITERIA_REG_TRAIN", "cls_n_loc"),
    ("MODEL.FCOS.REG_DISCRETE", False),
    ("MODEL.FCOS.DFL_WEIGHT", 0.0),
    ("MODEL.FCOS.LOC_FUN_ALL", "mean"),
    ("MODEL.FCOS.UNIFY_CTRCLS", False),
    ("MODEL.FCOS.REG_MAX", 16),
    ("MODEL.FCOS.QUALITY_EST", "centerness"),
    ("MODEL.FCOS.TSBETTER_CLS_SIGMA", 0.0),
    # joint pseudo-labelling
    ("SEMISUPNET.PSEUDO_BBOX_SAMPLE", "thresholding"),
    ("SEMISUPNET.BBOX_THRESHOLD", 0.5),
    ("SEMISUPNET.BBOX_CTR_THRESHOLD", 0.5),
    ("SEMISUPNET.PSEUDO_BBOX_SAMPLE_REG", "thresholding"),
    ("SEMISUPNET.BBOX_THRESHOLD_REG", 0.5),
    ("SEMISUPNET.BBOX_CTR_THRESHOLD_REG", 0.5),
    ("SEMISUPNET.ANALYSIS_PRINT_FRE", 5000),
    ("SEMISUPNET.ANALYSIS_ACCUMLATE_FRE", 200),
    ("SEMISUPNET.TS_BETTER", 0.1),
    ("SEMISUPNET.TS_BETTER_CERT", 0.8),
    ("SEMISUPNET.CONSIST_CLS_LOSS", "mse_loss_raw"),
    ("SEMISUPNET.CONSIST_CTR_LOSS", "kl_loss"),
    ("SEMISUPNET.CONSIST_REG_LOSS", "mse_loss_all_raw"),
    ("SEMISUPNET.RANDOM_FLIP_STRONG", False),
    ("MODEL.FCOS.KL_LOSS", False),
    ("MODEL.FCOS.KL_LOSS_TYPE", "klloss"),
    ("MODEL.FCOS.KLLOSS_WEIGHT", 0.1),
    ("SEMISUPNET.DYNAMIC_EMA", False),
    ("SEMISUPNET.DEMA_FINAL", 1.0),
    ("MODEL.ROI_BOX_HEAD.BBOX_PSEUDO_REG_LOSS_TYPE", "tsbetter"),
    ("SEMISUPNET.T_CERT", 0.5),
    ("SEMISUPNET.EMA_SCHEDULER", False),
    ("SEMISUPNET.EMA_RATE_STEP", (0.9996,)),
    ("SEMISUPNET.EMA_INTVEL", (120000,)),
    ("SEMISUPNET.EMA_KEEP_RATE", 0.0),
    ("SEMISUPNET.USE_SUP_STRONG", "both"),
]


def add_ubteacher_config(cfg):
    """Add the UTv2 keys onto a Detectron2-style default config tree (in place)."""
    for dotted, default in _UBTEACHER_KEYS:
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        if default is CN:
            node[parts[-1]] = CN()
        else:
            node[parts[-1]] = list(default) if isinstance(default, list) else default
    cfg.DATALOADER.RANDOM_DATA_SEED_PATH = _SEED_FILE if os.path.isfile(_SEED_FILE) else _SEED_FILE_FB
    return cfg
