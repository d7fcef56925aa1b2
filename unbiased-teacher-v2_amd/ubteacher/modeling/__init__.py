"""Registry surface of the reference's ubteacher/modeling/__init__.py:2-9."""
from .backbone import build_fcos_resnet_fpn_backbone, build_resnet_fpn_backbone  # noqa: F401
from .fcos import FCOS  # noqa: F401
from .rcnn import (TwoStagePseudoLabGeneralizedRCNN, PseudoLabRPN, StandardROIHeadsPseudoLab,  # noqa: F401
                   FastRCNNFocaltLossBoundaryVarOutputLayers, Box2BoxXYXYTransform)
from .one_stage_detector import OneStageDetector, PseudoProposalNetwork  # noqa: F401
from .ts_ensemble import EnsembleTSModel  # noqa: F401
from .pseudo_generator import PseudoGenerator  # noqa: F401
from .build import build_model  # noqa: F401
