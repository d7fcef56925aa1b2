"""Minimal self-contained substrate standing in for the Detectron2 pieces the UTv2 hot path
touches (config node, registries, Boxes/Instances/ImageList, event storage).  Written from the
behavioural description in SURVEY.md (Detectron2 is not available in this environment)."""
from .config import CfgNode, get_cfg  # noqa: F401
from .registry import Registry  # noqa: F401
from .structures import Boxes, Instances, ImageList  # noqa: F401
from .events import EventStorage, get_event_storage  # noqa: F401
