from .coco_eval import COCOBoxEvaluator, coco_box_ap
from .evaluator import inference_on_dataset, inference_context

__all__ = ["COCOBoxEvaluator", "coco_box_ap", "inference_on_dataset", "inference_context"]
