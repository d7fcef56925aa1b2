"""Inference loop of the evaluation path (reference ubteacher/evaluation/evaluator.py:14-104):
the model is switched to eval mode, every batch of the (fixed-length) loader is run under no_grad with the FCOS
test-time NMS criterion, outputs are handed to the evaluator, and the evaluator's result dict is returned.
The per-image timing the reference logs is returned under "_speed" instead of being printed."""
import time
from contextlib import contextmanager

import torch


@contextmanager
def inference_context(model):
    """eval mode for the duration of the block, previous mode restored afterwards"""
    was_training = getattr(model, "training", False)
    model.eval()
    try:
        yield
    finally:
        model.train(was_training)


def inference_on_dataset(model, data_loader, evaluator, cfg=None):
    total = len(data_loader)
    if evaluator is not None:
        evaluator.reset()
    # the FCOS detectors take the test-time NMS criterion (reference evaluator.py:57); the two-stage model is called plainly, as
    # Detectron2's own inference_on_dataset does for the Faster-RCNN trainer
    one_stage = cfg is not None and "FCOS" in cfg.MODEL and cfg.SEMISUPNET.Trainer != "ubteacher_rcnn"
    nms_method = cfg.MODEL.FCOS.NMS_CRITERIA_TEST if one_stage else None
    num_warmup = min(5, max(total - 1, 0))
    compute, images = 0.0, 0
    with inference_context(model), torch.no_grad():
        for idx, inputs in enumerate(data_loader):
            if idx == num_warmup:
                compute, images = 0.0, 0
            t0 = time.perf_counter()
            outputs = model(inputs, nms_method=nms_method) if nms_method is not None else model(inputs)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            compute += time.perf_counter() - t0
            images += len(inputs)
            if evaluator is not None:
                evaluator.process(inputs, outputs)
    results = evaluator.evaluate() if evaluator is not None else {}
    if results is None:
        results = {}
    results["_speed"] = {"images": images, "seconds_per_image": compute / max(images, 1)}
    return results
