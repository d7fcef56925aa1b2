"""Faster-RCNN UTv2 trainer - placeholder until the RCNN path lands."""


class UBRCNNTeacherTrainer:
    def __init__(self, cfg, data_loader=None):
        raise NotImplementedError("UBRCNNTeacherTrainer: Faster-RCNN path not built yet")
