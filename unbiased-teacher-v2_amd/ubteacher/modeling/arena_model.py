"""Base class for models whose whole state lives in a flat ParamStore arena."""
from collections import OrderedDict

import torch

from ..params import ParamStore
from .backbone import BNFolder


class ArenaModel:
    def __init__(self, cfg):
        self.cfg = cfg
        self.device = torch.device(cfg.MODEL.DEVICE)
        self.store = ParamStore()
        self.folder = BNFolder(self.store)
        self.training = True

    def _finalize(self):
        self.store.finalize(self.device)
        self.folder.materialize()

    # nn.Module-like surface ------------------------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        for m in self._children():
            m.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def _children(self):
        return []

    def state_dict(self):
        return self.store.state_dict()

    def load_state_dict(self, sd, strict=True):
        return self.store.load_state_dict(sd, strict)

    def named_parameters(self):
        for k, (p, g) in self.store.trainable_named().items():
            yield k, p

    def parameters(self):
        for _, p in self.named_parameters():
            yield p

    def flat_state(self):
        return self.store.flat

    def zero_grad(self):
        self.store.grad.zero_()
