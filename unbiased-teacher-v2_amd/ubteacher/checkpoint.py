"""Teacher/student checkpointing (reference ubteacher/checkpoint/detection_checkpoint.py:10-89):
state is saved as {model: {modelTeacher.*, modelStudent.*}, optimizer, scheduler, iteration}; an
ImageNet backbone checkpoint (no modelTeacher./modelStudent. prefixes) loads into the student only."""
import os

import torch


class DetectionTSCheckpointer:
    def __init__(self, model, save_dir="", optimizer=None, scheduler=None):
        self.model = model
        self.save_dir = save_dir
        self.optimizer = optimizer
        self.scheduler = scheduler

    def _last_file(self):
        return os.path.join(self.save_dir, "last_checkpoint")

    def has_checkpoint(self):
        return bool(self.save_dir) and os.path.exists(self._last_file())

    def get_checkpoint_file(self):
        with open(self._last_file()) as f:
            return os.path.join(self.save_dir, f.read().strip())

    def save(self, name, **kwargs):
        os.makedirs(self.save_dir, exist_ok=True)
        data = {"model": {k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()}}
        if self.optimizer is not None:
            data["optimizer"] = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in self.optimizer.state_dict().items()}
        if self.scheduler is not None:
            data["scheduler"] = self.scheduler.state_dict()
        data.update(kwargs)
        fn = "{}.pth".format(name)
        torch.save(data, os.path.join(self.save_dir, fn))
        with open(self._last_file(), "w") as f:
            f.write(fn)

    def load(self, path):
        if not path:
            return {}
        ck = torch.load(path, map_location="cpu")
        sd = ck.get("model", ck)
        if any(k.startswith("modelTeacher.") or k.startswith("modelStudent.") for k in sd):
            self.model.load_state_dict(sd, strict=False)
        else:  # backbone-only weights -> student only (detection_checkpoint.py:21-49)
            self.model.modelStudent.load_state_dict(sd, strict=False)
        if "optimizer" in ck and self.optimizer is not None:
            self.optimizer.load_state_dict(ck["optimizer"])
        if "scheduler" in ck and self.scheduler is not None:
            self.scheduler.load_state_dict(ck["scheduler"])
        return ck

    def resume_or_load(self, path, resume=True):
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        if path and os.path.exists(path):
            ck = self.load(path)
            return {k: v for k, v in ck.items() if k not in ("optimizer", "scheduler", "iteration")}
        return {}
