"""Teacher/student checkpointing (reference ubteacher/checkpoint/detection_checkpoint.py:10-89):
state is saved as {model: {modelTeacher.*, modelStudent.*}, optimizer, scheduler, iteration}; an
ImageNet backbone checkpoint (no modelTeacher./modelStudent. prefixes) loads into the student only."""
import os
import pickle
import re

import torch


# ---- Caffe2 / Detectron ImageNet backbones (R-50.pkl): Detectron2's c2_model_loading [D2-recall] ------------------------
def convert_c2_detectron_names(weights):
    """blob names of a Caffe2 ResNet (`conv1_w`, `res_conv1_bn_s`, `res2_0_branch2a_w`, `res2_0_branch2a_bn_b`, ...) ->
    Detectron2 ResNet names (`stem.conv1.weight`, `stem.conv1.norm.weight`, `res2.0.conv1.weight`, ...).  Momentum /
    classifier blobs are dropped.  Returns {new_name: value} in the original order."""
    out = {}
    for k, v in weights.items():
        if k.endswith("_momentum") or k.startswith("fc1000") or k in ("pred_w", "pred_b"):
            continue
        n = k.replace("_", ".")
        n = re.sub(r"\.b$", ".bias", n)
        n = re.sub(r"\.w$", ".weight", n)
        n = re.sub(r"bn\.s$", "norm.weight", n)
        n = re.sub(r"bn\.bias$", "norm.bias", n)
        n = re.sub(r"bn\.rm$", "norm.running_mean", n)
        n = re.sub(r"bn\.running\.mean$", "norm.running_mean", n)
        n = re.sub(r"bn\.riv$", "norm.running_var", n)
        n = re.sub(r"bn\.running\.var$", "norm.running_var", n)
        n = re.sub(r"bn\.gamma$", "norm.weight", n)
        n = re.sub(r"bn\.beta$", "norm.bias", n)
        n = re.sub(r"gn\.s$", "norm.weight", n)
        n = re.sub(r"gn\.bias$", "norm.bias", n)
        n = re.sub(r"^res\.conv1\.norm\.", "conv1.norm.", n)   # the stem's BN blobs are called res_conv1_bn_*
        n = re.sub(r"^conv1\.", "stem.conv1.", n)
        n = n.replace(".branch1.", ".shortcut.").replace(".branch2a.", ".conv1.").replace(".branch2b.", ".conv2.")
        n = n.replace(".branch2c.", ".conv3.")
        out[n] = v
    return out


def align_and_update_state_dicts(model_sd, ckpt_sd, c2_conversion=True):
    """Name-matching heuristic of Detectron2: a checkpoint key is given to the model key it is the longest dot-suffix of
    (`res2.0.conv1.weight` -> `backbone.bottom_up.res2.0.conv1.weight`); shape mismatches are skipped.
    Returns ({model_key: tensor}, unmatched_checkpoint_keys)."""
    if c2_conversion:
        ckpt_sd = convert_c2_detectron_names(ckpt_sd)
    ckeys = sorted(ckpt_sd.keys())
    matched, used = {}, set()
    for mk in model_sd:
        best = None
        for ck in ckeys:
            if mk == ck or mk.endswith("." + ck):
                if best is None or len(ck) > len(best):
                    best = ck
        if best is None:
            continue
        v = torch.as_tensor(ckpt_sd[best])
        if tuple(v.shape) != tuple(model_sd[mk].shape):
            continue
        matched[mk] = v
        used.add(best)
    return matched, [k for k in ckeys if k not in used]


def resolve_path(path):
    """Local file for a checkpoint path.  `detectron2://X` (the model-zoo URIs every shipped config uses for MODEL.WEIGHTS) maps to
    the cache location Detectron2's PathManager downloads it to - $FVCORE_CACHE (default ~/.torch/iopath_cache)/detectron2/X
    [D2-recall]; there is no network here, so the file has to be there already.  Raises FileNotFoundError otherwise."""
    if path.startswith("detectron2://"):
        cache = os.environ.get("FVCORE_CACHE", os.path.join(os.path.expanduser("~"), ".torch", "iopath_cache"))
        local = os.path.join(cache, "detectron2", path[len("detectron2://"):])
        if not os.path.exists(local):
            raise FileNotFoundError("{} is not in the local model-zoo cache ({}); place the file there or point MODEL.WEIGHTS at a local "
                                    "path (set MODEL.WEIGHTS \"\" to train from random initialisation)".format(path, local))
        return local
    if "://" in path:
        raise FileNotFoundError("cannot fetch {!r}: no network access; point MODEL.WEIGHTS at a local file".format(path))
    if not os.path.exists(path):
        raise FileNotFoundError("checkpoint {!r} not found".format(path))
    return path


def load_checkpoint_file(path):
    """.pth -> torch.load; .pkl -> Caffe2 / Detectron pickle: {"model": blobs, "__author__": "Caffe2", "matching_heuristics": True}
    (DetectionCheckpointer._load_file [D2-recall])."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
        if isinstance(data, dict) and "model" in data and "__author__" in data:
            return data
        if isinstance(data, dict) and "blobs" in data:
            data = data["blobs"]
        data = {k: v for k, v in data.items() if not k.endswith("_momentum")}
        return {"model": data, "__author__": "Caffe2", "matching_heuristics": True}
    return torch.load(path, map_location="cpu")


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return type(obj)((k, _to_cpu(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


class DetectionTSCheckpointer:
    def __init__(self, model, save_dir="", optimizer=None, scheduler=None, **checkpointables):
        self.model = model
        self.save_dir = save_dir
        self.optimizer = optimizer
        self.scheduler = scheduler
        # further objects with state_dict() / load_state_dict(), saved under their keyword (Detectron2 Checkpointer(**checkpointables)
        # [D2-recall]): the trainers register the fp16 loss scaler as `grad_scaler`
        self.extra = dict(checkpointables)
        self.last_optimizer_skipped = False   # set by every load(): True when the file's optimizer state could not be applied
        # resuming from a checkpoint whose optimizer entry is a torch.optim state (every reference-produced teacher/student checkpoint):
        # False = warn and resume weights / scheduler / iteration with zero momentum; True = refuse
        self.strict_optimizer = False

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
            data["optimizer"] = _to_cpu(self.optimizer.state_dict())
        if self.scheduler is not None:
            data["scheduler"] = self.scheduler.state_dict()
        for key, obj in self.extra.items():   # (not `name`: the file name below is the caller's)
            data[key] = obj.state_dict()
        data.update(kwargs)
        fn = "{}.pth".format(name)
        torch.save(data, os.path.join(self.save_dir, fn))
        with open(self._last_file(), "w") as f:
            f.write(fn)

    def load(self, path, checkpointables=None):
        """Detectron2 Checkpointer.load [D2-recall]: the model always; of the other checkpointables (optimizer, scheduler) the ones
        named in `checkpointables` (None = all that the file holds, [] = none - the non-resume path)."""
        if not path:
            return {}
        path = resolve_path(path)
        ck = load_checkpoint_file(path)
        self._load_model(ck, ck.get("model", ck))
        want = ("optimizer", "scheduler") + tuple(self.extra) if checkpointables is None else tuple(checkpointables)
        self.last_optimizer_skipped = False
        if "optimizer" in want and "optimizer" in ck and self.optimizer is not None:
            try:
                self.optimizer.load_state_dict(ck["optimizer"])
            except ValueError as e:
                if self.strict_optimizer:
                    raise ValueError("checkpoint {!r}: the optimizer state cannot be mapped onto this model's ArenaSGD state ({}); load it "
                                     "without resuming to take the weights only".format(path, e))
                # weights, scheduler and iteration resume, the momentum restarts from zero
                import logging
                logging.getLogger(__name__).warning(
                    "checkpoint %r holds an optimizer state that cannot be mapped onto this model (%s): resuming model / scheduler / "
                    "iteration WITHOUT momentum (set checkpointer.strict_optimizer = True to make this an error)", path, e)
                self.last_optimizer_skipped = True
        if "scheduler" in want and "scheduler" in ck and self.scheduler is not None:
            self.scheduler.load_state_dict(ck["scheduler"])
        for name, obj in self.extra.items():
            if name in want and name in ck:
                obj.load_state_dict(ck[name])
        return ck

    def _load_model(self, ck, sd):
        if ck.get("__author__", None) == "Caffe2":
            # pretrained ImageNet backbone: name-matching heuristics, STUDENT only (detection_checkpoint.py:11-37); the
            # teacher receives it at the burn-in boundary through _update_teacher_model(keep_rate=0)
            student = self.model.modelStudent
            matched, unmatched = align_and_update_state_dicts(student.state_dict(), sd, c2_conversion=True)
            student.load_state_dict(matched, strict=False)
            self.last_load_report = {"matched": sorted(matched), "unmatched_checkpoint_keys": unmatched}
        elif any(k.startswith("modelTeacher.") or k.startswith("modelStudent.") for k in sd):
            self.model.load_state_dict(sd, strict=False)
        else:  # backbone-only weights -> student only (detection_checkpoint.py:21-49)
            self.model.modelStudent.load_state_dict(sd, strict=False)

    def resume_or_load(self, path, resume=True):
        """Detectron2 Checkpointer.resume_or_load [D2-recall]: resume from the last checkpoint of OUTPUT_DIR if there is one, else
        load `path` (MODEL.WEIGHTS) WITHOUT optimizer / scheduler / iteration.  A non-empty path that cannot be resolved raises,
        like the reference's PathManager does: training silently from random, frozen stem / res2 weights is never what was asked."""
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file())
        if path:
            ck = self.load(path, checkpointables=[])
            return {k: v for k, v in ck.items() if k not in ("optimizer", "scheduler", "iteration")}
        return {}


class DetectionCheckpointer(DetectionTSCheckpointer):
    """Detectron2's plain DetectionCheckpointer over ONE model (the reference's Faster-RCNN `--eval-only` path, train_net.py:48-53)."""

    def _load_model(self, ck, sd):
        if ck.get("__author__", None) == "Caffe2":
            matched, unmatched = align_and_update_state_dicts(self.model.state_dict(), sd, c2_conversion=True)
            self.model.load_state_dict(matched, strict=False)
            self.last_load_report = {"matched": sorted(matched), "unmatched_checkpoint_keys": unmatched}
            return
        mine = set(self.model.state_dict())
        if not any(k in mine for k in sd):
            # Detectron2 would log every key as missing / unexpected and go on with the initial weights: refuse instead
            raise ValueError("checkpoint holds no tensor of this model (first keys: {}); a teacher/student checkpoint "
                             "(modelTeacher.* / modelStudent.*) loads through DetectionTSCheckpointer".format(list(sd)[:3]))
        self.model.load_state_dict(sd, strict=False)
