"""Teacher/student container (reference ubteacher/modeling/meta_arch/ts_ensemble.py:6-16): defines
the checkpoint key prefixes modelTeacher.* / modelStudent.*."""
from collections import OrderedDict


class EnsembleTSModel:
    def __init__(self, modelTeacher, modelStudent):
        self.modelTeacher = modelTeacher
        self.modelStudent = getattr(modelStudent, "module", modelStudent)

    def state_dict(self):
        sd = OrderedDict()
        for k, v in self.modelTeacher.state_dict().items():
            sd["modelTeacher." + k] = v
        for k, v in self.modelStudent.state_dict().items():
            sd["modelStudent." + k] = v
        return sd

    def load_state_dict(self, sd, strict=True):
        t = {k[len("modelTeacher."):]: v for k, v in sd.items() if k.startswith("modelTeacher.")}
        s = {k[len("modelStudent."):]: v for k, v in sd.items() if k.startswith("modelStudent.")}
        self.modelTeacher.load_state_dict(t, strict)
        self.modelStudent.load_state_dict(s, strict)
