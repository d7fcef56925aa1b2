"""Scalar event storage used by the trainers' _write_metrics (reference engine/trainer.py:431-466)."""
from collections import defaultdict

_STACK = []


def get_event_storage():
    assert _STACK, "get_event_storage() has to be called inside a 'with EventStorage(...)' context!"
    return _STACK[-1]


class EventStorage:
    def __init__(self, start_iter=0):
        self.iter = start_iter
        self._history = defaultdict(list)
        self._latest = {}

    def put_scalar(self, name, value, smoothing_hint=True):
        value = float(value)
        self._history[name].append((value, self.iter))
        self._latest[name] = (value, self.iter)

    def put_scalars(self, **kwargs):
        for k, v in kwargs.items():
            self.put_scalar(k, v)

    def latest(self):
        return self._latest

    def history(self, name):
        return self._history[name]

    def step(self):
        self.iter += 1

    def __enter__(self):
        _STACK.append(self)
        return self

    def __exit__(self, *a):
        assert _STACK[-1] is self
        _STACK.pop()
