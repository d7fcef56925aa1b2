"""Scalar event storage used by the trainers' _write_metrics (reference engine/trainer.py:431-466) and the two writers the reference's
`PeriodicWriter(self.build_writers(), period=20)` (engine/trainer.py:549-551) drives every 20 iterations on the main process: a console
line and OUTPUT_DIR/metrics.json (one JSON object per written iteration) - the Detectron2 writers' contract [D2-recall]: scalars with a
smoothing hint are reported as the median of their last 20 values, others as their latest value.  (TensorBoard is not in this image.)"""
import datetime
import json
import logging
import os
from collections import defaultdict

_STACK = []


def get_event_storage():
    assert _STACK, "get_event_storage() has to be called inside a 'with EventStorage(...)' context!"
    return _STACK[-1]


def _median(values):
    v = sorted(values)
    n = len(v)
    return 0.5 * (v[(n - 1) // 2] + v[n // 2])


class EventStorage:
    def __init__(self, start_iter=0):
        self.iter = start_iter
        self._history = defaultdict(list)
        self._latest = {}
        self._smoothing = {}

    def put_scalar(self, name, value, smoothing_hint=True):
        value = float(value)
        self._history[name].append((value, self.iter))
        self._latest[name] = (value, self.iter)
        self._smoothing.setdefault(name, bool(smoothing_hint))

    def put_scalars(self, *, smoothing_hint=True, **kwargs):
        for k, v in kwargs.items():
            self.put_scalar(k, v, smoothing_hint=smoothing_hint)

    def latest(self):
        return self._latest

    def history(self, name):
        return self._history[name]

    def median(self, name, window_size=20):
        return _median([v for v, _ in self._history[name][-window_size:]])

    def latest_with_smoothing_hint(self, window_size=20):
        """{name: (median of the last `window_size` values if the scalar was put with a smoothing hint, else the latest value, iteration)}"""
        return {k: ((self.median(k, window_size) if self._smoothing.get(k, True) else v), it) for k, (v, it) in self._latest.items()}

    def step(self):
        self.iter += 1

    def __enter__(self):
        _STACK.append(self)
        return self

    def __exit__(self, *a):
        assert _STACK[-1] is self
        _STACK.pop()


class CommonMetricPrinter:
    """one console line per write: eta, iteration, the losses and the other scalars (medians over the window), time per iteration,
    data time, learning rate, peak device memory"""

    def __init__(self, max_iter=None, window_size=20):
        self.logger = logging.getLogger("ubteacher.events")
        self._max_iter = max_iter
        self._window = window_size

    def write(self, storage=None):
        storage = storage or get_event_storage()
        it = storage.iter
        if self._max_iter is not None and it == self._max_iter:
            return      # the write after the last iteration repeats the previous line
        lat = storage.latest()
        eta = ""
        avg_time = ""
        if "time" in lat:
            t = storage.median("time", 1000)
            avg_time = "time: %.4f  " % t
            if self._max_iter is not None:
                eta = "eta: %s  " % datetime.timedelta(seconds=int(t * (self._max_iter - it - 1)))
        data_time = "data_time: %.4f  " % storage.median("data_time", self._window) if "data_time" in lat else ""
        lr = "lr: %.5g  " % lat["lr"][0] if "lr" in lat else ""
        losses = "  ".join("%s: %.4g" % (k, storage.median(k, self._window)) for k in lat if "loss" in k)
        skip = ("time", "data_time", "lr", "eta_seconds")
        others = "  ".join("%s: %.4g" % (k, storage.median(k, self._window)) for k in lat
                           if "loss" not in k and k not in skip and "/" not in k)
        mem = ""
        try:
            import torch
            if torch.cuda.is_available():
                mem = "max_mem: %.0fM" % (torch.cuda.max_memory_allocated() / 1048576.0)
        except Exception:   # pragma: no cover - never let a log line end a run
            pass
        self.logger.info(" %siter: %d  %s  %s  %s%s%s%s" % (eta, it, losses, others, avg_time, data_time, lr, mem))

    def close(self):
        pass


class JSONWriter:
    """OUTPUT_DIR/metrics.json: per written iteration one line {"iteration": n, <scalar>: value, ...} (sorted keys), appended"""

    def __init__(self, json_file, window_size=20):
        d = os.path.dirname(json_file)
        if d:
            os.makedirs(d, exist_ok=True)
        self._fh = open(json_file, "a")
        self._window = window_size
        self._last_write = -1

    def write(self, storage=None):
        storage = storage or get_event_storage()
        to_save = defaultdict(dict)
        for k, (v, it) in storage.latest_with_smoothing_hint(self._window).items():
            if it <= self._last_write:
                continue
            to_save[it][k] = v
        if to_save:
            self._last_write = max(to_save)
        for it in sorted(to_save):
            rec = dict(to_save[it])
            rec["iteration"] = it
            self._fh.write(json.dumps(rec, sort_keys=True) + "\n")
        self._fh.flush()
        try:
            os.fsync(self._fh.fileno())
        except OSError:
            pass

    def close(self):
        self._fh.close()
