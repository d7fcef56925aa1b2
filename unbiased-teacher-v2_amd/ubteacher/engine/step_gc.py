"""Garbage-collector policy of the training loop.

A UTv2 step creates ~10^4 short-lived Python objects (autograd nodes, ctypes arguments, tensor views) next to ~10^5 long-lived ones (the
two models' tensors, geometry tables, the configs).  Left to itself CPython's cyclic collector runs a young-generation pass every 700
allocations and, every few hundred steps, a FULL pass that walks every live object - milliseconds of host time in the middle of a step
whose host side is co-critical with the GPU (a 2 + 2-image step enqueues for 12 ms and runs for 13).  The loop therefore (the usual
recipe of large training loops) freezes what is alive when it starts - those objects are never traversed again -, switches the automatic
collector off and collects the young generations itself at a fixed period, between two steps.  Reference counting still frees
everything acyclic immediately; only cycles wait for the next periodic pass.  UTV2_STEP_GC=0 keeps the interpreter's default."""
import gc
import os


class StepGC:
    def __init__(self, period=None):
        self.enabled = os.environ.get("UTV2_STEP_GC", "1") != "0"
        self.period = int(os.environ.get("UTV2_STEP_GC_PERIOD", "200")) if period is None else period
        self._was_enabled = None
        self._n = 0

    def __enter__(self):
        if self.enabled:
            self._was_enabled = gc.isenabled()
            gc.collect()
            gc.freeze()
            gc.disable()
        return self

    def tick(self):
        """call once per iteration, between two steps"""
        if self.enabled:
            self._n += 1
            if self._n % self.period == 0:
                gc.collect(1)

    def __exit__(self, *exc):
        if self.enabled:
            gc.unfreeze()
            if self._was_enabled:
                gc.enable()
        return False
