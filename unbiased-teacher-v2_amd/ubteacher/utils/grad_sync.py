"""Bucketed, backward-overlapped all-reduce of the flat gradient arena (data parallel).

The reference wraps the student in DistributedDataParallel (engine/trainer.py:59-63), whose reducer all-reduces gradient
buckets while backward is still running.  Here all parameter gradients live in ONE flat fp32 arena that the backward
kernels accumulate into directly, so the same overlap is a matter of bookkeeping: the arena is cut into contiguous
buckets (at parameter boundaries); every autograd Function that will write gradients registers its parameters at forward
time and reports them at the end of its backward; when the last writer of a bucket has reported, the bucket's
all-reduce(SUM) is launched asynchronously (RCCL enqueues it behind the kernels already on the stream and runs it on its
own stream, overlapping the rest of backward over xGMI).  `finish()` launches whatever is left and waits.
The 1/world scale stays folded into the SGD kernel.  With world_size 1 nothing is registered or launched.
"""
import bisect
import os

import torch
import torch.distributed as dist

_DEBUG = os.environ.get("UTV2_GRAD_SYNC_DEBUG", "0") == "1"


class GradBuckets:
    def __init__(self, grad_flat, handles, bucket_bytes=16 << 20):
        """grad_flat: the flat gradient arena; handles: objects with .offset / .numel inside it (arena order)."""
        self.grad = grad_flat
        n = grad_flat.numel()
        spans = sorted((int(h.offset), int(h.offset) + int(h.numel)) for h in handles if h.offset < n)
        cap = max(1, bucket_bytes // 4)
        starts = [0]
        for s, _ in spans:  # cut only at parameter starts, once the running bucket is large enough
            if s - starts[-1] >= cap:
                starts.append(s)
        self.starts = starts
        self.bounds = [(starts[i], starts[i + 1] if i + 1 < len(starts) else n) for i in range(len(starts))]
        nb = len(self.bounds)
        self.pending = [0] * nb
        self.launched = [False] * nb
        self.works = []
        self.armed = False
        self.next = nb - 1       # next bucket to launch: strictly descending
        self.order = []          # launch order of the running step
        self.last_order = []     # ... of the last finished step (tests)
        self.prelaunched = []    # buckets reduced at arm() (no registered writer): UTV2_GRAD_SYNC_DEBUG=1 re-checks them in finish()
        self._ops = None
        # exposure bookkeeping (bench.py --gpus N): per step, device events around the wait in finish() = the time the main stream sat
        # between the end of backward and the last collective's completion, and how many buckets had already been issued by then
        self.timing = False
        self._expo = []          # (event at finish() entry, event after the waits, buckets issued during backward, buckets in all)

    def bucket_of(self, h):
        return bisect.bisect_right(self.starts, int(h.offset)) - 1

    # -- called by the autograd Functions ------------------------------------------------------
    def on_forward(self, handles):
        for h in handles:
            if h is not None and h.g is not None:
                self.pending[self.bucket_of(h)] += 1

    def on_backward_done(self, handles):
        for h in handles:
            if h is None or h.g is None:
                continue
            self.pending[self.bucket_of(h)] -= 1
        if self.armed:
            self._launch_ready()

    # -- called by the trainer -----------------------------------------------------------------
    def arm(self):
        """right before losses.backward(): every writer of this step has registered by now.  Buckets nobody will write this step
        (parameters of branches that did not run) are ready at once."""
        self.armed = True
        before = len(self.order)
        self._launch_ready()
        self.prelaunched = list(self.order[before:])
        if _DEBUG:
            # a bucket reduced before backward must not be written by it: a layer that accumulates into handle.g without registering
            # through on_forward would have its gradient reduced before it exists, and the ranks would diverge silently
            for w in self.works:
                w.wait()
            self._pre_sums = [float(self.grad[s:e].double().abs().sum()) for s, e in (self.bounds[b] for b in self.prelaunched)]

    def _launch_ready(self):
        """Collectives must be issued in the SAME order on every rank (a mismatch is an RCCL hang).  Which bucket completes first
        depends on how many writers a rank registered (e.g. one fused student pass on one rank, two passes on another), so the
        order is made static instead: buckets go strictly from the highest index down - the order backward finishes them in, the
        arena being laid out in forward order - and a finished bucket waits for every higher one, as DDP's reducer does."""
        while self.next >= 0 and self.pending[self.next] == 0:
            self._launch(self.next)
            self.next -= 1

    def _launch(self, b):
        s, e = self.bounds[b]
        self.launched[b] = True
        self.order.append(b)
        if e > s:
            if self._ops is None:
                from .. import ops as _ops
                self._ops = _ops
            ops = self._ops
            # this bucket's weight gradients may still be running on the wgrad side stream: the collective is issued from THAT stream
            # (made to wait for the main one), so RCCL's stream waits for both and the main stream's dgrad chain is never held up
            side = ops.wgrad_stream_behind_main(self.grad.device)
            if side is not None:
                with torch.cuda.stream(side):
                    self.works.append(dist.all_reduce(self.grad[s:e], op=dist.ReduceOp.SUM, async_op=True))
            else:
                ops.join_wgrad_stream()
                self.works.append(dist.all_reduce(self.grad[s:e], op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """after backward: reduce the buckets that are still waiting (in the same static order), wait for everything"""
        early = len(self.order)
        ev0 = None
        if self.timing and self.grad.is_cuda:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        while self.next >= 0:
            self._launch(self.next)
            self.next -= 1
        for w in self.works:
            w.wait()
        self.works = []
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            self._expo.append((ev0, ev1, early, len(self.order)))
        if _DEBUG and self.prelaunched:
            now = [float(self.grad[s:e].double().abs().sum()) for s, e in (self.bounds[b] for b in self.prelaunched)]
            bad = [b for b, x, y in zip(self.prelaunched, self._pre_sums, now) if x != y]
            if bad:
                raise RuntimeError("grad_sync: buckets %s were all-reduced at arm() (no registered writer) but backward wrote them: "
                                   "a layer accumulates into its gradient handle without on_forward()" % bad)
        self.armed = False
        nb = len(self.bounds)
        self.pending = [0] * nb
        self.launched = [False] * nb
        self.next = nb - 1
        self.last_order, self.order = self.order, []


    def exposure_summary(self, reset=True):
        """(bench) mean / max milliseconds per step the main stream waited in finish() for the gradient collectives - the part of the
        all-reduce that backward did NOT hide - with the bucket layout.  Synchronises on the recorded events."""
        if not self._expo:
            return None
        ms = [a.elapsed_time(b) for a, b, _, _ in self._expo]
        out = {"steps": len(ms), "exposed_ms_per_step_mean": sum(ms) / len(ms), "exposed_ms_per_step_max": max(ms),
               "buckets": len(self.bounds), "bucket_mbytes": [round(4 * (e - s) / 2 ** 20, 2) for s, e in self.bounds],
               "buckets_issued_during_backward_mean": sum(x[2] for x in self._expo) / len(self._expo),
               "gradient_mbytes_per_step": round(4 * self.grad.numel() / 2 ** 20, 1),
               "note": "device time between the end of backward on the main stream and the completion of the last bucket's all-reduce (what "
                       "the optimizer step waits for); buckets are issued from the weight-gradient stream as their last writer reports"}
        if reset:
            self._expo = []
        return out


def param_handles(layer):
    """the arena handles a conv / norm layer's backward accumulates into"""
    hs = []
    for name in ("w", "bias", "gamma", "beta"):
        h = getattr(layer, name, None)
        if h is not None and getattr(h, "g", None) is not None:
            hs.append(h)
    return hs
