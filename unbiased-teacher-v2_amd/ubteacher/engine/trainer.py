"""UTv2 trainers: `UBTeacherTrainer` (FCOS) and `UBRCNNTeacherTrainer` (Faster-RCNN).

Same public surface as the reference's ubteacher/engine/trainer.py (class names, build_model /
build_optimizer / build_lr_scheduler / build_train_loader, resume_or_load, train,
run_step_full_semisup, _update_teacher_model, test) with the step re-designed for MI355X:

  * student + teacher state are flat arenas: EMA is ONE axpby launch (reference: ~300 tensors x 3
    temporaries + load_state_dict, trainer.py:468-486), SGD one launch per decay group, and the
    data-parallel exchange is ONE flat RCCL all-reduce of the gradient arena per step (teacher is
    replicated and never communicates, trainer.py:59-63);
  * pseudo-labels, loss normalisers and metrics stay on the device: the only host syncs are the
    periodic metric read-out (every `log_period` steps, like PeriodicWriter's period 20) instead
    of ~30 `.item()` calls per step.
"""
import logging
import os
import time
from collections import OrderedDict

import torch
import torch.distributed as dist

from .. import hip, ops
from ..d2.events import EventStorage
from ..modeling.build import build_model
from ..modeling.pseudo_generator import PseudoGenerator
from ..modeling.ts_ensemble import EnsembleTSModel
from ..utils import comm
from ..data.synthetic import SyntheticTwoCropLoader
from ..checkpoint import DetectionTSCheckpointer
from .step_gc import StepGC


class ArenaSGD:
    """torch.optim.SGD semantics (momentum, dampening 0, no nesterov; weight decay per group as D2's
    get_default_optimizer_params: WEIGHT_DECAY on weights and biases, WEIGHT_DECAY_NORM on norm
    params) as one fused launch per group over the flat arena."""

    def __init__(self, cfg, model):
        s = cfg.SOLVER
        self.store = model.store
        self.lr = s.BASE_LR
        self.base_lr = s.BASE_LR
        self.momentum = s.MOMENTUM
        self.wd = s.WEIGHT_DECAY
        self.wd_norm = s.WEIGHT_DECAY_NORM
        self.store.mom = torch.zeros_like(self.store.grad)
        self.param_groups = [{"lr": self.lr}]  # scheduler surface

    def zero_grad(self):
        self.store.grad.zero_()

    def step(self, grad_scale=1.0, amp_state=None):
        """amp_state (fp16 mode): device fp32 {loss scale, found_inf, growth tracker} - the step unscales the gradients and is skipped on
        the device when they hold a non-finite value (GradScaler.step)"""
        st = self.store
        lr = self.param_groups[0]["lr"]
        if amp_state is not None:
            hip.amp_found_inf(st.grad, amp_state)
        m16 = st.mirror16(need_fresh=True)    # the 16-bit copy the convs read: written by the update itself (no conversion pass next step)
        for kind, wd in (("decay", self.wd), ("nodecay", self.wd_norm)):
            s, e = st.ranges[kind]
            mk = None if m16 is None else m16[s:e]
            if e > s and amp_state is not None:
                hip.sgd_momentum_amp(st.flat[s:e], st.grad[s:e], st.mom[s:e], lr, self.momentum, wd, grad_scale, amp_state, mirror16=mk)
            elif e > s:
                hip.sgd_momentum(st.flat[s:e], st.grad[s:e], st.mom[s:e], lr, self.momentum, wd, grad_scale, zero_grad=False, mirror16=mk)
        if amp_state is not None:
            hip.amp_update_scale(amp_state, 2.0, 0.5, 2000)   # torch.cuda.amp.GradScaler defaults [SURVEY appendix C]
        ops.bump_version()
        st.touch()
        if m16 is not None:
            st.mirror16_written()
        bank = getattr(st, "_flipbank", None)
        if bank is not None:
            bank.refresh_ahead()     # the dgrad weight images of the new weights, on a side stream: ready long before the next backward

    # ---- checkpoint surface -----------------------------------------------------------------------------------------------
    # The momentum arena is laid out like the parameter arena, and that layout is an implementation detail (the paired FCOS towers
    # merge cls_tower / bbox_tower tensors into one handle, conditional on the config): a flat copy would resume silently with the
    # momentum of one tensor applied to another after any layout change.  The state is therefore saved PER state_dict KEY, through
    # the same export views as the weights (reference key names and shapes), and loaded key by key with shape checks.
    def _momentum_views(self):
        """OrderedDict key -> (handle, exported view of the momentum arena, kind) in the model's state_dict (= reference) key order"""
        st = self.store
        per_key = {}
        for h in st.handles:
            if h.kind not in ("decay", "nodecay"):
                continue
            m = st.mom[h.offset: h.offset + h.numel].view(h.shape)
            for key, fn in h.exports:
                per_key[key] = (h, m if fn is None else fn(m), m)
        return OrderedDict((k, per_key[k]) for k in st.state_dict() if k in per_key)

    def state_dict(self):
        mv = self._momentum_views()
        return {"format": "utv2-arena-sgd/2", "lr": self.param_groups[0]["lr"],
                "momentum": OrderedDict((k, v.detach().clone().contiguous()) for k, (_, v, _) in mv.items())}

    def _load_momentum(self, per_key):
        mv = self._momentum_views()
        missing = [k for k in mv if k not in per_key]
        unexpected = [k for k in per_key if k not in mv]
        if missing or unexpected:
            raise ValueError("optimizer state does not match this model's trainable tensors: missing %s unexpected %s"
                             % (missing[:4], unexpected[:4]))
        for k, (h, view, raw) in mv.items():
            src = per_key[k]
            if tuple(src.shape) != tuple(view.shape):
                raise ValueError("optimizer state: size mismatch for %s: %s vs %s" % (k, tuple(src.shape), tuple(view.shape)))
        with torch.no_grad():
            for k, (h, view, raw) in mv.items():
                src = per_key[k].to(device=raw.device, dtype=raw.dtype)
                if k in h.loaders:
                    h.loaders[k](raw, src)
                else:
                    view.copy_(src)

    def load_state_dict(self, sd):
        """Accepts (1) this class's per-key state; (2) a torch.optim.SGD state_dict as the reference's checkpoints hold it
        (Detectron2 build_optimizer: one entry per trainable parameter in model.named_parameters() order, grouped by equal
        hyper-parameters in order of first appearance - weights / biases first, norm parameters (WEIGHT_DECAY_NORM) second
        [D2-recall]; every buffer is shape-checked against the tensor it is mapped to, any mismatch raises ValueError and nothing
        is written).  The round-1..3 flat `momentum_buffer` form carried no layout information and is refused."""
        if isinstance(sd, dict) and "momentum" in sd and str(sd.get("format", "")).startswith("utv2-arena-sgd"):
            self._load_momentum(sd["momentum"])
            self.param_groups[0]["lr"] = sd["lr"]
            return
        if isinstance(sd, dict) and "state" in sd and "param_groups" in sd:
            mv = self._momentum_views()
            groups = sd["param_groups"]
            ids = [i for g in groups for i in g["params"]]
            if len(ids) != len(mv):
                raise ValueError("optimizer state holds %d parameters, this model trains %d" % (len(ids), len(mv)))
            if len(groups) == 1 or len(groups) == len(ids):
                # one group (equal hyper-parameters everywhere) or one group PER parameter (Detectron2 releases without
                # reduce_param_groups): the ids follow model.named_parameters() = this model's key order, kinds interleaved
                order = list(mv)
                if len(groups) == len(ids) and len(ids) > 1:
                    # ids are mapped by POSITION: a [256] conv bias and a [256] GroupNorm weight pass every shape check, so the per-group
                    # weight decay must tell the same story as this model's kinds - constant within a kind, different between kinds
                    # whenever the checkpoint holds more than one value (a different registration order fails here, not silently)
                    wd_of = {}
                    for k, g in zip(order, groups):
                        wd_of.setdefault(mv[k][0].kind, set()).add(g.get("weight_decay"))
                    all_wd = set().union(*wd_of.values())
                    if len(all_wd) > 1 and (any(len(v) != 1 for v in wd_of.values()) or
                                            len({next(iter(v)) for v in wd_of.values()}) != len(wd_of)):
                        raise ValueError("optimizer state: one parameter group per tensor, but their weight decays %s cannot be told apart "
                                         "as this model's %s tensors in named_parameters() order" % (sorted(map(str, all_wd)), sorted(wd_of)))
            else:
                # reduce_param_groups: parameters grouped by equal hyper-parameters in order of first appearance.  The groups must
                # be told apart by their weight decay and hold exactly this model's tensors of each kind - a [256] conv bias
                # against a [256] GroupNorm weight passes every shape check, so the layout itself is verified
                kinds = []
                for k, (h, _, _) in mv.items():
                    if h.kind not in kinds:
                        kinds.append(h.kind)
                by_kind = {kd: [k for k, (h, _, _) in mv.items() if h.kind == kd] for kd in kinds}
                wds = [g.get("weight_decay") for g in groups]
                if len(groups) != len(kinds) or [len(g["params"]) for g in groups] != [len(by_kind[kd]) for kd in kinds] or \
                        len(set(wds)) != len(wds):
                    raise ValueError("optimizer state: %d parameter groups of sizes %s (weight decay %s) cannot be told apart as this "
                                     "model's %s groups of sizes %s" % (len(groups), [len(g["params"]) for g in groups], wds, kinds,
                                                                        [len(by_kind[kd]) for kd in kinds]))
                order = [k for kd in kinds for k in by_kind[kd]]
            per_key = {}
            for k, i in zip(order, ids):
                buf = sd["state"].get(i, {}).get("momentum_buffer")
                per_key[k] = torch.zeros_like(mv[k][1]) if buf is None else buf
            self._load_momentum(per_key)
            self.param_groups[0]["lr"] = sd["param_groups"][0]["lr"]
            return
        raise ValueError("not an ArenaSGD state (a flat `momentum_buffer` without key names cannot be mapped safely onto the arena)")


class AmpScalerState:
    """checkpointable view of the device-side GradScaler state {scale, found_inf, growth tracker} (fp16 mode; reference
    engine/trainer.py:207 GradScaler, saved by Detectron2's AMPTrainer.state_dict [D2-recall])"""

    def __init__(self, trainer):
        self.trainer = trainer

    def state_dict(self):
        st = self.trainer._amp_state
        if st is None:
            return {}
        scale, _, tracker = st.detach().cpu().tolist()
        return {"scale": scale, "_growth_tracker": int(tracker)}

    def load_state_dict(self, sd):
        st = self.trainer._amp_state
        if st is None or not sd:
            return
        st.copy_(torch.tensor([float(sd["scale"]), 0.0, float(sd.get("_growth_tracker", 0))], dtype=torch.float32))


class WarmupMultiStepLR:
    """D2 WarmupMultiStepLR [D2-recall]: lr = base * gamma^{#milestones <= it} * warmup(it)."""

    def __init__(self, cfg, optimizer):
        s = cfg.SOLVER
        self.opt = optimizer
        self.base = s.BASE_LR
        self.steps = [x for x in s.STEPS if x <= s.MAX_ITER]
        self.gamma = s.GAMMA
        self.warmup_factor = s.WARMUP_FACTOR
        self.warmup_iters = s.WARMUP_ITERS
        self.method = s.WARMUP_METHOD
        self.last_iter = 0
        self._apply()

    def _warmup(self, it):
        """D2 _get_warmup_factor_at_iter [D2-recall]"""
        if it >= self.warmup_iters:
            return 1.0
        if self.method == "constant":
            return self.warmup_factor
        if self.method == "linear":
            a = it / self.warmup_iters
            return self.warmup_factor * (1 - a) + a
        raise ValueError("Unknown warmup method: {}".format(self.method))

    def lr_at(self, it):
        k = sum(1 for m in self.steps if m <= it)
        return self.base * (self.gamma ** k) * self._warmup(it)

    def _apply(self):
        self.opt.param_groups[0]["lr"] = self.lr_at(self.last_iter)

    def step(self):
        self.last_iter += 1
        self._apply()

    def state_dict(self):
        return {"last_iter": self.last_iter}

    def load_state_dict(self, sd):
        self.last_iter = sd["last_iter"]
        self._apply()


class WarmupTwoStageMultiStepLR(WarmupMultiStepLR):
    """reference solver/lr_scheduler.py:9-57: lr = base * warmup(it) * FACTOR_LIST[#milestones <= it] (an explicit factor per
    stage instead of gamma^k; len(FACTOR_LIST) == len(STEPS) + 1, milestones increasing)."""

    def __init__(self, cfg, optimizer):
        s = cfg.SOLVER
        self.milestones = list(s.STEPS)
        self.factors = list(s.FACTOR_LIST)
        if self.milestones != sorted(self.milestones):
            raise ValueError("Milestones should be a list of increasing integers. Got {}".format(self.milestones))
        if len(self.milestones) + 1 != len(self.factors):
            raise ValueError("Length of milestones should match length of factor_list.")
        super().__init__(cfg, optimizer)

    def lr_at(self, it):
        k = sum(1 for m in self.milestones if m <= it)
        return self.base * self.factors[k] * self._warmup(it)


class WarmupCosineLR(WarmupMultiStepLR):
    """D2 WarmupCosineLR [D2-recall]: lr = base * warmup(it) * 0.5 * (1 + cos(pi * it / MAX_ITER))"""

    def __init__(self, cfg, optimizer):
        self.max_iter = cfg.SOLVER.MAX_ITER
        super().__init__(cfg, optimizer)

    def lr_at(self, it):
        import math
        return self.base * self._warmup(it) * 0.5 * (1.0 + math.cos(math.pi * it / self.max_iter))


def build_lr_scheduler(cfg, optimizer):
    """reference solver/build.py:9-45"""
    name = cfg.SOLVER.LR_SCHEDULER_NAME
    if name == "WarmupMultiStepLR":
        return WarmupMultiStepLR(cfg, optimizer)
    if name == "WarmupCosineLR":
        return WarmupCosineLR(cfg, optimizer)
    if name == "WarmupTwoStageMultiStepLR":
        return WarmupTwoStageMultiStepLR(cfg, optimizer)
    raise ValueError("Unknown LR scheduler: {}".format(name))


class _TrainerBase:
    """Shared machinery of the two trainers."""

    log_period = 20

    @classmethod
    def build_model(cls, cfg):
        return build_model(cfg)

    @classmethod
    def build_optimizer(cls, cfg, model):
        return ArenaSGD(cfg, model)

    @classmethod
    def build_lr_scheduler(cls, cfg, optimizer):
        return build_lr_scheduler(cfg, optimizer)

    @classmethod
    def build_train_loader(cls, cfg):
        """trainer.py:110-112: the two-crop semi-supervised loader (ubteacher/data, GPU mapper) when the configured training sets are
        registered in the DatasetCatalog; otherwise (no datasets exist in this environment) the synthetic COCO-shaped loader with the
        same output contract: 4 lists of dicts per iteration."""
        from ..data import DatasetCatalog, build_detection_semisup_train_loader_two_crops
        names = cfg.DATASETS.TRAIN_LABEL + cfg.DATASETS.TRAIN_UNLABEL if cfg.DATASETS.CROSS_DATASET else cfg.DATASETS.TRAIN
        if len(names) and all(DatasetCatalog.available(n) for n in names):
            return build_detection_semisup_train_loader_two_crops(cfg, mapper=None)
        return SyntheticTwoCropLoader(cfg)

    def _common_init(self, cfg, data_loader=None):
        self.cfg = cfg
        # SOLVER.AMP.ENABLED (reference: autocast + GradScaler, trainer.py:194-198,318-349,423-426) selects the 16-bit MFMA conv kernels in
        # the REFERENCE's own element type: IEEE fp16 with GradScaler's dynamic loss scale kept on the device - the mode whose losses
        # are shown to stay within 1e-3 of the fp32 step (tests/test_conv_bf16_gpu.py, bench.py `f32.f16_vs_f32_first_step_rel_dev`).
        # UTV2_PRECISION=bf16 opts into bfloat16 (fp32's exponent range: no loss scaling, nothing to overflow on unnormalised features;
        # 3 mantissa bits fewer: loss_fcos_cls moves by ~2e-2, outside the 1e-3 bound - BASELINE configs[4] names this mode).  The config
        # surface is the reference's, key for key: the 16-bit type is an environment choice, not a new config key.
        import os
        # (UTV2_PRECISION names the 16-bit type of the AMP path only: with SOLVER.AMP.ENABLED False the step is exact fp32 whatever it says;
        # UTV2_PRECISION=fp32 switches an AMP config back to fp32)
        env_kind = {"f32": "fp32", "f16": "fp16"}.get(os.environ.get("UTV2_PRECISION", ""), os.environ.get("UTV2_PRECISION", ""))
        if env_kind not in ("", "fp32", "fp16", "bf16"):
            raise ValueError("UTV2_PRECISION must be fp16, bf16 or fp32, got %r" % env_kind)
        amp_kind = "fp32" if (not cfg.SOLVER.AMP.ENABLED or env_kind == "fp32") else (env_kind or "fp16")
        ops.set_precision(amp_kind)
        self._amp_state = None
        if ops.PRECISION[0] == "fp16":
            self._amp_state = torch.tensor([65536.0, 0.0, 0.0], dtype=torch.float32).to(self.model.store.flat.device)   # GradScaler init_scale
        self.start_iter = 0
        self.max_iter = cfg.SOLVER.MAX_ITER
        self.iter = 0
        self.world_size = comm.get_world_size()
        # the data-parallel machinery (replica broadcast, bucketed all-reduce from the wgrad stream) also runs in a world of ONE
        # rank when asked to: a single-GPU box can then drive the real RCCL backend through every call the N-rank step makes
        self._dp_single_rank = comm.is_dist() and os.environ.get("UTV2_DP_SINGLE_RANK") == "1"
        self._data_loader = data_loader if data_loader is not None else self.build_train_loader(cfg)
        self._data_loader_iter = iter(self._data_loader)
        self.storage = None
        self._pending_metrics = None
        self._last_metrics = {}
        ensem = EnsembleTSModel(self.model_teacher, self.model)
        self.ensem_ts_model = ensem
        self.checkpointer = DetectionTSCheckpointer(ensem, cfg.OUTPUT_DIR, optimizer=self.optimizer, scheduler=self.scheduler,
                                                    grad_scaler=AmpScalerState(self))
        self._setup_grad_sync()
        self.sync_replicas()

    _dp_single_rank = False

    @property
    def data_parallel(self):
        return self.world_size > 1 or self._dp_single_rank

    def sync_replicas(self):
        """Data parallel: every rank starts from rank 0's student, teacher and momentum, as DistributedDataParallel does at
        construction in the reference (trainer.py:59-63).  Only gradients are exchanged afterwards, so replicas that differ here
        (per-rank seeding such as Detectron2's SEED + rank, a checkpoint only rank 0 can read) would stay different for good."""
        if self.data_parallel:
            for t in (self.model.flat_state(), self.model_teacher.flat_state(), self.model.store.mom):
                if t is not None:
                    dist.broadcast(t, 0)
            self.model.store.touch()
            self.model_teacher.store.touch()
            ops.bump_version()

    # -- reference API ------------------------------------------------------------------------
    def resume_or_load(self, resume=True):
        checkpoint = self.checkpointer.resume_or_load(self.cfg.MODEL.WEIGHTS, resume=resume)
        if resume and self.checkpointer.has_checkpoint():
            self.start_iter = checkpoint.get("iteration", -1) + 1
        self.sync_replicas()

    def train(self):
        self.train_loop(self.start_iter, self.max_iter)

    def build_writers(self):
        """reference DefaultTrainer.build_writers (what `PeriodicWriter(self.build_writers(), period=20)`, engine/trainer.py:549-551, drives):
        the console line and OUTPUT_DIR/metrics.json.  The step keeps its metrics on the device and reads them out every `log_period`
        iterations (one host sync per 20 steps instead of ~30 per step), so a written value is the SAMPLED step's, not a 20-step median."""
        from ..d2.events import CommonMetricPrinter, JSONWriter
        ws = [CommonMetricPrinter(self.max_iter, window_size=1)]
        if self.cfg.OUTPUT_DIR:
            ws.append(JSONWriter(os.path.join(self.cfg.OUTPUT_DIR, "metrics.json"), window_size=1))
        return ws

    def _evaluate_student_and_teacher(self):
        """the two EvalHooks of reference engine/trainer.py:533-546: the student's results under `<key>_student`, then the teacher's under
        their own keys; every rank takes part (the evaluator gathers), the scalars go to the storage flattened `a/b` like EvalHook does"""
        def flatten(d, prefix=""):
            out = {}
            for k, v in d.items():
                if isinstance(v, dict):
                    out.update(flatten(v, prefix + str(k) + "/"))
                else:
                    out[prefix + str(k)] = v
            return out
        # (inference_on_dataset restores each model's mode: the FCOS teacher lives in eval mode, the Faster-RCNN teacher in training mode -
        # its branch calls are training-mode forwards, reference rcnn.py:60-61)
        self._last_eval_results_student = self.test(self.cfg, self.model)
        student = {k + "_student": v for k, v in (self._last_eval_results_student or {}).items()}
        self._last_eval_results_teacher = self.test(self.cfg, self.model_teacher)
        for res in (student, self._last_eval_results_teacher or {}):
            res = {k: v for k, v in res.items() if not str(k).startswith("_")}       # "_speed": this package's timing record, not a metric
            flat = {k: float(v) for k, v in flatten(res).items() if isinstance(v, (int, float)) and v == v}
            if flat and comm.is_main_process() and self.storage is not None:
                self.storage.put_scalars(smoothing_hint=False, **flat)
        comm.synchronize()

    def train_loop(self, start_iter, max_iter):
        logger = logging.getLogger(__name__)
        logger.info("Starting training from iteration {}".format(start_iter))
        self.iter = self.start_iter = start_iter
        self.max_iter = max_iter
        writers = self.build_writers() if comm.is_main_process() else []
        eval_period = int(self.cfg.TEST.EVAL_PERIOD)
        t_mark, t_acc, it_mark = time.perf_counter(), 0.0, start_iter
        with EventStorage(start_iter) as self.storage, StepGC() as step_gc:
            try:
                for self.iter in range(start_iter, max_iter):
                    self.run_step_full_semisup()
                    lr = float(self.optimizer.param_groups[0]["lr"])     # the rate this step ran at (hooks.LRScheduler logs it before stepping)
                    self.scheduler.step()
                    step_gc.tick()
                    period = self.cfg.SOLVER.CHECKPOINT_PERIOD
                    if comm.is_main_process() and period > 0 and (self.iter + 1) % period == 0:
                        self.checkpointer.save("model_{:07d}".format(self.iter), iteration=self.iter)
                    if comm.is_main_process() and self.iter + 1 >= max_iter and self.cfg.OUTPUT_DIR:
                        self.checkpointer.save("model_final", iteration=self.iter)   # D2 PeriodicCheckpointer [D2-recall]
                    last = self.iter + 1 >= max_iter
                    log_due = (self.iter + 1) % self.log_period == 0 or last
                    if log_due:
                        if last:
                            self.flush_metrics()
                        # the metrics have just been read out (a device sync): wall time per iteration since the last read-out
                        now = time.perf_counter()
                        if self.iter + 1 > it_mark and comm.is_main_process():
                            self.storage.put_scalar("time", (t_acc + now - t_mark) / (self.iter + 1 - it_mark), smoothing_hint=False)
                        t_mark, t_acc, it_mark = now, 0.0, self.iter + 1
                    # hooks.EvalHook: every TEST.EVAL_PERIOD iterations, and once after the last one
                    if eval_period > 0 and ((self.iter + 1) % eval_period == 0 or last):
                        t_acc += time.perf_counter() - t_mark
                        self._evaluate_student_and_teacher()
                        t_mark = time.perf_counter()      # evaluation time is not iteration time
                    if writers and log_due:
                        self.storage.put_scalar("lr", lr, smoothing_hint=False)
                        for w in writers:
                            w.write(self.storage)
                    self.storage.step()
            except Exception:
                logger.exception("Exception during training:")
                raise
            finally:
                self.flush_metrics()
                for w in writers:
                    w.close()

    # -- one step as a hipGraph (round 4) ----------------------------------------------------------------
    def run_step_graph(self):
        """run_step_full_semisup as ONE hipGraph launch (torch.cuda.graph capture of the whole iteration: teacher EMA and forward on its
        side stream, pseudo-labelling, the student's forward / losses / backward with the weight gradients on their lanes, the AMP
        scaler, SGD - ~490 kernel nodes on four streams).  The step has no host synchronisation and no data-dependent shapes (padded
        detections, device-side counts), which is what makes the capture legal.  MEASURED (round 4, ROCm 7.2, bench.py
        `host.step_as_hipgraph`): the replayed step takes the GPU as long as the eager one (25.5-25.8 against 25.4 ms - the chip is busy
        either way, there are no launch gaps to win), and hipGraphLaunch keeps the host busy for 17 ms per step (the runtime walks the
        node list on the host) against 10 ms of Python for the eager step: no gain on either side on this stack, so it is NOT the
        default; kept as an opt-in (a future runtime with device-side graph launch would change the host figure) and as the proof
        that the step is capture-clean.
        Conditions, checked here: a loader that hands the SAME device tensors every step (`static_batches`: the synthetic loader of the
        benchmark; real loaders produce another canvas per batch), one rank (collectives inside a capture are untested on this
        stack), the post-burn-in branch, no metric flush inside the step.  A changed learning rate re-captures (the rate is a launch
        argument): in a real schedule that is once per milestone after the warm-up.  The first call after (re)capture conditions
        runs eagerly (allocator warm-up)."""
        if not getattr(self._data_loader, "static_batches", False):
            raise RuntimeError("run_step_graph needs a loader with static device batches (data.synthetic.SyntheticTwoCropLoader, one batch)")
        if self.data_parallel or self.model.device.type != "cuda":
            raise RuntimeError("run_step_graph: one CUDA rank only")
        if self.iter < self.cfg.SEMISUPNET.BURN_UP_STEP + 1:
            return self.run_step_full_semisup()
        # the flag is scoped to this call (it used to stay set for every later eager step and every other trainer of the
        # process); it covers the eager warm-up steps too - work they parked across the step boundary would be joined INSIDE the capture
        prev = ops.STEP_GRAPH[0]
        ops.STEP_GRAPH[0] = True
        try:
            return self._run_step_graph()
        finally:
            ops.STEP_GRAPH[0] = prev

    def _run_step_graph(self):
        S = self.cfg.SEMISUPNET
        # every host-side decision the captured step bakes in is part of the key: the learning rate (a launch argument) and the
        # iteration-dependent teacher update (TEACHER_UPDATE_ITER > 1: steps with and without the EMA are two graphs)
        ema_step = (self.iter - S.BURN_UP_STEP) % S.TEACHER_UPDATE_ITER == 0
        key = (float(self.optimizer.param_groups[0]["lr"]), ema_step)
        graphs = self.__dict__.setdefault("_step_graphs", {})
        if graphs and next(iter(graphs))[0] != key[0]:
            graphs.clear()                                   # a new learning rate retires every captured step
        st = graphs.setdefault(key, {"graph": None, "warm": 0})
        flush = (self.iter + 1) % self.log_period == 0
        if st["graph"] is None:
            if st["warm"] < 2 or flush:
                st["warm"] += 1
                return self.run_step_full_semisup()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st["loss"] = self.run_step_full_semisup()
            st["graph"], st["pending"] = g, self._pending_metrics
        st["graph"].replay()
        self._pending_metrics = st["pending"]        # the captured metric tensors were just rewritten by the replay
        if flush:
            self.flush_metrics()
        return st["loss"]

    # -- EMA (trainer.py:468-486 / :950-968) ---------------------------------------------------------
    @torch.no_grad()
    def _update_teacher_model(self, keep_rate=0.996):
        if not hasattr(self, "_sd_keys"):  # key sets are fixed by construction: check once, not per step
            self._sd_keys = (list(self.model.state_dict().keys()), list(self.model_teacher.state_dict().keys()))
        s_keys, t_keys = self._sd_keys
        if s_keys != t_keys:
            sk = set(s_keys)
            for k in t_keys:
                if k not in sk:
                    raise Exception("{} is not found in student model".format(k))
        ts = self.model_teacher.store
        m16 = ts.mirror16(need_fresh=False)
        hip.ema_axpby(self.model_teacher.flat_state(), self.model.flat_state(), keep_rate, mirror16=m16)
        ts.touch()
        if m16 is not None:
            ts.mirror16_written()

    # -- gradient exchange: ONE flat all-reduce (DDP mean semantics) -------------------------------------
    def _setup_grad_sync(self):
        """data parallel: cut the gradient arena into buckets that are all-reduced while backward is still running"""
        self._grad_sync = None
        if self.data_parallel and os.environ.get("UTV2_OVERLAP_ALLREDUCE", "1") != "0":
            from ..utils.grad_sync import GradBuckets
            st = self.model.store
            hs = [h for h in st.handles if h.g is not None]
            self._grad_sync = GradBuckets(st.grad, hs)
            ops.GRAD_SYNC[0] = self._grad_sync

    def _backward(self, losses):
        gs = getattr(self, "_grad_sync", None)
        if gs is not None:
            gs.arm()
        ops.wgrad_side_stream(True)      # weight gradients on a side stream, next to the dgrad chain (ops._wgrad_launch)
        try:
            if getattr(self, "_amp_state", None) is not None:
                losses = losses * self._amp_state[0]     # scaler.scale(losses): every gradient of the backward carries the loss scale
            losses.backward()
            ops.FanIn.check()            # a gradient handed from one producer to another's epilogue must have been picked up
        finally:
            ops.wgrad_side_stream(False)
            ops.join_wgrad_stream()      # the optimizer step (and whatever reads the gradient arena) follows on the main stream

    def _allreduce_grads(self):
        if self.data_parallel:
            gs = getattr(self, "_grad_sync", None)
            if gs is not None:
                gs.finish()
            else:
                dist.all_reduce(self.model.store.grad, op=dist.ReduceOp.SUM)
            return 1.0 / self.world_size
        return 1.0

    # -- metrics (trainer.py:431-466 / :914-948), device resident until flushed -------------------------
    def _write_metrics(self, metrics_dict):
        # host scalars (EMA rate, data time) stay on the host: a `torch.tensor(x, device=cuda)` here is a pageable host-to-device copy,
        # which PyTorch follows with a stream synchronize - the whole forward drained right before backward starts
        keys, vals, host_items = [], [], []
        for k, v in metrics_dict.items():
            if isinstance(v, torch.Tensor):
                keys.append(k)
                vals.append(v.detach().reshape(()).float())
            else:
                host_items.append((k, float(v)))
        self._pending_metrics = (keys, torch.stack(vals) if vals else None, host_items)
        if (self.iter + 1) % self.log_period == 0:
            self.flush_metrics()

    def flush_metrics(self):
        if self._pending_metrics is None:
            return self._last_metrics
        keys, vals, host_items = self._pending_metrics
        self._pending_metrics = None
        host = vals.cpu().tolist() if vals is not None else []  # the one host sync
        md = dict(zip(keys, host))
        md.update(host_items)
        all_md = comm.gather(md)
        if comm.is_main_process():
            if "data_time" in all_md[0]:
                data_time = max(x.pop("data_time") for x in all_md)
                if self.storage is not None:
                    self.storage.put_scalar("data_time", data_time)
            md = {k: sum(x[k] for x in all_md) / len(all_md) for k in all_md[0].keys()}
            total = sum(v for k, v in md.items() if k[:4] == "loss")
            if self.storage is not None:
                self.storage.put_scalar("total_loss", total)
                if len(md) > 1:
                    self.storage.put_scalars(**md)
            md["total_loss"] = total
            self._last_metrics = md
        return self._last_metrics

    # -- evaluation path (trainer.py:554-608) --------------------------------------------------------
    @classmethod
    def build_test_loader(cls, cfg, dataset_name):
        """reference engine/trainer.py:114-116 (DefaultTrainer.build_test_loader -> Detectron2 build_detection_test_loader): a registered
        set whose files are there (COCO-format json + image folder, data/datasets.py) goes through the test-time mapper; when the
        named set is not available - the shipped configs name coco_2017_val and no dataset exists in this environment - the synthetic
        COCO-shaped test set stands in, with a warning (its numbers exercise the plumbing, not accuracy)."""
        from ..data import DatasetCatalog, build_detection_test_loader
        if DatasetCatalog.available(dataset_name):
            return build_detection_test_loader(cfg, dataset_name)
        import logging
        logging.getLogger(__name__).warning("test set %r is not available (not registered, or its annotation file is missing): evaluating "
                                            "on the synthetic COCO-shaped test set", dataset_name)
        from ..data.synthetic import SyntheticTestLoader
        return SyntheticTestLoader(cfg)

    @classmethod
    def build_evaluator(cls, cfg, dataset_name, output_folder=None):
        from ..data import DatasetCatalog
        from ..evaluation import COCOBoxEvaluator
        rcnn = cfg.SEMISUPNET.Trainer == "ubteacher_rcnn"
        nc = cfg.MODEL.ROI_HEADS.NUM_CLASSES if rcnn else cfg.MODEL.FCOS.NUM_CLASSES
        return COCOBoxEvaluator(nc, dataset_name=dataset_name if DatasetCatalog.available(dataset_name) else None)

    @classmethod
    def test(cls, cfg, model, evaluators=None):
        """dict of result metrics per test set (the single dict itself when there is one), reference trainer.py:554-608"""
        from ..evaluation import inference_on_dataset
        names = list(cfg.DATASETS.TEST) or ["synthetic_val"]
        if evaluators is not None and not isinstance(evaluators, (list, tuple)):
            evaluators = [evaluators]
        if evaluators is not None:
            assert len(names) == len(evaluators), "{} != {}".format(len(names), len(evaluators))
        results = OrderedDict()
        for idx, name in enumerate(names):
            loader = cls.build_test_loader(cfg, name)
            evaluator = evaluators[idx] if evaluators is not None else cls.build_evaluator(cfg, name)
            results[name] = inference_on_dataset(model, loader, evaluator, cfg)
            if comm.is_main_process():
                # reference engine/trainer.py:596-603: the results of each set in Detectron2's print_csv_format shape [D2-recall]
                logger = logging.getLogger(__name__)
                logger.info("Evaluation results for {} in csv format:".format(name))
                for task, res in results[name].items():
                    if isinstance(res, dict) and not str(task).startswith("_"):
                        keys = [k for k in res if "-" not in k]
                        logger.info("copypaste: Task: {}".format(task))
                        logger.info("copypaste: " + ",".join(keys))
                        logger.info("copypaste: " + ",".join("{0:.4f}".format(float(res[k])) for k in keys))
        if len(results) == 1:
            results = list(results.values())[0]
        return results


class UBTeacherTrainer(_TrainerBase):
    """FCOS trainer (reference trainer.py:38-608)."""

    def __init__(self, cfg, data_loader=None):
        self.model = self.build_model(cfg)
        self.optimizer = self.build_optimizer(cfg, self.model)
        self.model_teacher = self.build_model(cfg)
        self.model_teacher.eval()  # trainer.py:55
        self.scheduler = self.build_lr_scheduler(cfg, self.optimizer)
        self.pseudo_generator = PseudoGenerator(cfg)
        self.fuse_student_passes = os.environ.get("UTV2_FUSE_STUDENT_PASSES", "1") != "0"
        self.fuse_teacher_nms = os.environ.get("UTV2_FUSE_TEACHER_NMS", "1") != "0"
        self.overlap_teacher = os.environ.get("UTV2_OVERLAP_TEACHER", "1") != "0"
        self._side_stream = None
        self._common_init(cfg, data_loader)

    # pseudo-label dict surgery (trainer.py:161-175)
    def remove_label(self, label_data):
        for d in label_data:
            if "instances" in d.keys():
                del d["instances"]
        return label_data

    def add_label(self, unlabled_data, label, labeltype=""):
        key = {"class": "instances_class", "reg": "instances_reg"}.get(labeltype, "instances")
        if isinstance(label, (list, tuple)):
            for d, inst in zip(unlabled_data, label):
                d[key] = inst
        else:  # batched, device-resident pseudo labels: every datum references the same batch object
            for d in unlabled_data:
                d[key] = label
        return unlabled_data

    @torch.no_grad()
    def _teacher_pseudo_labels(self, unlabel_data_k):
        """trainer.py:224-292: teacher forward on the weak views, detections under the two ranking criteria, thresholding into the
        classification and the regression pseudo sets"""
        cfg = self.cfg
        S = cfg.SEMISUPNET
        fo_t = self.model_teacher.proposal_generator.fcos_outputs     # eval mode: reads the *_TEST thresholds
        fo_p = self.pseudo_generator.fcos_output                       # never put in eval mode: *_TRAIN
        same = ((fo_t.pre_nms_thresh_test, fo_t.pre_nms_topk_test, fo_t.post_nms_topk_test, fo_t.nms_thresh) ==
                (fo_p.pre_nms_thresh_train, fo_p.pre_nms_topk_train, fo_p.post_nms_topk_train, fo_p.nms_thresh))
        if self.fuse_teacher_nms and same and not fo_t.training:
            # both criteria (classification / regression pseudo sets, trainer.py:232-251) in ONE set of launches over
            # (criterion, image) pairs: identical detections, half the latency-bound top-k / NMS kernels
            (pred_teacher, pred_teacher_loc), raw_pred_teacher = self.model_teacher(
                unlabel_data_k, output_raw=True,
                nms_method=(cfg.MODEL.FCOS.NMS_CRITERIA_TRAIN, cfg.MODEL.FCOS.NMS_CRITERIA_REG_TRAIN), branch="teacher_weak")
        else:
            pred_teacher, raw_pred_teacher = self.model_teacher(
                unlabel_data_k, output_raw=True, nms_method=cfg.MODEL.FCOS.NMS_CRITERIA_TRAIN, branch="teacher_weak")
            pred_teacher_loc = self.pseudo_generator.nms_from_dense(raw_pred_teacher, cfg.MODEL.FCOS.NMS_CRITERIA_REG_TRAIN)

        if S.PSEUDO_BBOX_SAMPLE == "thresholding":
            cur_threshold = S.BBOX_THRESHOLD
        elif S.PSEUDO_BBOX_SAMPLE == "thresholding_cls_ctr":
            cur_threshold = (S.BBOX_THRESHOLD, S.BBOX_CTR_THRESHOLD)
        else:
            raise ValueError
        if S.PSEUDO_BBOX_SAMPLE_REG == "thresholding":
            cur_threshold_reg = S.BBOX_THRESHOLD_REG
        elif S.PSEUDO_BBOX_SAMPLE_REG == "thresholding_cls_ctr":
            cur_threshold_reg = (S.BBOX_THRESHOLD_REG, S.BBOX_CTR_THRESHOLD_REG)
        else:
            raise ValueError

        pseudo_cls, _ = self.pseudo_generator.process_pseudo_label(pred_teacher, cur_threshold, "roih", S.PSEUDO_BBOX_SAMPLE)
        pseudo_reg, _ = self.pseudo_generator.process_pseudo_label(pred_teacher_loc, cur_threshold_reg, "roih", S.PSEUDO_BBOX_SAMPLE_REG)
        self._last_pseudo = (pseudo_cls, pseudo_reg)
        return pseudo_cls, pseudo_reg

    def run_step_full_semisup(self):
        cfg = self.cfg
        S = cfg.SEMISUPNET
        assert self.model.training, "[UBTeacherTrainer] model was changed to eval mode!"
        start = time.perf_counter()
        label_data_q, label_data_k, unlabel_data_q, unlabel_data_k = next(self._data_loader_iter)
        data_time = time.perf_counter() - start

        if self.iter < S.BURN_UP_STEP:
            record_dict = self.model(label_data_q + label_data_k, branch="labeled")
            loss_dict = {k: v for k, v in record_dict.items() if k[:4] == "loss" and k[-3:] != "val"}
            losses = sum(loss_dict.values())
        else:
            if self.iter == S.BURN_UP_STEP:
                self._update_teacher_model(keep_rate=0.00)
                ema_keep_rate = S.EMA_KEEP_RATE
            elif (self.iter - S.BURN_UP_STEP) % S.TEACHER_UPDATE_ITER == 0:
                ema_keep_rate = S.EMA_KEEP_RATE
                self._update_teacher_model(keep_rate=ema_keep_rate)
            else:
                ema_keep_rate = S.EMA_KEEP_RATE  # guards the reference's unbound-name bug
            record_dict = {"ema_rate_1000x": ema_keep_rate * 1000}

            all_label_data = label_data_q + label_data_k
            # The reference runs two student forwards (trainer.py:396-411).  Every layer is per-image (FrozenBN, per-image
            # GroupNorm), so when both lists pad to the same canvas they are ONE batch here: larger GEMMs, one weight
            # gradient per layer instead of two.  Different canvases (zero padding differs) keep the two passes.
            # (SEMISUPNET.PSEUDO_CLS_IGNORE_NEAR reaches the pseudo branch only - trainer.py:340,347 - where the reference never reads
            # the keep_locations it fills, fcos_outputs.py:487-631: it does not change a loss, so it does not constrain the fusion)
            fuse = (self.fuse_student_passes
                    and self.model.padded_canvas(all_label_data) == self.model.padded_canvas(unlabel_data_q))
            # The student's forward needs the pseudo labels only in its loss kernels: with one student batch the teacher (forward,
            # top-k, decode, NMS, thresholding - the last four latency-bound) runs on a side stream NEXT TO the student's backbone /
            # FPN / towers and fills the partial rounds and tails those leave on the chip.
            overlap = fuse and self.overlap_teacher and self.model.device.type == "cuda"
            ctx = None
            if overlap:
                main = torch.cuda.current_stream(self.model.device)
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(self.model.device)
                side = self._side_stream
                side.wait_stream(main)           # the EMA update above, the loader's copies
                torch.cuda.set_stream(side)
            try:
                pseudo_cls, pseudo_reg = self._teacher_pseudo_labels(unlabel_data_k)
                if overlap:
                    for pb in (pseudo_cls, pseudo_reg):      # allocated on the side stream, consumed on the main one
                        for t in pb.f.values():
                            if torch.is_tensor(t):
                                t.record_stream(main)
            finally:
                if overlap:
                    torch.cuda.set_stream(main)
            if overlap:
                ctx = self.model.forward_joint_begin(all_label_data, unlabel_data_q)   # concurrent with the teacher
                main.wait_stream(side)

            unlabel_data_q = self.remove_label(unlabel_data_q)
            unlabel_data_k = self.remove_label(unlabel_data_k)
            unlabel_data_q = self.add_label(unlabel_data_q, pseudo_cls, "class")
            unlabel_data_k = self.add_label(unlabel_data_k, pseudo_cls, "class")
            unlabel_data_q = self.add_label(unlabel_data_q, pseudo_reg, "reg")
            unlabel_data_k = self.add_label(unlabel_data_k, pseudo_reg, "reg")

            all_unlabel_data = unlabel_data_q
            lu, lr = S.UNSUP_LOSS_WEIGHT, S.UNSUP_REG_LOSS_WEIGHT
            # the weighting below as (mul, div) per key: the fused pass folds it into its single scalar-tail launch
            lw = {"loss_fcos_cls": (1.0, lu + 1.0), "loss_fcos_ctr": (1.0, lu + 1.0), "loss_fcos_loc": (1.0, lr + 1.0),
                  "loss_fcos_cls_pseudo": (lu, lu + 1.0), "loss_fcos_ctr_pseudo": (lu, lu + 1.0), "loss_fcos_loc_pseudo": (lr, lr + 1.0)}
            if os.environ.get("UTV2_FUSE_LOSS_TAIL", "1") == "0":
                lw = None
            if ctx is not None:
                rec_l, record_unl = self.model.forward_joint_finish(ctx, all_unlabel_data, loss_weights=lw)
                record_dict.update(rec_l)
            elif fuse:
                rec_l, record_unl = self.model.forward_joint(all_label_data, all_unlabel_data, loss_weights=lw)
                record_dict.update(rec_l)
            else:
                record_dict.update(self.model(all_label_data, branch="labeled"))
                record_unl, raw_pred_student, instance_reg = self.model(
                    all_unlabel_data, output_raw=True, ignore_near=S.PSEUDO_CLS_IGNORE_NEAR, branch="unlabeled")
            for k, v in record_unl.items():
                record_dict[k + "_pseudo"] = v

            fused_total = record_dict.pop("weighted_total", None)
            loss_dict = {}
            for key in ([] if fused_total is not None else record_dict.keys()):
                if key[:4] != "loss":
                    continue
                if key in ("loss_fcos_ctr", "loss_fcos_cls"):
                    loss_dict[key] = record_dict[key] / (lu + 1.0)
                elif key in ("loss_fcos_ctr_pseudo", "loss_fcos_cls_pseudo"):
                    loss_dict[key] = record_dict[key] * lu / (lu + 1.0)
                elif key == "loss_fcos_loc":
                    loss_dict[key] = record_dict[key] / (lr + 1.0)
                elif key == "loss_fcos_loc_pseudo":
                    loss_dict[key] = record_dict[key] * lr / (lr + 1.0)
                else:
                    loss_dict[key] = record_dict[key] / (lu + 1.0)
            losses = fused_total if fused_total is not None else sum(loss_dict.values())

        metrics_dict = record_dict
        metrics_dict["data_time"] = data_time
        self._write_metrics(metrics_dict)

        self.optimizer.zero_grad()
        self._backward(losses)
        gscale = self._allreduce_grads()
        self.optimizer.step(grad_scale=gscale, amp_state=self._amp_state)
        return losses


from .rcnn_trainer import _make as _make_rcnn  # noqa: E402

UBRCNNTeacherTrainer = _make_rcnn(_TrainerBase)
