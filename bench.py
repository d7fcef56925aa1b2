"""bench.py - images/sec of the UTv2 training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: spawns its N ranks itself, or joins the world
                                                            torch.distributed.run started it in)

A "step" is one UBTeacherTrainer.run_step_full_semisup on one synthetic COCO-shaped batch
(FCOS R50-FPN, 4 labeled + 4 unlabeled 1333x800 images per GPU, post-burn-in: teacher EMA,
teacher forward + two-criteria pseudo-labelling, two student forwards, backward, SGD).
Inputs are resident in HBM before the timed region.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA, dense (not the 2:1-sparse headline)


def tune_for_pseudo_labels(trainer, batch, target_std=1.5, per_image=40):
    """Random-init weights give no confident detections; rescale the student's cls_logits (on the
    device, with the product's own forward) so the teacher emits some pseudo boxes per image: logits of std `target_std`, the bias
    placed so that ~`per_image` logits per weak image clear p = 0.6 (as cpu_baseline_run does for the oracle's copy of the problem):
    both pseudo-label sets (criteria "cls" and "cls_n_loc", threshold 0.5) are non-empty and the classification branch is not degenerate."""
    from ubteacher.modeling.fcos import PaddedBoxes  # noqa: F401
    m = trainer.model
    sd = m.state_dict()
    w, b = sd["proposal_generator.fcos_head.cls_logits.weight"], sd["proposal_generator.fcos_head.cls_logits.bias"]
    g = torch.Generator(device="cpu").manual_seed(0)
    w.copy_((torch.randn(w.shape, generator=g) * 0.01).to(w.device))
    b.zero_()
    m.store.touch()
    m.eval()
    with torch.no_grad():
        _, raw = m(batch[3], output_raw=True, nms_method="cls", branch="teacher_weak")
        lg = torch.cat([x.reshape(-1) for x in raw["logits_pred"]]).float()
        s = lg.std()
        scale = target_std / s.clamp(min=1e-12)
        kth = torch.topk(lg, per_image * len(batch[3])).values[-1] * scale
    m.train()
    m.store.touch()
    w.mul_(scale)
    b.fill_(float(0.405 - kth))
    sd["proposal_generator.fcos_head.bbox_pred_std.bias"].fill_(-3.0)
    m.store.touch()
    trainer._update_teacher_model(keep_rate=0.0)  # teacher := student
    sd["proposal_generator.fcos_head.bbox_pred_std.bias"].fill_(0.0)  # student less certain than teacher
    m.store.touch()  # weights were edited in place: invalidate the bf16 mirror


def tune_rcnn_for_pseudo_labels(trainer, batch, target_std=8.0, bg_bias=0.0):
    """Faster-RCNN counterpart: a random-init R50 has no normalised features - the RPN's deltas explode (every proposal clamps to the
    image, NMS keeps a handful) and the class scores are flat.  Rescale the RPN prediction layers and the predictor's cls_score,
    data-driven with the product's own forward, so that the RPN emits its ~1000 proposals per image and the teacher some confident
    detections above BBOX_THRESHOLD: the pseudo-label branch of the step then does real work."""
    from ubteacher import hip, ops
    m = trainer.model
    sd = m.state_dict()
    g = torch.Generator(device="cpu").manual_seed(0)
    wo, wd = sd["proposal_generator.rpn_head.objectness_logits.weight"], sd["proposal_generator.rpn_head.anchor_deltas.weight"]
    wo.copy_((torch.randn(wo.shape, generator=g) * 0.01).to(wo.device))
    wd.copy_((torch.randn(wd.shape, generator=g) * 0.01).to(wd.device))
    m.store.touch()
    with torch.no_grad():
        images = [x["image"].to(m.device) for x in batch[3]]
        x4, _ = hip.preprocess_images(images, m._mean_host, m._std_host, m.backbone.size_divisibility, bf16_stem=ops.amp())
        m.folder.fold()
        big, _, _ = m.proposal_generator._head(m.backbone(x4))
        s_obj, s_dl = big[:, :3].float().std(), big[:, 3:15].float().std()
    wo.mul_(1.0 / s_obj.clamp(min=1e-12))
    wd.mul_(0.1 / s_dl.clamp(min=1e-12))
    w, b = sd["roi_heads.box_predictor.cls_score.weight"], sd["roi_heads.box_predictor.cls_score.bias"]
    wb, ws = sd["roi_heads.box_predictor.bbox_pred.weight"], sd["roi_heads.box_predictor.bbox_pred_std.weight"]
    for t in (w, wb, ws):
        t.copy_((torch.randn(t.shape, generator=g) * 0.01).to(t.device))
    b.zero_()
    m.store.touch()
    with torch.no_grad():
        _, rpn, _, preds = m(batch[3], branch="unsup_data_weak")
        rows = rpn["valid"].reshape(-1).bool()
        s, s_d, s_s = (p_.float()[rows].std() for p_ in preds)
    w.mul_(target_std / s.clamp(min=1e-12))
    wb.mul_(0.5 / s_d.clamp(min=1e-12))
    ws.mul_(0.5 / s_s.clamp(min=1e-12))
    b[-1] = bg_bias
    # confident teacher boundaries, a less certain student (as tune_for_pseudo_labels does for FCOS): the "tsbetter" selection of
    # box_reg_pseudo_loss (fast_rcnn.py:1018-1092: c_t > c_s + 0.1 and c_t > 0.5 per boundary, c = 1 - sigmoid(std)) is then non-empty
    # and loss_box_reg_pseudo / its backward do real work inside the timed step
    sb = sd["roi_heads.box_predictor.bbox_pred_std.bias"]
    sb.fill_(-3.0)
    m.store.touch()
    trainer._update_teacher_model(keep_rate=0.0)  # teacher := student
    sb.fill_(0.0)
    m.store.touch()


class BoardPower:
    """Board power over the timed region: amdgpu hwmon power1_average / power1_input of EVERY card at ~25 Hz from a thread (a one-GPU
    lease on an eight-GPU host still shows all eight in sysfs: the card under test is the one with the highest mean) and the cap
    (power1_cap).  The step's MFMA kernels run at the board's power limit, which holds the shader clock below the 2.4 GHz the
    datasheet peak is quoted at (profiles/r05_power_probe.txt, r05_pp_power.txt, r05_mfma_peak.txt) - this is the live figure."""

    def __init__(self):
        import glob
        self.files = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            for leaf in ("power1_average", "power1_input"):
                if os.path.exists(os.path.join(d, leaf)):
                    self.files.append((os.path.join(d, leaf), os.path.join(d, "power1_cap")))
                    break
        self.rows, self._stop, self._t = [], False, None

    @staticmethod
    def _rd(path):
        try:
            with open(path) as f:
                return float(f.read().strip()) / 1e6
        except Exception:  # noqa: BLE001
            return None

    def start(self):
        if not self.files:
            return
        import threading

        def run():
            while not self._stop:
                self.rows.append([self._rd(pf) for pf, _ in self.files])
                time.sleep(0.04)
        self._t = threading.Thread(target=run, daemon=True)
        self._t.start()

    def stop(self):
        if self._t is None:
            return None
        self._stop = True
        self._t.join()
        n = len(self.files)
        means = []
        for c in range(n):
            v = [r[c] for r in self.rows if r[c] is not None]
            means.append(sum(v) / len(v) if v else 0.0)
        c = max(range(n), key=lambda i: means[i])
        v = [r[c] for r in self.rows if r[c] is not None]
        if not v:
            return None
        return {"mean_w": means[c], "max_w": max(v), "last_w": v[-1], "cap_w": self._rd(self.files[c][1]), "samples": len(v),
                "source": self.files[c][0], "note": "busiest card's hwmon power over the timed region (the sensor averages over ~1 s: "
                "`last_w` / `max_w` are the settled readings of a sub-second region)"}


class ConvTimer:
    """HIP-event timing (torch events recorded on the stream the kernels are launched on) of every launch of the
    dominant kernel inside the timed region.  Dominant kernel (largest share of GPU time in profiles/): the multi-level
    3x3 implicit-GEMM conv of the shared FCOS towers - forward AND dgrad launches, 256 -> 256 channels over all five FPN
    levels of the student batch in one launch:
        bf16: conv_igemm_bf16_rs<true,__bf16> (the row-span form of the ping-pong kernel, round 5; UTV2_PP_RS=0: conv_igemm_bf16_pp)
              on the whole rounds of 256 x 256 tiles + conv_igemm_bf16_v2<128,true,64,__bf16> on the
              remaining output rows (two kernels, one C-ABI call = one timed launch)      f32: conv_igemm_f32<128,0,true>"""

    def __init__(self, dtype):
        self.pairs = []
        self.pairs_gnb = []             # the tower dgrads in their GroupNorm-backward form (another instantiation, more epilogue work): timed apart
        self.enabled = False
        self.bf16 = dtype != "f32"      # a 16-bit MFMA mode (either type: the same kernels)
        self.entry = "conv2d_ml_fwd_bf16" if self.bf16 else "conv2d_ml_fwd"
        t16 = "_Float16" if dtype == "f16" else "__bf16"    # element type of the kernel library's 16-bit build (csrc/common.h h16_t)
        big = "pp" if os.environ.get("UTV2_PP_RS", "1") == "0" else "rs"
        self.kernel = ("conv_igemm_bf16_%s<true,%s>+conv_igemm_bf16_v2<128,true,64,%s>" % (big, t16, t16)) if self.bf16 else "conv_igemm_f32<128,0,true>"
        self.kernel_gnb = "conv_igemm_bf16_rs<true,%s,true>+conv_igemm_bf16_v2<128,true,64,%s,false,true>" % (t16, t16)

    def install(self):
        from ubteacher import hip
        orig = getattr(hip, self.entry)
        timer = self

        def wrapped(x2d, w, level_hw, N, *args, **kw):
            P, C = x2d.shape
            K, Kred = w.shape
            tiles = -(-P // 128) * -(-K // 128)
            if timer.bf16:  # the dispatch rule of launch_igemm16 (csrc/conv_bf16.hip) for this template instance
                out_dt = kw["out"].dtype if kw.get("out") is not None else kw.get("out_dtype", x2d.dtype)
                mine = (x2d.dtype == hip.h16_dtype() and out_dt == hip.h16_dtype() and K >= 256 and C % 64 == 0 and Kred >= 1024
                        and (P // 256) * -(-K // 256) >= 256)
            else:
                mine = K > 64 and C % 16 == 0
            if not (timer.enabled and mine):
                return orig(x2d, w, level_hw, N, *args, **kw)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            y = orig(x2d, w, level_hw, N, *args, **kw)
            e1.record()
            eb = 2 if timer.bf16 else 4
            # algorithmic HBM bytes: activations in and out (+ residual read) and the weights once
            bts = eb * P * C + eb * K * Kred + eb * P * K * (1 + (kw.get("residual") is not None))
            if kw.get("gnb") is not None:   # the GroupNorm-backward form: + the GroupNorm input, the mask plane, the partial sums
                bts += eb * P * K + P * K // 8 + 8 * K * -(-P // 64)
            (timer.pairs_gnb if kw.get("gnb") is not None else timer.pairs).append((e0, e1, 2.0 * P * K * Kred, float(bts)))
            return y

        setattr(hip, self.entry, wrapped)

    def summary(self, gnb=False):
        pairs = self.pairs_gnb if gnb else self.pairs
        if not pairs:
            return None
        ms = sum(p[0].elapsed_time(p[1]) for p in pairs)
        fl = sum(p[2] for p in pairs)
        by = sum(p[3] for p in pairs)
        n = len(pairs)
        return dict(launches=n, total_ms=ms, avg_us=1e3 * ms / n, tflops=fl / ms / 1e9, alg_bytes=by / n,
                    alg_gbps=by / ms / 1e6)


class WgradTimer:
    """HIP-event timing of the FCOS tower weight-gradient launches (conv_wgrad_bf16_pp - the 256-tile kernel on the ping-pong schedule, round 5 - + reduce_slabs16_f32 + the bias column sums:
    one C-ABI call = one timed launch): the largest single symbol of the round-1 profile."""
    kernel = ("conv_wgrad_bf16_pp+colsum_bf16_partial (FCOS tower 3x3 weight gradients; their split-K tails run folded, one reduce_slabs_table launch "
              "per <= 8 layers of a lane, outside this timing)" if os.environ.get("UTV2_WGRAD_FOLD", "0") == "1" else
              "conv_wgrad_bf16_pp+reduce_slabs16_f32+colsum_bf16_* (FCOS tower 3x3 weight gradients)")

    def __init__(self):
        self.pairs = []
        self.enabled = False

    def install(self):
        from ubteacher import hip
        orig = hip.conv2d_wgrad_bf16
        timer = self

        def wrapped(x, dy2d, dw, rowinfo, C, kh, kw, *args, **kwargs):
            M, K = dy2d.shape
            G = kwargs.get("groups", 1)      # paired towers: K = 512 (cls | bbox), C = 256 input channels per group
            mine = (timer.enabled and kh == 3 and kw == 3 and C == 256 and K in (256, 512) and x.dtype == hip.h16_dtype()
                    and dy2d.dtype == hip.h16_dtype() and M >= 65536)
            if not mine:
                return orig(x, dy2d, dw, rowinfo, C, kh, kw, *args, **kwargs)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(x, dy2d, dw, rowinfo, C, kh, kw, *args, **kwargs)
            e1.record()
            timer.pairs.append((e0, e1, 2.0 * M * K * kh * kw * C, 2.0 * M * (G * C + K) + 4.0 * K * kh * kw * C))
            return r

        hip.conv2d_wgrad_bf16 = wrapped

    summary = ConvTimer.summary


class CallCounter:
    """C-ABI calls per step (every HIP kernel of the product is launched through ubteacher.hip.call; one call = 1-3 kernels)"""

    def __init__(self):
        self.n = 0

    def install(self):
        from ubteacher import hip
        orig = hip.call
        me = self

        def counted(name, *a):
            me.n += 1
            return orig(name, *a)
        hip.call = counted


def pmc_traffic(kernel, model="fcos"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS command and THIS model
    (profiles/rNN_traffic.json for the FCOS step, profiles/rNN_rcnn_traffic.json for the Faster-RCNN step: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs; FETCH_SIZE doubled as MI355X_MICROARCH.md's HBM section prescribes for
    16-byte-per-lane reads on gfx950).  The newest round's file wins."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    stem = "traffic.json" if model == "fcos" else "%s_traffic.json" % model
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(here, "%s_%s" % (rnd, stem))
        if not os.path.exists(path):
            continue
        try:
            with open(path) as f:
                j = json.load(f)
            t = j.get("tower_conv") if kernel.startswith("conv_igemm") else None    # r03 files: one key whatever the 16-bit type
            t = t if t is not None else j.get(kernel, j.get(kernel.replace("_Float16", "__bf16")))
            if t is not None:
                return float(t["hbm_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            pass
    return None


def board_mfma_ceiling(dtype):
    """What the matrix pipes of this board sustain at its 1400 W power cap with NOTHING else running: the register-only MFMA loop of
    tools/probe/mfma_peak.hip (one wave per SIMD, the shipped kernels' 32x32x16 instruction, operand data with half of the A values zero
    like post-ReLU activations), from the committed run profiles/r05_mfma_peak2.txt.  The datasheet peak (2.5 PFLOP/s) is only reached on
    all-zero data; this figure is the practical ceiling the roofline fraction can be read against."""
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r05_mfma_peak2.txt")
    want = "32x32x16_bf16" if dtype == "bf16" else "32x32x16_f16"
    try:
        sect = open(path).read().split("## 256 workgroups x 256 threads")[1]
        m = re.search(r"^%s\s+N\(0,1\), half of A zero\s+([0-9.]+) TFLOP/s\s+clock ([0-9.]+) GHz" % re.escape(want), sect, re.M)
        return {"tflops": float(m.group(1)), "clock_ghz": float(m.group(2)), "source": "profiles/r05_mfma_peak2.txt (%s, half of A zero, one wave per SIMD)" % want}
    except (OSError, IndexError, AttributeError, ValueError):
        return None


def dispatches_per_step(model):
    """GPU dispatches (kernels + copies) per step, from the committed kernel trace of THIS command
    (profiles/rNN_<model>_4p4_bf16_timeline.txt, newest round, tools/rocpd_timeline.py over the last 6 of `bench.py --timed-only` steps)"""
    import re
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    names = ["%s_%s_4p4_%s_timeline.txt" % (r, model, t) for r in ("r06", "r05", "r04", "r03", "r02") for t in ("f16", "bf16")]
    path = next((q for q in (os.path.join(here, n) for n in names) if os.path.exists(q)), os.path.join(here, "r02_%s_4p4_bf16_timeline.txt" % model))
    try:
        with open(path) as f:
            m = re.search(r"window [0-9.]+ ms \(([0-9.]+) ms / step\), (\d+) dispatches", f.read())
        if not m:
            return None
        return {"dispatches_per_step": int(m.group(2)) / 6.0, "ms_per_step_under_rocprofv3": float(m.group(1)), "source": os.path.relpath(path, os.path.dirname(os.path.abspath(__file__)))}
    except OSError:
        return None


def set_amp_type(dtype):
    """The 16-bit type of the SOLVER.AMP.ENABLED path is an environment choice (the config surface stays the reference's): the package
    default is the reference's own autocast type, fp16; bf16 (BASELINE configs[4]) is the opt-in; f32 runs with AMP disabled."""
    if dtype in ("f16", "bf16"):
        os.environ["UTV2_PRECISION"] = {"f16": "fp16", "bf16": "bf16"}[dtype]
    else:
        os.environ.pop("UTV2_PRECISION", None)


def _cpu_threads():
    # stock torch CPU convolutions stop scaling (and can slow down) far below the thread count of a 128-core host
    return max(1, min(os.cpu_count() or 1, 32))


def _synthetic_cpu_batch(label, unlabel):
    """SURVEY 8(d) synthetic inputs on the host: `label` labeled (weak + strong view) and `unlabel` unlabeled 1333x800 uint8 images"""
    import numpy as np
    from ubteacher.data.synthetic import make_gt, make_image, strong_view
    rng = np.random.default_rng(0)
    lq, lk, uq, uk = [], [], [], []
    for _ in range(label):
        wk = make_image(rng, 800, 1333).cpu()
        gt = make_gt(rng, 800, 1333)
        g = dict(boxes=gt.gt_boxes.tensor.cpu(), classes=gt.gt_classes.cpu())
        lq.append({"image": strong_view(rng, wk).cpu(), "gt": g}); lk.append({"image": wk, "gt": g})
    for _ in range(unlabel):
        wk = make_image(rng, 800, 1333).cpu()
        uq.append({"image": strong_view(rng, wk).cpu()}); uk.append({"image": wk})
    return lq, lk, uq, uk


def cpu_baseline_run(model_kind, label, unlabel, warmup, steps, dump=None):
    """The oracle (CPU port of the reference step: its orchestration + restated Detectron2 primitives on stock torch CPU kernels) timed
    on the host cores: `warmup` + `steps` iterations of the SAME post-burn-in step on `label` labeled (weak + strong views) +
    `unlabel` unlabeled 1333x800 images, with the phase split SURVEY 8(d) asks for.  Runs in its own process (see cpu_baseline).
    model_kind "fcos": BASELINE configs[1]'s step; "rcnn": configs[0] (Faster-RCNN UTv2, MODEL.DEVICE=cpu, 1 process).
    The teacher / student heads are rescaled (tests.utv2_testutil: data-driven, with the oracle's forward) so the teacher emits pseudo
    boxes and the pseudo-label branch does real work.  dump: path of a torch file that receives the initial student / teacher
    state, the batch, and the record_dict of the FIRST step - the parent process runs the product's f32 step on exactly that and
    reports the deviation (`parity_fullsize`): the oracle is the checker here, never the thing shipped."""
    from oracle import utv2_oracle as O
    from tests.utv2_testutil import rcnn_tune
    from ubteacher.modeling import build_model
    from ubteacher.presets import get_config
    cfg = get_config(model_kind, 1, ["MODEL.DEVICE", "cpu", "SEMISUPNET.BURN_UP_STEP", 0])
    O.FAST_ROI_ALIGN[0] = True   # same arithmetic, a backward without the per-ROI full-map zero fills (see oracle/utv2_oracle.py)
    torch.set_num_threads(_cpu_threads())
    torch.manual_seed(0)
    model = build_model(cfg)
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
    del model
    batch = _synthetic_cpu_batch(label, unlabel)
    cores = torch.get_num_threads()
    S = cfg.SEMISUPNET
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
    pstd = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    weak = [d["image"] for d in batch[3]]
    if model_kind == "fcos":
        # cls_logits rescaled to std 1.5 (as tests.utv2_testutil.tune_state_for_pseudo_labels), the bias placed so that ~40 logits per
        # weak image clear p = 0.6: both pseudo-label sets (criteria "cls" and "cls_n_loc", threshold 0.5) are non-empty
        q = "proposal_generator.fcos_head.cls_logits"
        g = torch.Generator().manual_seed(0)
        student = dict(sd)
        student[q + ".weight"] = torch.randn(sd[q + ".weight"].shape, generator=g) * 0.01
        student[q + ".bias"] = torch.zeros_like(sd[q + ".bias"])
        with torch.no_grad():
            lg = torch.cat([x.reshape(-1) for x in O.fcos_forward(student, weak, mean, pstd)[0]])
        scale = 1.5 / max(float(lg.std()), 1e-12)
        kth = float(torch.topk(lg, 40 * len(weak)).values[-1]) * scale
        student[q + ".weight"] = student[q + ".weight"] * scale
        student[q + ".bias"] = torch.full_like(sd[q + ".bias"], 0.405 - kth)
        teacher = dict(student)
        teacher["proposal_generator.fcos_head.bbox_pred_std.bias"] = torch.full((4,), -3.0)   # confident teacher boundaries
        keys = None
    else:
        student = rcnn_tune(sd, weak, mean, pstd)
        # as tune_rcnn_for_pseudo_labels does for the product: class scores wide enough (std 8, no background bias) that the teacher
        # emits detections above BBOX_THRESHOLD on noise images
        q = "roi_heads.box_predictor.cls_score"
        student[q + ".weight"] = student[q + ".weight"] * (8.0 / 2.5)
        student[q + ".bias"] = torch.zeros_like(student[q + ".bias"])
        teacher = dict(student)
        teacher["roi_heads.box_predictor.bbox_pred_std.bias"] = torch.full((4,), -3.0)
        g = torch.Generator().manual_seed(99)
        R = 3 * sum(-(-800 // s) * -(-1344 // s) for s in (4, 8, 16, 32, 64))

        roi_log = []    # (proposals, gts, keys) per image of the FIRST step, in call order: the parity run replays them

        def roi_k(nprop, ngt):
            k = torch.rand(nprop + ngt, generator=g)
            roi_log.append((int(nprop), int(ngt), k))
            return k
        keys = dict(rpn_sup=torch.rand(2 * label, R, generator=g), rpn_unsup=torch.rand(unlabel, R, generator=g),
                    roi_sup=[roi_k] * (2 * label), roi_unsup=[roi_k] * unlabel)
    init = (student, teacher)
    bufs = None
    phases, times, first, pseudo_n = {}, [], None, None
    for it in range(warmup + steps):
        ph = {}
        t0 = time.perf_counter()
        if model_kind == "fcos":
            rec, student, teacher, _, bufs, pseudo = O.fcos_semisup_step(
                O.FCOSCfg(), student, teacher, batch, keep_rate=S.EMA_KEEP_RATE, lam_u=S.UNSUP_LOSS_WEIGHT, lam_r=S.UNSUP_REG_LOSS_WEIGHT,
                bufs=bufs, mean=mean, pix_std=pstd, phase_times=ph)
            npseudo = {"cls": sum(len(p["boxes"]) for p in pseudo[0]), "reg": sum(len(p["boxes"]) for p in pseudo[1])}
        else:
            rec, student, teacher, _, pseudo = O.rcnn_semisup_step(
                student, teacher, batch, keys, keep_rate=S.EMA_KEEP_RATE, lam_u=S.UNSUP_LOSS_WEIGHT, lam_r=S.UNSUP_REG_LOSS_WEIGHT,
                thr=S.BBOX_THRESHOLD, lr=1e-12, mean=mean, pix_std=pstd)
            npseudo = sum(len(p["boxes"]) for p in pseudo)
        dt = time.perf_counter() - t0
        if it == 0:
            first, pseudo_n = rec, npseudo
            if dump:
                extra = {}
                if model_kind != "fcos":
                    extra = {"rpn_keys": (keys["rpn_sup"], keys["rpn_unsup"]), "roi_keys": list(roi_log), "label": label, "unlabel": unlabel,
                             # the oracle teacher's thresholded detections themselves: the parent replays them into the product's student
                             # (the decoupled half of parity_fullsize) and counts the anchors whose label flips between the two box sets
                             "pseudo_boxes": [{k: v.detach().clone() for k, v in p_.items() if torch.is_tensor(v)} for p_ in pseudo]}
                torch.save(dict({"student": init[0], "teacher": init[1], "batch": batch, "record": rec, "pseudo": npseudo,
                                 "keep_rate": S.EMA_KEEP_RATE, "model": model_kind}, **extra), dump)
        if it >= warmup:
            times.append(dt)
            for k, v in ph.items():
                phases[k] = phases.get(k, 0.0) + v
    mean_t = sum(times) / len(times)
    what = ("FCOS R50-FPN UTv2 (configs[1]'s step)" if model_kind == "fcos" else
            "Faster-RCNN R50-FPN UTv2 (configs[0]: MODEL.DEVICE=cpu, 1 process)")
    out = {"value": (label + unlabel) / mean_t, "unit": "images/sec", "cores": cores, "kind": "port", "model": model_kind,
           "sample": "%s: %d warm-up + %d timed steps of %d labeled (weak+strong views) + %d unlabeled 1333x800 images, fp32, reference "
                     "orchestration + restated Detectron2 primitives on stock torch CPU kernels; %.1f s per step"
                     % (what, warmup, steps, label, unlabel, mean_t),
           "step_seconds": times, "first_step_record": first, "pseudo_boxes_first_step": pseudo_n}
    if phases:
        out["phase_seconds_per_step"] = {k: v / len(times) for k, v in phases.items()}
    return out


def cpu_baseline(model_kind="fcos", label=2, unlabel=2, warmup=2, steps=5, timeout=900, dump=None):
    """cpu_baseline_run in a child process, BEFORE the GPU phase starts (it cannot disturb the timed region, and a host-side problem
    cannot lose the GPU measurement)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--model", model_kind, "--label", str(label),
           "--unlabel", str(unlabel), "--cpu-warmup", str(warmup), "--cpu-steps", str(steps)]
    if dump:
        cmd += ["--cpu-dump", dump]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-400:]}
        return json.loads(line[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def parity_fullsize(dump, device_index):
    """The product's exact-f32 step on the SAME 1333x800 batch and the SAME initial student / teacher the cpu_baseline child ran its
    first oracle step on: per-loss relative deviation of the two record_dicts (north-star tolerance 1e-3).  Every 256-tile / ping-pong /
    multi-round top-k / 1000-candidate NMS path that the 96x128 parity tests cannot reach is exercised here at benchmark resolution."""
    from ubteacher.d2.structures import Boxes, Instances
    from ubteacher.engine import UBRCNNTeacherTrainer, UBTeacherTrainer
    from ubteacher.presets import get_config
    from ubteacher import ops
    os.environ.pop("UTV2_PRECISION", None)    # exact-f32 mode (SOLVER.AMP.ENABLED False), whatever 16-bit type the headline ran in
    d = torch.load(dump, weights_only=False)
    kind = d.get("model", "fcos")
    lq, lk, uq, uk = d["batch"]
    dev = "cuda:%d" % device_index

    def conv(part):
        out = []
        for x in part:
            e = {"image": x["image"].to(dev), "height": int(x["image"].shape[1]), "width": int(x["image"].shape[2])}
            if "gt" in x:
                inst = Instances((e["height"], e["width"]))
                inst.gt_boxes = Boxes(x["gt"]["boxes"].clone().to(dev))
                inst.gt_classes = x["gt"]["classes"].clone().to(dev)
                e["instances"] = inst
            out.append(e)
        return out
    batch = tuple(conv(p) for p in (lq, lk, uq, uk))

    class Fixed:
        def __iter__(self):
            return self

        def __next__(self):
            return tuple([dict(x) for x in part] for part in batch)
    cfg = get_config(kind, 1, ["SOLVER.IMG_PER_BATCH_LABEL", len(lq), "SOLVER.IMG_PER_BATCH_UNLABEL", len(uq), "SEMISUPNET.BURN_UP_STEP", 0,
                               "SOLVER.AMP.ENABLED", False, "MODEL.DEVICE", dev])

    def fresh_trainer():
        torch.manual_seed(0)
        t = (UBTeacherTrainer if kind == "fcos" else UBRCNNTeacherTrainer)(cfg, data_loader=Fixed())
        t.model.load_state_dict(d["student"]); t.model_teacher.load_state_dict(d["teacher"])
        t.model.store.touch(); t.model_teacher.store.touch(); ops.bump_version()
        t.iter = 1
        t.log_period = 10 ** 9
        return t
    tr = fresh_trainer()
    ref = d["record"]
    decoupled = None
    if kind == "fcos":
        tr.run_step_full_semisup()
        rec = dict(tr.flush_metrics())
        torch.cuda.synchronize()
        pc, pr = tr._last_pseudo
        extra = {"pseudo_boxes": {"oracle": d["pseudo"], "product": {"cls": int(pc["valid"].sum()), "reg": int(pr["valid"].sum())}},
                 "teacher_better_student_pseudo": {"oracle": ref.get("teacher_better_student_pseudo"),
                                                   "product": rec.get("teacher_better_student_pseudo")}}
        tol = {}
    else:
        # The oracle's draws replayed: anchor keys per pass as they were; ROI keys per image in its compact (proposals ++ gts) convention,
        # laid out in the product's (proposal slots, then gt slots) - the proposal counts must agree for that to line up, which is part
        # of what is being checked.  lr does not enter the record of the first step.
        post = cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN
        calls = {"rpn": 0, "roi": 0}
        nsup = 2 * d["label"]

        def rpn_src(n, m, device):
            k = d["rpn_keys"][calls["rpn"]]
            calls["rpn"] += 1
            assert tuple(k.shape) == (n, m), ("anchor keys", tuple(k.shape), (n, m))
            return k.to(device)

        def roi_src(n, m, device):
            first = 0 if calls["roi"] == 0 else nsup
            calls["roi"] += 1
            out_k = torch.full((n, m), 0.5)
            for i in range(n):
                nprop, ngt, k = d["roi_keys"][first + i]
                out_k[i, :nprop] = k[:nprop]
                out_k[i, post:post + ngt] = k[nprop:]
            return out_k.to(device)
        tr.model.proposal_generator.sample_keys = rpn_src
        tr.model.roi_heads.sample_keys = roi_src
        tr.run_step_full_semisup()
        rec = dict(tr.flush_metrics())
        torch.cuda.synchronize()
        gl = tr._last_pseudo
        extra = {"pseudo_boxes": {"oracle": d["pseudo"], "product": int(gl["valid"].sum())}, "key_draws_replayed": dict(calls)}
        # COUPLED run (each side thresholds its OWN teacher's detections): every weighted term holds 1e-3.  loss_rpn_loc_pseudo - weight 0 in
        # the objective (trainer.py:888-890), a sum over <= 64 sampled positives per image - hangs on the Matcher's exact-equality
        # low-quality rule against pseudo boxes that differ by ~1e-5 px between the two teachers: anchors of equal area INSIDE a pseudo
        # box tie at the 1-ulp level of the fp32 IoU, and which of them tie changes with the last bits of the box
        # (tests/test_rcnn_conditioning.py: 18 vs 8 positives for one real box pair on the oracle alone).  It is therefore checked in
        # two other ways: (1) the anchors whose label differs between the two box sets are COUNTED (oracle Matcher on both), and
        # (2) the DECOUPLED run below replays the oracle's pseudo boxes into the product's student and holds 1e-3 on every term.
        tol = {"loss_rpn_loc_pseudo": 1e-1}
        pb = d.get("pseudo_boxes")
        if pb is not None:
            from oracle import utv2_oracle as O        # the checker (never the thing measured): anchor labels of both pseudo-box sets
            ch_, cw_ = tr.model.padded_canvas(batch[2])
            hw = [(-(-ch_ // s_), -(-cw_ // s_)) for s_ in (4, 8, 16, 32, 64)]     # p2..p6 of the padded canvas
            anchors = torch.cat(O.make_anchors(hw, (4, 8, 16, 32, 64)))
            flipped, positives, box_dev = [], [], 0.0
            for i, o in enumerate(pb):
                m = gl["valid"][i].bool()
                mine = gl["boxes"][i][m].float().cpu()
                assert len(mine) == len(o["boxes"]), ("pseudo-box count", i, len(mine), len(o["boxes"]))
                if len(mine) == 0:
                    flipped.append(0); positives.append(0)
                    continue
                box_dev = max(box_dev, float((mine - o["boxes"]).abs().max()))
                la = O.matcher(O.pairwise_iou(o["boxes"], anchors), [0.3, 0.7], [0, -1, 1], True)[1]
                lb = O.matcher(O.pairwise_iou(mine, anchors), [0.3, 0.7], [0, -1, 1], True)[1]
                flipped.append(int((la != lb).sum())); positives.append(int((la == 1).sum()))
            if sum(flipped):
                # loss_rpn_cls_pseudo sums score-weighted BCE terms over the 256 SAMPLED anchors per image: every anchor whose label differs
                # between the two pseudo-box sets can swap one term of that sample (1 + 1 fixture: 3 of 268 569 labels differ, the term
                # moves by 1.5e-3; 2 + 2: 7 labels, 1.9e-4).  With identical labels (the decoupled run) it holds 1e-3 like every other term.
                tol["loss_rpn_cls_pseudo"] = 5e-3
            extra["coupled_anchor_labels"] = {"anchors_per_image": int(anchors.shape[0]), "positives_under_oracle_boxes": positives,
                                              "labels_that_differ_under_product_boxes": flipped, "max_abs_box_dev_px": box_dev,
                                              "bound": "<= 64 labels per image (the sample of positives is 64 per image)"}
            # DECOUPLED: the same step with the oracle's thresholded detections substituted for the product teacher's (same count, same
            # order; coordinates / scores / boundary std replaced in place) - selection noise removed, arithmetic of EVERY term at 1e-3
            del tr
            torch.cuda.empty_cache()
            tr = fresh_trainer()
            calls["rpn"] = calls["roi"] = 0
            tr.model.proposal_generator.sample_keys = rpn_src
            tr.model.roi_heads.sample_keys = roi_src
            orig_ppl = tr.process_pseudo_label

            def replay(proposals, thr, ptype, method=""):
                out, frac = orig_ppl(proposals, thr, ptype, method)
                for i, o in enumerate(pb):
                    idx = out["valid"][i].bool().nonzero().squeeze(1)
                    assert idx.numel() == len(o["boxes"])
                    for key in ("boxes", "scores", "pred_boxes_std"):
                        if key in o and key in out.f:
                            out[key][i][idx] = o[key].to(out[key].device, out[key].dtype)
                return out, frac
            tr.process_pseudo_label = replay
            tr.run_step_full_semisup()
            rec_d = dict(tr.flush_metrics())
            torch.cuda.synchronize()
            dd = {k: abs(rec_d[k] - v) / max(abs(v), 1e-12) for k, v in ref.items() if k.startswith("loss") and k in rec_d}
            decoupled = {"rel_dev": dd, "max_rel_dev": max(dd.values()), "within_tolerance": all(v <= 1e-3 for v in dd.values()),
                         "note": "the oracle's pseudo boxes replayed into the product's student: every term, loss_rpn_loc_pseudo included, at 1e-3"}
    dev_ = {k: abs(rec[k] - v) / max(abs(v), 1e-12) for k, v in ref.items() if k.startswith("loss") and k in rec}
    out = {"mode": "f32", "model": kind, "images": "%d labeled (weak+strong) + %d unlabeled 1333x800" % (len(lq), len(uq)), "tolerance": 1e-3,
           "rel_dev": dev_, "max_rel_dev": max(dev_.values()) if dev_ else None,
           "within_tolerance": bool(dev_) and all(v <= tol.get(k, 1e-3) for k, v in dev_.items()),
           "oracle_losses": {k: ref[k] for k in dev_}, "product_losses": {k: rec[k] for k in dev_}}
    if tol:
        out["looser_terms"] = tol
    out.update(extra)
    if decoupled is not None:
        out["decoupled"] = decoupled
        cal = extra.get("coupled_anchor_labels")
        out["within_tolerance"] = bool(out["within_tolerance"] and decoupled["within_tolerance"]
                                       and all(f <= 64 for f in cal["labels_that_differ_under_product_boxes"]))
    del tr
    torch.cuda.empty_cache()
    return out


def step_subrecord(kind, args, device_index, timer=None, steps=10, warmup=5, dtype="bf16", label=None, unlabel=None, ragged=None):
    """images/sec of one trainer's run_step_full_semisup on `label` + `unlabel` images per GPU, timed by the headline's rule (synchronize on
    both sides at world 1).  kind "rcnn": UBRCNNTeacherTrainer (BASELINE configs[2] / [4]); "fcos": UBTeacherTrainer (configs[1] / [3]).
    timer (a ConvTimer of the matching precision): HIP-event timing of the dominant multi-level 3x3 conv inside the timed steps -> `roofline`.
    ragged = (min_lo, min_hi, max_size): every image at its own ResizeShortestEdge size, 8 different batches cycled (the reference recipes'
    INPUT.MIN_SIZE_TRAIN (400, 1200) "range"): labeled and unlabeled canvases differ, the student runs its two passes unfused."""
    from ubteacher.engine import UBRCNNTeacherTrainer, UBTeacherTrainer
    from ubteacher.presets import get_config
    label = args.label if label is None else label
    unlabel = args.unlabel if unlabel is None else unlabel
    rcnn = kind == "rcnn"
    cfg = get_config(kind, 1, ["SOLVER.IMG_PER_BATCH_LABEL", label, "SOLVER.IMG_PER_BATCH_UNLABEL", unlabel,
                               "SEMISUPNET.BURN_UP_STEP", 0, "SOLVER.AMP.ENABLED", dtype != "f32", "MODEL.DEVICE", "cuda:%d" % device_index])
    torch.manual_seed(0)
    import gc
    gc.collect()                                 # these steps are close to host-bound: start from a collected heap (several trainers came and went)
    set_amp_type(dtype)                          # BASELINE configs[4] names the bf16 MFMA conv path; configs[2] (no AMP in its YAML) is f32
    loader = None
    if ragged is not None:
        from ubteacher.data.synthetic import SyntheticTwoCropLoader
        loader = SyntheticTwoCropLoader(cfg, num_batches=8, ragged=ragged)
    tr = (UBRCNNTeacherTrainer if rcnn else UBTeacherTrainer)(cfg, data_loader=loader)
    tr.iter = 1
    tr.log_period = 10 ** 9
    tr.optimizer.param_groups[0]["lr"] = 1e-12   # see make_trainer: keeps the synthetic problem stationary
    if ragged is not None and not rcnn:
        # the bias is placed on the first batch's logits; the other seven batches have other canvases (fewer / more locations): four times
        # the margin of the static batch, so that every batch of the cycle has classification pseudo boxes
        tune_for_pseudo_labels(tr, tr._data_loader.batches[0], per_image=160)
    else:
        (tune_rcnn_for_pseudo_labels if rcnn else tune_for_pseudo_labels)(tr, tr._data_loader.batches[0])

    def counts():
        lp = getattr(tr, "_last_pseudo", None)
        if lp is None:
            return None
        if isinstance(lp, tuple):
            return {"cls": int(lp[0]["valid"].sum()), "reg": int(lp[1]["valid"].sum())}
        return int(lp["valid"].sum())
    pseudo_first = None
    pseudo_cycle = []
    for i in range(warmup):
        tr.run_step_full_semisup(); tr.iter += 1
        if i == 0:
            pseudo_first = counts()
        if ragged is not None and i < len(tr._data_loader.batches):
            pseudo_cycle.append(counts())       # one entry per batch of the cycle (a device read each: warm-up only)
    torch.cuda.synchronize()
    if timer is not None:
        timer.pairs, timer.pairs_gnb = [], []
        timer.enabled = True
    from ubteacher.engine.step_gc import StepGC
    with StepGC() as step_gc:                    # the collector policy of the product's own train_loop (engine/step_gc.py)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.run_step_full_semisup(); tr.iter += 1
            step_gc.tick()
        t_enq = time.perf_counter() - t0             # the host has enqueued every step
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    conv = None
    if timer is not None:
        timer.enabled = False
        conv = timer.summary()
    metrics = tr.flush_metrics()
    what = {"f32": "fp32 (exact-f32 MFMA; configs[2]'s YAML has no AMP)", "bf16": "bf16 MFMA conv path (configs[4])",
            "f16": "fp16 AMP, the reference's autocast type"}[dtype]
    out = {"value": (label + unlabel) * steps / dt, "unit": "images/sec", "ms_per_step": 1e3 * dt / steps, "steps": steps,
           "warmup": warmup, "dtype": dtype,
           "workload": "%s R50-FPN UTv2 sup1 (%s; %s): %d labeled + %d unlabeled %s images per GPU, post-burn-in semi-supervised step"
                       % ("Faster-RCNN" if rcnn else "FCOS", "the trainer of configs[2] / [4]" if rcnn else "the trainer of configs[1] / [3]", what, label, unlabel,
                          "1333x800" if ragged is None else "ResizeShortestEdge(%d-%d, max %d)-sized" % tuple(ragged)),
           "losses": {k: v for k, v in metrics.items() if k.startswith("loss")},
           "pseudo_boxes_first_step": pseudo_first, "pseudo_boxes_last_step": counts(),
           "enqueue_ms_per_step": 1e3 * t_enq / steps}
    if ragged is not None:
        px = [sum(int(x["image"].shape[1]) * int(x["image"].shape[2]) for part in (b[0], b[2]) for x in part) for b in tr._data_loader.batches]
        out["megapixels_per_sec"] = (sum(px) / len(px)) * steps / dt / 1e6
        out["canvases"] = [[list(tr.model.padded_canvas(b[0] + b[1])), list(tr.model.padded_canvas(b[2]))] for b in tr._data_loader.batches]
        out["pseudo_boxes_per_batch_of_the_cycle"] = pseudo_cycle
        out["note"] = ("8 different batches cycled, every image at its own size: labeled and unlabeled lists pad to different canvases, so the student runs the "
                       "reference's two passes (engine/trainer.py:396-411 / :838-866) instead of the fused one; geometry tables and workspaces are per shape")
    if conv:
        peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        out["roofline"] = {"bound": "mfma", "kernel": timer.kernel + (" (RPN head 3x3 conv over p2-p6, fwd+dgrad launches)" if rcnn else " (FCOS tower 3x3 convs, fwd+dgrad launches)"),
                           "achieved": conv["tflops"], "peak": peak, "unit": "TFLOP/s",
                           "frac": conv["tflops"] / peak, "traffic": pmc_traffic(timer.kernel, kind) if dtype != "f32" else None,
                           "algorithmic_bytes": conv["alg_bytes"], "launches": conv["launches"], "avg_us": conv["avg_us"]}
    del tr
    torch.cuda.empty_cache()
    return out


def rcnn_first_step_deviation(args, device_index):
    """Faster-RCNN: deviation of ONE 16-bit step (bf16 = BASELINE configs[4]'s arithmetic, fp16 = the package's AMP default) from the exact-f32
    step on the same 1333x800 batch, from the same initial student / teacher and with the same device RNG stream for the sampling keys.
    Unlike FCOS the step SELECTS (RPN top-k / NMS, ROI sampling, pseudo-label threshold) on rounded scores, so a term can move by more than
    its arithmetic error: `same_pseudo_count` / `pseudo_boxes` say whether the two runs even saw the same pseudo-label set."""
    from ubteacher.engine import UBRCNNTeacherTrainer
    from ubteacher.presets import get_config
    from ubteacher import ops

    def make(dtype):
        cfg = get_config("rcnn", 1, ["SOLVER.IMG_PER_BATCH_LABEL", args.label, "SOLVER.IMG_PER_BATCH_UNLABEL", args.unlabel, "SEMISUPNET.BURN_UP_STEP", 0,
                                     "SOLVER.AMP.ENABLED", dtype != "f32", "MODEL.DEVICE", "cuda:%d" % device_index])
        set_amp_type(dtype)
        torch.manual_seed(0)
        t = UBRCNNTeacherTrainer(cfg)
        t.iter = 1
        t.log_period = 10 ** 9
        t.optimizer.param_groups[0]["lr"] = 1e-12
        return t
    tr = make("f32")
    tune_rcnn_for_pseudo_labels(tr, tr._data_loader.batches[0])
    s0, t0 = tr.model.flat_state().clone(), tr.model_teacher.flat_state().clone()
    out = {}
    ref = None
    for dtype in ("f32", "bf16", "f16"):
        if dtype != "f32":
            del tr
            torch.cuda.empty_cache()
            tr = make(dtype)
            tr.model.flat_state().copy_(s0); tr.model_teacher.flat_state().copy_(t0)
            tr.model.store.touch(); tr.model_teacher.store.touch(); ops.bump_version()
        torch.manual_seed(1)                      # the same sampling-key draws in every mode
        tr.run_step_full_semisup()
        rec = dict(tr.flush_metrics())
        npseudo = int(tr._last_pseudo["valid"].sum())
        losses = {k: v for k, v in rec.items() if k.startswith("loss")}
        if dtype == "f32":
            ref, ref_n = losses, npseudo
            out["f32_first_step_losses"] = losses
            out["pseudo_boxes"] = {"f32": npseudo}
        else:
            out["%s_vs_f32_first_step_rel_dev" % dtype] = {k: abs(losses[k] - ref[k]) / max(abs(ref[k]), 1e-12) for k in ref}
            out["pseudo_boxes"][dtype] = npseudo
    del tr
    torch.cuda.empty_cache()
    out["note"] = ("one step per mode from the same state; selection steps (top-k / NMS / sampling / thresholding) act on rounded scores, so these "
                   "are not pure arithmetic errors (tests/test_rcnn_step_gpu.py::test_rcnn_step_bf16_vs_rounding_oracle decouples them: 1e-2 / 3e-3)")
    return out


def subrecord_child(kind, dtype, label, unlabel, steps, warmup, ragged=False, timeout=600):
    """step_subrecord in a FRESH process.  The sub-records with the thinnest host margin (2 + 2 images per GPU, ragged canvases: ~10 ms of host
    work under a 13 ms step) are measured the way a user runs the product - one trainer in one process: inside the bench process, which
    has built and dropped half a dozen trainers by then, the host is slower and the same Faster-RCNN 2 + 2 step becomes host-bound (246
    against 282-290 img/s, profiles/r06_bench_f16.json against r06_fold_ab.txt / r06_bench_f16_final.json); the 4 + 4 sub-records do not move."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--subrecord-only", "--model", kind, "--dtype", dtype, "--label", str(label), "--unlabel", str(unlabel),
           "--steps", str(steps), "--warmup", str(warmup)] + (["--ragged"] if ragged else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-400:]}
        out = json.loads(line[-1])
        out["process"] = "fresh child process (one trainer, as a user runs it)"
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # BASELINE.md protocol: 10 warm-up + >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--label", type=int, default=4, help="labeled images per GPU")
    ap.add_argument("--unlabel", type=int, default=4, help="unlabeled images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) run the CPU port on --label/--unlabel images and print its record")
    ap.add_argument("--cpu-warmup", type=int, default=2)
    ap.add_argument("--cpu-steps", type=int, default=5)
    ap.add_argument("--cpu-images", type=int, default=2, help="labeled and unlabeled images of the CPU baseline sample (SURVEY 8d protocol: 2)")
    ap.add_argument("--cpu-dump", default=None, help="(internal) file that receives the first oracle step's inputs / record_dict")
    ap.add_argument("--subrecord-only", action="store_true", help="(internal) run step_subrecord for --model / --dtype / --label / --unlabel / --steps / "
                    "--warmup (/ --ragged) on this process's GPU and print its record")
    ap.add_argument("--ragged", action="store_true", help="(internal, with --subrecord-only) every image at its own ResizeShortestEdge size")
    ap.add_argument("--dry-nccl", action="store_true",
                    help="join the world, run the RCCL self-check (one tiny all-reduce per rank, a device-bound barrier, an object "
                         "all-gather), print {n_gpus, rccl_ranks, ...} on rank 0 and stop: first contact with the backend in seconds")
    ap.add_argument("--no-rcnn", action="store_true", help="skip the Faster-RCNN sub-records (GPU step of configs[2]/[4], CPU step of configs[0])")
    ap.add_argument("--timed-only", action="store_true", help="only the warmup and the timed steps (profiling runs): no exclusive pass, no host probe")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph-replay sub-record (host.step_as_hipgraph)")
    ap.add_argument("--no-small", action="store_true", help="skip the 2+2-per-GPU (`small_batch`) and ragged-canvas (`ragged_canvases`) sub-records")
    ap.add_argument("--no-f32", action="store_true", help="skip the f32 sub-record / the bf16-vs-f32 one-step loss deviation")
    ap.add_argument("--model", choices=["fcos", "rcnn"], default="fcos",
                    help="fcos: BASELINE configs[1] (the headline workload); rcnn: the Faster-RCNN UTv2 trainer of configs[2] / [4] on the same "
                         "per-GPU batch (its shipped configs run fp32: --dtype f32; configs[4] is the bf16 MFMA path)")
    ap.add_argument("--dtype", choices=["f16", "bf16", "f32"], default=None,
                    help="conv arithmetic of the config's SOLVER.AMP.ENABLED path.  f16 (default for --model fcos): IEEE fp16 MFMA operands / "
                         "activations / gradients with GradScaler-style dynamic loss scaling - the reference's own autocast type "
                         "(engine/trainer.py:195,207), losses within 1e-3 of the f32 step; bf16 (default for --model rcnn: BASELINE configs[4] names "
                         "the bf16 MFMA path): no loss scaling; f32 = exact-f32 MFMA")
    a = ap.parse_args(argv)
    if a.dtype is None:
        a.dtype = "bf16" if a.model == "rcnn" else "f16"
    return a


def _launcher_name(world):
    if os.environ.get("TORCHELASTIC_RUN_ID"):
        return "torch.distributed.run"
    return "ubteacher.engine.launch" if world > 1 else "single process"


def worker(args):
    """one rank (ubteacher.engine.launch has bound the device and joined the RCCL world)"""
    from ubteacher.engine.launch import dist_info
    info = dist_info()
    rank, local_rank, world = info["rank"], info["local_rank"], info["world_size"]
    assert world == args.gpus, "bench: %d ranks were requested, this process is in a world of %d" % (args.gpus, world)
    if world > 1:
        assert dist.is_initialized() and dist.get_world_size() == args.gpus
    device_index = info["device"]
    from ubteacher.utils import comm
    rccl = None
    if dist.is_initialized() and os.environ.get("UTV2_BENCH_LAUNCH_ONLY") != "1":
        # first contact with the backend, before any model is built: a wrong device binding / IPC mode / missing rank shows here
        rccl = comm.rccl_selfcheck()
        comm.barrier()
        rccl["devices"] = comm.all_gather_object(device_index)
        assert rccl["ok"] and rccl["ranks"] == world, "RCCL self-check failed: %r" % (rccl,)
    if args.dry_nccl:
        if rank == 0:
            print(json.dumps({"n_gpus": world, "rccl_ranks": None if rccl is None else rccl["ranks"], "rccl": rccl,
                              "ranks": {"world_size": world, "backend": info["backend"], "launcher": _launcher_name(world)}}), flush=True)
        return
    if os.environ.get("UTV2_BENCH_LAUNCH_ONLY") == "1":   # tests of the launch contract on boxes without GPUs: report the world, stop
        ids = [None] * world
        if world > 1:
            dist.all_gather_object(ids, rank)
        else:
            ids = [0]
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks": {"world_size": world, "backend": info["backend"], "rank_ids": ids,
                                                         "launcher": _launcher_name(world)}}), flush=True)
        return

    from ubteacher.engine import UBRCNNTeacherTrainer, UBTeacherTrainer
    from ubteacher.presets import get_config
    from ubteacher import hip, ops
    hip.load()
    rcnn = args.model == "rcnn"
    cpu_rec = cpu_rcnn = dump = dump_rcnn = None

    def make_trainer(dtype):
        cfg = get_config(args.model, 1, ["SOLVER.IMG_PER_BATCH_LABEL", args.label * world, "SOLVER.IMG_PER_BATCH_UNLABEL",
                                         args.unlabel * world, "SEMISUPNET.BURN_UP_STEP", 0, "SOLVER.AMP.ENABLED", dtype != "f32",
                                         "MODEL.DEVICE", "cuda:%d" % device_index])
        torch.manual_seed(0)
        set_amp_type(dtype)
        t = (UBRCNNTeacherTrainer if rcnn else UBTeacherTrainer)(cfg)
        t.iter = 1
        t.log_period = 10 ** 9
        # random-init R50 features are not normalised (|x| ~ 1e4 in the box head): at any practical learning rate ONE SGD step moves
        # the student far enough that the EMA teacher stops emitting pseudo boxes and the pseudo-label branch degenerates (FCOS at the
        # config's lr: 89 classification pseudo boxes in the first step, 1 after 25).  The step's work does not depend on the learning
        # rate: a vanishing one keeps the synthetic problem stationary over the timed window - `pseudo_boxes_first_step` /
        # `pseudo_boxes_last_step` report both ends.
        t.optimizer.param_groups[0]["lr"] = 1e-12
        return t

    timer = ConvTimer(args.dtype)
    timer.install()
    wtimer = WgradTimer()
    wtimer.install()
    calls = CallCounter()
    calls.install()
    tr = make_trainer(args.dtype)
    batch = tr._data_loader.batches[0]
    (tune_rcnn_for_pseudo_labels if rcnn else tune_for_pseudo_labels)(tr, batch)
    tr.sync_replicas()   # identical students / teachers on every rank (DDP broadcasts rank 0's parameters at construction)
    parity = rank == 0 and world == 1 and args.dtype != "f32" and not args.no_f32 and not rcnn
    if parity:
        s0, t0 = tr.model.flat_state().clone(), tr.model_teacher.flat_state().clone()
    def pseudo_counts(t):
        lp_ = getattr(t, "_last_pseudo", None)
        if lp_ is None:
            return None
        if isinstance(lp_, tuple):
            return {"cls": int(lp_[0]["valid"].sum()), "reg": int(lp_[1]["valid"].sum())}
        return int(lp_["valid"].sum())

    first = pseudo_first = None
    for i in range(args.warmup):
        tr.run_step_full_semisup(); tr.iter += 1
        if i == 0:
            pseudo_first = pseudo_counts(tr)
            if parity:
                first = dict(tr.flush_metrics())

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            comm.barrier()     # device_ids = this rank's device under nccl
            torch.cuda.synchronize()

    sync()
    gsync = getattr(tr, "_grad_sync", None)
    if gsync is not None:
        gsync.timing = True          # per-step exposure of the gradient all-reduce (what backward did not hide)
    timer.enabled = wtimer.enabled = True
    n0 = calls.n
    power = BoardPower() if rank == 0 else None
    if power is not None:
        power.start()
    from ubteacher.engine.step_gc import StepGC
    with StepGC() as step_gc:               # the collector policy of the product's own train_loop (engine/step_gc.py)
        sync()
        t0_ = time.perf_counter()
        for _ in range(args.steps):
            tr.run_step_full_semisup(); tr.iter += 1
            step_gc.tick()
        t_host = time.perf_counter() - t0_      # the host has enqueued every step (it runs ahead of the GPU)
        sync()
        dt = time.perf_counter() - t0_
    board_power = power.stop() if power is not None else None
    timer.enabled = wtimer.enabled = False
    # shader clock the chip sustained under the last multi-level (tower / RPN head) launch of the 256-tile conv kernel inside the timed
    # region (utv2_conv_clock_probe: s_memtime against the 100 MHz real-time counter over workgroup 0's lifetime)
    clock_ghz = None
    if args.dtype != "f32":
        try:
            from ubteacher import hip as _hip
            g, us = _hip.conv_clock_probe()
            clock_ghz = {"ghz": g, "workgroup_lifetime_us": us} if g > 0 else None
        except Exception as e:  # noqa: BLE001
            clock_ghz = {"error": repr(e)}
    calls_per_step = (calls.n - n0) / max(args.steps, 1)
    devices = [device_index]
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        devices = comm.all_gather_object(device_index)
    replicas = None
    if world > 1:
        # data parallel keeps the replicas in lock step (the reference wraps the student in DDP, engine/trainer.py:59-63,631-635; the
        # teacher's EMA is local and deterministic): bit-exact fingerprints of every rank's student and teacher after the timed steps
        import hashlib

        def digest(t):
            return hashlib.sha1(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()
        fps = comm.all_gather_object((digest(tr.model.flat_state()), digest(tr.model_teacher.flat_state())))
        replicas = {"students_bit_identical": len({f[0] for f in fps}) == 1, "teachers_bit_identical": len({f[1] for f in fps}) == 1,
                    "ranks_compared": len(fps)}
    allreduce = None
    if gsync is not None:
        gsync.timing = False
        expo = gsync.exposure_summary()
        if expo is not None and world > 1:
            allr = comm.all_gather_object(expo["exposed_ms_per_step_mean"])
            expo["exposed_ms_per_step_mean_by_rank"] = allr
        allreduce = expo
    metrics = tr.flush_metrics()
    amp_state = tr._amp_state.cpu().tolist() if getattr(tr, "_amp_state", None) is not None else None
    pseudo_count = pseudo_counts(tr)
    conv, wg = timer.summary(), wtimer.summary()
    conv_gnb = timer.summary(gnb=True)

    # the same launches with the step's side streams off (teacher pass / weight gradients back on the main stream): the dominant kernels
    # alone on the GPU.  Not part of the timed region - reported beside the in-step figures as `exclusive`.
    conv_x = wg_x = conv_gnb_x = None
    if rank == 0 and world == 1 and args.dtype != "f32" and not args.timed_only:
        saved = {k: os.environ.get(k) for k in ("UTV2_OVERLAP_TEACHER", "UTV2_WGRAD_STREAM")}
        os.environ["UTV2_OVERLAP_TEACHER"] = os.environ["UTV2_WGRAD_STREAM"] = "0"
        ot = getattr(tr, "overlap_teacher", None)
        if ot is not None:
            tr.overlap_teacher = False
        timer.pairs, timer.pairs_gnb, wtimer.pairs = [], [], []
        timer.enabled = wtimer.enabled = True
        for _ in range(min(args.steps, 5)):
            tr.run_step_full_semisup(); tr.iter += 1
        torch.cuda.synchronize()
        timer.enabled = wtimer.enabled = False
        conv_x, wg_x = timer.summary(), wtimer.summary()
        conv_gnb_x = timer.summary(gnb=True)
        if ot is not None:
            tr.overlap_teacher = ot
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    host_ms = None
    if rank == 0 and world == 1 and not args.timed_only:
        # host cost of one step: the same trainer code on 96 x 128 images, where the GPU work is negligible and the step time IS the
        # Python / launch overhead (on the 1333 x 800 batch the host runs ahead until the launch queue is full, so its enqueue time
        # only mirrors the GPU time)
        from ubteacher.data.synthetic import SyntheticTwoCropLoader
        cfg_s = get_config(args.model, 1, ["SOLVER.IMG_PER_BATCH_LABEL", args.label, "SOLVER.IMG_PER_BATCH_UNLABEL", args.unlabel,
                                       "SEMISUPNET.BURN_UP_STEP", 0, "SOLVER.AMP.ENABLED", args.dtype != "f32",
                                       "MODEL.DEVICE", "cuda:%d" % device_index])
        torch.manual_seed(0)
        ts = (UBRCNNTeacherTrainer if rcnn else UBTeacherTrainer)(cfg_s, data_loader=SyntheticTwoCropLoader(cfg_s, height=96, width=128))
        ts.iter = 1
        ts.log_period = 10 ** 9
        (tune_rcnn_for_pseudo_labels if rcnn else tune_for_pseudo_labels)(ts, ts._data_loader.batches[0])
        for _ in range(3):
            ts.run_step_full_semisup(); ts.iter += 1
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(10):
            ts.run_step_full_semisup(); ts.iter += 1
        torch.cuda.synchronize()
        host_ms = 1e3 * (time.perf_counter() - th) / 10
        del ts

    f32_rec = None
    if parity and first is not None:
        # the mode in which the 1e-3 loss parity against the oracle is demonstrated (tests/test_fcos_step_gpu.py), driver-timed here, and
        # the deviation of ONE bf16 step from the f32 step on the same 1333x800 batch from the same initial weights
        del tr
        torch.cuda.empty_cache()
        tr32 = make_trainer("f32")
        tr32.model.flat_state().copy_(s0); tr32.model_teacher.flat_state().copy_(t0)
        tr32.model.store.touch(); tr32.model_teacher.store.touch(); ops.bump_version()
        tr32.run_step_full_semisup(); tr32.iter += 1
        first32 = dict(tr32.flush_metrics())
        tr32.run_step_full_semisup(); tr32.iter += 1
        torch.cuda.synchronize()
        k32 = 5
        t1 = time.perf_counter()
        for _ in range(k32):
            tr32.run_step_full_semisup(); tr32.iter += 1
        torch.cuda.synchronize()
        d32 = time.perf_counter() - t1
        keys = [k for k in first32 if k.startswith("loss")]
        main16, other16 = args.dtype, ("bf16" if args.dtype == "f16" else "f16")

        def rel(a_):
            return {k: abs(a_[k] - first32[k]) / max(abs(first32[k]), 1e-12) for k in keys}
        f32_rec = {"value": (args.label + args.unlabel) * k32 / d32, "unit": "images/sec", "ms_per_step": 1e3 * d32 / k32, "steps": k32,
                   "warmup": 2, "dtype": "f32",
                   "first_step_losses": {k: first32[k] for k in keys},
                   "%s_first_step_losses" % main16: {k: first[k] for k in keys},
                   "%s_vs_f32_first_step_rel_dev" % main16: rel(first),
                   "pseudo_boxes_per_step": {"f32": first32.get("teacher_better_student_pseudo"),
                                             main16: first.get("teacher_better_student_pseudo"),
                                             "note": "teacher_better_student count of the regression pseudo set; each mode thresholds its OWN teacher's detections, so the pseudo "
                                                     "classification loss also moves with which borderline detections pass the score threshold, not only with rounding"}}
        f32_rec["headline_within_1e-3_of_f32"] = max(f32_rec["%s_vs_f32_first_step_rel_dev" % main16].values()) <= 1e-3
        del tr32
        torch.cuda.empty_cache()
        # the same step in the OTHER 16-bit type (bf16: no loss scaling, 8 mantissa bits; f16: the reference's own autocast type, 11 bits,
        # dynamic loss scale): deviation of its first step from the f32 step, and its speed
        try:
            tro = make_trainer(other16)
            tro.model.flat_state().copy_(s0); tro.model_teacher.flat_state().copy_(t0)
            tro.model.store.touch(); tro.model_teacher.store.touch(); ops.bump_version()
            tro.run_step_full_semisup(); tro.iter += 1
            firsto = dict(tro.flush_metrics())
            tro.run_step_full_semisup(); tro.iter += 1
            torch.cuda.synchronize()
            ko = 5
            t1 = time.perf_counter()
            for _ in range(ko):
                tro.run_step_full_semisup(); tro.iter += 1
            torch.cuda.synchronize()
            do = time.perf_counter() - t1
            lasto = dict(tro.flush_metrics())
            f32_rec["%s_first_step_losses" % other16] = {k: firsto[k] for k in keys}
            f32_rec["%s_vs_f32_first_step_rel_dev" % other16] = rel(firsto)
            f32_rec["pseudo_boxes_per_step"][other16] = firsto.get("teacher_better_student_pseudo")
            f32_rec[other16] = {"value": (args.label + args.unlabel) * ko / do, "unit": "images/sec", "ms_per_step": 1e3 * do / ko, "steps": ko,
                                "warmup": 2, "dtype": other16, "losses_finite": all(v == v and abs(v) != float("inf") for v in lasto.values())}
            if tro._amp_state is not None:
                f32_rec[other16]["loss_scale_state"] = dict(zip(("scale", "found_inf", "clean_steps"), tro._amp_state.cpu().tolist()))
            del tro
        except Exception as e:  # noqa: BLE001
            f32_rec[other16] = {"error": repr(e)}
        finally:
            os.environ.pop("UTV2_PRECISION", None)
            ops.set_precision("bf16")

    rcnn_rec = None
    if rank == 0 and world == 1 and not rcnn and not args.no_rcnn and not args.timed_only and args.dtype != "f32":
        # the Faster-RCNN UTv2 trainer (BASELINE configs[2] / [4]: bf16 MFMA conv path) on the same per-GPU batch, as a sub-record
        args_r = argparse.Namespace(**vars(args)); args_r.model = "rcnn"
        timer_bf = ConvTimer("bf16")               # the sub-record's own 16-bit type (the headline's timer carries the headline's kernel name)
        timer_bf.install()
        try:
            torch.cuda.empty_cache()
            rcnn_rec = step_subrecord("rcnn", args_r, device_index, timer_bf, steps=max(10, min(args.steps, 30)), warmup=5)
        except Exception as e:  # noqa: BLE001
            rcnn_rec = {"error": repr(e)}
        # the same trainer at the precision of its own shipped YAML (configs[2]: no SOLVER.AMP -> fp32; exact-f32 MFMA here), driver-timed,
        # with the roofline of ITS dominant kernel (the RPN head's multi-level 3x3 conv on v_mfma_f32_32x32x2_f32: 157.3 TFLOP/s dense)
        try:
            torch.cuda.empty_cache()
            timer32 = ConvTimer("f32")
            timer32.install()
            rcnn_rec["f32"] = step_subrecord("rcnn", args_r, device_index, timer32, steps=5, warmup=2, dtype="f32")
        except Exception as e:  # noqa: BLE001
            rcnn_rec["f32"] = {"error": repr(e)}

        try:
            torch.cuda.empty_cache()
            rcnn_rec["first_step_vs_f32"] = rcnn_first_step_deviation(args_r, device_index)
        except Exception as e:  # noqa: BLE001
            rcnn_rec["first_step_vs_f32"] = {"error": repr(e)}

    small_rec = ragged_rec = None
    if rank == 0 and world == 1 and not args.timed_only and args.dtype != "f32" and not args.no_small and (args.label, args.unlabel) == (4, 4):
        # 2 + 2 images per GPU: the per-GPU workload of configs[2] / [4] (16 + 16 over 8 GPUs) and of the reference's FCOS recipe - the
        # size at which launch overheads and the host weigh twice as much per image as at 4 + 4
        small_rec = {}
        for kind_, dt_ in (("fcos", args.dtype), ("rcnn", "bf16")):
            if kind_ == "rcnn" and args.no_rcnn:
                continue
            torch.cuda.empty_cache()
            small_rec[kind_] = subrecord_child(kind_, dt_, 2, 2, 40, 8)
        # the reference recipes' input sizes (INPUT.MIN_SIZE_TRAIN (400, 1200) "range", MAX_SIZE_TRAIN 1333): ragged canvases, the two-pass student
        torch.cuda.empty_cache()
        ragged_rec = subrecord_child(args.model, args.dtype, args.label, args.unlabel, 24, 16, ragged=True)
        if not rcnn and not args.no_rcnn and isinstance(rcnn_rec, dict) and "error" not in rcnn_rec:
            rcnn_rec["ragged_canvases"] = subrecord_child("rcnn", "bf16", args.label, args.unlabel, 24, 16, ragged=True)

    graph_rec = None
    if rank == 0 and world == 1 and not args.timed_only and args.dtype != "f32" and not args.no_graph:
        # the same step as ONE hipGraph launch (engine.trainer.run_step_graph): last, on a fresh trainer, so that a failed capture cannot
        # disturb any other measurement.  host_ms = what the host spends per step inside hipGraphLaunch (on this ROCm stack the runtime
        # enqueues the ~490 kernel nodes one by one on the host: measured 17 ms per step, MORE than the eager step's 10 ms of Python)
        try:
            torch.cuda.empty_cache()
            trg = make_trainer(args.dtype)
            (tune_rcnn_for_pseudo_labels if rcnn else tune_for_pseudo_labels)(trg, trg._data_loader.batches[0])
            for _ in range(4):                      # two eager warm-ups, the capture (+ first replay), one more replay
                trg.run_step_graph(); trg.iter += 1
            torch.cuda.synchronize()
            kg = max(args.steps, 10)
            t1 = time.perf_counter()
            for _ in range(kg):
                trg.run_step_graph(); trg.iter += 1
            th_ = time.perf_counter() - t1
            torch.cuda.synchronize()
            dg = time.perf_counter() - t1
            mg = trg.flush_metrics()
            graph_rec = {"value": (args.label + args.unlabel) * kg / dg, "unit": "images/sec", "ms_per_step": 1e3 * dg / kg, "steps": kg,
                         "host_ms_per_step": 1e3 * th_ / kg, "losses_finite": all(v == v and abs(v) != float("inf") for v in mg.values()),
                         "note": "the whole iteration (teacher EMA + forward, pseudo-labelling, student forward / backward on four streams, AMP "
                                 "scaler, SGD) captured once with torch.cuda.graph and replayed: one launch per step on the host"}
            del trg
        except Exception as e:  # noqa: BLE001
            graph_rec = {"error": repr(e)[:300]}
        finally:
            ops.STEP_GRAPH[0] = False

    # The CPU baselines run AFTER every timed GPU phase (round 4; they ran first before): ~100 s of 32-thread CPU work right in front of
    # the GPU phases left the host slower for a while - the Faster-RCNN sub-record, whose host margin is the thinnest (19 ms of enqueue
    # work per 26 ms step), measured 277-282 img/s behind them and 305-308 without (same box, same process otherwise).  Each baseline
    # is a child process; a failure is reported in its record and cannot lose the GPU numbers, which are complete at this point.  The
    # children dump their first oracle step, which the parity checks below replay through the product's exact-f32 step.
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # in its own process
        import tempfile
        if not rcnn:
            dump = os.path.join(tempfile.gettempdir(), "utv2_bench_parity_%d.pt" % os.getpid())
            cpu_rec = cpu_baseline("fcos", args.cpu_images, args.cpu_images, args.cpu_warmup, args.cpu_steps, dump=dump)
        if rcnn or not args.no_rcnn:
            # BASELINE configs[0]: Faster-RCNN 2+2, MODEL.DEVICE=cpu, 1 process (a step costs ~2x the FCOS one: fewer repetitions)
            # its first oracle step is dumped either way: the parent replays it through the product's f32 step (`parity_fullsize` of the
            # Faster-RCNN trainer: in the headline line under rcnn.parity_fullsize)
            dump_rcnn = os.path.join(tempfile.gettempdir(), "utv2_bench_parity_rcnn_%d.pt" % os.getpid())
            if rcnn:
                dump = dump_rcnn
            cpu_rcnn = cpu_baseline("rcnn", 2, 2, 1, 2, dump=dump_rcnn)
        if rcnn:
            cpu_rec = cpu_rcnn

    parity_full = None
    if dump is not None and os.path.exists(dump):
        try:
            try:
                del tr
            except NameError:
                pass
            torch.cuda.empty_cache()
            parity_full = parity_fullsize(dump, device_index)
        except Exception as e:  # noqa: BLE001  (a failed check is reported, it must not lose the measurement)
            parity_full = {"error": repr(e)}
        finally:
            os.remove(dump)

    if rcnn_rec is not None and dump_rcnn is not None and dump_rcnn != dump and os.path.exists(dump_rcnn):
        # the Faster-RCNN f32 step against the cpu_baseline_rcnn child's first oracle step on the same 2+2 1333x800 batch
        try:
            torch.cuda.empty_cache()
            rcnn_rec["parity_fullsize"] = parity_fullsize(dump_rcnn, device_index)
        except Exception as e:  # noqa: BLE001
            rcnn_rec["parity_fullsize"] = {"error": repr(e)}
    if dump_rcnn is not None and dump_rcnn != dump and os.path.exists(dump_rcnn):
        os.remove(dump_rcnn)

    if rank == 0:
        per_step_images = (args.label + args.unlabel) * world
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS   # f16 MFMA: the bf16 rate
        out = {
            "metric": "images/sec/node (labeled+unlabeled) UTv2 step, R50-FPN 1333x800",
            "value": per_step_images * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("Faster-RCNN R50-FPN UTv2 sup1 (the trainer of configs[2] / [4]): %d labeled + %d unlabeled 1333x800 images "
                                    "per GPU, post-burn-in semi-supervised step" if rcnn else
                                    "FCOS R50-FPN UTv2 sup1 (configs[1]): %d labeled + %d unlabeled 1333x800 images per GPU, "
                                    "post-burn-in semi-supervised step") % (args.label, args.unlabel),
                       "global_batch": per_step_images, "parallelism": "dp%d" % world,
                       "precision": {"bf16": "AMP (config SOLVER.AMP.ENABLED): bf16 MFMA operands, bf16 activations and activation gradients in HBM, fp32 "
                                             "accumulate / losses / weight gradients / master weights",
                                     "f16": "AMP (config SOLVER.AMP.ENABLED) on the fp16 build of the kernels: IEEE fp16 MFMA operands / activations / "
                                            "activation gradients, dynamic loss scale with GradScaler's semantics kept on the device, fp32 accumulate / "
                                            "losses / weight gradients / master weights",
                                     "f32": "fp32 MFMA, fp32 everywhere"}[args.dtype]},
            "ranks": {"world_size": world, "backend": info["backend"], "devices": devices,
                      "launcher": _launcher_name(world), "rccl_selfcheck": rccl, "replicas": replicas, "allreduce": allreduce},
            "host": {"ms_per_step_on_96x128_images": host_ms, "enqueue_ms_per_step": 1e3 * t_host / args.steps,
                     "cabi_calls_per_step": calls_per_step, "gpu_dispatches": dispatches_per_step(args.model) if args.dtype != "f32" else None},
            "losses": {k: v for k, v in metrics.items() if k.startswith("loss") or k.startswith("teacher")},
            "pseudo_boxes_first_step": pseudo_first, "pseudo_boxes_last_step": pseudo_count,
            "learning_rate": "1e-12 (keeps the synthetic pseudo-label problem stationary over the timed window; the step's work does not depend on it)",
        }
        if amp_state is not None:
            out["loss_scale_state"] = dict(zip(("scale", "found_inf", "clean_steps"), amp_state))
        if conv:
            out["roofline"] = {"bound": "mfma", "kernel": timer.kernel + (" (RPN head 3x3 conv over p2-p6, fwd+dgrad launches)" if rcnn else
                                                                         (" (FCOS tower 3x3 convs: the forward launches and the first layer's dgrad; the other three dgrads per step: `gn_backward_dgrads`)"
                                                                          if conv_gnb else " (FCOS tower 3x3 convs, all fwd+dgrad launches)")),
                               "achieved": conv["tflops"], "peak": peak, "unit": "TFLOP/s",
                               "frac": conv["tflops"] / peak, "traffic": pmc_traffic(timer.kernel, args.model),
                               "algorithmic_bytes": conv["alg_bytes"], "algorithmic_GBps": conv["alg_gbps"],
                               "launches": conv["launches"], "avg_us": conv["avg_us"],
                               "time_share": conv["total_ms"] / (1e3 * dt)}
            if clock_ghz and clock_ghz.get("ghz"):
                # `peak` is the datasheet rate at the 2.4 GHz maximum clock; under full-chip MFMA load the board's power limit holds the
                # clock lower (measured live, above): what the matrix pipes could have delivered at THAT clock, and the fraction of it
                g = clock_ghz["ghz"]
                out["roofline"]["sustained_clock_ghz"] = g
                out["roofline"]["peak_at_sustained_clock"] = peak * g / 2.4
                out["roofline"]["frac_of_peak_at_sustained_clock"] = conv["tflops"] / (peak * g / 2.4)
                out["roofline"]["clock_probe"] = "utv2_conv_clock_probe: s_memtime ticks per 10 ns s_memrealtime tick over the lifetime (%.0f us) of workgroup 0 of the last multi-level 256-tile launch of the timed region" % clock_ghz["workgroup_lifetime_us"]
            if board_power:
                out["roofline"]["board_power"] = board_power
            ceil_ = board_mfma_ceiling(args.dtype) if args.dtype != "f32" else None
            if ceil_:
                out["roofline"]["board_mfma_ceiling"] = ceil_
                out["roofline"]["frac_of_board_mfma_ceiling"] = conv["tflops"] / ceil_["tflops"]
            if conv_x:
                out["roofline"]["exclusive"] = {"achieved": conv_x["tflops"], "frac": conv_x["tflops"] / peak, "avg_us": conv_x["avg_us"],
                                                "launches": conv_x["launches"],
                                                "note": "same launches, side streams off (UTV2_OVERLAP_TEACHER=0 UTV2_WGRAD_STREAM=0), outside the timed region: "
                                                        "in the timed region the teacher pass / weight gradients share the CUs with this kernel"}
        if conv and conv_gnb:
            # 3 of the 4 tower dgrads per step run the kernels' GroupNorm-backward instantiation: the same MFMA work + the
            # ReLU mask plane, a second operand row and per-channel partial sums in the epilogue - another symbol in the kernel trace,
            # so it is reported apart from `roofline` (whose launches are the forward convs and the first layer's dgrad)
            out["roofline"]["gn_backward_dgrads"] = {
                "kernel": timer.kernel_gnb, "achieved": conv_gnb["tflops"], "frac": conv_gnb["tflops"] / peak, "launches": conv_gnb["launches"],
                "avg_us": conv_gnb["avg_us"], "algorithmic_bytes": conv_gnb["alg_bytes"], "time_share": conv_gnb["total_ms"] / (1e3 * dt),
                "all_tower_launches": {"achieved": (conv["tflops"] * conv["total_ms"] + conv_gnb["tflops"] * conv_gnb["total_ms"]) / (conv["total_ms"] + conv_gnb["total_ms"]),
                                       "frac": (conv["tflops"] * conv["total_ms"] + conv_gnb["tflops"] * conv_gnb["total_ms"]) / (conv["total_ms"] + conv_gnb["total_ms"]) / peak,
                                       "launches": conv["launches"] + conv_gnb["launches"]},
                "note": "the epilogue work replaces gn_bwd_partial (one pass over the gradient and the GroupNorm input per layer); UTV2_GN_BWD_FUSE=0 "
                        "puts these launches back on the plain kernel"}
            if conv_gnb_x:
                out["roofline"]["gn_backward_dgrads"]["exclusive"] = {"achieved": conv_gnb_x["tflops"], "frac": conv_gnb_x["tflops"] / peak,
                                                                      "avg_us": conv_gnb_x["avg_us"], "launches": conv_gnb_x["launches"]}
        if wg:
            out["roofline_wgrad"] = {"bound": "mfma", "kernel": wtimer.kernel, "achieved": wg["tflops"], "peak": peak, "unit": "TFLOP/s",
                                     "frac": wg["tflops"] / peak, "traffic": pmc_traffic("conv_wgrad_bf16_w8", args.model),
                                     "algorithmic_bytes": wg["alg_bytes"], "launches": wg["launches"], "avg_us": wg["avg_us"],
                                     "time_share": wg["total_ms"] / (1e3 * dt)}
            if wg_x:
                out["roofline_wgrad"]["exclusive"] = {"achieved": wg_x["tflops"], "frac": wg_x["tflops"] / peak, "avg_us": wg_x["avg_us"],
                                                      "launches": wg_x["launches"]}
        if f32_rec is not None:
            out["f32"] = f32_rec
        if cpu_rec is not None:
            out["cpu_baseline"] = cpu_rec
        if parity_full is not None:
            out["parity_fullsize"] = parity_full
        if rcnn_rec is not None:
            out["rcnn"] = rcnn_rec
        if small_rec:
            # images / s at 2 + 2 per GPU over images / s at 4 + 4 per GPU, same trainer and 16-bit type: 1.0 = no per-image penalty for the small batch
            ref = {"fcos": out["value"], "rcnn": (rcnn_rec or {}).get("value")}
            for k_, r_ in small_rec.items():
                if isinstance(r_, dict) and r_.get("value") and ref.get(k_):
                    r_["per_image_rate_vs_4p4"] = r_["value"] / ref[k_]
            out["small_batch"] = small_rec
        if ragged_rec is not None:
            out["ragged_canvases"] = ragged_rec
        if graph_rec is not None:
            out["host"]["step_as_hipgraph"] = graph_rec
        if cpu_rcnn is not None and not rcnn:
            out["cpu_baseline_rcnn"] = cpu_rcnn
        print(json.dumps(out), flush=True)


def main(argv=None):
    """`python bench.py --gpus N`: N ranks, one per GPU.  Started bare, the ranks are spawned here (ubteacher.engine.launch, the
    counterpart of the reference's train_net.py:62-73 `launch(...)`); started by `python -m torch.distributed.run --nproc-per-node N
    bench.py --gpus N` (the driver's form), this process IS one rank and joins that world.  Either way a world that is not exactly
    N ranks on N distinct GPUs is an error, never a silently smaller measurement."""
    args = parse_args(argv)
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_run(args.model, args.label, args.unlabel, args.cpu_warmup, args.cpu_steps, dump=args.cpu_dump)), flush=True)
        return
    if args.subrecord_only:
        from ubteacher import hip
        hip.load()
        print(json.dumps(step_subrecord(args.model, args, 0, None, steps=args.steps, warmup=args.warmup, dtype=args.dtype,
                                        ragged=(400, 1200, 1333) if args.ragged else None)), flush=True)
        return
    from ubteacher.engine.launch import launch
    launch(worker, args.gpus, args=(args,))


if __name__ == "__main__":
    main()
