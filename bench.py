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


def tune_for_pseudo_labels(trainer, batch, target_std=1.5, bias=-6.0):
    """Random-init weights give no confident detections; rescale the student's cls_logits (on the
    device, with the product's own forward) so the teacher emits some pseudo boxes per image."""
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
        s = torch.cat([x.reshape(-1) for x in raw["logits_pred"]]).std()
    m.train()
    m.store.touch()
    w.mul_(target_std / s.clamp(min=1e-12))
    b.fill_(bias)
    sd["proposal_generator.fcos_head.bbox_pred_std.bias"].fill_(-3.0)
    m.store.touch()
    trainer._update_teacher_model(keep_rate=0.0)  # teacher := student
    sd["proposal_generator.fcos_head.bbox_pred_std.bias"].fill_(0.0)  # student less certain than teacher
    m.store.touch()  # weights were edited in place: invalidate the bf16 mirror


class ConvTimer:
    """HIP-event timing (torch events recorded on the stream the kernels are launched on) of every launch of the
    dominant kernel inside the timed region.  Dominant kernel (largest share of GPU time in profiles/): the multi-level
    3x3 implicit-GEMM conv of the shared FCOS towers - forward AND dgrad launches, 256 -> 256 channels over all five FPN
    levels of the student batch in one launch:
        bf16: conv_igemm_bf16_w8<true,__bf16> on the whole rounds of 256 x 256 tiles + conv_igemm_bf16_v2<128,true,64,__bf16> on the
              remaining output rows (two kernels, one C-ABI call = one timed launch)      f32: conv_igemm_f32<128,0,true>"""

    def __init__(self, dtype):
        self.pairs = []
        self.enabled = False
        self.bf16 = dtype == "bf16"
        self.entry = "conv2d_ml_fwd_bf16" if self.bf16 else "conv2d_ml_fwd"
        self.kernel = "conv_igemm_bf16_w8<true,__bf16>+conv_igemm_bf16_v2<128,true,64,__bf16>" if self.bf16 else "conv_igemm_f32<128,0,true>"

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
                mine = (x2d.dtype == torch.bfloat16 and out_dt == torch.bfloat16 and K >= 256 and C % 64 == 0 and Kred >= 1024
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
            timer.pairs.append((e0, e1, 2.0 * P * K * Kred, float(bts)))
            return y

        setattr(hip, self.entry, wrapped)

    def summary(self):
        if not self.pairs:
            return None
        ms = sum(p[0].elapsed_time(p[1]) for p in self.pairs)
        fl = sum(p[2] for p in self.pairs)
        by = sum(p[3] for p in self.pairs)
        n = len(self.pairs)
        return dict(launches=n, total_ms=ms, avg_us=1e3 * ms / n, tflops=fl / ms / 1e9, alg_bytes=by / n,
                    alg_gbps=by / ms / 1e6)


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS command
    (profiles/r01_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs; FETCH_SIZE doubled as
    MI355X_MICROARCH.md's HBM section prescribes for 16-byte-per-lane reads on gfx950)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f).get(kernel)
        return None if t is None else float(t["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(cfg):
    """The oracle (CPU port of the reference step) timed on the host cores on a bounded sample:
    ONE step with 1 labeled + 1 unlabeled 1333x800 image."""
    from oracle import utv2_oracle as O
    from ubteacher.data.synthetic import make_gt, make_image, strong_view
    from ubteacher.modeling import build_model
    import numpy as np
    ccfg = cfg.clone()
    ccfg.defrost()
    ccfg.MODEL.DEVICE = "cuda"
    rng = np.random.default_rng(0)
    model = build_model(ccfg)
    sd = {k: v.detach().cpu().clone().contiguous() for k, v in model.state_dict().items()}
    del model
    torch.cuda.empty_cache()

    def im():
        return make_image(rng, 800, 1333)
    gt = make_gt(rng, 800, 1333)
    g = dict(boxes=gt.gt_boxes.tensor, classes=gt.gt_classes)
    wk = im()
    batch = ([{"image": strong_view(rng, wk), "gt": g}], [{"image": wk, "gt": g}], [{"image": im()}], [{"image": im()}])
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    O.fcos_semisup_step(O.FCOSCfg(), sd, dict(sd), batch, keep_rate=0.9999)
    dt = time.perf_counter() - t0
    return {"value": 2.0 / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "1 step, 1 labeled (weak+strong views) + 1 unlabeled 1333x800 image, fp32, torch CPU kernels, %.1f s" % dt}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--label", type=int, default=4, help="labeled images per GPU")
    ap.add_argument("--unlabel", type=int, default=4, help="unlabeled images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16",
                    help="conv arithmetic: bf16 = the config's SOLVER.AMP.ENABLED path (bf16 MFMA, fp32 accumulate); f32 = exact-f32 MFMA")
    return ap.parse_args(argv)


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

    from ubteacher.engine import UBTeacherTrainer
    from ubteacher.presets import get_config
    from ubteacher import hip
    hip.load()
    cfg = get_config("fcos", 1, ["SOLVER.IMG_PER_BATCH_LABEL", args.label * world, "SOLVER.IMG_PER_BATCH_UNLABEL",
                                 args.unlabel * world, "SEMISUPNET.BURN_UP_STEP", 0, "SOLVER.AMP.ENABLED", args.dtype == "bf16",
                                 "MODEL.DEVICE", "cuda:%d" % device_index])
    torch.manual_seed(0)
    timer = ConvTimer(args.dtype)
    timer.install()
    tr = UBTeacherTrainer(cfg)
    batch = tr._data_loader.batches[0]
    tune_for_pseudo_labels(tr, batch)
    tr.sync_replicas()   # identical students / teachers on every rank (DDP broadcasts rank 0's parameters at construction)
    tr.iter = 1
    tr.log_period = 10 ** 9
    for _ in range(args.warmup):
        tr.run_step_full_semisup(); tr.iter += 1

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.run_step_full_semisup(); tr.iter += 1
    sync()
    dt = time.perf_counter() - t0
    timer.enabled = False
    devices = [device_index]
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        devs = [None] * world
        dist.all_gather_object(devs, device_index)
        devices = devs
    metrics = tr.flush_metrics()
    conv = timer.summary()

    if rank == 0:
        per_step_images = (args.label + args.unlabel) * world
        out = {
            "metric": "images/sec/node (labeled+unlabeled) UTv2 step, R50-FPN 1333x800",
            "value": per_step_images * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "FCOS R50-FPN UTv2 sup1 (configs[1]): %d labeled + %d unlabeled 1333x800 images per GPU, "
                                   "post-burn-in semi-supervised step" % (args.label, args.unlabel),
                       "global_batch": per_step_images, "parallelism": "dp%d" % world,
                       "precision": "AMP (config SOLVER.AMP.ENABLED): bf16 MFMA operands, bf16 activations and activation gradients in HBM, fp32 accumulate / losses / weight gradients / master weights" if args.dtype == "bf16" else "fp32 MFMA, fp32 everywhere"},
            "ranks": {"world_size": world, "backend": info["backend"], "devices": devices,
                      "launcher": _launcher_name(world)},
            "losses": {k: v for k, v in metrics.items() if k.startswith("loss") or k.startswith("teacher")},
        }
        if conv:
            peak = PEAK_BF16_MFMA_TFLOPS if args.dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
            out["roofline"] = {"bound": "mfma", "kernel": timer.kernel + " (FCOS tower 3x3 convs, all fwd+dgrad launches)",
                               "achieved": conv["tflops"], "peak": peak, "unit": "TFLOP/s",
                               "frac": conv["tflops"] / peak, "traffic": pmc_traffic(timer.kernel),
                               "algorithmic_bytes": conv["alg_bytes"], "algorithmic_GBps": conv["alg_gbps"],
                               "launches": conv["launches"], "avg_us": conv["avg_us"],
                               "time_share": conv["total_ms"] / (1e3 * dt)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:  # never lose the GPU measurement to a host-side problem
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)


def main(argv=None):
    """`python bench.py --gpus N`: N ranks, one per GPU.  Started bare, the ranks are spawned here (ubteacher.engine.launch, the
    counterpart of the reference's train_net.py:62-73 `launch(...)`); started by `python -m torch.distributed.run --nproc-per-node N
    bench.py --gpus N` (the driver's form), this process IS one rank and joins that world.  Either way a world that is not exactly
    N ranks on N distinct GPUs is an error, never a silently smaller measurement."""
    args = parse_args(argv)
    from ubteacher.engine.launch import launch
    launch(worker, args.gpus, args=(args,))


if __name__ == "__main__":
    main()
