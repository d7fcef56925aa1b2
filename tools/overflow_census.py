"""fp16 range census of the benchmark step: would fp16 activations / activation gradients - the reference's AMP mode
is fp16 autocast + GradScaler, engine/trainer.py:195,207,424-426 - fit the tensors of THIS workload?

Runs the bench workload (FCOS or Faster-RCNN UTv2 step, 1333x800, random-init weights tuned as bench.py does) in exact-f32 mode and records,
for every conv / GroupNorm / elementwise entry point that would store a 16-bit tensor under AMP, the largest |value| of its output
(fp16 max = 65504) and, for gradient-producing entry points, the share of non-zero elements below fp16's normal range (6.1e-5) and
below its subnormal floor (6e-8) at loss scale 1 and at GradScaler's initial scale 65536, plus the largest scaled value.

    python tools/overflow_census.py [--model fcos|rcnn] [--label 4 --unlabel 4]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "unbiased-teacher-v2_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FWD = ["conv2d_fwd", "conv2d_stem_fwd", "conv2d_ml_fwd", "groupnorm_relu_seg_fwd", "groupnorm_relu_fwd", "upsample2x_add", "maxpool3x3s2"]
BWD = ["conv2d_dgrad", "conv2d_ml_dgrad", "groupnorm_relu_seg_bwd", "groupnorm_relu_bwd", "relu_bwd_scale", "downsample2x_sum",
       "zero_interleave2x", "add"]
F16_MAX, F16_MIN_NORMAL, F16_MIN_SUB = 65504.0, 6.1035e-5, 5.96e-8
SCALE = 65536.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="fcos")
    ap.add_argument("--label", type=int, default=4)
    ap.add_argument("--unlabel", type=int, default=4)
    args = ap.parse_args()
    import bench
    from ubteacher import hip
    from ubteacher.engine import UBRCNNTeacherTrainer, UBTeacherTrainer
    from ubteacher.presets import get_config
    hip.load()
    rcnn = args.model == "rcnn"
    cfg = get_config(args.model, 1, ["SOLVER.IMG_PER_BATCH_LABEL", args.label, "SOLVER.IMG_PER_BATCH_UNLABEL", args.unlabel,
                                     "SEMISUPNET.BURN_UP_STEP", 0, "SOLVER.AMP.ENABLED", False, "MODEL.DEVICE", "cuda:0"])
    torch.manual_seed(0)
    tr = (UBRCNNTeacherTrainer if rcnn else UBTeacherTrainer)(cfg)
    tr.iter = 1
    tr.log_period = 10 ** 9
    if rcnn:
        tr.optimizer.param_groups[0]["lr"] = 1e-12
    (bench.tune_rcnn_for_pseudo_labels if rcnn else bench.tune_for_pseudo_labels)(tr, tr._data_loader.batches[0])
    stats = {}
    enabled = [False]

    def first_tensor(out):
        if torch.is_tensor(out):
            return out
        if isinstance(out, (tuple, list)):
            for o in out:
                if torch.is_tensor(o) and o.dtype.is_floating_point and o.numel() > 4096:
                    return o
        return None

    def wrap(name, grad):
        orig = getattr(hip, name)

        def f(*a, **k):
            out = orig(*a, **k)
            t = first_tensor(out)
            if enabled[0] and t is not None and t.dtype.is_floating_point:
                v = t.detach().float().abs()
                nz = v > 0
                n = int(nz.sum())
                rec = stats.setdefault(name, {"calls": 0, "max_abs": 0.0, "nonfinite": 0, "nonzero": 0, "below_normal": 0, "below_sub": 0,
                                              "below_normal_scaled": 0, "below_sub_scaled": 0, "grad": grad})
                rec["calls"] += 1
                rec["max_abs"] = max(rec["max_abs"], float(v.max()))
                rec["nonfinite"] += int((~torch.isfinite(t)).sum())
                rec["nonzero"] += n
                rec["below_normal"] += int((nz & (v < F16_MIN_NORMAL)).sum())
                rec["below_sub"] += int((nz & (v < F16_MIN_SUB)).sum())
                if grad:
                    rec["below_normal_scaled"] += int((nz & (v * SCALE < F16_MIN_NORMAL)).sum())
                    rec["below_sub_scaled"] += int((nz & (v * SCALE < F16_MIN_SUB)).sum())
            return out
        setattr(hip, name, f)
    for n in FWD:
        if hasattr(hip, n):
            wrap(n, False)
    for n in BWD:
        if hasattr(hip, n):
            wrap(n, True)
    tr.run_step_full_semisup(); tr.iter += 1
    enabled[0] = True
    tr.run_step_full_semisup(); tr.iter += 1
    torch.cuda.synchronize()
    out = {"model": args.model, "images": "%d+%d 1333x800" % (args.label, args.unlabel), "fp16_max": F16_MAX, "loss_scale": SCALE, "entries": {}}
    for k, r in stats.items():
        nzc = max(r["nonzero"], 1)
        e = {"calls": r["calls"], "max_abs": r["max_abs"], "nonfinite": r["nonfinite"], "overflows_fp16": r["max_abs"] > F16_MAX,
             "frac_below_fp16_normal": r["below_normal"] / nzc, "frac_below_fp16_subnormal": r["below_sub"] / nzc}
        if r["grad"]:
            e.update({"max_abs_scaled": r["max_abs"] * SCALE, "overflows_fp16_scaled": r["max_abs"] * SCALE > F16_MAX,
                      "largest_power_of_two_scale_that_fits": float(2 ** int(torch.log2(torch.tensor(F16_MAX / max(r["max_abs"], 1e-30))).floor())),
                      "frac_below_fp16_normal_scaled": r["below_normal_scaled"] / nzc, "frac_below_fp16_subnormal_scaled": r["below_sub_scaled"] / nzc})
        out["entries"][k] = e
    out["any_forward_overflow"] = any(e["overflows_fp16"] for k, e in out["entries"].items() if k in FWD)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
