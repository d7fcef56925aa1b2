"""Pins the CPU oracle (oracle/utv2_oracle.py) against golden vectors produced by executing the
reference's own modules (tests/golden/gen_golden.py).  fp32: rtol 1e-5 / atol 1e-6 on losses and
gradients; integer targets and NMS selections exact."""
import os

import numpy as np
import pytest
import torch

from oracle import utv2_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.allclose(a, b, rtol=rtol, atol=atol), float(np.abs(a - b).max())


@pytest.fixture(scope="module")
def fc():
    return dict(np.load(os.path.join(G, "fcos_outputs.npz")))


def head(fc, grad=False):
    outs = []
    for nm in ("logits", "reg", "std", "ctr"):
        outs.append([T(fc["%s%d" % (nm, l)]).clone().requires_grad_(grad) for l in range(5)])
    H, W = int(fc["H"]), int(fc["W"])
    locs = [O.compute_locations(-(-H // s), -(-W // s), s) for s in (8, 16, 32, 64, 128)]
    return outs, locs


def gts(fc, prefix, N):
    out = []
    for i in range(N):
        g = dict(boxes=T(fc["%s%d_boxes" % (prefix, i)]).float().reshape(-1, 4), classes=T(fc["%s%d_classes" % (prefix, i)]).long())
        if "%s%d_std" % (prefix, i) in fc:
            g["reg_pred_std"] = T(fc["%s%d_std" % (prefix, i)]).float().reshape(-1, 4)
            g["scores"] = T(fc["%s%d_scores" % (prefix, i)])
        out.append(g)
    return out


@pytest.mark.parametrize("case", ["sup", "supempty"])
def test_supervised_losses_and_grads(fc, case):
    cfg = O.FCOSCfg()
    (lg, rg, sd, ct), locs = head(fc, True)
    losses, tg = O.fcos_losses(cfg, lg, rg, sd, ct, locs, gts(fc, case + "_gt", int(fc["N"])))
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(losses[k].detach(), fc["%s_%s" % (case, k)])
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    for nm, lst in zip(("logits", "reg", "std", "ctr"), (lg, rg, sd, ct)):
        for l in range(5):
            g = lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l])
            close(g, fc["%s_g%s%d" % (case, nm, l)], rtol=1e-4, atol=1e-7)
    for l in range(5):
        assert np.array_equal(tg["labels"][l].numpy(), fc["%s_labels%d" % (case, l)])
        assert np.array_equal(tg["target_inds"][l].numpy(), fc["%s_tinds%d" % (case, l)])
        close(tg["reg_targets"][l], fc["%s_regt%d" % (case, l)])


def test_pseudo_losses_and_grads(fc):
    cfg = O.FCOSCfg()
    (lg, rg, sd, ct), locs = head(fc, True)
    N = int(fc["N"])
    losses, ex = O.fcos_pseudo_losses(cfg, lg, rg, sd, ct, locs, {"cls": gts(fc, "pcls_gt", N), "reg": gts(fc, "preg_gt", N)})
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr", "teacher_better_student"):
        close(losses[k].detach().float(), fc["pseudo_%s" % k])
    assert float(fc["pseudo_teacher_better_student"]) > 0  # the selection branch is exercised
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    for nm, lst in zip(("logits", "reg", "std", "ctr"), (lg, rg, sd, ct)):
        for l in range(5):
            g = lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l])
            close(g, fc["pseudo_g%s%d" % (nm, l)], rtol=1e-4, atol=1e-7)
    for l in range(5):
        close(ex["reg"]["boundary_vars"][l], fc["preg_bvars%d" % l])
        assert np.array_equal(ex["reg"]["labels"][l].numpy(), fc["preg_labels%d" % l])


@pytest.mark.parametrize("method", ["cls", "cls_n_ctr", "cls_n_loc"])
def test_decode_nms_threshold(fc, method):
    cfg = O.FCOSCfg()
    (lg, rg, sd, ct), locs = head(fc)
    N, H, W = int(fc["N"]), int(fc["H"]), int(fc["W"])
    res = O.fcos_predict(cfg, lg, rg, sd, ct, locs, [(H, W)] * N, method)
    for i, r in enumerate(res):
        assert np.array_equal(r["classes"].numpy(), fc["det_%s_%d_classes" % (method, i)])  # exact selection + order
        close(r["boxes"], fc["det_%s_%d_boxes" % (method, i)], rtol=1e-5, atol=1e-4)
        close(r["scores"], fc["det_%s_%d_scores" % (method, i)])
        close(r["centerness"], fc["det_%s_%d_ctr" % (method, i)])
        close(r["cls_confid"], fc["det_%s_%d_conf" % (method, i)])
        close(r["reg_pred_std"], fc["det_%s_%d_std" % (method, i)])
        th = O.threshold_bbox(r, 0.3)
        close(th["boxes"], fc["thr_%s_%d_boxes" % (method, i)], rtol=1e-5, atol=1e-4)
        close(th["scores"], fc["thr_%s_%d_scores" % (method, i)])
        # the two-threshold selection, pseudo_generator.py:107-131 (kept set exact: same classes in the same order, same count)
        thr = tuple(float(v) for v in fc["thrcc_thresholds"])
        (tc,), num = O.process_pseudo_label([r], thr, "thresholding_cls_ctr")
        assert np.array_equal(tc["classes"].numpy(), fc["thrcc_%s_%d_classes" % (method, i)])
        assert num == float(fc["thrcc_%s_%d_num" % (method, i)])
        close(tc["boxes"], fc["thrcc_%s_%d_boxes" % (method, i)], rtol=1e-5, atol=1e-4)
        for k, g in (("scores", "scores"), ("centerness", "ctr"), ("cls_confid", "conf"), ("reg_pred_std", "std")):
            close(tc[k], fc["thrcc_%s_%d_%s" % (method, i, g)])
        assert 0 < len(tc["scores"]) < len(r["scores"])          # the thresholds bite


def test_small_ops():
    d = dict(np.load(os.path.join(G, "small_ops.npz")))
    pred = T(d["pred"]).requires_grad_(True)
    tgt, w = T(d["tgt"]), T(d["w"])
    std = T(d["std"]).requires_grad_(True)
    l = O.giou_loss_ltrb(pred, tgt, w)
    l.backward()
    close(l.detach(), d["giou"]); close(pred.grad, d["giou_gpred"], rtol=1e-4)
    pred.grad = None
    iw = O.iou_targets(pred.detach(), tgt)
    close(iw, d["iou_targets"])
    l = O.nl_loss(pred, std, tgt, iw)
    l.backward()
    close(l.detach(), d["nll"]); close(pred.grad, d["nll_gpred"], rtol=1e-4); close(std.grad, d["nll_gstd"], rtol=1e-4)
    close(O.ctrness_targets(tgt), d["ctr_targets"])
    close(O.integral(T(d["integral_in"])), d["integral_out"])


def test_ema_bit_exact():
    d = dict(np.load(os.path.join(G, "ema.npz")))
    for keep in (0.0, 0.9996, 0.9999):
        tag = str(keep).replace(".", "p")
        keys = sorted(k[len(tag) + 3:] for k in d if k.startswith(tag + "_s_"))
        s = {k: T(d["%s_s_%s" % (tag, k)]) for k in keys}
        t = {k: T(d["%s_t_%s" % (tag, k)]) for k in keys}
        out = O.ema_update(s, t, keep)
        for k in keys:
            assert np.array_equal(out[k].numpy(), d["%s_out_%s" % (tag, k)])  # bit exact


def test_center_sample_variant():
    """MODEL.FCOS.CENTER_SAMPLE True (config-reachable): targets, losses and gradients of the oracle vs the
    reference's own get_sample_region path (tests/golden/fcos_center_sample.npz)."""
    cs = dict(np.load(os.path.join(G, "fcos_center_sample.npz")))
    cfg = O.FCOSCfg(center_sample=True, radius=float(cs["radius"]))
    (lg, rg, sd, ct), locs = head(cs, True)
    N = int(cs["N"])
    losses, tg = O.fcos_losses(cfg, lg, rg, sd, ct, locs, gts(cs, "gt", N))
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(losses[k].detach(), cs["loss_%s" % k])
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    for nm, lst in zip(("logits", "reg", "std", "ctr"), (lg, rg, sd, ct)):
        for l in range(5):
            g = lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l])
            close(g, cs["g%s%d" % (nm, l)], rtol=1e-4, atol=1e-7)
    npos = 0
    for l in range(5):
        assert np.array_equal(tg["labels"][l].numpy(), cs["labels%d" % l])
        assert np.array_equal(tg["target_inds"][l].numpy(), cs["tinds%d" % l])
        close(tg["reg_targets"][l], cs["regt%d" % l])
        npos += int((cs["labels%d" % l] < 80).sum())
    assert npos > 0
    # the plain in-box rule gives MORE positives on the same inputs: the variant is really exercised
    _, tg0 = O.fcos_losses(O.FCOSCfg(), lg, rg, sd, ct, locs, gts(cs, "gt", N))
    assert sum(int((tg0["labels"][l] < 80).sum()) for l in range(5)) > npos


def test_ignore_near_variant():
    """SEMISUPNET.PSEUDO_CLS_IGNORE_NEAR (fcos_outputs.py:841-851) on the centre-sampling inputs: kept locations, supervised losses and
    gradients vs the reference's own; the pseudo losses do not depend on it (the reference never reads keep_locations there)."""
    cs = dict(np.load(os.path.join(G, "fcos_center_sample.npz")))
    cfg = O.FCOSCfg(center_sample=True, radius=float(cs["radius"]))
    (lg, rg, sd, ct), locs = head(cs, True)
    N = int(cs["N"])
    losses, tg = O.fcos_losses(cfg, lg, rg, sd, ct, locs, gts(cs, "gt", N), ignore_near=True)
    dropped = 0
    for l in range(5):
        assert np.array_equal(tg["keep_locations"][l].numpy().astype(np.uint8), cs["ign_keep%d" % l])
        dropped += int((cs["ign_keep%d" % l] == 0).sum())
    assert dropped > 0
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(losses[k].detach(), cs["ign_loss_%s" % k])
    assert abs(float(cs["ign_loss_loss_fcos_cls"]) - float(cs["loss_loss_fcos_cls"])) > 1.0     # the switch matters on these inputs
    (losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]).backward()
    for nm, lst in zip(("logits", "reg", "std", "ctr"), (lg, rg, sd, ct)):
        for l in range(5):
            g = lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l])
            close(g, cs["ign_g%s%d" % (nm, l)], rtol=1e-4, atol=1e-7)
    (lg, rg, sd, ct), locs = head(cs, False)
    pg = gts(cs, "ign_pgt", N)
    for ign in (False, True):
        pl, _ = O.fcos_pseudo_losses(cfg, lg, rg, sd, ct, locs, {"cls": pg, "reg": pg}, ignore_near=ign)
        for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
            close(pl[k].detach().float(), cs["ign_pseudo_%s" % k])


LOSS_VARIANTS = {
    "klloss": dict(kl_loss_type="klloss"),
    "nokl": dict(kl_loss=False),
    "iouq": dict(quality_est="iou"),
    "lociou": dict(loc_loss_type="iou"),
    "loclinear": dict(loc_loss_type="linear_iou"),
    "klloss_iouq_linear": dict(kl_loss_type="klloss", quality_est="iou", loc_loss_type="linear_iou"),
    "klloss_sum": dict(kl_loss_type="klloss", loc_fun_all="sum"),                       # MODEL.FCOS.LOC_FUN_ALL (kl_loss.py:48-64)
    "klloss_wsum": dict(kl_loss_type="klloss", loc_fun_all="weight_ctr_sum"),
    "klloss_wmean_iouq": dict(kl_loss_type="klloss", loc_fun_all="weight_ctr_mean", quality_est="iou"),
}


@pytest.mark.parametrize("case", sorted(LOSS_VARIANTS))
def test_supervised_loss_variants(case):
    """KL_LOSS_TYPE / KL_LOSS / QUALITY_EST / LOC_LOSS_TYPE variants (config-reachable): oracle vs the reference's own
    FCOSOutputs.losses with that config (tests/golden/fcos_loss_variants.npz)."""
    lv = dict(np.load(os.path.join(G, "fcos_loss_variants.npz")))
    cfg = O.FCOSCfg(**LOSS_VARIANTS[case])
    (lg, rg, sd, ct), locs = head(lv, True)
    losses, _ = O.fcos_losses(cfg, lg, rg, sd, ct, locs, gts(lv, "gt", int(lv["N"])))
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(losses[k].detach(), lv["%s_%s" % (case, k)])
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    for nm, lst in zip(("reg", "std", "ctr"), (rg, sd, ct)):
        for l in range(5):
            g = lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l])
            close(g, lv["%s_g%s%d" % (case, nm, l)], rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("case,kw", [("pseudo_nll", {}), ("pseudo_kl", dict(kl_loss_type="klloss")),
                                     ("pseudo_kl_wmean", dict(kl_loss_type="klloss", loc_fun_all="weight_ctr_mean"))])
def test_pseudo_regression_kl_term(case, kw):
    """SEMISUPNET.CONSIST_REG_LOSS other than the TS-better selection: loss_fcos_loc = KLLOSS_WEIGHT * (NLL | KL) on the regression
    pseudo set (fcos_outputs.py:571-585)."""
    lv = dict(np.load(os.path.join(G, "fcos_loss_variants.npz")))
    cfg = O.FCOSCfg(reg_unsup_loss="mse_loss_all_raw", **kw)
    (lg, rg, sd, ct), locs = head(lv, True)
    N = int(lv["N"])
    losses, _ = O.fcos_pseudo_losses(cfg, lg, rg, sd, ct, locs, {"cls": gts(lv, "pcls_gt", N), "reg": gts(lv, "preg_gt", N)})
    for k in ("loss_fcos_cls", "loss_fcos_loc", "loss_fcos_ctr"):
        close(losses[k].detach().float(), lv["%s_%s" % (case, k)])
    assert "teacher_better_student" not in losses
    tot = losses["loss_fcos_cls"] + 2.0 * losses["loss_fcos_loc"] + 3.0 * losses["loss_fcos_ctr"]
    tot.backward()
    for nm, lst in zip(("reg", "std", "ctr"), (rg, sd, ct)):
        for l in range(5):
            g = lst[l].grad if lst[l].grad is not None else torch.zeros_like(lst[l])
            close(g, lv["%s_g%s%d" % (case, nm, l)], rtol=1e-4, atol=1e-7)
