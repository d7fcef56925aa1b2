"""End-to-end parity of the HIP UTv2 Faster-RCNN path vs the CPU oracle on a small seeded problem
(teacher RPN+ROI inference + thresholding, both student forwards with injected sampling keys, the
weighted loss, backward, SGD, EMA).  Losses 1e-3 relative; NMS/top-k selections identical."""
import numpy as np
import pytest
import torch

from oracle import utv2_oracle as O
from tests.utv2_testutil import FixedLoader, cpu_state, make_batch, rcnn_tune as tune

pytestmark = pytest.mark.gpu
H, W = 96, 128


@pytest.fixture
def separable_roi_align():
    """The oracle's RoIAlign in its separable form (oracle FAST_ROI_ALIGN: the same arithmetic regrouped, pinned against the per-sample form
    forward and backward by tests/test_oracle_golden_rcnn.py::test_fast_roi_align_equals_pinned_form; what bench.py's CPU baseline runs):
    the per-sample form's autograd backward materialises a full feature map per ROI and tap and makes an oracle step 4x slower on the
    host.  The whole-step tests below that only need the oracle's numbers at 1e-3 use it; test_rcnn_step_vs_reference_trainer_golden and
    the kernel-level RoIAlign tests keep the per-sample form."""
    O.FAST_ROI_ALIGN[0] = True
    yield
    O.FAST_ROI_ALIGN[0] = False


def rcnn_cfg():
    from ubteacher.presets import get_config
    return get_config("rcnn", 1, ["SOLVER.IMG_PER_BATCH_LABEL", 2, "SOLVER.IMG_PER_BATCH_UNLABEL", 2,
                                  "SEMISUPNET.BURN_UP_STEP", 0, "MODEL.DEVICE", "cuda"])


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("predictor", ["FocalLoss_BoundaryVar", "FocalLoss"])
def test_rcnn_full_semisup_step_parity(predictor, separable_roi_align):
    """FocalLoss_BoundaryVar: the shipped UTv2 configuration.  FocalLoss: the UTv1 predictor the reference still ships
    (roi_heads/roi_heads.py:52-66 -> fast_rcnn.py:1296-1429): class-specific centre-size deltas, confidence-weighted focal loss on the
    pseudo-labeled branch, no boundary-variance head - pseudo boxes without pred_boxes_std (trainer.py:743-746)."""
    from ubteacher.engine import UBRCNNTeacherTrainer
    cfg = rcnn_cfg()
    cfg.MODEL.ROI_HEADS.LOSS = predictor
    utv1 = predictor == "FocalLoss"
    nb = 1 if utv1 else 2                  # images per group: the UTv1 variant runs 1 + 1 (half the oracle's CPU time; same code paths)
    cfg.SOLVER.IMG_PER_BATCH_LABEL = cfg.SOLVER.IMG_PER_BATCH_UNLABEL = nb
    if utv1:
        cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE = "smooth_l1"       # the UTv2 YAML's "nlloss" is a ValueError there (fast_rcnn.py:184-186)
        cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = False          # Detectron2's default: 4 deltas per class
    torch.manual_seed(0)
    prod, orac = make_batch(31, nb, nb, H, W, "cuda")
    tr = UBRCNNTeacherTrainer(cfg, data_loader=FixedLoader(prod))
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
    pstd = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    sd_s = tune(cpu_state(tr.model), [d["image"] for d in orac[3]], mean, pstd)
    sd_t = dict(sd_s)
    if utv1:
        assert "roi_heads.box_predictor.bbox_pred_std.weight" not in sd_s and tuple(sd_s["roi_heads.box_predictor.bbox_pred.weight"].shape) == (320, 1024)
    else:
        sd_t["roi_heads.box_predictor.bbox_pred_std.bias"] = torch.full((4,), -3.0)
    tr.model.load_state_dict(sd_s)
    tr.model_teacher.load_state_dict(sd_t)
    tr.iter = 1
    tr.optimizer.param_groups[0]["lr"] = 0.01

    g = torch.Generator().manual_seed(99)
    rpn_keys, roi_keys = [], []

    def rpn_src(n, m, device):
        k = torch.rand(n, m, generator=g)
        rpn_keys.append(k)
        return k.to(device)

    def roi_src(n, m, device):
        k = torch.rand(n, m, generator=g)
        roi_keys.append(k)
        return k.to(device)

    tr.model.proposal_generator.sample_keys = rpn_src
    tr.model.roi_heads.sample_keys = roi_src
    tr.run_step_full_semisup()
    rec = tr.flush_metrics()
    torch.cuda.synchronize()
    assert len(rpn_keys) == 2 and len(roi_keys) == 2

    # oracle keys in its compact (proposals ++ gts) convention
    post = cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN
    samp_sup = None

    def compact_roi(keys, nprops, ngts):
        return [torch.cat((keys[i, :nprops[i]], keys[i, post:post + ngts[i]])) for i in range(keys.shape[0])]

    # supervised pass: proposals counts come from the oracle itself (must equal the product's if parity holds)
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
    pstd = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    t_sd = O.ema_update(sd_s, sd_t, cfg.SEMISUPNET.EMA_KEEP_RATE)
    with torch.no_grad():
        pseudo, _ = O.rcnn_teacher(t_sd, [d["image"] for d in orac[3]], mean, pstd, thr=cfg.SEMISUPNET.BBOX_THRESHOLD)
        _, props_sup, _ = O.rcnn_student_losses(sd_s, [d["image"] for d in orac[0] + orac[1]], [d["gt"] for d in orac[0] + orac[1]],
                                                rpn_keys[0], [torch.zeros(2000)] * (2 * nb), False, mean, pstd)
        _, props_uns, _ = O.rcnn_student_losses(sd_s, [d["image"] for d in orac[2]], pseudo, rpn_keys[1], [torch.zeros(2000)] * nb,
                                                True, mean, pstd)
    assert sum(len(p["boxes"]) for p in pseudo) > 0, "test setup: teacher produced no pseudo boxes"
    gl = tr._last_pseudo
    for i, p in enumerate(pseudo):
        assert int(gl["valid"][i].sum()) == len(p["boxes"])
        assert ("pred_boxes_std" in p) == (not utv1) and ("pred_boxes_std" in gl) == (not utv1)
    keys = dict(rpn_sup=rpn_keys[0], rpn_unsup=rpn_keys[1],
                roi_sup=compact_roi(roi_keys[0], [len(p["boxes"]) for p in props_sup], [len(d["gt"]["boxes"]) for d in orac[0] + orac[1]]),
                roi_unsup=compact_roi(roi_keys[1], [len(p["boxes"]) for p in props_uns], [len(p["boxes"]) for p in pseudo]))
    rec_o, new_s, new_t, grads, _ = O.rcnn_semisup_step(sd_s, sd_t, orac, keys, keep_rate=cfg.SEMISUPNET.EMA_KEEP_RATE,
                                                       lam_u=cfg.SEMISUPNET.UNSUP_LOSS_WEIGHT, lam_r=cfg.SEMISUPNET.UNSUP_REG_LOSS_WEIGHT,
                                                       thr=cfg.SEMISUPNET.BBOX_THRESHOLD, lr=0.01, mean=mean, pix_std=pstd)
    for k, v in rec_o.items():
        assert k in rec, k
        # loss_rpn_loc_pseudo (weight 0 in the objective, trainer.py:888-890) sums over anchors made positive by
        # the Matcher's exact-equality low-quality rule against pseudo boxes that differ by 1e-5 between the
        # two teachers: an ill-conditioned selection, compared loosely.
        # loss_rpn_cls_pseudo (weight UNSUP_LOSS_WEIGHT) holds the north star's 1e-3 like every other weighted term.  The decoupled
        # tests (test_rcnn_step_fp32_tight_with_the_product_pseudo_boxes here, tests/test_fullsize_gpu.py at 1333x800) hold
        # loss_rpn_loc_pseudo to 1e-3 too once both sides see the SAME pseudo boxes.
        tol = 2e-2 if k == "loss_rpn_loc_pseudo" else 1e-3
        assert abs(rec[k] - v) <= tol * max(abs(v), 1e-6), (k, rec[k], v)
    assert rec_o["loss_box_reg_pseudo"] > 0 and rec_o["loss_rpn_cls_pseudo"] > 0
    t_after = cpu_state(tr.model_teacher)
    for k in new_t:
        assert torch.equal(t_after[k], new_t[k]), k
    s_after = cpu_state(tr.model)
    for k in new_s:
        err = float((s_after[k].double() - new_s[k].double()).abs().max())
        upd = float((new_s[k].double() - sd_s[k].double()).abs().max())
        tol = 1e-4 * float(new_s[k].abs().max()) + 4e-2 * upd + 1e-12  # discrete selections (matcher ties, ReLU gates) are ill-conditioned: measured worst 3.2e-2; the decoupled test below holds 1e-2
        assert err <= tol, (k, err, tol)


def test_rcnn_step_amp_close_to_fp32(monkeypatch):
    """The AMP (bf16 activations / MFMA operands) Faster-RCNN step runs through every typed kernel path - fp32 FPN levels
    into RoIAlign, bf16 backbone / heads, fp32 loss-side outputs - and its supervised losses stay within a few per cent
    of the fp32 step on the same seeded batch and sampling keys (proposal selection is discrete, hence the loose bound)."""
    monkeypatch.setenv("UTV2_PRECISION", "bf16")   # written for bf16 rounding (the package default AMP type is fp16, the reference's)
    from ubteacher import ops
    from ubteacher.engine import UBRCNNTeacherTrainer
    recs = {}
    try:
        for amp in (False, True):
            cfg = rcnn_cfg()
            cfg.SOLVER.AMP.ENABLED = amp
            torch.manual_seed(0)
            prod, orac = make_batch(31, 2, 2, H, W, "cuda")
            tr = UBRCNNTeacherTrainer(cfg, data_loader=FixedLoader(prod))
            assert ops.PRECISION[0] == ("bf16" if amp else "fp32")
            mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
            pstd = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
            sd_s = tune(cpu_state(tr.model), [d["image"] for d in orac[3]], mean, pstd)
            sd_t = dict(sd_s)
            sd_t["roi_heads.box_predictor.bbox_pred_std.bias"] = torch.full((4,), -3.0)
            tr.model.load_state_dict(sd_s)
            tr.model_teacher.load_state_dict(sd_t)
            tr.iter = 1
            tr.optimizer.param_groups[0]["lr"] = 0.01
            g = torch.Generator().manual_seed(99)
            src = lambda n, m, device: torch.rand(n, m, generator=g).to(device)  # noqa: E731
            tr.model.proposal_generator.sample_keys = src
            tr.model.roi_heads.sample_keys = src
            tr.run_step_full_semisup()
            recs[amp] = tr.flush_metrics()
            after = cpu_state(tr.model)
            assert all(torch.isfinite(v).all() for v in after.values())
            assert any(not torch.equal(after[k], sd_s[k]) for k in sd_s if k.endswith("weight"))
    finally:
        ops.set_precision("fp32")
    for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"):
        a, b = recs[True][k], recs[False][k]
        assert np.isfinite(a) and abs(a - b) <= 5e-2 * max(abs(b), 1e-6), (k, a, b)


def test_rcnn_fused_student_pass_equals_two_passes():
    """The fused student pass (labeled + pseudo-labeled images as ONE backbone / RPN head / RoIAlign / box head batch, losses per branch on
    their own image range) against the two passes of the reference trainer (trainer.py:838-866) on the same weights, batch and sampling
    keys: every loss to 1e-5 (per-image layers: only the summation order of sums over the batch differs), the same pseudo labels, the
    student after SGD to the accumulation-order bound of the weight gradients."""
    from ubteacher.engine import UBRCNNTeacherTrainer
    outs = {}
    for fuse in (True, False):
        cfg = rcnn_cfg()
        torch.manual_seed(0)
        prod, orac = make_batch(31, 2, 2, H, W, "cuda")
        tr = UBRCNNTeacherTrainer(cfg, data_loader=FixedLoader(prod))
        tr.fuse_student_passes = fuse
        mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
        pstd = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
        sd_s = tune(cpu_state(tr.model), [d["image"] for d in orac[3]], mean, pstd)
        sd_t = dict(sd_s)
        sd_t["roi_heads.box_predictor.bbox_pred_std.bias"] = torch.full((4,), -3.0)
        tr.model.load_state_dict(sd_s)
        tr.model_teacher.load_state_dict(sd_t)
        tr.iter = 1
        tr.optimizer.param_groups[0]["lr"] = 0.01
        calls = []

        def src(n, m, device, calls=calls):   # keys depend on the shape only: both schedules draw the same ones whatever their call order
            calls.append((n, m))
            return torch.rand(n, m, generator=torch.Generator().manual_seed(1000 * n + m)).to(device)

        tr.model.proposal_generator.sample_keys = src
        tr.model.roi_heads.sample_keys = src
        seen = []
        if fuse:
            orig = tr.model.forward_joint_begin
            tr.model.forward_joint_begin = lambda *a, **k: (seen.append(1), orig(*a, **k))[1]
        tr.run_step_full_semisup()
        torch.cuda.synchronize()
        assert bool(seen) == fuse and len(calls) == 4
        outs[fuse] = (tr.flush_metrics(), cpu_state(tr.model), sd_s, tr._last_pseudo)
    ra, rb = outs[True][0], outs[False][0]
    assert set(ra) == set(rb)
    for k in rb:
        if k.startswith("loss") or k == "total_loss":
            assert abs(ra[k] - rb[k]) <= 1e-5 * max(abs(rb[k]), 1e-6), (k, ra[k], rb[k])
    assert rb["loss_box_reg_pseudo"] > 0 and rb["loss_rpn_cls_pseudo"] > 0 and rb["loss_rpn_loc"] > 0
    for f in ("boxes", "classes", "valid", "scores"):
        assert torch.equal(outs[True][3][f], outs[False][3][f]), f
    sa, sb, s0 = outs[True][1], outs[False][1], outs[True][2]
    for k in sb:
        upd = float((sb[k].double() - s0[k].double()).abs().max())
        err = float((sa[k].double() - sb[k].double()).abs().max())
        assert err <= 1e-6 * float(sb[k].abs().max()) + 1e-3 * upd + 1e-12, (k, err, upd)


def test_rcnn_evaluation_loop_runs_and_rescales():
    """UBRCNNTeacherTrainer.test(): the eval-mode model runs `inference` (RPN test top-k -> box head -> fast_rcnn_inference) over a
    fixed-length loader, detections are rescaled to the ORIGINAL image size and the COCO box-AP dict comes back; the inference
    boxes / scores / classes of the first image equal the CPU oracle's teacher path on the same weights."""
    from ubteacher.data.synthetic import SyntheticTestLoader
    from ubteacher.engine import UBRCNNTeacherTrainer
    from ubteacher.evaluation import COCOBoxEvaluator

    class Tr(UBRCNNTeacherTrainer):
        @classmethod
        def build_test_loader(cls, cfg, dataset_name):
            return SyntheticTestLoader(cfg, num_images=3, height=H, width=W, orig_scale=1.5)

    cfg = rcnn_cfg()
    torch.manual_seed(0)
    prod, orac = make_batch(31, 2, 2, H, W, "cuda")
    tr = Tr(cfg, data_loader=FixedLoader(prod))
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
    pstd = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    loader = Tr.build_test_loader(cfg, "x")
    first = loader.items[0][0]["image"].cpu()
    sd = tune(cpu_state(tr.model), [first], mean, pstd)
    tr.model_teacher.load_state_dict(sd)
    ev = COCOBoxEvaluator(80)
    was = tr.model_teacher.training
    res = Tr.test(cfg, tr.model_teacher, evaluators=ev)
    assert tr.model_teacher.training == was
    assert set(res["bbox"]) == {"AP", "AP50", "AP75", "APs", "APm", "APl"}
    assert res["_speed"]["images"] >= 1
    n_det = sum(len(p["scores"]) for p in ev._pred.values())
    assert n_det > 0
    for p in ev._pred.values():
        if len(p["scores"]):
            assert p["boxes"][:, 2].max() <= W * 1.5 + 1e-3 and p["boxes"][:, 3].max() <= H * 1.5 + 1e-3
    # first image vs the oracle (test-time RPN top-k 1000 -> 1000, score 0.05, NMS 0.5, 100 / image), boxes scaled by 1.5
    dets, _ = O.rcnn_teacher(sd, [first], mean, pstd, pre_topk=cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, post_topk=cfg.MODEL.RPN.POST_NMS_TOPK_TEST,
                                thr=-1.0)
    mine = ev._pred[0]
    ob, osc, ocl = dets[0]["boxes"] * 1.5, dets[0]["scores"], dets[0]["classes"]
    assert len(osc) == len(mine["scores"]) and len(osc) > 0
    assert np.array_equal(np.asarray(mine["classes"]), ocl.numpy())
    assert np.allclose(np.asarray(mine["scores"]), osc.numpy(), rtol=1e-3, atol=1e-5)
    assert np.allclose(np.asarray(mine["boxes"]), ob.numpy(), rtol=1e-3, atol=5e-2)


def _golden_setup(amp=False):
    from tests.utv2_testutil import golden_batches, golden_init_state, load_step_golden
    from ubteacher.engine import UBRCNNTeacherTrainer
    d = load_step_golden("rcnn")
    _, sd0 = golden_init_state("rcnn", d)
    prod, orac = golden_batches(d, "cuda")
    cfg = rcnn_cfg()
    cfg.SOLVER.AMP.ENABLED = amp
    tr = UBRCNNTeacherTrainer(cfg, data_loader=FixedLoader(prod))
    mean = torch.tensor(cfg.MODEL.PIXEL_MEAN).view(3, 1, 1)
    pstd = torch.tensor(cfg.MODEL.PIXEL_STD).view(3, 1, 1)
    sd_s = tune(sd0, [x["image"] for x in orac[3]], mean, pstd)
    sd_t = dict(sd_s)
    sd_t["roi_heads.box_predictor.bbox_pred_std.bias"] = torch.full((4,), -3.0)
    tr.model.load_state_dict(sd_s)
    tr.model_teacher.load_state_dict(sd_t)
    tr.iter = 1
    tr.optimizer.param_groups[0]["lr"] = float(d["lr"])
    K = {k: torch.from_numpy(d["keys_" + k]) for k in ("rpn_sup", "roi_sup", "rpn_unsup", "roi_unsup")}
    calls = {"rpn": 0, "roi": 0}

    def rpn_src(n, m, device):
        k = K["rpn_sup" if calls["rpn"] == 0 else "rpn_unsup"]
        calls["rpn"] += 1
        assert tuple(k.shape) == (n, m), (tuple(k.shape), n, m)
        return k.to(device)

    def roi_src(n, m, device):
        k = K["roi_sup" if calls["roi"] == 0 else "roi_unsup"]
        calls["roi"] += 1
        assert k.shape[0] == n and k.shape[1] >= m
        return k[:, :m].contiguous().to(device)

    tr.model.proposal_generator.sample_keys = rpn_src
    tr.model.roi_heads.sample_keys = roi_src
    return d, cfg, tr, orac, sd_s, sd_t, K, mean, pstd


def test_rcnn_step_vs_reference_trainer_golden():
    """One full Faster-RCNN UTv2 iteration of the PRODUCT against the golden produced by executing the reference's own
    UBRCNNTeacherTrainer.run_step_full_semisup (tests/golden/gen_golden_step.py): record_dict within 1e-3 (the two pseudo RPN
    terms as in the oracle test above), the same pseudo-label set, teacher after EMA bit exact, student after SGD."""
    from tests.utv2_testutil import check_state_fingerprints, golden_record
    d, cfg, tr, orac, sd_s, sd_t, K, mean, pstd = _golden_setup()
    tr.run_step_full_semisup()
    rec = tr.flush_metrics()
    torch.cuda.synchronize()
    ref = golden_record(d)
    for k, v in ref.items():
        if k in ("data_time", "total_loss"):
            continue
        tol = 2e-2 if k == "loss_rpn_loc_pseudo" else 1e-3     # the one weight-0 term (trainer.py:888-890), see above
        assert abs(rec[k] - v) <= tol * max(abs(v), 1e-6), (k, rec[k], v)
    assert abs(rec["total_loss"] - ref["total_loss"]) <= 1e-3 * ref["total_loss"]
    gl = tr._last_pseudo
    i = 0
    while "pseudo%d_boxes" % i in d:
        m = gl["valid"][i].bool()
        assert int(m.sum()) == len(d["pseudo%d_boxes" % i])
        assert np.array_equal(gl["classes"][i][m].long().cpu().numpy(), d["pseudo%d_classes" % i])
        np.testing.assert_allclose(gl["boxes"][i][m].cpu().numpy(), d["pseudo%d_boxes" % i], rtol=0, atol=2e-2)
        np.testing.assert_allclose(gl["scores"][i][m].cpu().numpy(), d["pseudo%d_scores" % i], rtol=1e-3)
        i += 1
    assert i == 2
    check_state_fingerprints(d, "teacher", cpu_state(tr.model_teacher), 0.0, exact=True)
    check_state_fingerprints(d, "student", cpu_state(tr.model), 1e-4, rtol_update=4e-2)


@pytest.mark.parametrize("kind,tol", [("bf16", 1e-2), ("fp16", 3e-3)])
def test_rcnn_step_bf16_vs_rounding_oracle(kind, tol, monkeypatch, separable_roi_align):
    """BASELINE configs[4] (Faster-RCNN, bf16 MFMA conv path; and the same on the fp16 build of the kernels with the dynamic loss scale,
    UTV2_PRECISION=fp16): the full AMP step against the oracle with the same operand rounding emulated in its convs / linears (O.CONV_ROUND).  Discrete selections are decoupled from rounding noise the way the FCOS AMP
    test does it: the oracle is handed the product's pseudo labels and the product's RPN proposals of the two student passes, and
    both sides use the same injected sampling keys; then every loss must agree within 1e-2 relative and the teacher after EMA is
    bit exact.  The product's own teacher detections must mostly coincide with the rounding oracle's."""
    from ubteacher import ops
    monkeypatch.setenv("UTV2_PRECISION", kind)
    try:
        O.CONV_ROUND[0] = kind
        d, cfg, tr, orac, sd_s, sd_t, K, mean, pstd = _golden_setup(amp=True)
        assert ops.PRECISION[0] == kind
        pg = tr.model.proposal_generator
        calls = []
        orig_cls_call = type(pg).__call__

        def patched(self, *a, **k):
            out = orig_cls_call(self, *a, **k)
            if self is pg:
                calls.append(out[0])
            return out
        type(pg).__call__ = patched
        orig_joint = pg.forward_joint_begin

        def joint(image_sizes, features, n_labeled, gt_labeled, **kw):   # the fused student pass: one RPN call for both image sets
            out = orig_joint(image_sizes, features, n_labeled, gt_labeled, **kw)
            calls.extend((out[1].images(0, n_labeled), out[1].images(n_labeled, out[1].n)))
            return out
        pg.forward_joint_begin = joint
        try:
            tr.run_step_full_semisup()
        finally:
            type(pg).__call__ = orig_cls_call
        rec = tr.flush_metrics()
        torch.cuda.synchronize()
        assert len(calls) == 2
        post = int(cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN)

        def props_of(pb):
            out = []
            for i in range(pb.n):
                n = int(pb["count"][i])
                out.append(dict(boxes=pb["boxes"][i][:n].cpu(), scores=pb["objectness_logits"][i][:n].cpu()))
            return out
        props = (props_of(calls[0]), props_of(calls[1]))
        gl = tr._last_pseudo
        pseudo = []
        for i in range(gl.n):
            m = gl["valid"][i].bool()
            pseudo.append(dict(boxes=gl["boxes"][i][m].cpu(), classes=gl["classes"][i][m].long().cpu(), scores=gl["scores"][i][m].cpu(),
                               pred_boxes_std=gl["pred_boxes_std"][i][m].cpu()))
        assert sum(len(p["boxes"]) for p in pseudo) > 0

        def compact(name):
            return [(lambda i: lambda nprop, ngt: torch.cat((K[name][i, :nprop], K[name][i, post:post + ngt])))(i)
                    for i in range(K[name].shape[0])]
        keys = dict(rpn_sup=K["rpn_sup"], rpn_unsup=K["rpn_unsup"], roi_sup=compact("roi_sup"), roi_unsup=compact("roi_unsup"))
        S = cfg.SEMISUPNET
        rec_o, _, new_t, _, _ = O.rcnn_semisup_step(
            sd_s, sd_t, orac, keys, keep_rate=S.EMA_KEEP_RATE, lam_u=S.UNSUP_LOSS_WEIGHT, lam_r=S.UNSUP_REG_LOSS_WEIGHT,
            thr=S.BBOX_THRESHOLD, lr=float(d["lr"]), mean=mean, pix_std=pstd, pseudo_override=pseudo, props_override=props)
        t_sd = O.ema_update(sd_s, sd_t, S.EMA_KEEP_RATE)
        with torch.no_grad():
            own, _ = O.rcnn_teacher(t_sd, [x["image"] for x in orac[3]], mean, pstd, thr=S.BBOX_THRESHOLD)
    finally:
        O.CONV_ROUND[0] = None
        ops.set_precision("fp32")
    tot, hit = 0, 0
    for p, q in zip(own, pseudo):
        tot += max(len(p["boxes"]), len(q["boxes"]))
        if len(p["boxes"]) and len(q["boxes"]):
            hit += int((O.pairwise_iou(p["boxes"], q["boxes"]).max(dim=1)[0] > 0.9).sum())
    assert tot > 0 and hit >= 0.6 * tot, (hit, tot)
    for k, v in rec_o.items():
        assert abs(rec[k] - v) <= tol * max(abs(v), 1e-6), (k, rec[k], v)
    t_after = cpu_state(tr.model_teacher)
    for k in new_t:
        assert torch.equal(t_after[k], new_t[k]), k
    if kind == "fp16":
        assert tr._amp_state.cpu().tolist() == [65536.0, 0.0, 1.0]   # finite gradients: step applied, flag cleared


def test_rcnn_step_is_bit_deterministic():
    """Two identical Faster-RCNN UTv2 steps (same weights, batch and sampling keys) end in bit-identical students and teachers: every
    kernel on the path is deterministic now that the RoIAlign backward is a gather (it was an fp32 atomic scatter) - what data-parallel
    replicas rely on to stay in lock step."""
    states = []
    for _ in range(2):
        d, cfg, tr, orac, sd_s, sd_t, K, mean, pstd = _golden_setup(amp=True)
        try:
            tr.run_step_full_semisup()
            torch.cuda.synchronize()
            states.append((tr.model.flat_state().clone(), tr.model_teacher.flat_state().clone(), tr.model.store.grad.clone()))
        finally:
            from ubteacher import ops
            ops.set_precision("fp32")
    assert torch.equal(states[0][2], states[1][2])      # gradients
    assert torch.equal(states[0][0], states[1][0]) and torch.equal(states[0][1], states[1][1])


def test_rcnn_step_fp32_tight_with_the_product_pseudo_boxes(separable_roi_align):
    """The two pseudo RPN terms are compared at 5e-3 / 2e-2 above because the product's and the oracle's teachers emit pseudo boxes that
    differ by ~1e-5, which flips exact-equality low-quality matches and near-tied arg-max IoUs (an ill-conditioned SELECTION).  With the
    selection decoupled - the oracle is handed the product's pseudo boxes, as the AMP test does - the ARITHMETIC of every loss,
    including loss_rpn_cls_pseudo and loss_rpn_loc_pseudo, must hold the north-star 1e-3 in fp32, and the student after SGD tightens too."""
    d, cfg, tr, orac, sd_s, sd_t, K, mean, pstd = _golden_setup()
    tr.run_step_full_semisup()
    rec = tr.flush_metrics()
    torch.cuda.synchronize()
    post = int(cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN)
    gl = tr._last_pseudo
    pseudo = []
    for i in range(gl.n):
        m = gl["valid"][i].bool()
        pseudo.append(dict(boxes=gl["boxes"][i][m].cpu(), classes=gl["classes"][i][m].long().cpu(), scores=gl["scores"][i][m].cpu(),
                           pred_boxes_std=gl["pred_boxes_std"][i][m].cpu()))
    assert sum(len(p["boxes"]) for p in pseudo) > 0

    def compact(name):
        return [(lambda i: lambda nprop, ngt: torch.cat((K[name][i, :nprop], K[name][i, post:post + ngt])))(i)
                for i in range(K[name].shape[0])]
    keys = dict(rpn_sup=K["rpn_sup"], rpn_unsup=K["rpn_unsup"], roi_sup=compact("roi_sup"), roi_unsup=compact("roi_unsup"))
    S = cfg.SEMISUPNET
    rec_o, new_s, new_t, _, _ = O.rcnn_semisup_step(
        sd_s, sd_t, orac, keys, keep_rate=S.EMA_KEEP_RATE, lam_u=S.UNSUP_LOSS_WEIGHT, lam_r=S.UNSUP_REG_LOSS_WEIGHT,
        thr=S.BBOX_THRESHOLD, lr=float(d["lr"]), mean=mean, pix_std=pstd, pseudo_override=pseudo)
    assert rec_o["loss_rpn_cls_pseudo"] > 0 and rec_o["loss_rpn_loc_pseudo"] > 0
    for k, v in rec_o.items():
        assert abs(rec[k] - v) <= 1e-3 * max(abs(v), 1e-6), (k, rec[k], v)
    t_after = cpu_state(tr.model_teacher)
    for k in new_t:
        assert torch.equal(t_after[k], new_t[k]), k
    s_after = cpu_state(tr.model)
    for k in new_s:
        err = float((s_after[k].double() - new_s[k].double()).abs().max())
        upd = float((new_s[k].double() - sd_s[k].double()).abs().max())
        assert err <= 1e-4 * float(new_s[k].abs().max()) + 1e-2 * upd + 1e-12, (k, err, upd)


def test_rcnn_step_with_trainable_stem_runs_and_updates_it(monkeypatch):
    """MODEL.BACKBONE.FREEZE_AT 0 through the Faster-RCNN trainer (fp32 and AMP): the step runs, the stem and res2 weights move, nothing
    is left parked or non-finite (the FCOS trainer's parity test pins the stem's gradient against the oracle)."""
    monkeypatch.setenv("UTV2_PRECISION", "bf16")   # written for bf16 rounding (the package default AMP type is fp16, the reference's)
    from ubteacher import ops
    from ubteacher.engine import UBRCNNTeacherTrainer
    try:
        for amp in (False, True):
            cfg = rcnn_cfg()
            cfg.MODEL.BACKBONE.FREEZE_AT = 0
            cfg.SOLVER.AMP.ENABLED = amp
            torch.manual_seed(0)
            prod, orac = make_batch(31, 2, 2, H, W, "cuda")
            tr = UBRCNNTeacherTrainer(cfg, data_loader=FixedLoader(prod))
            before = cpu_state(tr.model)
            tr.iter = 1
            tr.optimizer.param_groups[0]["lr"] = 0.01
            tr.run_step_full_semisup()
            torch.cuda.synchronize()
            after = cpu_state(tr.model)
            for k in ("backbone.bottom_up.stem.conv1.weight", "backbone.bottom_up.res2.1.conv1.weight"):
                d = (after[k] - before[k]).abs()
                assert torch.isfinite(after[k]).all() and float(d.max()) > 0, k
    finally:
        ops.set_precision("fp32")


def test_rcnn_eval_detections_vs_reference_golden():
    """Faster-RCNN test-mode inference of the PRODUCT (Trainer.test -> inference_on_dataset -> eval-mode two-stage model -> RPN with the
    *_TEST top-k -> box head -> predictor inference -> detector_postprocess) against the golden produced by executing the reference's
    own eval-mode call chain (tests/golden/gen_golden_eval.py::gen_rcnn_eval; meta_arch/rcnn.py:8-13, proposal_generator/rpn.py:21-76,
    roi_heads/roi_heads.py:75-139, roi_heads/fast_rcnn.py:1094-1125): a two-image ragged batch rescaled to original sizes; kept detections
    identical in class and order, scores 1e-3, boxes 1e-3 of the image size, pred_boxes_std 1e-3."""
    from tests.test_step_golden import _rcnn_eval_golden, rcnn_eval_golden_state
    from ubteacher.engine import UBRCNNTeacherTrainer
    from ubteacher.evaluation import inference_on_dataset
    d = _rcnn_eval_golden()
    _, sd = rcnn_eval_golden_state(d)
    cfg = rcnn_cfg()
    assert cfg.MODEL.RPN.PRE_NMS_TOPK_TEST == int(d["pre_topk"]) and cfg.MODEL.RPN.POST_NMS_TOPK_TEST == int(d["post_topk"])
    torch.manual_seed(0)
    prod, _ = make_batch(31, 2, 2, H, W, "cuda")
    tr = UBRCNNTeacherTrainer(cfg, data_loader=FixedLoader(prod))
    tr.model_teacher.load_state_dict(sd)
    batch = []
    for i in range(2):
        oh, ow = [int(x) for x in d["orig%d" % i]]
        batch.append({"image": torch.from_numpy(d["img%d" % i]).cuda(), "height": oh, "width": ow, "image_id": i})

    class Capture:
        def reset(self):
            self.out = []

        def process(self, inputs, outputs):
            self.out.extend(outputs)

        def evaluate(self):
            return {}
    ev = Capture()
    inference_on_dataset(tr.model_teacher, [batch], ev, cfg)
    assert len(ev.out) == 2
    for i, r in enumerate(ev.out):
        x = r["instances"]
        oh, ow = [int(v) for v in d["orig%d" % i]]
        assert tuple(x.image_size) == (oh, ow)
        cls, sc, bx, sd_ = d["classes%d" % i], d["scores%d" % i], d["boxes%d" % i], d["std%d" % i]
        assert len(x) == len(cls) and len(cls) > 0, (len(x), len(cls))
        assert np.array_equal(x.pred_classes.long().cpu().numpy(), cls)
        np.testing.assert_allclose(x.scores.cpu().numpy(), sc, rtol=1e-3)
        np.testing.assert_allclose(x.pred_boxes.tensor.cpu().numpy(), bx, rtol=0, atol=1e-3 * max(oh, ow))
        np.testing.assert_allclose(x.pred_boxes_std.cpu().numpy(), sd_, rtol=1e-3, atol=1e-4)
