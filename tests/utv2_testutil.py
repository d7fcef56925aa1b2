"""Shared helpers for the end-to-end parity tests / smoke (small seeded UTv2 FCOS problems)."""
import numpy as np
import torch

from oracle import utv2_oracle as O


def small_fcos_cfg(bl=2, bu=2, device="cuda"):
    from ubteacher.presets import get_config
    return get_config("fcos", 1, ["SOLVER.IMG_PER_BATCH_LABEL", bl, "SOLVER.IMG_PER_BATCH_UNLABEL", bu,
                                  "SEMISUPNET.BURN_UP_STEP", 0, "SOLVER.AMP.ENABLED", False, "MODEL.DEVICE", device])


def make_batch(seed, bl, bu, H, W, device, sizes=None, empty_gt=()):
    """(loader-style batch for the product, same batch for the oracle).
    sizes: optional per-sample (H, W) for the bl labeled then the bu unlabeled samples (ragged batches: both views of a
    sample share its size); empty_gt: indices of labeled samples that get no ground-truth boxes."""
    from ubteacher.d2.structures import Boxes, Instances
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)

    cur = [H, W]

    def img():
        # smooth-ish random image so the backbone sees structure, uint8 BGR CHW
        H, W = cur
        base = torch.rand(3, H // 8 + 1, W // 8 + 1, generator=g)
        im = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear", align_corners=False)[0]
        im = (im * 255 + torch.randn(3, H, W, generator=g) * 20).clamp(0, 255)
        return im.to(torch.uint8)

    def gt():
        H, W = cur
        G = int(rng.integers(1, 5))
        cx, cy = rng.uniform(0, W, G), rng.uniform(0, H, G)
        bw, bh = np.exp(rng.uniform(2.5, 4.5, G)), np.exp(rng.uniform(2.5, 4.5, G))
        b = np.stack([np.clip(cx - bw / 2, 0, W - 4), np.clip(cy - bh / 2, 0, H - 4), np.clip(cx + bw / 2, 4, W), np.clip(cy + bh / 2, 4, H)], 1)
        return torch.tensor(b, dtype=torch.float32), torch.from_numpy(rng.integers(0, 80, G)).long()

    prod = ([], [], [], [])
    orac = ([], [], [], [])
    for i in range(bl):
        if sizes is not None:
            cur[:] = sizes[i]
        wk, st = img(), img()
        boxes, classes = gt()
        if i in empty_gt:
            boxes, classes = boxes[:0], classes[:0]
        for dst_p, dst_o, im in ((prod[1], orac[1], wk), (prod[0], orac[0], st)):
            H, W = cur
            inst = Instances((H, W))
            inst.gt_boxes = Boxes(boxes.clone())
            inst.gt_classes = classes.clone()
            dst_p.append({"image": im.to(device), "height": H, "width": W, "instances": inst})
            dst_o.append({"image": im, "gt": dict(boxes=boxes, classes=classes)})
    for i in range(bu):
        if sizes is not None:
            cur[:] = sizes[bl + i]
        H, W = cur
        wk, st = img(), img()
        prod[3].append({"image": wk.to(device), "height": H, "width": W}); orac[3].append({"image": wk})
        prod[2].append({"image": st.to(device), "height": H, "width": W}); orac[2].append({"image": st})
    return prod, orac


class FixedLoader:
    def __init__(self, batch):
        self.batch = batch

    def __iter__(self):
        return self

    def __next__(self):
        return tuple([dict(d) for d in part] for part in self.batch)


def cpu_state(model):
    return {k: v.detach().cpu().clone().contiguous() for k, v in model.state_dict().items()}


def tune_state_for_pseudo_labels(sd, images, target_std=1.5, bias=-4.5, seed=0):
    """Rescale cls_logits so the (random-init) teacher emits a handful of confident, well separated
    detections (otherwise the unsupervised branch degenerates to zero losses - SURVEY 8d)."""
    p = "proposal_generator.fcos_head.cls_logits"
    g = torch.Generator().manual_seed(seed)
    sd = dict(sd)
    sd[p + ".weight"] = torch.randn(sd[p + ".weight"].shape, generator=g) * 0.01
    sd[p + ".bias"] = torch.zeros_like(sd[p + ".bias"])
    with torch.no_grad():
        logits = O.fcos_forward(sd, images, torch.tensor([103.53, 116.28, 123.675]).view(3, 1, 1), torch.ones(3, 1, 1))[0]
        s = torch.cat([x.reshape(-1) for x in logits]).std().item()
    sd[p + ".weight"] = sd[p + ".weight"] * (target_std / max(s, 1e-12))
    sd[p + ".bias"] = torch.full_like(sd[p + ".bias"], bias)
    return sd


def rcnn_tune(sd, images, mean, pstd, seed=0):
    """Random-init R50 features are not normalised (no pretrained BN statistics), so the head outputs
    explode; rescale the prediction layers (data-driven, via the oracle forward) so the detector emits a
    few confident, well separated, non-degenerate detections."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    sd = dict(sd)
    q = "proposal_generator.rpn_head."
    p = "roi_heads.box_predictor."
    with torch.no_grad():
        feats, sizes = O.rcnn_backbone(sd, images, mean, pstd)
        fl = [feats[k] for k in ("p2", "p3", "p4", "p5", "p6")]
        t = torch.cat([F.relu(F.conv2d(f, sd[q + "conv.weight"], sd[q + "conv.bias"], 1, 1)).permute(0, 2, 3, 1).reshape(-1, 256) for f in fl])
        s_t = t.std().item()
        sd[q + "objectness_logits.weight"] = torch.randn(3, 256, 1, 1, generator=g) * (1.0 / (s_t * 16))
        sd[q + "anchor_deltas.weight"] = torch.randn(12, 256, 1, 1, generator=g) * (0.1 / (s_t * 16))
        hw = [(f.shape[2], f.shape[3]) for f in fl]
        anchors = O.make_anchors(hw, [4, 8, 16, 32, 64])
        obj, dl = O.rpn_head(sd, fl)
        props = O.find_top_rpn_proposals(anchors, obj, dl, sizes, 2000, 1000)
        x = O.roi_pool(fl[:4], [pp["boxes"] for pp in props]).flatten(1)
        x = F.relu(F.linear(x, sd["roi_heads.box_head.fc1.weight"], sd["roi_heads.box_head.fc1.bias"]))
        x = F.relu(F.linear(x, sd["roi_heads.box_head.fc2.weight"], sd["roi_heads.box_head.fc2.bias"]))
        s_x = x.std().item()
    sd[p + "cls_score.weight"] = torch.randn(81, 1024, generator=g) * (2.5 / (s_x * 32))
    b = torch.zeros(81); b[80] = 3.0
    sd[p + "cls_score.bias"] = b
    nb = sd[p + "bbox_pred.weight"].shape[0]          # 4 (boundary-variance predictors) or 4 per class (the UTv1 predictor)
    sd[p + "bbox_pred.weight"] = torch.randn(nb, 1024, generator=g) * (0.5 / (s_x * 32))
    if p + "bbox_pred_std.weight" in sd:
        sd[p + "bbox_pred_std.weight"] = torch.randn(4, 1024, generator=g) * (0.5 / (s_x * 32))
    return sd


# ---- step-level goldens (tests/golden/step_*.npz: the reference's own run_step_full_semisup, gen_golden_step.py) ----------------
def load_step_golden(kind):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_%s.npz" % kind), allow_pickle=False)


def state_fingerprint(t):
    a = t.detach().cpu().contiguous().double().numpy().reshape(-1)
    head = np.zeros(8)
    head[:min(8, a.size)] = a[:8]
    return np.concatenate([[a.sum(), np.abs(a).sum()], head])


def golden_init_state(kind, d):
    """The initial weights the golden step started from: the product's CPU initialisation under the stored seed, verified
    against the stored per-tensor fingerprints (a torch RNG change fails here, loudly, not as a loss mismatch)."""
    from ubteacher.modeling import build_model
    from ubteacher.presets import get_config
    cfg = get_config(kind, 1, ["SOLVER.IMG_PER_BATCH_LABEL", 2, "SOLVER.IMG_PER_BATCH_UNLABEL", 2, "SEMISUPNET.BURN_UP_STEP", 0,
                               "SOLVER.AMP.ENABLED", False, "MODEL.DEVICE", "cpu"])
    torch.manual_seed(int(d["seed_state"]))
    model = build_model(cfg)
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items()}
    keys = [str(k) for k in d["init_keys"]]
    assert keys == [k for k in sd if sd[k].dtype.is_floating_point], "state-dict surface differs from the golden's"
    fp = np.stack([state_fingerprint(sd[k]) for k in keys])
    assert np.array_equal(fp, d["init_fp"]), "CPU initialisation is not the one the golden step started from"
    return cfg, sd


def golden_batches(d, device):
    """(loader-style batch for the product, oracle batch) from the arrays stored in a step golden"""
    from ubteacher.d2.structures import Boxes, Instances
    H, W = int(d["H"]), int(d["W"])
    prod, orac = ([], [], [], []), ([], [], [], [])
    i = 0
    while "lab%d_weak" % i in d:
        boxes, classes = torch.from_numpy(d["lab%d_boxes" % i]), torch.from_numpy(d["lab%d_classes" % i])
        for slot, view in ((1, "weak"), (0, "strong")):
            im = torch.from_numpy(d["lab%d_%s" % (i, view)])
            inst = Instances((H, W))
            inst.gt_boxes = Boxes(boxes.clone().to(device))
            inst.gt_classes = classes.clone().to(device)
            prod[slot].append({"image": im.to(device), "height": H, "width": W, "instances": inst})
            orac[slot].append({"image": im, "gt": dict(boxes=boxes, classes=classes)})
        i += 1
    i = 0
    while "unl%d_weak" % i in d:
        for slot, view in ((3, "weak"), (2, "strong")):
            im = torch.from_numpy(d["unl%d_%s" % (i, view)])
            prod[slot].append({"image": im.to(device), "height": H, "width": W})
            orac[slot].append({"image": im})
        i += 1
    return prod, orac


def golden_record(d):
    return {k[4:]: float(d[k]) for k in d.files if k.startswith("rec_")}


def check_state_fingerprints(d, prefix, sd, rtol, exact=False, rtol_update=0.0):
    """per-tensor (sum, |sum|, first 8 values) of `sd` against the golden's.  Tolerance per tensor: rtol x its magnitude plus
    rtol_update x the size of the step it took in the golden (sum |after - before|; zero-initialised biases ARE their update)."""
    keys = [str(k) for k in d[prefix + "_keys"]]
    ref = d[prefix + "_fp"]
    upd = d[prefix + "_upd"] if (prefix + "_upd") in d.files else np.zeros(len(keys))
    for k, r, u in zip(keys, ref, upd):
        if k not in sd:
            assert "integral" in k, k
            continue
        f = state_fingerprint(sd[k])
        if exact:
            assert np.array_equal(f, r), (prefix, k)
            continue
        n = max(sd[k].numel(), 1)
        tol = rtol * r[1] + rtol_update * u + 1e-12
        assert abs(f[0] - r[0]) <= tol, (prefix, k, "sum", f[0], r[0], tol)
        assert abs(f[1] - r[1]) <= tol, (prefix, k, "abs", f[1], r[1], tol)
        assert np.all(np.abs(f[2:] - r[2:]) <= 50 * tol / n + 1e-9), (prefix, k, "head", f[2:], r[2:])
