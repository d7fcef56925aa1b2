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
