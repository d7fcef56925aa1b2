"""Golden vectors of the FCOS evaluation path: the reference's OWN eval-mode `OneStageDetector.forward` executed here (CPU, build
container only), SURVEY 8(f) rank 3.

The reference's `inference_on_dataset` (evaluation/evaluator.py:14-104) switches the model to eval mode and calls
`model(inputs, nms_method=cfg.MODEL.FCOS.NMS_CRITERIA_TEST)`; `OneStageDetector.forward` (one_stage_detector.py:230-240) falls through to
`PseudoProposalNetwork.forward` (:70-145): backbone -> `FCOS.forward` in eval mode (fcos/fcos.py: `predict_proposals` with the `*_TEST`
thresholds, fcos_outputs.py:1058-1065) -> `detector_postprocess` per image to the dict's `height` / `width` (:16-43,136-145) ->
`{"instances": ...}`.  All of that is the reference's code, imported in place; the Detectron2 backbone is the oracle's functional
ResNet-50 + FPN (as in gen_golden_step.py) and Detectron2's own `detector_postprocess` [D2-recall: boxes scaled by
(out_w / in_w, out_h / in_h), clipped to the output size, empty boxes dropped] is a stand-in written here.

Stored (tests/golden/fcos_eval.npz): the input images (two different sizes - ImageList padding - with original sizes that differ
from the network input), per image the returned pred_boxes / scores / pred_classes, for two configs: the shipped defaults and a
variant whose *_TEST thresholds differ from the *_TRAIN ones (proves the eval path reads *_TEST).  Initial weights: the product's
CPU initialisation under the stored seed + the cls_logits rescale of the parity tests.

    python tests/golden/gen_golden_eval.py
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import gen_golden as G  # noqa: E402
import gen_golden_step as S  # noqa: E402
from oracle import utv2_oracle as O  # noqa: E402

VARIANTS = {
    "default": [],
    "testth": ["MODEL.FCOS.INFERENCE_TH_TEST", 0.2, "MODEL.FCOS.PRE_NMS_TOPK_TEST", 60, "MODEL.FCOS.POST_NMS_TOPK_TEST", 12,
               "MODEL.FCOS.NMS_CRITERIA_TEST", "cls_n_loc"],
}


def d2_detector_postprocess(structures):
    def post(results, output_height, output_width, mask_threshold=0.5):
        sx, sy = output_width / results.image_size[1], output_height / results.image_size[0]
        out = structures.Instances((output_height, output_width))
        for k, v in results.get_fields().items():
            out.set(k, v)
        b = out.pred_boxes.tensor.clone()
        b[:, 0::2] *= sx
        b[:, 1::2] *= sy
        b[:, 0::2] = b[:, 0::2].clamp(min=0, max=output_width)
        b[:, 1::2] = b[:, 1::2].clamp(min=0, max=output_height)
        out.pred_boxes = structures.Boxes(b)
        keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
        return out[keep]
    return post


def main():
    structures, fo, pg, tr = G.install_shims()
    fcos_mod, osd, _ = S.load_ref_fcos_modules()
    osd.d2_postprocesss = d2_detector_postprocess(structures)
    cfg, sd0 = S.product_cfg_and_state("fcos", seed=0)
    g = torch.Generator().manual_seed(41)

    def img(H, W):
        base = torch.rand(3, H // 8 + 1, W // 8 + 1, generator=g)
        im = torch.nn.functional.interpolate(base[None], size=(H, W), mode="bilinear", align_corners=False)[0]
        return (im * 255 + torch.randn(3, H, W, generator=g) * 20).clamp(0, 255).to(torch.uint8)
    sizes = [(96, 128), (80, 112)]
    origs = [(144, 192), (100, 140)]        # height / width the detections are rescaled to (1.5x, 1.25x)
    images = [img(*s) for s in sizes]
    mean, pstd = sd0["pixel_mean"], sd0["pixel_std"]
    p = "proposal_generator.fcos_head.cls_logits"
    gw = torch.Generator().manual_seed(0)
    sd = OrderedDict(sd0)
    sd[p + ".weight"] = torch.randn(sd0[p + ".weight"].shape, generator=gw) * 0.01
    sd[p + ".bias"] = torch.zeros_like(sd0[p + ".bias"])
    with torch.no_grad():
        s = torch.cat([x.reshape(-1) for x in O.fcos_forward(sd, images, mean, pstd)[0]]).std().item()
    sd[p + ".weight"] = sd[p + ".weight"] * (1.5 / max(s, 1e-12))
    sd[p + ".bias"] = torch.full_like(sd[p + ".bias"], -3.0)
    d = {"seed_state": 0, "cls_scale": 1.5 / max(s, 1e-12), "cls_bias": -3.0}
    S.state_fingerprints("init", sd0, d)
    for i, (im, (oh, ow)) in enumerate(zip(images, origs)):
        d["img%d" % i] = im.numpy()
        d["orig%d" % i] = np.array([oh, ow])
    for name, over in VARIANTS.items():
        c = cfg.clone()
        c.defrost()
        c.merge_from_list(over)
        model = S.build_ref_one_stage(c, sd, fcos_mod, osd)
        model.eval()
        batch = []
        for im, (oh, ow) in zip(images, origs):
            inst = structures.Instances((oh, ow))
            inst.gt_boxes = structures.Boxes(torch.tensor([[4.0, 5.0, 60.0, 70.0]]))
            inst.gt_classes = torch.tensor([3])
            batch.append({"image": im, "height": oh, "width": ow, "instances": inst})
        with torch.no_grad():
            out = model(batch, nms_method=c.MODEL.FCOS.NMS_CRITERIA_TEST)
        for i, r in enumerate(out):
            x = r["instances"]
            d["%s_boxes%d" % (name, i)] = G.npy(x.pred_boxes.tensor)
            d["%s_scores%d" % (name, i)] = G.npy(x.scores)
            d["%s_classes%d" % (name, i)] = G.npy(x.pred_classes)
            d["%s_size%d" % (name, i)] = np.array(x.image_size)
            print(name, i, "detections:", len(x), "image_size", x.image_size, "top score %.4f" % float(x.scores.max()) if len(x) else "")
            assert len(x) > 0
        d[name + "_overrides"] = np.array([str(v) for v in over])
    np.savez_compressed(os.path.join(HERE, "fcos_eval.npz"), **d)
    print("fcos_eval.npz:", len(d), "arrays")


if __name__ == "__main__":
    main()
