"""`PseudoProposalNetwork` / `OneStageDetector` meta-architectures (reference
ubteacher/modeling/one_stage_detector.py:46-240): same registry names, same forward signature
`model(batched_inputs, output_raw=False, nms_method="cls_n_ctr", ignore_near=False, branch=...)`
and the same return shapes per branch; numerics on the HIP path."""
import torch

from .. import hip, ops
from ..d2.registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY
from ..d2.structures import Instances
from .arena_model import ArenaModel
from .fcos import PaddedBoxes


def detector_postprocess(results, output_height, output_width):
    """D2 detector_postprocess for boxes: rescale to the requested size, clip, drop empty."""
    sx = output_width / results.image_size[1]
    sy = output_height / results.image_size[0]
    out = Instances((output_height, output_width), **results.get_fields())
    boxes = out.pred_boxes.clone()
    boxes.tensor = boxes.tensor * torch.tensor([sx, sy, sx, sy], device=boxes.tensor.device)
    boxes.clip(out.image_size)
    out.pred_boxes = boxes
    return out[boxes.nonempty()]


@META_ARCH_REGISTRY.register()
class PseudoProposalNetwork(ArenaModel):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg, self.store, self.folder)
        self.proposal_generator = PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(
            cfg, self.store, self.backbone.out_channels)
        # persistent buffers in the reference (one_stage_detector.py:59-64) => part of state_dict / EMA
        mean, std = list(cfg.MODEL.PIXEL_MEAN), list(cfg.MODEL.PIXEL_STD)
        self.pixel_mean = self.store.new((3, 1, 1), "buffer", lambda t: t.copy_(torch.tensor(mean).view(3, 1, 1))).export("pixel_mean")
        self.pixel_std = self.store.new((3, 1, 1), "buffer", lambda t: t.copy_(torch.tensor(std).view(3, 1, 1))).export("pixel_std")
        self._mean_host, self._std_host = mean, std
        self._finalize()

    def _children(self):
        return [self.proposal_generator]

    def _features(self, batched_inputs):
        images = [x["image"].to(self.device) for x in batched_inputs]
        # (x - pixel_mean) / pixel_std + pad to size_divisibility, fused into the NHWC4 conversion.
        # Host copies of mean/std are used (the device buffers only change by EMA rounding).
        x4, image_sizes = hip.preprocess_images(images, self._mean_host, self._std_host, self.backbone.size_divisibility,
                                                 bf16_stem=ops.amp() and not self.backbone.bottom_up.stem.trainable)
        self.folder.fold()
        return self.backbone(x4), image_sizes

    def _gt(self, batched_inputs, key):
        first = batched_inputs[0][key]
        if isinstance(first, PaddedBoxes):
            return first  # batched pseudo labels attached once to the first datum (device resident)
        return PaddedBoxes.from_instances([x[key] for x in batched_inputs], self.device)

    def forward(self, batched_inputs, output_raw=False, nms_method="cls_n_ctr", ignore_near=False, branch="labeled"):
        features, image_sizes = self._features(batched_inputs)
        if "instances" in batched_inputs[0] and branch != "teacher_weak":
            gt = self._gt(batched_inputs, "instances")
        else:
            gt = None
        out = self.proposal_generator(image_sizes, features, gt, output_raw=output_raw, nms_method=nms_method,
                                      ignore_near=ignore_near)
        if output_raw:
            proposals, proposal_losses, raw_pred = out
        else:
            proposals, proposal_losses = out
        if self.training:
            return (proposal_losses, raw_pred) if output_raw else proposal_losses
        if output_raw:
            return proposals, raw_pred  # raw: not rescaled
        processed = []
        for res, inp, size in zip(proposals.to_instances(), batched_inputs, image_sizes):
            h, w = inp.get("height", size[0]), inp.get("width", size[1])
            processed.append({"proposals": detector_postprocess(res, h, w)})
        return processed

    __call__ = forward

    def padded_canvas(self, batched_inputs):
        """(H, W) the batch is zero-padded to by preprocess_image (ImageList.from_tensors with size_divisibility)."""
        d = self.backbone.size_divisibility
        h = max(int(x["image"].shape[1]) for x in batched_inputs)
        w = max(int(x["image"].shape[2]) for x in batched_inputs)
        return ((h + d - 1) // d * d, (w + d - 1) // d * d) if d > 1 else (h, w)


@META_ARCH_REGISTRY.register()
class OneStageDetector(PseudoProposalNetwork):
    def forward_joint(self, labeled_inputs, unlabeled_inputs, loss_weights=None):
        """model(labeled, branch="labeled") and model(unlabeled, branch="unlabeled") of one iteration
        (reference trainer.py:396-411) as ONE forward over the concatenated batch; returns the two loss dicts.
        Only valid when padded_canvas() of the two lists coincide (the caller checks)."""
        assert self.training
        both = list(labeled_inputs) + list(unlabeled_inputs)
        features, image_sizes = self._features(both)
        gt_l = self._gt(labeled_inputs, "instances")
        gt_u = {"cls": self._gt(unlabeled_inputs, "instances_class"), "reg": self._gt(unlabeled_inputs, "instances_reg")}
        return self.proposal_generator.forward_joint(image_sizes, features, len(labeled_inputs), gt_l, gt_u, loss_weights)

    def forward_joint_begin(self, labeled_inputs, unlabeled_inputs):
        """first half of forward_joint: everything that does not need the pseudo labels (backbone, FPN, towers, prediction convs of the
        concatenated batch) - the trainer runs it while the teacher is still producing them on another stream"""
        assert self.training
        both = list(labeled_inputs) + list(unlabeled_inputs)
        features, image_sizes = self._features(both)
        gt_l = self._gt(labeled_inputs, "instances")
        return self.proposal_generator.forward_joint_begin(image_sizes, features, len(labeled_inputs), gt_l)

    def forward_joint_finish(self, ctx, unlabeled_inputs, loss_weights=None):
        """second half: the loss kernels, given the unlabeled images' pseudo labels"""
        gt_u = {"cls": self._gt(unlabeled_inputs, "instances_class"), "reg": self._gt(unlabeled_inputs, "instances_reg")}
        return self.proposal_generator.forward_joint_finish(ctx, gt_u, loss_weights)

    def forward(self, batched_inputs, output_raw=False, nms_method="cls_n_ctr", ignore_near=False, branch="labeled"):
        if self.training:
            features, image_sizes = self._features(batched_inputs)
            b0 = batched_inputs[0]
            if "instances_class" in b0 and "instances_reg" in b0:
                gt = {"cls": self._gt(batched_inputs, "instances_class"), "reg": self._gt(batched_inputs, "instances_reg")}
            elif "instances" in b0 and branch != "teacher_weak":
                gt = self._gt(batched_inputs, "instances")
            else:
                gt = None
            out = self.proposal_generator(image_sizes, features, gt, output_raw=output_raw, ignore_near=ignore_near,
                                          branch=branch)
            if output_raw:
                proposals, proposal_losses, raw_pred = out
                return proposal_losses, raw_pred, proposals
            return out[1]
        if output_raw:
            return PseudoProposalNetwork.forward(self, batched_inputs, output_raw=True, nms_method=nms_method, branch=branch)
        res = PseudoProposalNetwork.forward(self, batched_inputs, output_raw=False, nms_method=nms_method, branch=branch)
        return [{"instances": r["proposals"]} for r in res]

    __call__ = forward
