"""Pseudo-label generation for FCOS (reference ubteacher/modeling/pseudo_generator.py:7-131).

Quirk kept: this object owns its own FCOSOutputs that is never put in eval mode, so
`nms_from_dense` reads the *_TRAIN thresholds while the eval-mode teacher reads *_TEST.
"""
from .fcos import FCOSOutputs, METHODS


class PseudoGenerator:
    def __init__(self, cfg):
        self.fcos_output = FCOSOutputs(cfg)

    def nms_from_dense(self, raw_output, nms_method):
        assert nms_method in METHODS
        if dict.__contains__(raw_output, "head_out"):   # the product's own raw output: the fused level-first buffers ride along
            return self.fcos_output.predict_proposals(raw_output["head_out"], raw_output["level_hw"],
                                                      raw_output["image_sizes"], nms_method)
        # a reference-style dict of per-level NCHW tensors (pseudo_generator.py:11-36): rebuild the fused buffers (copies)
        import torch
        from .. import ops
        logits, reg, ctr = raw_output["logits_pred"], raw_output["reg_pred"], raw_output["ctrness_pred"]
        std = raw_output.get("reg_pred_std")
        level_hw = [tuple(t.shape[-2:]) for t in logits]
        N = logits[0].shape[0]
        meta = ops.LevelMeta(N, level_hw)

        def rows(ts):
            return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts], dim=0).float()
        R = reg[0].shape[1]
        box = torch.zeros((meta.P, 80), device=logits[0].device)
        box[:, :R] = rows(reg)
        if std is not None:
            box[:, R:R + 4] = rows(std)
        box[:, R + 4:R + 5] = rows(ctr)
        head_out = {"logits": rows(logits).contiguous(), "box": box, "meta": meta}
        return self.fcos_output.predict_proposals(head_out, level_hw, raw_output["image_sizes"], nms_method)

    def process_pseudo_label(self, proposals, cur_threshold, proposal_type, psedo_label_method=""):
        """Returns (thresholded padded boxes, mean #boxes as a device scalar - no host sync)."""
        if proposal_type != "roih":
            raise ValueError("FCOS pseudo labels are 'roih' detections")
        if psedo_label_method == "thresholding":
            out = proposals.threshold(cur_threshold)
        elif psedo_label_method == "thresholding_cls_ctr":
            out = proposals.threshold(cur_threshold[0], cur_threshold[1])
        else:
            raise ValueError("Unkown pseudo label boxes methods")
        num = out["valid"].float().sum() / max(out.n, 1)
        # as ground truth for the student: rename to the gt_* convention expected by the targets kernel
        return out, num
