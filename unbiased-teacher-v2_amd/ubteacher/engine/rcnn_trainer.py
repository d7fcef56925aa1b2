"""Faster-RCNN UTv2 trainer (reference engine/trainer.py:612-1023) on the arena/HIP machinery."""
import os
import time

import torch

from ..modeling.fcos import PaddedBoxes


def _make(base):
    class UBRCNNTeacherTrainer(base):
        def __init__(self, cfg, data_loader=None):
            self.model = self.build_model(cfg)
            self.optimizer = self.build_optimizer(cfg, self.model)
            self.model_teacher = self.build_model(cfg)  # stays in train mode
            self.scheduler = self.build_lr_scheduler(cfg, self.optimizer)
            self.fuse_student_passes = os.environ.get("UTV2_FUSE_STUDENT_PASSES", "1") != "0"
            self._common_init(cfg, data_loader)

        # -- pseudo-labelling (trainer.py:727-780) ----------------------------------------------------
        def threshold_bbox(self, proposal_bbox_inst, thres=0.7, proposal_type="roih"):
            if proposal_type != "roih":
                raise ValueError("Error in proposal type.")
            return proposal_bbox_inst.threshold(thres)

        def process_pseudo_label(self, proposals, cur_threshold, proposal_type, psedo_label_method=""):
            if psedo_label_method != "thresholding":
                raise ValueError("Unkown pseudo label boxes methods")
            out = self.threshold_bbox(proposals, thres=cur_threshold, proposal_type=proposal_type)
            return out, out["valid"].float().sum() / max(out.n, 1)

        def remove_label(self, label_data):
            for d in label_data:
                if "instances" in d.keys():
                    del d["instances"]
            return label_data

        def add_label(self, unlabled_data, label):
            if isinstance(label, PaddedBoxes):
                for d in unlabled_data:
                    d["instances"] = label
            else:
                for d, inst in zip(unlabled_data, label):
                    d["instances"] = inst
            return unlabled_data

        def run_step_full_semisup(self):
            cfg = self.cfg
            S = cfg.SEMISUPNET
            assert self.model.training, "[UBTeacherTrainer] model was changed to eval mode!"
            start = time.perf_counter()
            label_data_q, label_data_k, unlabel_data_q, unlabel_data_k = next(self._data_loader_iter)
            data_time = time.perf_counter() - start

            if self.iter < S.BURN_UP_STEP:
                all_label_data = label_data_q + label_data_k if S.USE_SUP_STRONG == "both" else label_data_k
                record_dict, _, _, _ = self.model(all_label_data, branch="supervised")
                losses = sum(v for k, v in record_dict.items() if k[:4] == "loss")
            else:
                if self.iter == S.BURN_UP_STEP:
                    self._update_teacher_model(keep_rate=0.0)
                cur_ema_rate = S.EMA_KEEP_RATE
                if (self.iter - S.BURN_UP_STEP) % S.TEACHER_UPDATE_ITER == 0:
                    self._update_teacher_model(keep_rate=cur_ema_rate)
                record_dict = {"EMA_rate": cur_ema_rate}

                # The supervised student pass does not depend on the pseudo labels: the teacher (forward, RPN top-k + NMS, ROI inference
                # + NMS, thresholding - mostly latency-bound kernels) runs on a side stream next to it (engine/trainer.py does the same
                # for FCOS); the pseudo-labeled pass follows once both are done.  UTV2_OVERLAP_TEACHER=0: one stream (identical results).
                # When the labeled and the unlabeled images pad to the same canvas the two student passes (trainer.py:838-866) are ONE
                # batch (model.forward_joint_*): the part that needs no pseudo labels runs next to the teacher, the rest after it.
                all_label_data = label_data_q + label_data_k if S.USE_SUP_STRONG == "both" else label_data_k
                fuse = self.fuse_student_passes and self.model.padded_canvas(all_label_data) == self.model.padded_canvas(unlabel_data_q)
                overlap = os.environ.get("UTV2_OVERLAP_TEACHER", "1") != "0" and self.model.device.type == "cuda"
                if overlap:
                    main = torch.cuda.current_stream(self.model.device)
                    if getattr(self, "_side_stream", None) is None:
                        self._side_stream = torch.cuda.Stream(self.model.device)
                    side = self._side_stream
                    side.wait_stream(main)           # the EMA update above
                    torch.cuda.set_stream(side)
                try:
                    with torch.no_grad():
                        _, proposals_rpn_unsup_k, proposals_roih_unsup_k, _ = self.model_teacher(unlabel_data_k, branch="unsup_data_weak")
                    pseudo, _ = self.process_pseudo_label(proposals_roih_unsup_k, S.BBOX_THRESHOLD, "roih", "thresholding")
                    # student ground truth fields
                    extra = {"pred_boxes_std": pseudo["pred_boxes_std"]} if "pred_boxes_std" in pseudo else {}   # trainer.py:743-746: when there is one
                    gt = PaddedBoxes(pseudo.image_sizes, boxes=pseudo["boxes"], classes=pseudo["classes"], valid=pseudo["valid"],
                                     scores=pseudo["scores"], **extra)
                    if overlap:
                        for t in gt.f.values():          # allocated on the side stream, consumed on the main one
                            if torch.is_tensor(t):
                                t.record_stream(main)
                finally:
                    if overlap:
                        torch.cuda.set_stream(main)
                self._last_pseudo = gt

                # the trainer's weighting of the loss dict (below) as one weight per key: the fused student pass folds it, the normalisers
                # and the sum into its single scalar-tail launch (UTV2_FUSE_LOSS_TAIL=0: the ATen op chain; identical values)
                lw = None
                if fuse and os.environ.get("UTV2_FUSE_LOSS_TAIL", "1") != "0":
                    lw = {"loss_cls": 1.0, "loss_box_reg": 1.0, "loss_rpn_cls": 1.0, "loss_rpn_loc": 1.0,
                          "loss_cls_pseudo": S.UNSUP_LOSS_WEIGHT, "loss_box_reg_pseudo": S.UNSUP_REG_LOSS_WEIGHT,
                          "loss_rpn_cls_pseudo": S.UNSUP_LOSS_WEIGHT, "loss_rpn_loc_pseudo": 0.0}
                if fuse:
                    ctx = self.model.forward_joint_begin(all_label_data, unlabel_data_q, loss_weights=lw)
                else:
                    rec_l, _, _, _ = self.model(all_label_data, branch="supervised")
                if overlap:
                    main.wait_stream(side)
                unlabel_data_q = self.add_label(self.remove_label(unlabel_data_q), gt)
                unlabel_data_k = self.add_label(self.remove_label(unlabel_data_k), gt)
                if fuse:
                    rec_l, rec_u = self.model.forward_joint_finish(ctx, gt)
                else:
                    rec_u, _, _, _ = self.model(unlabel_data_q, branch="unsup_data_train")
                record_dict.update(rec_l)
                for k, v in rec_u.items():
                    record_dict[k + "_pseudo"] = v

                fused_total = record_dict.pop("weighted_total", None)
                loss_dict = {}
                for key in ([] if fused_total is not None else record_dict.keys()):
                    if key[:4] != "loss":
                        continue
                    if key == "loss_rpn_loc_pseudo":
                        loss_dict[key] = record_dict[key] * 0
                    elif key == "loss_box_reg_pseudo":
                        loss_dict[key] = record_dict[key] * S.UNSUP_REG_LOSS_WEIGHT
                    elif key[-6:] == "pseudo":
                        loss_dict[key] = record_dict[key] * S.UNSUP_LOSS_WEIGHT
                    else:
                        loss_dict[key] = record_dict[key]
                losses = fused_total if fused_total is not None else sum(loss_dict.values())

            metrics_dict = record_dict
            metrics_dict["data_time"] = data_time
            self._write_metrics(metrics_dict)
            self.optimizer.zero_grad()
            self._backward(losses)
            gscale = self._allreduce_grads()
            self.optimizer.step(grad_scale=gscale, amp_state=self._amp_state)
            return losses

        # test() / build_test_loader() / build_evaluator() come from the shared base (reference trainer.py:986-1023 ≡ :554-608): the
        # eval-mode model runs `inference` (RPN test top-k -> box head -> fast_rcnn_inference -> detector_postprocess).

    return UBRCNNTeacherTrainer
