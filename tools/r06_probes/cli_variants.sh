#!/bin/bash
# config-reachable variants through the launcher: 14 iterations each (Faster-RCNN: 60) across the burn-in boundary (BURN_UP_STEP 6), 2+2
# images of 1333x800, the recipes' own learning rates, MODEL.WEIGHTS = tools/make_synthetic_backbone.py's stand-in for R-50.pkl
mkdir -p gpurun_out; rm -f gpurun_out/cli_variants.txt
python tools/make_synthetic_backbone.py fcos /tmp/synth_fcos.pth > /dev/null
python tools/make_synthetic_backbone.py rcnn /tmp/synth_frcnn.pth > /dev/null
run() {  # run <tag> <kind> <env...> -- <opts...>
  local tag=$1 kind=$2; shift 2
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  local N=14; [ $kind = frcnn ] && N=60
  env "${envs[@]}" timeout 600 python train_net.py --config-file configs/utv2_${kind}_r50.yaml SOLVER.MAX_ITER $N SEMISUPNET.BURN_UP_STEP 6 SOLVER.CHECKPOINT_PERIOD 0 \
     TEST.EVAL_PERIOD 0 OUTPUT_DIR "" MODEL.WEIGHTS /tmp/synth_$kind.pth SOLVER.IMG_PER_BATCH_LABEL 2 SOLVER.IMG_PER_BATCH_UNLABEL 2 "$@" > gpurun_out/var_$tag.log 2>&1
  local rc=$?
  echo "== $tag rc=$rc  $(grep -a "iter: $((N-1)) " gpurun_out/var_$tag.log | sed "s/.*iter: $((N-1))//" | cut -c1-260)" >> gpurun_out/cli_variants.txt
  [ $rc -ne 0 ] && tail -4 gpurun_out/var_$tag.log | cut -c1-300 >> gpurun_out/cli_variants.txt
}
run fcos_default fcos UTV2_X=1 --
run fcos_fp32 fcos UTV2_X=1 -- SOLVER.AMP.ENABLED False
run fcos_bf16 fcos UTV2_PRECISION=bf16 --
run fcos_klloss fcos UTV2_X=1 -- MODEL.FCOS.KL_LOSS True MODEL.FCOS.KL_LOSS_TYPE klloss
run fcos_cls_ctr fcos UTV2_X=1 -- SEMISUPNET.PSEUDO_BBOX_SAMPLE thresholding_cls_ctr SEMISUPNET.BBOX_CTR_THRESHOLD 0.1
run fcos_nocenter fcos UTV2_X=1 -- MODEL.FCOS.CENTER_SAMPLE False
run fcos_iouq fcos UTV2_X=1 -- MODEL.FCOS.QUALITY_EST iou
run fcos_ignore_near fcos UTV2_X=1 -- SEMISUPNET.PSEUDO_CLS_IGNORE_NEAR True
run fcos_yield fcos UTV2_X=1 -- MODEL.FCOS.YIELD_PROPOSAL True
run fcos_twostage_lr fcos UTV2_X=1 -- SOLVER.LR_SCHEDULER_NAME WarmupTwoStageMultiStepLR SOLVER.STEPS "(8, 11)" SOLVER.FACTOR_LIST "(1, 0.5, 0.1)"
run fcos_ema_every2 fcos UTV2_X=1 -- SEMISUPNET.TEACHER_UPDATE_ITER 2
run fcos_unpaired fcos UTV2_PAIR_TOWERS=0 --
run fcos_nofuse fcos UTV2_GN_BWD_FUSE=0 --
run frcnn_default frcnn UTV2_X=1 --
run frcnn_amp frcnn UTV2_X=1 -- SOLVER.AMP.ENABLED True
run frcnn_bf16 frcnn UTV2_PRECISION=bf16 -- SOLVER.AMP.ENABLED True
run frcnn_ce_bvar frcnn UTV2_X=1 -- MODEL.ROI_HEADS.LOSS CrossEntropy_BoundaryVar
run frcnn_focal_v1 frcnn UTV2_X=1 -- MODEL.ROI_HEADS.LOSS FocalLoss MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE smooth_l1
run frcnn_smoothl1 frcnn UTV2_X=1 -- MODEL.ROI_BOX_HEAD.BBOX_REG_LOSS_TYPE smooth_l1 MODEL.ROI_BOX_HEAD.BBOX_PSEUDO_REG_LOSS_TYPE smooth_l1
cat gpurun_out/cli_variants.txt
