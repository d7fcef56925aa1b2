#!/bin/bash
# FCOS on the tiny learnable dataset: where does the post-burn-in divergence come from?  (same run, one thing changed each time)
mkdir -p gpurun_out; rm -f gpurun_out/learn_fcos_ab.txt
python tools/make_tiny_coco.py /tmp/tiny_ds64 64 16 colour > /dev/null
python tools/make_synthetic_backbone.py fcos /tmp/synth_fcos.pth > /dev/null
run() {
  local tag=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done
  shift
  env "${envs[@]}" DETECTRON2_DATASETS=/tmp/tiny_ds64 timeout 1500 python train_net.py --config-file configs/utv2_fcos_r50.yaml SOLVER.MAX_ITER 900 SEMISUPNET.BURN_UP_STEP 500 \
    SOLVER.CHECKPOINT_PERIOD 0 TEST.EVAL_PERIOD 0 OUTPUT_DIR "" MODEL.WEIGHTS /tmp/synth_fcos.pth SOLVER.IMG_PER_BATCH_LABEL 4 SOLVER.IMG_PER_BATCH_UNLABEL 4 \
    DATALOADER.SUP_PERCENT 50.0 DATALOADER.RANDOM_DATA_SEED_PATH /tmp/tiny_ds64/seed.json INPUT.MIN_SIZE_TRAIN "(160, 224)" INPUT.MAX_SIZE_TRAIN 320 \
    INPUT.MIN_SIZE_TEST 192 INPUT.MAX_SIZE_TEST 320 "$@" > gpurun_out/lab_$tag.log 2>&1
  echo "== $tag rc=$?" >> gpurun_out/learn_fcos_ab.txt
  grep -a "iter: 499 \|iter: 519 \|iter: 559 \|iter: 599 \|iter: 699 \|iter: 899 " gpurun_out/lab_$tag.log | sed 's/.*iter: /iter /' | cut -c1-250 >> gpurun_out/learn_fcos_ab.txt
}
run fp16 UTV2_X=1 --
run fp32 UTV2_X=1 -- SOLVER.AMP.ENABLED False
run bf16 UTV2_PRECISION=bf16 --
run fp16_nopseudo UTV2_X=1 -- SEMISUPNET.UNSUP_LOSS_WEIGHT 0.0 SEMISUPNET.UNSUP_REG_LOSS_WEIGHT 0.0
run fp16_lr UTV2_X=1 -- SOLVER.BASE_LR 0.002
cat gpurun_out/learn_fcos_ab.txt
