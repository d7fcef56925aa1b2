#!/bin/bash
# the launcher on FILES: a tiny COCO-layout tree under $DETECTRON2_DATASETS (tools/make_tiny_coco.py), the shipped configs' dataset names
# (coco_2017_train split 50 % / 50 % by a seed table, coco_2017_val evaluated every 10 iterations), both trainers, 30 iterations
mkdir -p gpurun_out; rm -f gpurun_out/cli_realdata.txt
python tools/make_tiny_coco.py /tmp/tiny_ds >> gpurun_out/cli_realdata.txt
python tools/make_synthetic_backbone.py fcos /tmp/synth_fcos.pth > /dev/null
python tools/make_synthetic_backbone.py rcnn /tmp/synth_frcnn.pth > /dev/null
for kind in fcos frcnn; do
  O=/tmp/real_$kind; rm -rf $O
  DETECTRON2_DATASETS=/tmp/tiny_ds timeout 900 python train_net.py --config-file configs/utv2_${kind}_r50.yaml SOLVER.MAX_ITER 30 SEMISUPNET.BURN_UP_STEP 10 \
    SOLVER.CHECKPOINT_PERIOD 0 TEST.EVAL_PERIOD 10 OUTPUT_DIR $O MODEL.WEIGHTS /tmp/synth_$kind.pth SOLVER.IMG_PER_BATCH_LABEL 2 SOLVER.IMG_PER_BATCH_UNLABEL 2 \
    DATALOADER.SUP_PERCENT 50.0 DATALOADER.RANDOM_DATA_SEED_PATH /tmp/tiny_ds/seed.json INPUT.MIN_SIZE_TRAIN "(160, 224)" INPUT.MAX_SIZE_TRAIN 320 \
    INPUT.MIN_SIZE_TEST 192 INPUT.MAX_SIZE_TEST 320 > gpurun_out/real_$kind.log 2>&1
  echo "rc $kind $?" >> gpurun_out/cli_realdata.txt
  grep -a "iter: \|copypaste\|Evaluation results\|not available\|Error\|error" gpurun_out/real_$kind.log | cut -c1-330 >> gpurun_out/cli_realdata.txt
  [ -f $O/metrics.json ] && cut -c1-360 $O/metrics.json >> gpurun_out/cli_realdata.txt
  tail -3 gpurun_out/real_$kind.log | cut -c1-300 >> gpurun_out/cli_realdata.txt
done
cat gpurun_out/cli_realdata.txt
