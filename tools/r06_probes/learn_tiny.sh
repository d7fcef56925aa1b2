#!/bin/bash
# does the whole thing LEARN?  64 train / 16 val images of coloured patches on noise (tools/make_tiny_coco.py), 50 % labeled, the launcher,
# the recipe's learning rate, 1500 iterations (burn-in 500), box AP of student and teacher on the val files every 500 iterations
mkdir -p gpurun_out; rm -f gpurun_out/learn_tiny.txt
python tools/make_tiny_coco.py /tmp/tiny_ds64 64 16 colour > /dev/null
for kind in ${1:-fcos frcnn}; do
  python tools/make_synthetic_backbone.py $( [ $kind = frcnn ] && echo rcnn || echo fcos ) /tmp/synth_$kind.pth > /dev/null
  O=/tmp/learn_$kind; rm -rf $O
  DETECTRON2_DATASETS=/tmp/tiny_ds64 timeout 1500 python train_net.py --config-file configs/utv2_${kind}_r50.yaml SOLVER.MAX_ITER 1500 SEMISUPNET.BURN_UP_STEP 500 \
    SOLVER.CHECKPOINT_PERIOD 0 TEST.EVAL_PERIOD 500 OUTPUT_DIR $O MODEL.WEIGHTS /tmp/synth_$kind.pth SOLVER.IMG_PER_BATCH_LABEL 4 SOLVER.IMG_PER_BATCH_UNLABEL 4 \
    DATALOADER.SUP_PERCENT 50.0 DATALOADER.RANDOM_DATA_SEED_PATH /tmp/tiny_ds64/seed.json INPUT.MIN_SIZE_TRAIN "(160, 224)" INPUT.MAX_SIZE_TRAIN 320 \
    INPUT.MIN_SIZE_TEST 192 INPUT.MAX_SIZE_TEST 320 SOLVER.STEPS "(1200,)" > gpurun_out/learn_$kind.log 2>&1
  echo "rc $kind $?" >> gpurun_out/learn_tiny.txt
  grep -a "iter: [0-9]*99 \|copypaste: [0-9-]" gpurun_out/learn_$kind.log | cut -c1-260 >> gpurun_out/learn_tiny.txt
  tail -2 gpurun_out/learn_$kind.log | cut -c1-300 >> gpurun_out/learn_tiny.txt
done
cat gpurun_out/learn_tiny.txt
