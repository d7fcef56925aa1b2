#!/bin/bash
# what the pseudo labels buy on a toy set: 128 train / 32 val files (4 texture categories), 12.5 % labeled (16 images), FCOS, 2400 iterations
# at the recipe's learning rate - supervised only (burn-in never ends) against the UTv2 schedule (burn-in 800, then teacher / student);
# box AP of student and teacher on the val files at the end.  usage: semisup_gain.sh [fcos|frcnn]
kind=${1:-fcos}; [ $# -gt 0 ] && shift
mkdir -p gpurun_out
python tools/make_tiny_coco.py /tmp/tiny_ds128 128 32 > /dev/null
python tools/make_synthetic_backbone.py $( [ $kind = frcnn ] && echo rcnn || echo fcos ) /tmp/synth_$kind.pth > /dev/null
for mode in suponly utv2; do
  B=800; [ $mode = suponly ] && B=100000
  for seed in 1 2; do
    DETECTRON2_DATASETS=/tmp/tiny_ds128 timeout 1500 python train_net.py --config-file configs/utv2_${kind}_r50.yaml SOLVER.MAX_ITER 2400 SEMISUPNET.BURN_UP_STEP $B \
      SOLVER.CHECKPOINT_PERIOD 0 TEST.EVAL_PERIOD 800 OUTPUT_DIR /tmp/gain_${kind}_${mode}_$seed MODEL.WEIGHTS /tmp/synth_$kind.pth SOLVER.IMG_PER_BATCH_LABEL 4 SOLVER.IMG_PER_BATCH_UNLABEL 4 \
      DATALOADER.SUP_PERCENT 12.5 DATALOADER.RANDOM_DATA_SEED_PATH /tmp/tiny_ds128/seed.json INPUT.MIN_SIZE_TRAIN "(160, 224)" INPUT.MAX_SIZE_TRAIN 320 \
      INPUT.MIN_SIZE_TEST 192 INPUT.MAX_SIZE_TEST 320 SOLVER.STEPS "(2000,)" SEED $seed "$@" > gpurun_out/gain_${kind}_${mode}_$seed.log 2>&1
    echo "== $kind $mode seed $seed rc=$?" >> gpurun_out/semisup_gain_$kind.txt
    python - >> gpurun_out/semisup_gain_$kind.txt <<PY
import json
for l in open('/tmp/gain_${kind}_${mode}_$seed/metrics.json'):
    d = json.loads(l)
    if 'bbox/AP' in d:
        print('  iter %4d  student AP %.1f AP50 %.1f | teacher AP %.1f AP50 %.1f | total_loss %s' % (d['iteration'], d['bbox_student/AP'], d['bbox_student/AP50'], d['bbox/AP'], d['bbox/AP50'], round(d.get('total_loss', float('nan')), 3)))
PY
  done
done
cat gpurun_out/semisup_gain_$kind.txt
