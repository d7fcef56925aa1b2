#!/bin/bash
# FCOS / Faster-RCNN on the texture toy set with 50 % labeled: burn-in 500 (a GOOD teacher at the boundary), then teacher / student to 1500; AP every 250
kind=${1:-fcos}; [ $# -gt 0 ] && shift
mkdir -p gpurun_out
python tools/make_tiny_coco.py /tmp/tiny_ds128 128 32 > /dev/null
python tools/make_synthetic_backbone.py $( [ $kind = frcnn ] && echo rcnn || echo fcos ) /tmp/synth_$kind.pth > /dev/null
tag=${TAG:-base}
DETECTRON2_DATASETS=/tmp/tiny_ds128 timeout 1500 python train_net.py --config-file configs/utv2_${kind}_r50.yaml SOLVER.MAX_ITER 1500 SEMISUPNET.BURN_UP_STEP 500 \
  SOLVER.CHECKPOINT_PERIOD 0 TEST.EVAL_PERIOD 250 OUTPUT_DIR /tmp/s50_${kind}_$tag MODEL.WEIGHTS /tmp/synth_$kind.pth SOLVER.IMG_PER_BATCH_LABEL 4 SOLVER.IMG_PER_BATCH_UNLABEL 4 \
  DATALOADER.SUP_PERCENT 50.0 DATALOADER.RANDOM_DATA_SEED_PATH /tmp/tiny_ds128/seed.json INPUT.MIN_SIZE_TRAIN "(160, 224)" INPUT.MAX_SIZE_TRAIN 320 \
  INPUT.MIN_SIZE_TEST 192 INPUT.MAX_SIZE_TEST 320 SOLVER.STEPS "(1300,)" SEED 1 "$@" > gpurun_out/s50_${kind}_$tag.log 2>&1
echo "== $kind $tag rc=$? $*" >> gpurun_out/semisup_50.txt
python - >> gpurun_out/semisup_50.txt <<PY
import json
for l in open('/tmp/s50_${kind}_$tag/metrics.json'):
    d = json.loads(l)
    if 'bbox/AP' in d:
        print('  iter %4d  student AP %.1f AP50 %.1f | teacher AP %.1f AP50 %.1f | %s' % (d['iteration'], d['bbox_student/AP'], d['bbox_student/AP50'], d['bbox/AP'], d['bbox/AP50'],
              '  '.join('%s %.3g' % (k.replace('loss_fcos_', '').replace('loss_', ''), v) for k, v in d.items() if k.startswith('loss'))))
PY
