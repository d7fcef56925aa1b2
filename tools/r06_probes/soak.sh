#!/bin/bash
# both trainers through the launcher at 2+2 images (synthetic COCO-shaped data, default AMP, burn-in 40): 400 iterations, the console
# lines and OUTPUT_DIR/metrics.json of the periodic writers; then 60 iterations with TEST.EVAL_PERIOD 30 (student + teacher evaluation hooks).
# No pretrained R-50.pkl exists here: a random ResNet-50 under FrozenBN (identity statistics) has activations of 1e4-1e5 at the FPN and
# the Faster-RCNN recipe diverges from it within a few iterations at its own learning rate for most seeds (tools/r06_probes/rcnn_nan_debug.py;
# the reference would too) - its run uses BASE_LR 1e-4: the launcher, the writers and the hooks are what is checked, not convergence.
mkdir -p gpurun_out
for kind in fcos frcnn; do
  LR=""; [ $kind = frcnn ] && LR="SOLVER.BASE_LR 0.0001"
  rm -rf /tmp/soak_$kind /tmp/soak_eval_$kind
  timeout 900 python train_net.py --config-file configs/utv2_${kind}_r50.yaml SOLVER.MAX_ITER 400 SEMISUPNET.BURN_UP_STEP 40 SOLVER.CHECKPOINT_PERIOD 100000 \
     TEST.EVAL_PERIOD 0 OUTPUT_DIR /tmp/soak_$kind MODEL.WEIGHTS "" SOLVER.IMG_PER_BATCH_LABEL 2 SOLVER.IMG_PER_BATCH_UNLABEL 2 $LR > gpurun_out/soak_$kind.log 2>&1
  echo "rc $kind $?" >> gpurun_out/soak_$kind.log
  cp /tmp/soak_$kind/metrics.json gpurun_out/soak_${kind}_metrics.json
  timeout 900 python train_net.py --config-file configs/utv2_${kind}_r50.yaml SOLVER.MAX_ITER 60 SEMISUPNET.BURN_UP_STEP 20 SOLVER.CHECKPOINT_PERIOD 100000 \
     TEST.EVAL_PERIOD 30 OUTPUT_DIR /tmp/soak_eval_$kind MODEL.WEIGHTS "" SOLVER.IMG_PER_BATCH_LABEL 2 SOLVER.IMG_PER_BATCH_UNLABEL 2 $LR > gpurun_out/soak_eval_$kind.log 2>&1
  echo "rc eval $kind $?" >> gpurun_out/soak_eval_$kind.log
  cp /tmp/soak_eval_$kind/metrics.json gpurun_out/soak_eval_${kind}_metrics.json
  grep "iter: " gpurun_out/soak_$kind.log | sed -n '1p;2p;3p;10p;$p' | cut -c1-420
  tail -3 gpurun_out/soak_eval_$kind.log | cut -c1-300
  cut -c1-500 gpurun_out/soak_eval_${kind}_metrics.json
done
