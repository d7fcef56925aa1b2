#!/bin/bash
# both trainers through the launcher at 2+2 images of 1333x800 (synthetic COCO-shaped data), the recipes' own learning rates, burn-in 40:
# 400 iterations (FCOS fp16 AMP = its YAML; Faster-RCNN fp32 = its YAML, and with SOLVER.AMP.ENABLED True), the console lines and
# OUTPUT_DIR/metrics.json of the periodic writers; then 60 iterations with TEST.EVAL_PERIOD 30 (student + teacher evaluation hooks).
# MODEL.WEIGHTS = tools/make_synthetic_backbone.py's stand-in for R-50.pkl: no pretrained backbone exists here, and a random ResNet-50 under
# FrozenBN's identity statistics has activations of 1e4-1e5 at the FPN - the Faster-RCNN recipe diverges from it within a few iterations
# for most seeds (tools/r06_probes/rcnn_nan_debug.py; the reference would too).
mkdir -p gpurun_out
python tools/make_synthetic_backbone.py fcos /tmp/synth_fcos.pth > /dev/null
python tools/make_synthetic_backbone.py rcnn /tmp/synth_frcnn.pth > /dev/null
for run in "fcos fcos" "frcnn frcnn" "frcnn_amp frcnn SOLVER.AMP.ENABLED True"; do
  set -- $run; tag=$1; kind=$2; shift 2
  rm -rf /tmp/soak_$tag /tmp/soak_eval_$tag
  A="--config-file configs/utv2_${kind}_r50.yaml SOLVER.CHECKPOINT_PERIOD 100000 MODEL.WEIGHTS /tmp/synth_$kind.pth SOLVER.IMG_PER_BATCH_LABEL 2 SOLVER.IMG_PER_BATCH_UNLABEL 2 $*"
  timeout 900 python train_net.py $A SOLVER.MAX_ITER 400 SEMISUPNET.BURN_UP_STEP 40 TEST.EVAL_PERIOD 0 OUTPUT_DIR /tmp/soak_$tag > gpurun_out/soak_$tag.log 2>&1
  echo "rc $tag $?" >> gpurun_out/soak_$tag.log
  cp /tmp/soak_$tag/metrics.json gpurun_out/soak_${tag}_metrics.json
  timeout 900 python train_net.py $A SOLVER.MAX_ITER 60 SEMISUPNET.BURN_UP_STEP 20 TEST.EVAL_PERIOD 30 OUTPUT_DIR /tmp/soak_eval_$tag > gpurun_out/soak_eval_$tag.log 2>&1
  echo "rc eval $tag $?" >> gpurun_out/soak_eval_$tag.log
  cp /tmp/soak_eval_$tag/metrics.json gpurun_out/soak_eval_${tag}_metrics.json
  grep -a "iter: " gpurun_out/soak_$tag.log | sed -n '1p;3p;10p;$p' | cut -c1-420
  tail -1 gpurun_out/soak_$tag.log; tail -1 gpurun_out/soak_eval_$tag.log
done
