#!/bin/bash
# end-to-end run of the launcher with the shipped configs (synthetic COCO-shaped data: no dataset is registered on the box):
# burn-in -> teacher copy -> mutual learning -> checkpoint -> --resume, both trainers, default AMP (fp16)
set -x
mkdir -p gpurun_out
for kind in fcos frcnn; do
  O=/tmp/cli_$kind
  timeout 600 python train_net.py --config-file configs/utv2_${kind}_r50.yaml SOLVER.MAX_ITER 12 SEMISUPNET.BURN_UP_STEP 4 \
     SOLVER.CHECKPOINT_PERIOD 6 TEST.EVAL_PERIOD 0 OUTPUT_DIR $O MODEL.WEIGHTS "" SOLVER.IMG_PER_BATCH_LABEL 2 SOLVER.IMG_PER_BATCH_UNLABEL 2 \
     > gpurun_out/cli_${kind}_train.log 2>&1; echo "rc train $kind $?" >> gpurun_out/cli_e2e.txt
  ls -la $O >> gpurun_out/cli_e2e.txt
  timeout 600 python train_net.py --resume --config-file configs/utv2_${kind}_r50.yaml SOLVER.MAX_ITER 16 SEMISUPNET.BURN_UP_STEP 4 \
     SOLVER.CHECKPOINT_PERIOD 6 TEST.EVAL_PERIOD 0 OUTPUT_DIR $O MODEL.WEIGHTS "" SOLVER.IMG_PER_BATCH_LABEL 2 SOLVER.IMG_PER_BATCH_UNLABEL 2 \
     > gpurun_out/cli_${kind}_resume.log 2>&1; echo "rc resume $kind $?" >> gpurun_out/cli_e2e.txt
  tail -5 gpurun_out/cli_${kind}_train.log >> gpurun_out/cli_e2e.txt
  tail -5 gpurun_out/cli_${kind}_resume.log >> gpurun_out/cli_e2e.txt
done
cat gpurun_out/cli_e2e.txt
