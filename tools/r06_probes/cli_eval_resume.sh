#!/bin/bash
# launcher paths a user takes after a first run: --resume (iteration counter, metrics.json appended) and --eval-only from the written checkpoint
mkdir -p gpurun_out; rm -f gpurun_out/cli_eval_resume.txt
for kind in fcos frcnn; do
  O=/tmp/cer_$kind; rm -rf $O
  LR=""; [ $kind = frcnn ] && LR="SOLVER.BASE_LR 0.0001"
  A="--config-file configs/utv2_${kind}_r50.yaml SEMISUPNET.BURN_UP_STEP 10 SOLVER.CHECKPOINT_PERIOD 20 TEST.EVAL_PERIOD 0 OUTPUT_DIR $O MODEL.WEIGHTS '' SOLVER.IMG_PER_BATCH_LABEL 2 SOLVER.IMG_PER_BATCH_UNLABEL 2 $LR"
  eval timeout 600 python train_net.py $A SOLVER.MAX_ITER 40 > gpurun_out/cer_${kind}_1.log 2>&1; echo "rc train $kind $?" >> gpurun_out/cli_eval_resume.txt
  eval timeout 600 python train_net.py --resume $A SOLVER.MAX_ITER 80 > gpurun_out/cer_${kind}_2.log 2>&1; echo "rc resume $kind $?" >> gpurun_out/cli_eval_resume.txt
  grep -a "Starting training\|iter: " gpurun_out/cer_${kind}_2.log | cut -c1-200 >> gpurun_out/cli_eval_resume.txt
  python -c "
import json
its=[json.loads(l)['iteration'] for l in open('$O/metrics.json')]
print('metrics.json iterations', its)" >> gpurun_out/cli_eval_resume.txt
  ls $O >> gpurun_out/cli_eval_resume.txt
  eval timeout 600 python train_net.py --eval-only $A MODEL.WEIGHTS $O/model_final.pth > gpurun_out/cer_${kind}_3.log 2>&1; echo "rc eval-only $kind $?" >> gpurun_out/cli_eval_resume.txt
  tail -4 gpurun_out/cer_${kind}_3.log | cut -c1-300 >> gpurun_out/cli_eval_resume.txt
done
cat gpurun_out/cli_eval_resume.txt
