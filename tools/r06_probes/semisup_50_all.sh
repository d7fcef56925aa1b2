rm -f gpurun_out/semisup_50.txt
X='SEMISUPNET.BURN_UP_STEP 1500 SOLVER.MAX_ITER 3000 TEST.EVAL_PERIOD 500 SOLVER.STEPS (2600,)'
TAG=long bash tools/r06_probes/semisup_50.sh fcos $X
TAG=long_sup bash tools/r06_probes/semisup_50.sh fcos $X SEMISUPNET.BURN_UP_STEP 100000
TAG=long bash tools/r06_probes/semisup_50.sh frcnn $X
TAG=long_sup bash tools/r06_probes/semisup_50.sh frcnn $X SEMISUPNET.BURN_UP_STEP 100000
cat gpurun_out/semisup_50.txt
