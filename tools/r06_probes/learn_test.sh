for i in 7 1 2 3; do UTV2_LEARN_TEST_SEED=$i timeout 900 python -m pytest tests/test_learning_gpu.py -x -q -s -k rcnn 2>&1 | grep -a "learned:\|passed\|failed\|Error"; done
