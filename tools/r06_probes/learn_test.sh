for i in 7 2; do UTV2_LEARN_TEST_SEED=$i timeout 900 python -m pytest tests/test_learning_gpu.py -x -q -s 2>&1 | grep -a "learned:\|passed\|failed"; done
