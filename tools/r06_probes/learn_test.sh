for i in 1 2 3 4 5 6; do UTV2_LEARN_TEST_SEED=$i timeout 900 python -m pytest tests/test_learning_gpu.py -x -q -s 2>&1 | grep -a "learned:\|passed\|failed"; done
