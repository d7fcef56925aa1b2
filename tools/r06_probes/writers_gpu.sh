timeout 900 python -m pytest tests/test_coco_dataset.py -x -q -m gpu 2>&1 | tail -15
