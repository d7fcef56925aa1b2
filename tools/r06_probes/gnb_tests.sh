timeout 1200 python -m pytest tests/test_gn_bwd_fuse_gpu.py -x -q 2>&1 | tail -40
