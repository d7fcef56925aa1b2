cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_conv_bf16_gpu.py -x -q -m gpu -k "fused_frozen" 2>&1 | tail -3
timeout 300 python tools/bench_bottleneck.py fp16 2>&1 | grep "N="
