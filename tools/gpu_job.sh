#!/bin/bash
# scratch: the GPU job of the moment
cd /root/repo
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_rcnn_step_gpu.py -x -q -m gpu -k "deterministic or trainable_stem" 2>&1 | tail -2
timeout 600 python bench.py --steps 23 --warmup 5 --no-cpu-baseline --no-rcnn --no-f32 --timed-only 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['dtype'], round(d['value'],2), round(d['roofline']['frac'],4))"
