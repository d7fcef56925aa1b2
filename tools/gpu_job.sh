#!/bin/bash
# scratch: the GPU job of the moment
cd /root/repo
timeout 900 python bench.py --model rcnn --steps 5 --warmup 2 > gpurun_out/rcnn_parity.json 2> gpurun_out/rcnn_parity.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/rcnn_parity.json").read().strip().splitlines()[-1])
print(json.dumps(d.get("parity_fullsize"), indent=1)[:3000])
PY
tail -5 gpurun_out/rcnn_parity.err
