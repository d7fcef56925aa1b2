#!/bin/bash
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/rs4_gputests.log 2>&1
tail -5 gpurun_out/rs4_gputests.log
