#!/bin/bash
cd /root/repo
timeout 300 python tools/bench_narrowk.py 2>&1 | tail -4
