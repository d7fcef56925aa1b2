#!/bin/bash
cd /root/repo
timeout 600 python tools/cpu_headroom.py 2>&1 | tail -3
