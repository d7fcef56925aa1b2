#!/bin/bash
mkdir -p gpurun_out
V=$PWD/unbiased-teacher-v2_amd/lib_v
( UTV2_W8=0 timeout 300 python tools/check_w8.py save /tmp/ref.pt | tail -1
  echo "--- default build (row span, zero-bank fix)"; timeout 300 python tools/check_w8.py cmp /tmp/ref.pt | grep -v amdgpu | tr '\n' ';'; echo
  for rep in 1 2; do for x in rs_base rs_alast rs_nosel; do
    echo "$x: $(UTV2_LIB_DIR=$V/$x UTV2_H16_KIND=bf16 timeout 100 python tools/power_probe.py 2.5 2>/dev/null | grep 'tower again' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_launch"],4), round(d["tflops"]), round(d["power_w_mean"]), round(d["sclk_mhz_mean"]))')"
  done; done
  echo "pp: $(UTV2_PP_RS=0 UTV2_LIB_DIR=$V/rs_base UTV2_H16_KIND=bf16 timeout 100 python tools/power_probe.py 2.5 2>/dev/null | grep 'tower again' | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["ms_per_launch"],4), round(d["tflops"]), round(d["power_w_mean"]), round(d["sclk_mhz_mean"]))')"
) > gpurun_out/rs3.txt 2>&1
cat gpurun_out/rs3.txt
