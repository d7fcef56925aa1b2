#!/bin/bash
# builds tools/probe/pp_power_<variant> (see pp_power.hip); the fp16 element type of the headline run
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DUTV2_H16=_Float16 -I ../../include"
b() { hipcc $F $2 -o pp_power_$1 pp_power.hip & }
b full ""
b nomfma "-DPP_NO_MFMA"
b nodma "-DPP_NO_DMA"
b noread "-DPP_NO_READ"
b mfmaonly "-DPP_NO_DMA -DPP_NO_READ"
b dmaonly "-DPP_NO_MFMA -DPP_NO_READ"
b readonly "-DPP_NO_MFMA -DPP_NO_DMA"
b skeleton "-DPP_NO_MFMA -DPP_NO_DMA -DPP_NO_READ"
b a3 "-DPP_A_EVERY=3"
b a9 "-DPP_A_EVERY=9"
wait
ls -la pp_power_*
