#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]   - builds unbiased-teacher-v2_amd/lib_v/<name>/libutv2_hip.so (bf16 build only) with extra
# defines on conv_bf16.hip; the other objects are taken from the regular build (lib/*.o).  For same-box A/Bs through UTV2_LIB_DIR.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
D=$ROOT/unbiased-teacher-v2_amd/lib_v/$NAME
mkdir -p $D
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -I $ROOT/include -c $ROOT/unbiased-teacher-v2_amd/csrc/conv_bf16.hip -o $D/conv_bf16.o
OBJS=$(ls $ROOT/unbiased-teacher-v2_amd/lib/*.o | grep -v "\.f16\.o" | grep -v "/conv_bf16\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libutv2_hip.so $OBJS $D/conv_bf16.o
cp $D/libutv2_hip.so $D/libutv2_hip_f16.so   # (hip.load looks both up; the fp16 name is a stand-in here)
rm $D/conv_bf16.o
echo built $D
