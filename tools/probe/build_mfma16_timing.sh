#!/bin/bash
# Timing build (results are GARBAGE, never shipped): conv_igemm_bf16_rs with every v_mfma_f32_32x32x16 of its COMPUTE slots replaced by two
# v_mfma_f32_16x16x32 on the same operand registers, each updating 4 of the block's 16 accumulator registers - the FLOPs, the operand
# reads, the slot structure and the ACCUMULATOR traffic per FLOP (half of the 32x32x16's) of a real 16x16x32 kernel, without its new
# fragment layout / epilogue.  Bounds what the 16x16x32 instruction's +11 % FLOP/W (profiles/r05_mfma_peak.txt) buys the power-capped tower
# launch before anyone rewrites the kernel.  usage: tools/probe/build_mfma16_timing.sh  ->  unbiased-teacher-v2_amd/lib_v/mfma16/
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
D=$ROOT/unbiased-teacher-v2_amd/lib_v/mfma16
mkdir -p $D
python3 - "$ROOT" "$D" <<'PY'
import sys
root, d = sys.argv[1], sys.argv[2]
s = open(root + "/unbiased-teacher-v2_amd/csrc/conv_bf16.hip").read()
old = "          acc[i][j] = mfma_32x32x16(__builtin_bit_cast(bf16x8_t, fb[ks][j]), __builtin_bit_cast(bf16x8_t, fa[ks][i]), acc[i][j]);\n          const int n = (ks * TM + i) * TN + j;\n          if (n == 2 || n == 6 || n == 10) {   // (the span piece LAST instead of first: same time, measured)"
assert s.count(old) == 1
new = """          {
            typedef float f32x4t __attribute__((ext_vector_type(4)));
            f32x4t lo = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]}, hi = {acc[i][j][4], acc[i][j][5], acc[i][j][6], acc[i][j][7]};
            lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[ks][j]), __builtin_bit_cast(bf16x8_t, fa[ks][i]), lo, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb[ks][j]), __builtin_bit_cast(bf16x8_t, fa[ks][i]), hi, 0, 0, 0);
            acc[i][j][0] = lo[0]; acc[i][j][1] = lo[1]; acc[i][j][2] = lo[2]; acc[i][j][3] = lo[3];
            acc[i][j][4] = hi[0]; acc[i][j][5] = hi[1]; acc[i][j][6] = hi[2]; acc[i][j][7] = hi[3];
          }
          const int n = (ks * TM + i) * TN + j;
          if (n == 2 || n == 6 || n == 10) {   // (the span piece LAST instead of first: same time, measured)"""
open(d + "/conv_bf16_mfma16.hip", "w").write(s.replace(old, new).replace('#include "common.h"', '#include "%s/unbiased-teacher-v2_amd/csrc/common.h"' % root))
PY
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I $ROOT/include -I $ROOT/unbiased-teacher-v2_amd/csrc -c $D/conv_bf16_mfma16.hip -o $D/conv_bf16.o
OBJS=$(ls $ROOT/unbiased-teacher-v2_amd/lib/*.o | grep -v "\.f16\.o" | grep -v "/conv_bf16\.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libutv2_hip.so $OBJS $D/conv_bf16.o
cp $D/libutv2_hip.so $D/libutv2_hip_f16.so
rm $D/conv_bf16.o $D/conv_bf16_mfma16.hip
echo built $D
