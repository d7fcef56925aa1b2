// Shared helpers for the utv2 HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UTV2_OK 0
#define UTV2_EARG (-1000)

// All entry points return 0 on success, -(hipError_t) on a launch error,
// UTV2_EARG on a bad argument.  They never synchronise the stream.
static inline int utv2_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? UTV2_OK : -(int)e;
}

// Opt-in of kernels to more than 64 KB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize), once PER DEVICE: the attribute
// belongs to the function object of the device that is current when it is set, so a process that drives several devices has to set it on
// each (a process-wide once-flag covered the first device only).  Racing first calls both set it - idempotent.
#ifdef __cplusplus
#include <atomic>
#include <initializer_list>
struct LdsOptIn {
  std::atomic<unsigned long long> done{0};
  void operator()(std::initializer_list<const void*> fns, int bytes) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    for (const void* f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
  }
};
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// The 16-bit float type of THIS build of the library.  Every "bf16" kernel, entry point and dtype code (UTV2_BF16) is written against
// h16_t: the default build (libutv2_hip.so) makes it bfloat16, the second build of the same sources (libutv2_hip_f16.so, compiled with
// -DUTV2_H16=_Float16) makes it IEEE fp16 - the element type of the reference's own AMP mode (torch.cuda.amp.autocast,
// engine/trainer.py:195,319).  Nothing below depends on which one it is: conversions are plain casts (round-to-nearest-even either way),
// the matrix instruction is picked by overload (mfma_32x32x16), LDS / DMA / transposing-read traffic is 16-bit data either way.
#ifndef UTV2_H16
#define UTV2_H16 __bf16
#endif
typedef UTV2_H16 h16_t;
typedef h16_t bf16x4_t __attribute__((ext_vector_type(4)));
typedef h16_t bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 mfma_f16x8 __attribute__((ext_vector_type(8)));

// v_mfma_f32_32x32x16_{bf16,f16}: same shape, same rate (2.5 PFLOP/s dense), fp32 accumulate
__device__ __forceinline__ float __attribute__((ext_vector_type(16))) mfma_32x32x16(mfma_bf16x8 a, mfma_bf16x8 b,
                                                                                 float __attribute__((ext_vector_type(16))) c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float __attribute__((ext_vector_type(16))) mfma_32x32x16(mfma_f16x8 a, mfma_f16x8 b,
                                                                                 float __attribute__((ext_vector_type(16))) c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// activation element types of the C-ABI (`*_dtype` arguments)
#define UTV2_F32 0
#define UTV2_BF16 1

// quad (4 consecutive elements) load / store of an activation tensor, arithmetic always in fp32
__device__ __forceinline__ f32x4 ld4(const float* p, size_t i4) { return ((const f32x4*)p)[i4]; }
__device__ __forceinline__ f32x4 ld4(const h16_t* p, size_t i4) {
  const bf16x4_t v = ((const bf16x4_t*)p)[i4];
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (float)v[e];
  return o;
}
__device__ __forceinline__ void st4(float* p, size_t i4, f32x4 v) { ((f32x4*)p)[i4] = v; }
__device__ __forceinline__ void st4(h16_t* p, size_t i4, f32x4 v) {
  bf16x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (h16_t)v[e];  // round-to-nearest-even
  ((bf16x4_t*)p)[i4] = o;
}

#define WAVE 64

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, WAVE));
  return v;
}

// Block-wide sum; result valid in thread 0.  `red` must hold >= blockDim.x/64 floats.
__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  float r = (threadIdx.x < nw) ? red[threadIdx.x] : 0.f;
  if (wid == 0) r = wave_reduce_sum(r);
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------------------------
// Coalesced epilogue for the 32x32-MFMA conv kernels.  Each wave owns a 64 x (TN*32) output sub-tile held
// as acc[TM=2][TN] (row = (e&3) + 8*(e>>2) + 4*(lane>>5), col = lane&31).  Writing it straight from the
// accumulators costs 16*TM*TN dword stores per lane that each cover only two 128-byte row pieces; instead
// the wave bounces one 32-row slab at a time through a private LDS patch and writes full rows with 16-byte
// stores (and reads the residual the same way).  `lds` = wave-private float[32 * (TN*32 + 4)].
// TO = element type of y / residual (float or __bf16); arithmetic is fp32, one rounding at the store.
// Every lane moves 16 bytes per access: 4 fp32 or 8 bf16 channels.
// ReLU masks as BIT planes (16-bit outputs with K % 8 == 0 and ldy % 8 == 0; byte index = element offset / 8, bit q = channel co + q):
// relu_bits is WRITTEN (bit = stored value > 0), mask_bits / post_bits are read in place of mask / post_mask - a sixteenth of the bytes
// of the 16-bit tensors those arguments would re-read for a sign.
// GroupNorm-backward partial sums (gnb_part, with mask_bits and gnb_x; 64 x 64 wave tiles of the multi-level kernels): the conv is the
// dgrad that produces the gradient of a GroupNorm + ReLU OUTPUT; with the ReLU mask applied here (mask_bits: the plane that GroupNorm's apply
// pass wrote) the stored rows are g = dy * (y > 0), and per 64-row block b and output channel c the epilogue also leaves
// gnb_part[b][c] = {sum g, sum g * x} over the block's rows (g as stored; x = gnb_x, the GroupNorm's INPUT, same shape and pitch as the
// output) - GroupNorm backward's first tensor pass (gn_bwd_partial: dy and x, read again from HBM) without the reads.
struct EpiBits {
  unsigned char* relu_bits;
  const unsigned char* mask_bits;
  const unsigned char* post_bits;
  const void* gnb_x;
  float* gnb_part;
};

// One operand combination of epilogue_rows' 16-bit path, fixed at compile time (see the LEAN dispatch there).  Order of operations as in
// the general path: value = acc * scale + bias; mask plane; + residual; post-mask plane; ReLU (lo = 0) or nothing (lo = -inf: non-finite values propagate as in the general path); round; store;
// ReLU bit plane; GroupNorm partial sums.
// TR: the accumulators are TRANSPOSED blocks (the kernel issued its MFMAs with the operands swapped: row = channel (e&3) + 8 (e>>2) +
// 4 (lane>>5), col = pixel lane&31): a lane's 4 consecutive values are 4 consecutive channels of one pixel and reach the patch with 4
// 16-byte LDS writes per block instead of 16 4-byte ones
template <int TN, bool RES, bool MB, bool PB, bool TR = false, bool GNB = false>
__device__ __forceinline__ void epilogue_rows_lean(const f32x16 (&acc)[2][TN], float* lds, int lane, h16_t* __restrict__ y,
                                                   const f32x4 (&sc)[2], const f32x4 (&bi)[2], const h16_t* __restrict__ residual, float lo,
                                                   int m_base, int co, int M, int K, int LDY, float* __restrict__ gn_part, EpiBits eb) {
  constexpr int COLS = TN * 32, LD = COLS + 4, CV = COLS / 8, RPI = 64 / CV;
  const int frow = lane & 31, fh = lane >> 5;
  const int cv = lane % CV, rsub = lane / CV;
  const bool relu_on = lo == 0.f;   // lo: 0 = ReLU, -inf = none (wave-uniform)
  static_assert(!GNB || (TN == 2 && MB && !RES && !PB), "GroupNorm-backward partials: 64-column wave tiles, mask plane only");
  float s0[GNB ? 8 : 1], s1[GNB ? 8 : 1];   // GNB: this lane's 8 channels over its rows of BOTH passes: sum g, sum g * x
  // GNB: the rows of the GroupNorm input and the mask bytes of BOTH passes are requested up front - these kernels run one workgroup per
  // CU and every wave of it is in its epilogue at the same time, so nothing else hides a load issued next to its use (measured: the
  // per-pass form cost the tower dgrad +85 us per launch, more than half of what the fusion saves)
  bf16x8_t xpre[GNB ? 2 : 1][32 / RPI];
  unsigned mpre[GNB ? 2 : 1][32 / RPI];
  if constexpr (GNB) {
#pragma unroll
    for (int q = 0; q < 8; ++q) s0[q] = s1[q] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int m = m_base + i * 32 + it * RPI + rsub;
        const size_t off = (size_t)(m < M ? m : M - 1) * LDY + co;
        xpre[i][it] = *(const bf16x8_t*)((const h16_t*)eb.gnb_x + off);
        mpre[i][it] = eb.mask_bits[off >> 3];
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float gs = 0.f, gq = 0.f;
    bf16x8_t rpre[32 / RPI];   // this pass's residual rows, fetched before the accumulators bounce through LDS: the loads overlap the bounce
    if constexpr (RES) {
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int m = m_base + i * 32 + it * RPI + rsub;
        rpre[it] = *(const bf16x8_t*)(residual + (size_t)(m < M ? m : M - 1) * LDY + co);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (TR) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4*)(lds + frow * LD + j * 32 + 8 * q + 4 * fh) = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) lds[((e & 3) + 8 * (e >> 2) + 4 * fh) * LD + j * 32 + frow] = acc[i][j][e];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed (wave-private patch)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int row = it * RPI + rsub;
      const int m = m_base + i * 32 + row;
      const f32x4 a0 = *(const f32x4*)(lds + row * LD + cv * 8), a1 = *(const f32x4*)(lds + row * LD + cv * 8 + 4);
      const bool in = m < M;
      const size_t off = (size_t)(in ? m : 0) * LDY + co;
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = a0[q] * sc[0][q] + bi[0][q];
        v[4 + q] = a1[q] * sc[1][q] + bi[1][q];
      }
      if constexpr (MB) {
        unsigned mb;
        if constexpr (GNB) mb = mpre[i][it];
        else mb = in ? eb.mask_bits[off >> 3] : 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ((mb >> q) & 1u) ? v[q] : 0.f;
      }
      if constexpr (RES) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += (float)rpre[it][q];
      }
      if constexpr (PB) {
        const unsigned pb = in ? eb.post_bits[off >> 3] : 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = ((pb >> q) & 1u) ? v[q] : 0.f;
      }
      bf16x8_t o;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (h16_t)(relu_on ? fmaxf(v[q], 0.f) : v[q]);   // no ReLU: the value as is (a NaN stays a NaN, as in the general path)
      if (in) {
        *(bf16x8_t*)(y + off) = o;
        if constexpr (GNB) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float f = (float)o[q];
            s0[q] += f;
            s1[q] += f * (float)xpre[i][it][q];
          }
        }
        if (eb.relu_bits) {
          unsigned b = 0;
#pragma unroll
          for (int q = 0; q < 8; ++q) b |= ((float)o[q] > 0.f ? 1u : 0u) << q;
          eb.relu_bits[off >> 3] = (unsigned char)b;
        }
        if (gn_part) {
          float f[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) f[q] = (float)o[q];
          gs += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
          gq += ((f[0] * f[0] + f[1] * f[1]) + (f[2] * f[2] + f[3] * f[3])) + ((f[4] * f[4] + f[5] * f[5]) + (f[6] * f[6] + f[7] * f[7]));
        }
      }
    }
    if (gn_part) {   // lanes cv, cv + CV, ... hold the row parts of channel group co / 8: fixed-order butterfly, lane rsub == 0 writes
#pragma unroll
      for (int d = CV; d < 64; d <<= 1) {
        gs += __shfl_xor(gs, d, 64);
        gq += __shfl_xor(gq, d, 64);
      }
      const int mb = m_base + i * 32;
      if (rsub == 0 && mb < M) {
        float* dst = gn_part + ((size_t)(mb >> 5) * (K >> 3) + (co >> 3)) * 2;
        dst[0] = gs;
        dst[1] = gq;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if constexpr (GNB) {
    // the 8 row lanes of a channel octet meet in the (now free) patch: lane (cv, rsub) leaves {s0, s1} of its 8 channels in row rsub,
    // lane L then owns channel co_base + L and adds the 8 rows in a fixed order - one 8-byte store per lane, 512 contiguous bytes per wave
    constexpr int LE = 2 * COLS + 4;
    float* ex = lds + rsub * LE + cv * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) *(f32x4*)(ex + 4 * q) = f32x4{s0[2 * q], s1[2 * q], s0[2 * q + 1], s1[2 * q + 1]};
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      const float2 v = *(const float2*)(lds + r * LE + 2 * lane);
      a += v.x;
      b += v.y;
    }
    if (m_base < M) *(float2*)(eb.gnb_part + ((size_t)(m_base >> 6) * K + (co - cv * 8) + lane) * 2) = float2{a, b};
    __builtin_amdgcn_wave_barrier();
  }
}

// LEAN_OPS: also instantiate the lean forms with operands (kernels with register headroom; the 128-VGPR kernels keep the plain form only:
// the extra variants cost them ~75 spills)
// GNB_OK: also instantiate the GroupNorm-backward partial-sum form (EpiBits::gnb_part; the multi-level kernels the tower dgrads run on)
template <int TN, typename TO = float, bool LEAN_OPS = true, bool TR = false, bool GNB_OK = false>
__device__ __forceinline__ void epilogue_rows(const f32x16 (&acc)[2][TN], float* lds, int lane, TO* __restrict__ y,
                                              const float* __restrict__ scale, const float* __restrict__ bias,
                                              const TO* __restrict__ residual, int relu, int accumulate, int m_base,
                                              int co_base, int M, int K, const TO* __restrict__ mask = nullptr,
                                              const TO* __restrict__ post_mask = nullptr, int ldy = 0,
                                              float* __restrict__ gn_part = nullptr, EpiBits eb = EpiBits{nullptr, nullptr, nullptr, nullptr, nullptr}) {
  // gn_part (optional, bf16 outputs with K % 8 == 0): fp32 [ceil(M / 32)][K / 8][2] - per 32-row block and 8-channel group the sum and
  // the sum of squares of the values AS STORED (after the bf16 rounding): the statistics pass of the GroupNorm(8 channels per group)
  // that consumes this conv's output (fcos/fcos.py:263-264), taken while the rows are in registers.  m_base is a multiple of 32; rows
  // >= M contribute nothing; fixed reduction order (deterministic).
  // ldy (optional): elements between consecutive rows of y / residual / mask / post_mask (0 = K); y may be a column slice of a wider
  // matrix (the operands that share y's shape share its pitch)
  const int LDY = ldy > 0 ? ldy : K;
  const bool no_plain = (relu & 2) != 0;   // bit 1 of relu (A/B runs, UTV2_EPI_PLAIN=0): keep the general path
  relu &= 1;
  // mask (optional, y's type and shape): y = mask > 0 ? value : 0, applied before the residual add - the ReLU backward of
  // the layer that produced this conv's input, fused into the dgrad that computes its gradient
  // post_mask (optional, same type and shape): y = post_mask > 0 ? value : 0 AFTER the residual add - the ReLU backward of the layer
  // whose output is this conv's INPUT in the forward pass (a dgrad + residual-branch sum that flows into a ReLU output is masked where it is made)
  constexpr int VEC = sizeof(TO) == 2 ? 8 : 4, NQ = VEC / 4;  // channels per lane, as NQ quads
  constexpr int COLS = TN * 32, LD = COLS + 4, CV = COLS / VEC, RPI = 64 / CV;  // rows per store instruction
  const int frow = lane & 31, fh = lane >> 5;
  const int cv = lane % CV, rsub = lane / CV;
  const int co = co_base + cv * VEC;
  f32x4 sc[NQ], bi[NQ];
  bool cok[NQ];  // K % 4 == 0 is required by the caller; a bf16 octet may be half outside
#pragma unroll
  for (int h = 0; h < NQ; ++h) {
    sc[h] = f32x4{1.f, 1.f, 1.f, 1.f};
    bi[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    cok[h] = co + 4 * h < K;
    if (cok[h]) {
      if (scale) sc[h] = *(const f32x4*)(scale + co + 4 * h);
      if (bias) bi[h] = *(const f32x4*)(bias + co + 4 * h);
    }
  }
  const bool full = cok[NQ - 1] && (K & 7) == 0 && (LDY & 7) == 0;  // keeps the 16-byte accesses 16-byte aligned
  if constexpr (NQ == 2) {
    // LEAN paths (wave-uniform choice): 16-bit outputs of a tile that lies inside K whose optional operands are among {residual, mask bit
    // plane, post-mask bit plane} in one of the combinations the step uses - every forward conv (plain; bottleneck conv3: residual), the
    // bottleneck dgrads (mask plane; residual + post-mask plane), the FPN / RPN dgrads.  The same arithmetic in the same order as the
    // general path below (bit-identical outputs, same gn_part reduction tree) with the operand set fixed at compile time: the general
    // path tests eight runtime flags per stored row, each a taken s_cbranch (measured: +2.1 % FCOS step, +1.8 % Faster-RCNN for the plain form).
    const bool lean = (K & 7) == 0 && (LDY & 7) == 0 && co_base + COLS <= K && !mask && !post_mask && !accumulate && !no_plain;
    if (lean) {
      const float lo = relu ? 0.f : -__builtin_inff();
      const int mode = (residual ? 1 : 0) | (eb.mask_bits ? 2 : 0) | (eb.post_bits ? 4 : 0);
#define UTV2_LEAN(R, MB, PB)                                                                                                          \
  epilogue_rows_lean<TN, R, MB, PB, TR>(acc, lds, lane, (h16_t*)y, sc, bi, (const h16_t*)residual, lo, m_base, co, M, K, LDY, gn_part, eb)
      if (mode == 0) { UTV2_LEAN(false, false, false); return; }
      if constexpr (GNB_OK && LEAN_OPS && TN == 2) {
        if (mode == 2 && eb.gnb_part) {
          epilogue_rows_lean<TN, false, true, false, TR, true>(acc, lds, lane, (h16_t*)y, sc, bi, (const h16_t*)residual, lo, m_base, co, M, K, LDY,
                                                               gn_part, eb);
          return;
        }
      }
      if constexpr (LEAN_OPS) {
        if (mode == 1) { UTV2_LEAN(true, false, false); return; }
        if (mode == 2) { UTV2_LEAN(false, true, false); return; }
        if (mode == 5) { UTV2_LEAN(true, false, true); return; }
      }
#undef UTV2_LEAN
    }
  }
  // bf16 residual rows are fetched BEFORE the accumulators bounce through LDS: the loads then overlap the bounce instead of
  // sitting, one dependent load after another, between it and the stores
  bf16x8_t rpre[2][32 / RPI];
  const bool prefetch = NQ == 2 && full && residual != nullptr;
  if constexpr (NQ == 2) {
    if (prefetch) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
          const int m = m_base + i * 32 + it * RPI + rsub;
          const size_t off = (size_t)(m < M ? m : M - 1) * LDY + co;
          rpre[i][it] = *(const bf16x8_t*)((const h16_t*)residual + off);
        }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float gs = 0.f, gq = 0.f;   // this lane's share of the block's group statistics (gn_part)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if constexpr (TR) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(f32x4*)(lds + frow * LD + j * 32 + 8 * q + 4 * fh) = f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) lds[((e & 3) + 8 * (e >> 2) + 4 * fh) * LD + j * 32 + frow] = acc[i][j][e];
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own LDS writes have landed (wave-private patch)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
      const int row = it * RPI + rsub;
      const int m = m_base + i * 32 + row;
      if (m < M && cok[0]) {
        const size_t off = (size_t)m * LDY + co;
        f32x4 v[NQ];
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
          v[h] = *(const f32x4*)(lds + row * LD + cv * VEC + 4 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q) v[h][q] = v[h][q] * sc[h][q] + bi[h][q];
        }
        if constexpr (NQ == 2) {
          if (full) {  // 16-byte bf16 accesses
            if (mask) {
              const bf16x8_t mk = *(const bf16x8_t*)(mask + off);
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q >> 2][q & 3] = (float)mk[q] > 0.f ? v[q >> 2][q & 3] : 0.f;
            } else if (eb.mask_bits) {
              const unsigned mb = eb.mask_bits[off >> 3];
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q >> 2][q & 3] = ((mb >> q) & 1u) ? v[q >> 2][q & 3] : 0.f;
            }
            if (residual) {
              const bf16x8_t r = rpre[i][it];
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q >> 2][q & 3] += (float)r[q];
            }
            if (post_mask) {
              const bf16x8_t mk = *(const bf16x8_t*)(post_mask + off);
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q >> 2][q & 3] = (float)mk[q] > 0.f ? v[q >> 2][q & 3] : 0.f;
            } else if (eb.post_bits) {
              const unsigned mb = eb.post_bits[off >> 3];
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q >> 2][q & 3] = ((mb >> q) & 1u) ? v[q >> 2][q & 3] : 0.f;
            }
            if (relu) {
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q >> 2][q & 3] = fmaxf(v[q >> 2][q & 3], 0.f);
            }
            if (accumulate) {
              const bf16x8_t o = *(const bf16x8_t*)(y + off);
#pragma unroll
              for (int q = 0; q < 8; ++q) v[q >> 2][q & 3] += (float)o[q];
            }
            bf16x8_t o;
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = (h16_t)v[q >> 2][q & 3];
            *(bf16x8_t*)(y + off) = o;
            if (eb.relu_bits) {
              unsigned b = 0;
#pragma unroll
              for (int q = 0; q < 8; ++q) b |= ((float)o[q] > 0.f ? 1u : 0u) << q;
              eb.relu_bits[off >> 3] = (unsigned char)b;
            }
            if (gn_part) {
              float f[8];
#pragma unroll
              for (int q = 0; q < 8; ++q) f[q] = (float)o[q];
              gs += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
              gq += ((f[0] * f[0] + f[1] * f[1]) + (f[2] * f[2] + f[3] * f[3])) + ((f[4] * f[4] + f[5] * f[5]) + (f[6] * f[6] + f[7] * f[7]));
            }
            continue;
          }
        }
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
          if (!cok[h]) continue;
          const size_t o4 = (off >> 2) + h;
          if (mask) {
            const f32x4 mk = ld4(mask, o4);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[h][q] = mk[q] > 0.f ? v[h][q] : 0.f;
          }
          if (residual) v[h] += ld4(residual, o4);
          if (post_mask) {
            const f32x4 mk = ld4(post_mask, o4);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[h][q] = mk[q] > 0.f ? v[h][q] : 0.f;
          }
          if (relu) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[h][q] = fmaxf(v[h][q], 0.f);
          }
          if (accumulate) v[h] += ld4(y, o4);
          st4(y, o4, v[h]);
        }
      }
    }
    if constexpr (NQ == 2) {
      if (gn_part && full) {   // lanes cv, cv + CV, ... hold the row parts of channel group co / 8: fixed-order butterfly, lane rsub == 0 writes
#pragma unroll
        for (int d = CV; d < 64; d <<= 1) {
          gs += __shfl_xor(gs, d, 64);
          gq += __shfl_xor(gq, d, 64);
        }
        const int mb = m_base + i * 32;
        if (rsub == 0 && mb < M) {
          float* dst = gn_part + ((size_t)(mb >> 5) * (K >> 3) + (co >> 3)) * 2;
          dst[0] = gs;
          dst[1] = gq;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
