// Mixed-precision (AMP) implicit-GEMM convolution for gfx950: bf16 operands on
// v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense), fp32 accumulate.
//
// This is the MI355X counterpart of the reference's `SOLVER.AMP.ENABLED: True` FCOS configs
// (autocast around the model calls, ubteacher/engine/trainer.py:194-198,318-349): conv operands
// are 16-bit floats, products accumulate in fp32, conv outputs are stored as 16-bit floats.
// Activations (and activation gradients) are bf16 in HBM - half the traffic of the HBM-bound 1x1
// convs and elementwise passes; the kernels are templated on the input / output element type so
// fp32 tensors (loss-side head outputs and their gradients, RoIAlign output) enter and leave
// without a cast pass: an fp32 input is rounded to bf16 (RNE) while it is staged into LDS.
// Weights come from a bf16 mirror of the fp32 master arena.
//
// Same tiling as conv.hip (128 x BN tile, 4 waves as 2x2, each 2 x TN 32x32 accumulators) with
// BK = 32: a row of the LDS tile is 32 bf16 = 64 B (+16 B pad -> the same conflict-free 80-byte
// stride); each thread stages 8 consecutive k of a row (two 16-byte global loads -> one
// ds_write_b128); each MFMA reads one ds_read_b128 per operand fragment.
#include <mutex>
#include <stdlib.h>
#include "common.h"
#include <type_traits>

// 64 zero bytes: out-of-image / out-of-range operand pieces are loaded from here, so every staging load is
// unconditional (no exec-mask branches) and needs no masking of the loaded data (which would force the wave to wait for
// its loads right after issuing them)
__device__ uint4 g_zero64[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};  // never written (non-const keeps it in the global address space: global_load, not flat_load)

#define CONV_MAX_LEVELS 8
struct LevelTab {
  int n;
  int start[CONV_MAX_LEVELS + 1];
  int H[CONV_MAX_LEVELS], W[CONV_MAX_LEVELS];
};

struct ConvArgs16 {
  LevelTab lt;
  const void* x;             // NHWC activations, element type TI
  const h16_t* w;           // bf16 [K][Kred]
  void* y;                   // element type TO
  const float* scale;
  const float* bias;
  const void* residual;      // element type TO
  const void* mask;          // element type TO (optional): y = mask > 0 ? y : 0 before the residual add
  const void* post_mask;     // element type TO (optional): y = post_mask > 0 ? y : 0 after the residual add
  int N, H, W, C, OH, OW, K, KH, KW, stride, pad, in_dil, relu, Kred, M, accumulate;
  int xs;                    // elements between consecutive input pixels (= C except for the image stem's overlapping 8-pixel reads, a
                             // channel slice of a wider matrix, or a grouped conv: groups * C)
  int m_begin;               // first output row of this launch (a conv may be split over two kernels by output-row range)
  int groups;                // grouped conv: output channels [g*K/groups, (g+1)*K/groups) read input channels [g*C, (g+1)*C) of a pixel
                             // (C = channels PER GROUP, Kred = KH*KW*C, xs >= groups*C); the tile width divides K/groups.  The paired
                             // FCOS towers (cls | bbox, two independent 256 -> 256 chains) run as ONE launch per depth this way.
  int ldy;                   // elements between consecutive rows of y / residual / mask / post_mask (>= K: y may be a column slice)
  float* gn_part;            // optional: per (32-row block, 8-channel group) sum / sum of squares of the stored output (see epilogue_rows)
  int ntiles;                // conv_igemm_bf16_pp as a persistent grid: total tiles (0 = one tile per workgroup)
  EpiBits bits;              // optional ReLU bit planes (see epilogue_rows): written from / read in place of 16-bit sign tensors
  int mtot;                  // conv_igemm_bf16_rs: rows of x / rowinfo (>= M: a launch may cover a row range of the matrix)
  const int2* rowinfo;       // optional: per OUTPUT row m {input pixel index of tap (0,0), (W << 16) | tap-validity mask} - the table the weight
                             // gradient kernels read (utv2_conv2d_wgrad_bf16): the tile prologue then loads its rows' geometry instead of
                             // decoding it (level search, two integer divisions and a KH x KW bounds loop per staged row: 2.0-2.2 us of a
                             // 70 us tile on the 256-tile kernel, tools/probe/pp_trace)
};

__device__ __forceinline__ void ml_decode16(const LevelTab& lt, int m, int& pixbase, int& H, int& W, int& oh, int& ow) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < CONV_MAX_LEVELS; ++i)
    if (i < lt.n && m >= lt.start[i]) l = i;
  H = lt.H[l];
  W = lt.W[l];
  const int r = m - lt.start[l], hw = H * W;
  const int n = r / hw, rem = r - n * hw;
  oh = rem / W;
  ow = rem - oh * W;
  pixbase = lt.start[l] + n * hw;
}

// Matrix-pipe priority (round 5; measured, OFF by default).  Two waves of one SIMD that both have 32x32x16 MFMAs ready are served
// alternately by the issue arbiter, and alternating costs the pipe a third of its rate: a register-only MFMA loop sustains 0.95 of the
// issue ceiling with one wave per SIMD, 0.69-0.72 with two (same workgroup or two workgroups per CU), and 0.93-0.94 again when each
// wave raises its priority for its own burst (tools/probe/mfma_peak.hip, profiles/r05_mfma_peak2.txt).  The kernels whose waves run
// their MFMA bursts unsynchronised (the 128-tile forward / dgrad kernel and both weight-gradient kernels; NOT the ping-pong kernel,
// whose barriers give each wave of a SIMD the pipe in turn) can bracket every burst with s_setprio: -DUTV2_MFMA_PRIO=1|3
// (-DUTV2_MFMA_PRIO_LOOSE: without the scheduling fences around the burst).  In the real kernels it buys nothing - their waves wait on
// LDS / global data far more often than on each other's matrix instructions: per-shape replay of a whole step 21.4 ms (off) against
// 21.7-22.0 (three variants), step 338.5 img/s (off) against 337.1-337.6 over three interleaved runs (profiles/r05_prio_ab.txt).
#ifndef UTV2_MFMA_PRIO
#define UTV2_MFMA_PRIO 0
#endif
#if UTV2_MFMA_PRIO && defined(UTV2_MFMA_PRIO_LOOSE)
#define MFMA_BURST_BEGIN __builtin_amdgcn_s_setprio(UTV2_MFMA_PRIO)
#define MFMA_BURST_END __builtin_amdgcn_s_setprio(0)
#elif UTV2_MFMA_PRIO
#define MFMA_BURST_BEGIN __builtin_amdgcn_s_setprio(UTV2_MFMA_PRIO); __builtin_amdgcn_sched_barrier(0)
#define MFMA_BURST_END __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0)
#else
#define MFMA_BURST_BEGIN
#define MFMA_BURST_END
#endif

template <int BN, bool ML, typename TI, typename TO>
__global__ __launch_bounds__(256) void conv_igemm_bf16(ConvArgs16 p) {
  constexpr bool IN16 = sizeof(TI) == 2;
  constexpr int BM = 128, BK = 32, LDB = 80;  // LDS row stride in BYTES
  constexpr int TM = 2, TN = BN / 64, BROWS = BN / 64;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * LDB];
  unsigned char* As = smem;
  unsigned char* Bs = smem + 2 * BM * LDB;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tilesN = (p.K + BN - 1) / BN;
  const int nwg = gridDim.x;
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mt = tile / tilesN, nt = tile - mt * tilesN;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lrow = tid >> 2, kg = tid & 3;

  int ih0[2], iw0[2], Hr[2], Wr[2];
  long long pixbase[2];
  bool mvalid[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int m = m0 + lrow + 64 * r;
    mvalid[r] = m < p.M;
    const int mm = mvalid[r] ? m : 0;
    if constexpr (ML) {
      int pb, oh, ow;
      ml_decode16(p.lt, mm, pb, Hr[r], Wr[r], oh, ow);
      ih0[r] = oh - p.pad;
      iw0[r] = ow - p.pad;
      pixbase[r] = pb;
    } else {
      const int hw = p.OH * p.OW;
      const int n = mm / hw, rem = mm - n * hw;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      ih0[r] = oh * p.stride - p.pad;
      iw0[r] = ow * p.stride - p.pad;
      pixbase[r] = (long long)n * p.H * p.W;
      Hr[r] = p.H;
      Wr[r] = p.W;
    }
  }
  bool bvalid[BROWS];
  const h16_t* wrow[BROWS];
#pragma unroll
  for (int r = 0; r < BROWS; ++r) {
    const int co = n0 + lrow + 64 * r;
    bvalid[r] = co < p.K;
    wrow[r] = p.w + (size_t)(bvalid[r] ? co : 0) * p.Kred;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nchunks = p.KH * p.KW * ((p.C + BK - 1) / BK);  // a 32-channel slab per tap; the last slab may be partial (C % 8 == 0)
  int kh = 0, kw = 0, c0 = 0;
  f32x4 ra[2][2];     // TI = float: two quads per row, converted when written to LDS
  bf16x8_t ra16[2];   // TI = bf16: staged as is
  bf16x8_t rb[BROWS];

  auto gload = [&](int kc) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      int ihn = ih0[r] + kh, iwn = iw0[r] + kw;
      bool ok = mvalid[r];
      int ih = ihn, iw = iwn;
      if (p.in_dil > 1) {
        ok = ok && (ihn % p.in_dil == 0) && (iwn % p.in_dil == 0);
        ih = ihn / p.in_dil;
        iw = iwn / p.in_dil;
      }
      ok = ok && (unsigned)ih < (unsigned)Hr[r] && (unsigned)iw < (unsigned)Wr[r] && (c0 + kg * 8 < p.C);
      const size_t eoff = (size_t)(pixbase[r] + (long long)ih * Wr[r] + iw) * p.C + c0 + kg * 8;
      if constexpr (IN16) {
        bf16x8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (h16_t)0.f;
        if (ok) v = *(const bf16x8_t*)((const h16_t*)p.x + eoff);
        ra16[r] = v;
      } else {
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
          const float* src = (const float*)p.x + eoff;
          v0 = *(const f32x4*)src;
          v1 = *(const f32x4*)(src + 4);
        }
        ra[r][0] = v0;
        ra[r][1] = v1;
      }
    }
#pragma unroll
    for (int r = 0; r < BROWS; ++r) {
      bf16x8_t v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (h16_t)0.f;
      if (bvalid[r] && (c0 + kg * 8 < p.C)) v = *(const bf16x8_t*)(wrow[r] + (kh * p.KW + kw) * p.C + c0 + kg * 8);
      rb[r] = v;
    }
    // taps innermost: the KH*KW shifted reads of one 32-channel slab stay L1/L2 resident
    if (++kw == p.KW) {
      kw = 0;
      if (++kh == p.KH) { kh = 0; c0 += BK; }
    }
  };
  auto lds_store = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      bf16x8_t v;
      if constexpr (IN16) {
        v = ra16[r];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = (h16_t)ra[r][0][e];
          v[4 + e] = (h16_t)ra[r][1][e];
        }
      }
      *(bf16x8_t*)(As + buf * BM * LDB + (lrow + 64 * r) * LDB + kg * 16) = v;
    }
#pragma unroll
    for (int r = 0; r < BROWS; ++r) *(bf16x8_t*)(Bs + buf * BN * LDB + (lrow + 64 * r) * LDB + kg * 16) = rb[r];
  };

  gload(0);
  lds_store(0);
  __syncthreads();
  const int frow = lane & 31, fh = lane >> 5;
  for (int kc = 0; kc < nchunks; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nchunks) gload(kc + 1);
    bf16x8_t a[TM][2], b[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned char* ap = As + buf * BM * LDB + (wm * 64 + i * 32 + frow) * LDB + fh * 16;
      a[i][0] = *(const bf16x8_t*)ap;
      a[i][1] = *(const bf16x8_t*)(ap + 32);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const unsigned char* bp = Bs + buf * BN * LDB + (wn * (BN / 2) + j * 32 + frow) * LDB + fh * 16;
      b[j][0] = *(const bf16x8_t*)bp;
      b[j][1] = *(const bf16x8_t*)(bp + 32);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x16(a[i][s], b[j][s], acc[i][j]);
    if (kc + 1 < nchunks) lds_store(buf ^ 1);
    __syncthreads();
  }

  if ((p.K & 3) == 0) {
    // all MFMAs retired and every wave is past the last barrier of the K loop: the staging LDS is free
    float* patch = (float*)smem + wid * (32 * ((BN / 2) + 4));
    epilogue_rows<TN, TO, false>(acc, patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual, p.relu, p.accumulate,
                          m0 + wm * 64, n0 + wn * (BN / 2), p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits);
    return;
  }
  TO* yo = (TO*)p.y;
  const TO* res = (const TO*)p.residual;
  const TO* msk = (const TO*)p.mask;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = n0 + wn * (BN / 2) + j * 32 + frow;
    if (co >= p.K) continue;
    const float sc = p.scale ? p.scale[co] : 1.f;
    const float bi = p.bias ? p.bias[co] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (m >= p.M) continue;
        const size_t off = (size_t)m * p.ldy + co;
        float v = acc[i][j][e] * sc + bi;
        if (msk) v = (float)msk[off] > 0.f ? v : 0.f;
        if (res) v += (float)res[off];
        if (p.post_mask) v = (float)((const TO*)p.post_mask)[off] > 0.f ? v : 0.f;
        if (p.relu & 1) v = fmaxf(v, 0.f);
        if (p.accumulate) v += (float)yo[off];
        yo[off] = (TO)v;
      }
    }
  }
}

// dst bf16 = RNE(src fp32)
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, h16_t* __restrict__ dst, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    const f32x4 v = ((const f32x4*)src)[i];
    typedef h16_t bf16x4_t __attribute__((ext_vector_type(4)));
    bf16x4_t o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (h16_t)v[e];
    ((bf16x4_t*)dst)[i] = o;
  }
}

// dst bf16 [rows][cpad] = RNE(src [rows][c]) with zeros in columns [c, cpad)   (c, cpad multiples of 8)
template <typename T>
__global__ __launch_bounds__(256) void pad_cols_bf16_kernel(const T* __restrict__ src, h16_t* __restrict__ dst, size_t rows, int c, int cpad) {
  typedef h16_t bf16x8_t __attribute__((ext_vector_type(8)));
  const int per_row = cpad / 8;
  const size_t total = rows * per_row, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t r = i / per_row;
    const int j = (int)(i - r * per_row) * 8;
    bf16x8_t o;
    if (j < c) {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (h16_t)(float)src[r * c + j + e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (h16_t)0.f;
    }
    *(bf16x8_t*)(dst + r * cpad + j) = o;
  }
}

// wt16[ci][KH-1-kh][KW-1-kw][co] = bf16(w[co][kh][kw][ci] * scale[co])   (scale optional: the folded FrozenBN multiplier,
// so dgrad consumes the UNscaled output gradient and no "dy * scale" pass is ever materialised)
__global__ void weight_flip_transpose_bf16_kernel(const float* __restrict__ w, h16_t* __restrict__ wt, const float* __restrict__ scale,
                                                  int K, int KH, int KW, int C) {
  const size_t n = (size_t)K * KH * KW * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    size_t t = i;
    const int co = (int)(t % K); t /= K;
    const int kwp = (int)(t % KW); t /= KW;
    const int khp = (int)(t % KH); t /= KH;
    const int ci = (int)t;
    float v = w[(((size_t)co * KH + (KH - 1 - khp)) * KW + (KW - 1 - kwp)) * C + ci];
    if (scale) v *= scale[co];
    wt[i] = (h16_t)v;
  }
}

// All dgrad weight images of a model in ONE launch: blockIdx.y = layer, the table lives on the device.
struct FlipDesc {
  long long w_off;      // element offset of the layer's [K][KH][KW][C] fp32 weights in the parameter arena
  long long dst_off;    // element offset of its [C][KH][KW][K] bf16 image in the bank buffer
  long long scale_off;  // element offset of its per-output-channel multiplier in `scales`, or -1
  int K, KH, KW, C;
  int Kpad;             // pitch of the image's innermost (output-channel) axis, >= K: channels [K, Kpad) are written as zeros
  int reserved;
};

__global__ __launch_bounds__(256) void weight_flip_transpose_bf16_batched_kernel(const float* __restrict__ arena, const float* __restrict__ scales,
                                                                               h16_t* __restrict__ bank, const FlipDesc* __restrict__ table) {
  // per tap a (K x C) -> (C x K) transpose: 32 x 32 tiles through LDS so that both the fp32 reads (along ci) and the bf16 writes (along
  // co) are coalesced (the element-wise gather ran at 0.75 TB/s)
  __shared__ float tile[32][33];
  const FlipDesc d = table[blockIdx.y];
  const float* w = arena + d.w_off;
  const float* scale = d.scale_off >= 0 ? scales + d.scale_off : nullptr;
  h16_t* wt = bank + d.dst_off;
  const int T = d.KH * d.KW, tk = (d.Kpad + 31) / 32, tc = (d.C + 31) / 32;
  const int ntiles = T * tk * tc;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int tap = t / (tk * tc), r = t - tap * (tk * tc);
    const int co0 = (r / tc) * 32, ci0 = (r - (r / tc) * tc) * 32;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int ftap = (d.KH - 1 - kh) * d.KW + (d.KW - 1 - kw);  // flipped position in the output
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int co = co0 + ty + 8 * i, ci = ci0 + tx;
      float v = 0.f;
      if (co < d.K && ci < d.C) {
        v = w[((size_t)co * T + tap) * d.C + ci];
        if (scale) v *= scale[co];
      }
      tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = ci0 + ty + 8 * i, co = co0 + tx;
      if (ci < d.C && co < d.Kpad) wt[((size_t)ci * T + ftap) * d.Kpad + co] = (h16_t)tile[tx][ty + 8 * i];
    }
    __syncthreads();
  }
}

static int fill_levels16(LevelTab& lt, int nlev, int N, const int* H, const int* W) {
  lt.n = nlev;
  int off = 0;
  for (int l = 0; l < CONV_MAX_LEVELS; ++l) {
    lt.start[l] = off;
    if (l < nlev) {
      lt.H[l] = H[l];
      lt.W[l] = W[l];
      off += N * H[l] * W[l];
    } else {
      lt.H[l] = lt.W[l] = 1;
    }
  }
  lt.start[CONV_MAX_LEVELS] = off;
  return off;
}


// ---------------------------------------------------------------------------------------------
// Main kernel for bf16 activations (every backbone / FPN / tower layer): BK = 64 (C % 64 == 0; long, MFMA-bound K loops)
// or BK = 32 (C % 32 == 0; short HBM-bound 1x1 layers, where 32 KB of LDS keeps 4 workgroups per CU in flight).
//   * the im2col address work is hoisted out of the K loop: per staged row a 32-bit element offset of tap (0,0),
//     W*C and a 16-bit tap-validity mask are computed once; a chunk then costs a multiply-add and a bit test per row
//     (the BK = 32 kernel spent ~3x more VALU than MFMA cycles on 64-bit address arithmetic and bounds checks);
//   * one register set, software-pipelined one full iteration deep: chunk k+1 (loaded during iteration k-1) is
//     written to LDS at the top of iteration k and the loads of chunk k+2 are issued right behind it, so they have
//     a whole 16-MFMA iteration to land;
//   * BK/8 consecutive lanes stage one row of BK bf16 (BK = 64: full 128-byte lines from HBM; conflict-free
//     ds_write_b128); LDS rows are unpadded, 16-byte slot s of row r is stored at slot s ^ ((r >> 1) & 7) (BK = 64) or
//     s ^ ((r >> 2) & 3) (BK = 32): the 16 rows a ds_read_b128 lane group touches then fall on 16 distinct slots of the
//     256-byte bank row (the padded 80-byte rows of the fp32-input kernel cost 2-way ds_write conflicts).
// Stride-1-in-the-input only (in_dil == 1); KH*KW <= 16.
// BN = 96 (round 4): the narrow prediction convs (256 -> 80: cls_logits, bbox_pred | std | ctrness of fcos/fcos.py:306-376).  On the
// 128-wide tile 37 % of their MFMAs and fragment reads computed zero columns.  Here the four waves stack in M (256 x 96 tile, 64 x 96
// per wave = 2 x 3 accumulators): 5 fragment reads feed 6 MFMAs per k16 step (4 : 4 on the 128 x 128 tile) and 83 % of the columns are
// real.  Same staging, swizzle, pipeline and epilogue (called on the 64- and the 32-column part of the wave tile).
// TALL (BN = 64, round 4): K <= 64 layers (the stem, res2's 3x3) likewise as a 256 x 64 tile of four 64 x 64 wave tiles - 4 fragment reads per
// 4 MFMAs instead of 3 per 2 on the 128 x 64 tile's 64 x 32 wave tiles.
template <int BN, bool ML, int BK, typename TO, bool TALL = false, bool GNB = false>   // GNB: see EpiBits::gnb_part (an instantiation of its own)
__global__ __launch_bounds__(256, TALL ? 3 : ((BK == 32 && BN != 96) ? 4 : 2)) void conv_igemm_bf16_v2(ConvArgs16 p) {
  static_assert(!TALL || BN == 64, "the tall layout exists for the 64-wide tile");
  constexpr int WN = (BN == 96 || TALL) ? 1 : 2, WM = 4 / WN, WCOLS = BN / WN;   // waves along N / M; output columns per wave
  constexpr int BM = WM * 64, ROWB = BK * 2;          // LDS row = BK bf16
  constexpr int SLOTS = BK / 8, RPP = 256 / SLOTS;    // 16-byte slots per row; rows staged per pass of the 256 threads
  constexpr int TM = 2, TN = WCOLS / 32, AP = BM / RPP, BP = (BN + RPP - 1) / RPP, KS = BK / 16;  // pieces per thread; k16 steps per chunk
  constexpr int ABUF = BM * ROWB, BBUF = BP * RPP * ROWB;
  constexpr int PCOLS = WCOLS > 64 ? 64 : WCOLS;      // widest column block one epilogue call handles
  constexpr int STAGE = 2 * (ABUF + BBUF), PATCH = 4 * 32 * (PCOLS + 4) * 4;  // staging buffers; epilogue patches
  __shared__ __attribute__((aligned(16))) unsigned char smem[STAGE > PATCH ? STAGE : PATCH];
  unsigned char* As = smem;
  unsigned char* Bs = smem + 2 * ABUF;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int tilesN = (p.K + BN - 1) / BN;
  const int nwg = gridDim.x;
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mt = tile / tilesN, nt = tile - mt * tilesN;
  const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
  const int lrow = tid / SLOTS, slot = tid % SLOTS;  // staged rows lrow + RPP*j, channels slot*8 .. +7 of the chunk
  const int ntaps = p.KH * p.KW;
  const int goff = p.groups > 1 ? (n0 / (p.K / p.groups)) * p.C : 0;  // first input channel of this tile's group

  int aoff[AP], awc[AP];
  unsigned amask[AP];
#pragma unroll
  for (int j = 0; j < AP; ++j) {
    const int m = m0 + lrow + RPP * j;
    const bool mv = m < p.M;
    const int mm = mv ? m : 0;
    int pb, H, W, ih0, iw0;
    if (p.rowinfo) {
      const int2 ri = p.rowinfo[mm];
      aoff[j] = ri.x * p.xs + slot * 8 + goff;
      awc[j] = (ri.y >> 16) * p.xs;
      amask[j] = mv ? (unsigned)(ri.y & 0xffff) : 0u;
      continue;
    }
    if constexpr (ML) {
      int oh, ow;
      ml_decode16(p.lt, mm, pb, H, W, oh, ow);
      ih0 = oh - p.pad;
      iw0 = ow - p.pad;
    } else {
      if (p.KH * p.KW == 1 && p.stride == 1 && p.pad == 0) {  // pointwise: output pixel m reads input pixel m, no geometry to decode
        aoff[j] = mm * p.xs + slot * 8 + goff;
        awc[j] = 0;
        amask[j] = mv ? 1u : 0u;
        continue;
      }
      const int hw = p.OH * p.OW;
      const int n = mm / hw, rem = mm - n * hw;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      ih0 = oh * p.stride - p.pad;
      iw0 = ow * p.stride - p.pad;
      pb = n * p.H * p.W;
      H = p.H;
      W = p.W;
    }
    aoff[j] = (pb + ih0 * W + iw0) * p.xs + slot * 8 + goff;
    awc[j] = W * p.xs;
    unsigned mk = 0;
    for (int kh = 0; kh < p.KH; ++kh)
      for (int kw = 0; kw < p.KW; ++kw)
        if (mv && (unsigned)(ih0 + kh) < (unsigned)H && (unsigned)(iw0 + kw) < (unsigned)W) mk |= 1u << (kh * p.KW + kw);
    amask[j] = mk;
  }
  int boff[BP];
  bool bvalid[BP];
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int co = n0 + lrow + RPP * j;
    bvalid[j] = co < p.K && lrow + RPP * j < BN;
    boff[j] = (bvalid[j] ? co : 0) * p.Kred + slot * 8;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const h16_t* __restrict__ xb = (const h16_t*)p.x;
  const int nchunks = ntaps * (p.C / BK);
  int kh = 0, kw = 0, c0 = 0, tap = 0;
  bf16x8_t ra[AP], rb[BP];

  constexpr int NP = AP + BP;  // staged 16-byte pieces per thread per chunk: A pieces 0..AP-1, then B pieces
  int ua = 0, ub = 0, ukh = 0, utap = 0;  // cursor of the chunk being loaded (wave-uniform)
  auto cursor_next = [&]() {  // latch the chunk (kh, kw, c0) to load next and advance; taps innermost: the KH*KW shifted
    ua = kw * p.xs + c0;      // reads of one BK-channel slab stay L1/L2 resident
    ub = tap * p.C + c0;
    ukh = kh;
    utap = tap;
    ++tap;
    if (++kw == p.KW) {
      kw = 0;
      if (++kh == p.KH) { kh = 0; tap = 0; c0 += BK; }
    }
  };
  const h16_t* zero = (const h16_t*)g_zero64;  // halo / out-of-range pieces read zeros: no branches, no data masking
  auto load_piece = [&](int q) {
    if (q < AP) {
      const h16_t* src = ((amask[q] >> utap) & 1u) ? xb + (unsigned)(aoff[q] + ukh * awc[q] + ua) : zero;
      ra[q] = *(const bf16x8_t*)src;
    } else {
      const h16_t* src = bvalid[q - AP] ? p.w + (unsigned)(boff[q - AP] + ub) : zero;
      rb[q - AP] = *(const bf16x8_t*)src;
    }
  };
  auto swizzle = [](int row) { return BK == 64 ? (row >> 1) & 7 : (row >> 2) & 3; };
  const int wslot = (slot ^ swizzle(lrow)) * 16;  // the swizzle is the same for rows lrow + RPP*j
  auto store_piece = [&](int buf, int q) {
    if (q < AP) *(bf16x8_t*)(As + buf * ABUF + (lrow + RPP * q) * ROWB + wslot) = ra[q];
    else *(bf16x8_t*)(Bs + buf * BBUF + (lrow + RPP * (q - AP)) * ROWB + wslot) = rb[q - AP];
  };

  const int frow = lane & 31, fh = lane >> 5;
  const int swz = swizzle(frow);
  int koff[KS];  // byte offset of the lane's 8 k-elements of k16-step s inside its (swizzled) row
#pragma unroll
  for (int s = 0; s < KS; ++s) koff[s] = ((2 * s + fh) ^ swz) * 16;
  const int arow = (wm * 64 + frow) * ROWB, brow = (wn * WCOLS + frow) * ROWB;

  cursor_next();
#pragma unroll
  for (int q = 0; q < NP; ++q) load_piece(q);
#pragma unroll
  for (int q = 0; q < NP; ++q) store_piece(0, q);
  if (nchunks > 1) {
    cursor_next();
#pragma unroll
    for (int q = 0; q < NP; ++q) load_piece(q);
  }
  __syncthreads();
  // One K iteration.  Chunk kc+1 (its loads were issued one full iteration ago) moves registers -> LDS and the loads of
  // chunk kc+2 are re-issued into the same registers, piece by piece BEHIND the MFMAs of the k16 steps, so the staging
  // instructions issue in the shadow of the matrix pipe instead of ahead of it.  STORE / LOAD are compile-time so the
  // steady-state loop body is one basic block (branches inside it make the compiler fall back to vmcnt(0) waits).
  auto iteration = [&](int buf, auto do_store, auto do_load) {
    constexpr bool STORE = decltype(do_store)::value, LOAD = decltype(do_load)::value;
    if constexpr (LOAD) cursor_next();
    const unsigned char* ab = As + buf * ABUF + arow;
    const unsigned char* bb = Bs + buf * BBUF + brow;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8_t a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8_t*)(ab + i * 32 * ROWB + koff[s]);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8_t*)(bb + j * 32 * ROWB + koff[s]);
      MFMA_BURST_BEGIN;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x16(b[j], a[i], acc[i][j]);   // operands swapped: TRANSPOSED blocks (epilogue_rows<.., TR>)
      MFMA_BURST_END;
#pragma unroll
      for (int q = s * NP / KS; q < (s + 1) * NP / KS; ++q) {
        if constexpr (STORE) store_piece(buf ^ 1, q);
        if constexpr (LOAD) load_piece(q);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;
  int kc = 0;
  for (; kc + 2 < nchunks; ++kc) iteration(kc & 1, yes{}, yes{});
  if (kc + 1 < nchunks) { iteration(kc & 1, yes{}, no{}); ++kc; }
  iteration(kc & 1, no{}, no{});

  if ((p.K & 3) == 0) {
    static_assert(sizeof(smem) >= 4 * 32 * (PCOLS + 4) * sizeof(float), "epilogue patches must fit the staging LDS");
    float* patch = (float*)smem + wid * (32 * (PCOLS + 4));
    if constexpr (TN == 3) {
      // the wave's 64 x 96 tile as a 64-column and a 32-column block (the epilogue's lane mapping needs a power-of-two column count)
      f32x16 lo[2][2], hi[2][1];
#pragma unroll
      for (int i = 0; i < 2; ++i) { lo[i][0] = acc[i][0]; lo[i][1] = acc[i][1]; hi[i][0] = acc[i][2]; }
      epilogue_rows<2, TO, true, true>(lo, patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual, p.relu, p.accumulate, m0 + wm * 64, n0,
                                 p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, nullptr, p.bits);
      if (n0 + 64 < p.K)
        epilogue_rows<1, TO, true, true>(hi, patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual, p.relu, p.accumulate, m0 + wm * 64,
                                   n0 + 64, p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, nullptr, p.bits);
    } else {
      epilogue_rows<TN, TO, (BK != 32), true, GNB>(acc, patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual, p.relu, p.accumulate,
                            m0 + wm * 64, n0 + wn * WCOLS, p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits);
    }
    return;
  }
  TO* yo = (TO*)p.y;
  const TO* res = (const TO*)p.residual;
  const TO* msk = (const TO*)p.mask;
  // (transposed blocks: value e of a lane = channel (e & 3) + 8 (e >> 2) + 4 fh of pixel frow)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = n0 + wn * WCOLS + j * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (co >= p.K) continue;
        const float sc = p.scale ? p.scale[co] : 1.f;
        const float bi = p.bias ? p.bias[co] : 0.f;
        const int m = m0 + wm * 64 + i * 32 + frow;
        if (m >= p.M) continue;
        const size_t off = (size_t)m * p.ldy + co;
        float v = acc[i][j][e] * sc + bi;
        if (msk) v = (float)msk[off] > 0.f ? v : 0.f;
        if (res) v += (float)res[off];
        if (p.post_mask) v = (float)((const TO*)p.post_mask)[off] > 0.f ? v : 0.f;
        if (p.relu & 1) v = fmaxf(v, 0.f);
        if (p.accumulate) v += (float)yo[off];
        yo[off] = (TO)v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// (the lock-step schedule of the 256 x 256 tile, conv_igemm_bf16_w8, lives in tools/probe/conv_lockstep.h since round 6)
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---------------------------------------------------------------------------------------------
// Same 256 x 256 tile, 8 waves, 128 x 64 per wave, LDS-DMA staging - scheduled as a PING-PONG: the two waves that share a SIMD never
// do the same thing at the same time.  In conv_igemm_bf16_w8 all 8 waves run one stream (reads, MFMAs, DMA issue) in loose lockstep,
// so the matrix pipe idles whenever both waves of a SIMD are in their non-MFMA part (s_memtime trace: ~4200 cycles per 64-deep chunk
// against the 2048 the 64 MFMAs of a SIMD need).  Here a chunk is cut into two 32-channel SEGMENTS; waves 0-3 (one per SIMD) and waves
// 4-7 run   LOAD(s) | COMPUTE(s)   one slot apart, slots separated by s_barrier:
//     slot        4c         4c+1        4c+2        4c+3        4c+4
//     waves 0-3   LOAD(2c)   COMP(2c)    LOAD(2c+1)  COMP(2c+1)  LOAD(2c+2)
//     waves 4-7   COMP(2c-1) LOAD(2c)    COMP(2c)    LOAD(2c+1)  COMP(2c+1)
// LOAD(s): 12 ds_read_b128 (all fragments of the segment: 48 VGPRs) and, in their shadow, the source addresses of the wave's next 4
// LDS-DMA pieces; COMPUTE(s): 16 MFMAs with those 4 pieces (segment s+3) issued bare behind the 3rd, 7th, 11th and 15th MFMA.  Measured
// placement costs (tools/probe/pp_trace.hip): a piece issued with its address arithmetic in a LOAD slot stalls the wave 100-200 cycles
// (all four waves of a half queue on the CU's one address unit), a bare piece between MFMAs ~40.
// The LDS image is a ring of four 32 KB segment slots ([stage][operand][segment][256 rows][64 B]), a piece = 16 rows x 64 B, the
// 16-byte slot XOR-swizzled with (row >> 2) & 3 on the source side (conflict-free ds_read_b128).  DMA bookkeeping: segment s+3 goes
// into the slot of segment s-1, whose last readers finished two barriers earlier, and is first read >= 4 slots after its issue; at the
// end of every odd slot each wave waits for everything but the batches younger than the segment the next LOAD slot reads (vmcnt(8) /
// vmcnt(4), loads return in order).  Fragment reads are asm (the compiler would put vmcnt(0) in front of any LDS load it can see while
// DMA is in flight), their waits tied to the registers.  Same accumulation order as the other kernels: bit-identical outputs.
#ifdef UTV2_PP_TRACE
// tools/probe/pp_trace.hip: s_memtime at both ends of every slot of chunks 8..15 of workgroup 0, kept in the unused LDS above the ring
__device__ unsigned g_pp_trace[8 * 128];
// phase stamps (kernel entry, first DMA issue, main loop start, main loop end, kernel end) of workgroup 0 and of a mid-grid workgroup
__device__ unsigned long long g_pp_phase[2][5];
#define PP_STAMP                                                                                         \
  if (blockIdx.x == 0 && c >= 8 && c < 16) {                                                             \
    const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                                          \
    asm volatile("ds_write_b32 %0, %1" : : "v"(tr_addr + 4 * tn), "v"(t_) : "memory");                   \
    ++tn;                                                                                                \
  }
#else
#define PP_STAMP
#endif
#if defined(UTV2_PP_TRACE) && defined(PP_NO_DMA)
#define PP_DMA(cond) false
#else
#define PP_DMA(cond) (cond)
#endif
// Shader-clock probe (utv2_conv_clock_probe): wave 0 of workgroup 0 of every persistent-grid launch leaves {s_memtime ticks, 100 MHz
// s_memrealtime ticks} of its own lifetime here - the clock the chip sustained under THIS kernel (the full-chip launches run at
// 1.45-1.9 GHz of the 2.4 GHz the peak is quoted at: the board's power limit, bench.py roofline.sustained_clock_ghz)
__device__ __attribute__((aligned(16))) unsigned long long g_pp_clock[2];   // written as ONE 16-byte store: a reader never pairs two launches' halves
template <bool ML, typename TO>
__global__ __launch_bounds__(512) void conv_igemm_bf16_pp(ConvArgs16 p) {
#ifdef UTV2_PP_TRACE
  const unsigned long long ph_entry = __builtin_amdgcn_s_memtime();
#endif
  const bool clk_on = ML && blockIdx.x == 0 && p.ntiles > 0;   // the multi-level (tower / RPN head) launches
  unsigned long long clk_c0 = 0, clk_r0 = 0;
  if (clk_on) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
  constexpr int BM = 256, BN = 256, BK = 64, SEGB = 64;
  constexpr int OPSEG = 256 * SEGB, OPB = 2 * OPSEG, STAGE = 2 * OPB;  // 16 KB, 32 KB, 64 KB
  constexpr int TM = 4, TN = 2;
  constexpr int PATCH = 8 * 32 * (TN * 32 + 4) * 4;
  static_assert(2 * STAGE >= PATCH, "epilogue patches must fit the staging LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int tilesN = (p.K + BN - 1) / BN;
  // p.ntiles > 0: PERSISTENT grid (UTV2_PP=2) - gridDim.x workgroups walk the ntiles tiles round by round (virtual block vb = the block
  // index a one-tile-per-workgroup launch would have had: same XCD-aware tile order); 0: one tile per workgroup
  const int nwg = p.ntiles > 0 ? p.ntiles : (int)gridDim.x;
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
  int tile;
  {
    const int bid = vb, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mt = tile / tilesN, nt = tile - mt * tilesN;
  const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
  // DMA role of the lane: rows 32 * wid + 16 * j + (lane >> 2) of either operand, the k-slot that belongs in physical slot lane & 3
  const int prow = lane >> 2;
  const int kslot = (lane & 3) ^ ((lane >> 4) & 3);
  const int ntaps = p.KH * p.KW;
  const int goff = p.groups > 1 ? (n0 / (p.K / p.groups)) * p.C : 0;

  int aoff[2], awc[2];
  unsigned amask[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wid * 32 + j * 16 + prow;
    const bool mv = m < p.M;
    const int mm = mv ? m : 0;
    int pb, H, W, ih0, iw0;
    if (p.rowinfo) {
      const int2 ri = p.rowinfo[mm];
      aoff[j] = ri.x * p.xs + kslot * 8 + goff;
      awc[j] = (ri.y >> 16) * p.xs;
      amask[j] = mv ? (unsigned)(ri.y & 0xffff) : 0u;
      continue;
    }
    if constexpr (ML) {
      int oh, ow;
      ml_decode16(p.lt, mm, pb, H, W, oh, ow);
      ih0 = oh - p.pad;
      iw0 = ow - p.pad;
    } else {
      const int hw = p.OH * p.OW;
      const int n = mm / hw, rem = mm - n * hw;
      const int oh = rem / p.OW, ow = rem - oh * p.OW;
      ih0 = oh * p.stride - p.pad;
      iw0 = ow * p.stride - p.pad;
      pb = n * p.H * p.W;
      H = p.H;
      W = p.W;
    }
    aoff[j] = (pb + ih0 * W + iw0) * p.xs + kslot * 8 + goff;
    awc[j] = W * p.xs;
    unsigned mk = 0;
    for (int kh = 0; kh < p.KH; ++kh)
      for (int kw = 0; kw < p.KW; ++kw)
        if (mv && (unsigned)(ih0 + kh) < (unsigned)H && (unsigned)(iw0 + kw) < (unsigned)W) mk |= 1u << (kh * p.KW + kw);
    amask[j] = mk;
  }
  int boff[2];
  bool bvalid[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = n0 + wid * 32 + j * 16 + prow;
    bvalid[j] = co < p.K;
    boff[j] = (bvalid[j] ? co : 0) * p.Kred + kslot * 8;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const h16_t* __restrict__ xb = (const h16_t*)p.x;
  const int nchunks = ntaps * (p.C / BK);
  int kh = 0, kw = 0, c0 = 0, tap = 0;
  int ua = 0, ub = 0, ukh = 0, utap = 0;
  auto cursor_next = [&]() {
    ua = kw * p.xs + c0;
    ub = tap * p.C + c0;
    ukh = kh;
    utap = tap;
    ++tap;
    if (++kw == p.KW) {
      kw = 0;
      if (++kh == p.KH) { kh = 0; tap = 0; c0 += BK; }
    }
  };
  const h16_t* zero = (const h16_t*)g_zero64;
  unsigned char* const dma_row = smem + (wid * 32) * SEGB;  // wave-uniform (M0)
#ifdef UTV2_PP_TRACE
  const unsigned tr_addr = (unsigned)(size_t)(lptr_t)smem + 2 * STAGE + wid * 512;
  int tn = 0;
  int c = 0;
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  const h16_t* psrc[4];  // sources of the wave's next 4 pieces (0,1: im2col, 2,3: weights), computed in the LOAD slot, issued from the COMPUTE slot
  auto prep_pieces = [&](int sg) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      psrc[j] = ((amask[j] >> utap) & 1u) ? xb + (unsigned)(aoff[j] + ukh * awc[j] + ua + sg * 32) : zero;
      psrc[2 + j] = bvalid[j] ? p.w + (unsigned)(boff[j] + ub + sg * 32) : zero;
    }
  };
  auto issue_piece = [&](int stage, int sg, int q) {
    unsigned char* d = dma_row + stage * STAGE + sg * OPSEG + (q >> 1) * OPB + (q & 1) * 16 * SEGB;
#if defined(UTV2_PP_TRACE) && defined(PP_A_EVERY)
    // tools/probe/pp_power.hip: what an LDS-resident input span would save at best - the im2col pieces only for one tap in PP_A_EVERY
    // (3: one row span per kernel row serves its three taps; 9: a patch with halo serves all nine).  Garbage results: timing / power only
    if (q < 2 && (utap % PP_A_EVERY) != PP_A_EVERY / 2) return;
#endif
    __builtin_amdgcn_global_load_lds((gptr_t)psrc[q], (lptr_t)d, 16, 0, 0);
  };

  const int frow = lane & 31, fh = lane >> 5, fx = (frow >> 2) & 3;
  const unsigned lbase = (unsigned)(size_t)(lptr_t)smem;
  unsigned a_addr[2], b_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_addr[ks] = lbase + (wm * 128 + frow) * SEGB + (((ks * 2 + fh) ^ fx) << 4);
    b_addr[ks] = lbase + OPB + (wn * 64 + frow) * SEGB + (((ks * 2 + fh) ^ fx) << 4);
  }
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 fa[2][TM], fb[2][TN];
#if defined(UTV2_PP_TRACE) && defined(PP_NO_READ)
  for (int ks = 0; ks < 2; ++ks) {
    // fp16 / bf16-plausible bit patterns (activations: half of the values zero; weights: all non-zero) - the matrix pipe's power depends on its data
    for (int i = 0; i < TM; ++i) fa[ks][i] = i32x4{0x3c003a00 + lane * 0x00010003 + i, 0x00003d00 + lane, (0x3e00 + 37 * lane) << 16, 0x3b803c80 ^ (lane << 3)};
    for (int j = 0; j < TN; ++j) fb[ks][j] = i32x4{0x3a10b9f0 + lane * 0x00030001 + j, (int)0xb8c03b40 ^ lane, 0x39e0ba20 + lane * 5, (int)0xbb003900 ^ (lane << 2)};
  }
#endif
  unsigned aa0, aa1, bb0, bb1;
#define PP_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#if defined(UTV2_PP_TRACE) && defined(PP_NO_READ)
#define PP_LOAD_SEG(SG)
#else
#define PP_LOAD_SEG(SG)                                                                                        \
  PP_READ(fa[0][0], aa0, (SG) * OPSEG); PP_READ(fa[0][1], aa0, (SG) * OPSEG + 2048);                           \
  PP_READ(fa[0][2], aa0, (SG) * OPSEG + 4096); PP_READ(fa[0][3], aa0, (SG) * OPSEG + 6144);                    \
  PP_READ(fb[0][0], bb0, (SG) * OPSEG); PP_READ(fb[0][1], bb0, (SG) * OPSEG + 2048);                           \
  PP_READ(fa[1][0], aa1, (SG) * OPSEG); PP_READ(fa[1][1], aa1, (SG) * OPSEG + 2048);                           \
  PP_READ(fa[1][2], aa1, (SG) * OPSEG + 4096); PP_READ(fa[1][3], aa1, (SG) * OPSEG + 6144);                    \
  PP_READ(fb[1][0], bb1, (SG) * OPSEG); PP_READ(fb[1][1], bb1, (SG) * OPSEG + 2048)
#endif
#define PP_WAIT_FRAGS                                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                         \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), \
                 "+v"(fa[1][3]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]))
#define PP_VMWAIT(last, N)                                          \
  if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        \
  else asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define PP_BARRIER                          \
  __builtin_amdgcn_sched_barrier(0);        \
  __builtin_amdgcn_s_barrier();             \
  __builtin_amdgcn_sched_barrier(0)
  auto compute = [&](bool with_dma, int stage, int sg) {  // 16 MFMAs; the wave's 4 pieces of a later segment go out behind the 3rd, 7th, 11th, 15th
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // operands swapped: acc[i][j] is the TRANSPOSED block (rows = channels, columns = pixels), see epilogue_rows<.., TR>
#if defined(UTV2_PP_TRACE) && defined(PP_NO_MFMA)
          asm volatile("" : "+v"(acc[i][j]) : "v"(fb[ks][j]), "v"(fa[ks][i]));   // tools/probe/pp_power.hip: the slot without its matrix instructions
#else
          acc[i][j] = mfma_32x32x16(__builtin_bit_cast(bf16x8_t, fb[ks][j]), __builtin_bit_cast(bf16x8_t, fa[ks][i]),
                                                              acc[i][j]);
#endif
          const int n = (ks * TM + i) * TN + j;
          if ((n & 3) == 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (with_dma) issue_piece(stage, sg, n >> 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
  };

  // prologue: segments 0, 1 (chunk 0) and 2 (chunk 1); the cursor stays on chunk 1
  cursor_next();
#pragma unroll
  for (int sg = 0; sg < 2; ++sg) {
    prep_pieces(sg);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece(0, sg, q);
  }
  if (nchunks > 1) {
    cursor_next();
    prep_pieces(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece(1, 0, q);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PP_BARRIER;
#ifdef UTV2_PP_TRACE
  const unsigned long long ph_loop = __builtin_amdgcn_s_memtime();
#endif
  // C(2c) sends segment 2c+3 = (chunk c+1, segment 1), C(2c+1) segment 2c+4 = (chunk c+2, segment 0) - into the ring slot whose last
  // readers finished two barriers earlier.  Waits (end of every odd slot): everything but the batches younger than the segment the
  // next LOAD slot reads.
  if (wm == 0) {
#ifdef UTV2_PP_TRACE
    for (c = 0; c < nchunks; ++c) {
#else
    for (int c = 0; c < nchunks; ++c) {
#endif
      const int st = c & 1;
      const bool more1 = c + 1 < nchunks, more2 = c + 2 < nchunks;
      aa0 = a_addr[0] + st * STAGE; aa1 = a_addr[1] + st * STAGE; bb0 = b_addr[0] + st * STAGE; bb1 = b_addr[1] + st * STAGE;
      PP_STAMP;
      PP_LOAD_SEG(0);
      if (more1) prep_pieces(1);
      PP_WAIT_FRAGS;
      PP_STAMP;
      PP_BARRIER;
      PP_STAMP;
      compute(PP_DMA(more1), st ^ 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (more1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PP_STAMP;
      PP_BARRIER;
      PP_STAMP;
      PP_LOAD_SEG(1);
      if (more2) { cursor_next(); prep_pieces(0); }
      PP_WAIT_FRAGS;
      PP_STAMP;
      PP_BARRIER;
      PP_STAMP;
      compute(PP_DMA(more2), st, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (more1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PP_STAMP;
      PP_BARRIER;
    }
    PP_BARRIER;
  } else {
    PP_BARRIER;
#ifdef UTV2_PP_TRACE
    for (c = 0; c < nchunks; ++c) {
#else
    for (int c = 0; c < nchunks; ++c) {
#endif
      const int st = c & 1;
      const bool more1 = c + 1 < nchunks, more2 = c + 2 < nchunks;
      aa0 = a_addr[0] + st * STAGE; aa1 = a_addr[1] + st * STAGE; bb0 = b_addr[0] + st * STAGE; bb1 = b_addr[1] + st * STAGE;
      PP_STAMP;
      PP_LOAD_SEG(0);
      if (more1) prep_pieces(1);
      PP_WAIT_FRAGS;
      if (more1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PP_STAMP;
      PP_BARRIER;
      PP_STAMP;
      compute(PP_DMA(more1), st ^ 1, 1);
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP;
      PP_BARRIER;
      PP_STAMP;
      PP_LOAD_SEG(1);
      if (more2) { cursor_next(); prep_pieces(0); }
      PP_WAIT_FRAGS;
      if (more1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PP_STAMP;
      PP_BARRIER;
      PP_STAMP;
      compute(PP_DMA(more2), st, 0);
      __builtin_amdgcn_sched_barrier(0);
      PP_STAMP;
      PP_BARRIER;
    }
  }
#undef PP_READ
#undef PP_LOAD_SEG
#undef PP_WAIT_FRAGS
#undef PP_VMWAIT
#undef PP_BARRIER
#ifdef UTV2_PP_TRACE
  const unsigned long long ph_loop_end = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();
#ifdef UTV2_PP_TRACE
  if (blockIdx.x == 0) {
    for (int i = tid; i < 8 * 128; i += 512) g_pp_trace[i] = ((const unsigned*)(smem + 2 * STAGE))[i];
    __syncthreads();
    if (tid == 0) {  // shader clock: s_memtime ticks per 100 MHz s_memrealtime tick over the main loop
      g_pp_trace[126] = (unsigned)(__builtin_amdgcn_s_memtime() - clk0);
      g_pp_trace[127] = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);
    }
  }
#endif

  float* patch = (float*)smem + wid * (32 * (TN * 32 + 4));
  // two explicit calls (a loop the optimizer declines to unroll would index the accumulator registers dynamically: scratch)
  epilogue_rows<TN, TO, true, true>(*(const f32x16(*)[2][TN]) & acc[0], patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual,
                        p.relu, p.accumulate, m0 + wm * 128, n0 + wn * 64, p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits);
  epilogue_rows<TN, TO, true, true>(*(const f32x16(*)[2][TN]) & acc[2], patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual,
                        p.relu, p.accumulate, m0 + wm * 128 + 64, n0 + wn * 64, p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits);
#ifdef UTV2_PP_TRACE
  if ((blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have left: what a successor workgroup on this CU waits for is the wave's end
    unsigned long long* o = g_pp_phase[blockIdx.x == 0 ? 0 : 1];
    o[0] = ph_entry; o[1] = clk0; o[2] = ph_loop; o[3] = ph_loop_end; o[4] = __builtin_amdgcn_s_memtime();
  }
#endif
  __syncthreads();   // persistent grid: every wave is done with its epilogue patch before the next tile's first DMA pieces land there
  }
  if (clk_on && threadIdx.x == 0) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 v = {__builtin_amdgcn_s_memtime() - clk_c0, __builtin_amdgcn_s_memrealtime() - clk_r0};
    *(u64x2*)g_pp_clock = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Row-span form of the ping-pong kernel (round 5): 3x3, stride 1, rowinfo given.  The three taps of one kernel ROW read the same
// input pixels shifted by one column, so the im2col operand of (64-channel chunk, kh) is staged ONCE as a span of 258 consecutive
// rows (span row s = the centre-column pixel of output row m0 - 1 + s) and tap kw reads span rows r + kw: a third of the kernel's
// im2col LDS-DMA traffic.  Why it pays on a power-capped board: the global -> LDS DMA is 27 % of a tile's energy; the
// shipped kernel with its im2col pieces issued for one tap in three (timing build, garbage results) ran 0.582 -> 0.511 ms on the
// student's paired tower launch (profiles/r05_pp_power_span.txt).
//   * LDS: [2 group buffers][2 segments][272 span rows][64 B] (17 pieces of 16 rows; rows 258.. are never read) | the ping-pong
//     kernel's ring of B segments [2 stages][2 segments][256][64 B] | 256 zero bytes.  Same source-side swizzle: the 16-byte k-slot q of
//     LDS row s lives at slot q ^ ((s >> 2) & 3) - a function of the LDS row, so a fragment read that starts one row up or down stays
//     conflict-free (any 8 consecutive rows cover all 64 banks).
//   * Column borders: output pixel x = 0 / W-1 must read zeros for kw = 0 / 2 where the span holds the neighbouring image row's
//     pixel: the lane's fragment address is switched to the zero bytes (tap-validity bits of the lane's four rows from rowinfo).  Row
//     borders (kh) stay where they were: the DMA source of a span row outside the image is the zero page.
//   * Schedule, slots, barriers, MFMA order and the B stream are the ping-pong kernel's; results are BIT-IDENTICAL to it (same
//     products in the same order).  The span of group g+1 (4 pieces per wave + one 16-row tail piece from waves 0 / 1) is issued in
//     the COMPUTE slots of group g's first two taps; the counted vmcnt waits follow the per-slot piece counts (below).
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

template <bool ML, typename TO, bool GNB = false>   // GNB: see EpiBits::gnb_part (an instantiation of its own: the plain one keeps its registers)
__global__ __launch_bounds__(512) void conv_igemm_bf16_rs(ConvArgs16 p) {
  const bool clk_on = ML && blockIdx.x == 0 && p.ntiles > 0;
  unsigned long long clk_c0 = 0, clk_r0 = 0;
  if (clk_on) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
  constexpr int BM = 256, BN = 256, SEGB = 64;
  constexpr int OPSEG = 256 * SEGB;                                  // one B segment (16 KB)
  constexpr int SPR = 272, ASEG = SPR * SEGB;                        // span rows / bytes of one span segment
  constexpr int BOFF = 4 * ASEG, ZOFF = BOFF + 4 * OPSEG;            // A spans | B ring | zero bytes
  constexpr int TM = 4, TN = 2;
  constexpr int PATCH = 8 * 32 * (TN * 32 + 4) * 4;
  static_assert(BOFF >= PATCH, "epilogue patches must fit the span buffers");
  static_assert(ZOFF % 256 == 0, "the zero bytes mirror the bank of the address they replace");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  if (tid < 64) ((unsigned*)(smem + ZOFF))[tid] = 0u;   // 256 zero bytes (ZOFF is 256-aligned); visible to every wave after the first tile's prologue barrier:
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the store has RETIRED before this wave reaches that barrier (bare s_barrier there, vmcnt-only waits)
  const int tilesN = (p.K + BN - 1) / BN;
  const int nwg = p.ntiles > 0 ? p.ntiles : (int)gridDim.x;
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
  int tile;
  {
    const int bid = vb, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mt = tile / tilesN, nt = tile - mt * tilesN;
  const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
  const int prow = lane >> 2;
  const int kslot = (lane & 3) ^ ((lane >> 4) & 3);
  const int goff = p.groups > 1 ? (n0 / (p.K / p.groups)) * p.C : 0;

  // DMA role: span rows 32 * wid + 16 * j + prow (j = 0, 1) and, for waves 0 / 1, tail rows 256 + prow; a span row is the centre-column
  // pixel (+ p.xs) of output row m0 - 1 + s, valid for kernel row kh when that row's tap (kh, 1) is
  int aoff[3], awc[3];
  unsigned amask[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int s = j < 2 ? wid * 32 + j * 16 + prow : 256 + prow;
    const int m = m0 - 1 + s;
    const bool mv = (unsigned)m < (unsigned)p.mtot && (j < 2 || prow < 2);
    const int2 ri = p.rowinfo[mv ? m : 0];
    aoff[j] = (ri.x + 1) * p.xs + kslot * 8 + goff;
    awc[j] = (ri.y >> 16) * p.xs;
    amask[j] = mv ? (unsigned)(ri.y & 0xffff) : 0u;
  }
  int boff[2];
  bool bvalid[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = n0 + wid * 32 + j * 16 + prow;
    bvalid[j] = co < p.K;
    boff[j] = (bvalid[j] ? co : 0) * p.Kred + kslot * 8;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const h16_t* __restrict__ xb = (const h16_t*)p.x;
  const int ngroups = 3 * (p.C / 64);
  // B cursor (the ping-pong kernel's, taps innermost) and span-group cursor (kernel rows innermost)
  int kh = 0, kw = 0, c0 = 0, tap = 0, ub = 0;
  auto cursor_next = [&]() {
    ub = tap * p.C + c0;
    ++tap;
    if (++kw == 3) {
      kw = 0;
      if (++kh == 3) { kh = 0; tap = 0; c0 += 64; }
    }
  };
  int gkh = 0, gc0 = 0, lkh = 0, lc0 = 0;
  auto group_next = [&]() {
    lkh = gkh;
    lc0 = gc0;
    if (++gkh == 3) { gkh = 0; gc0 += 64; }
  };
  const h16_t* zero = (const h16_t*)g_zero64;
  unsigned char* const dma_row = smem + (wid * 32) * SEGB;  // wave-uniform (M0)
  const h16_t* psrcb[2];
  const h16_t* psrca;
  auto prep_b = [&](int sg) {
#pragma unroll
    for (int j = 0; j < 2; ++j) psrcb[j] = bvalid[j] ? p.w + (unsigned)(boff[j] + ub + sg * 32) : zero;
  };
  auto prep_a = [&](int j, int sg) {   // piece (j = 0, 1: the wave's rows; 2: tail rows) of segment sg of the latched group
    psrca = ((amask[j] >> (lkh * 3 + 1)) & 1u) ? xb + (unsigned)(aoff[j] + lkh * awc[j] + lc0 + sg * 32) : zero;
  };
  auto issue_a = [&](int gb, int sg, int j) {
    unsigned char* d = (j < 2 ? dma_row + j * 16 * SEGB : smem + 256 * SEGB) + (gb * 2 + sg) * ASEG;
    __builtin_amdgcn_global_load_lds((gptr_t)psrca, (lptr_t)d, 16, 0, 0);
  };
  auto issue_b = [&](int stage, int sg, int j) {
    unsigned char* d = dma_row + BOFF + (stage * 2 + sg) * OPSEG + j * 16 * SEGB;
    __builtin_amdgcn_global_load_lds((gptr_t)psrcb[j], (lptr_t)d, 16, 0, 0);
  };

  // fragment role
  const int frow = lane & 31, fh = lane >> 5, fx = (frow >> 2) & 3;
  const unsigned lbase = (unsigned)(size_t)(lptr_t)smem;
  unsigned a_rel[3][2], b_addr[2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      a_rel[t][ks] = lbase + (wm * 128 + frow + t) * SEGB + (((ks * 2 + fh) ^ (((frow + t) >> 2) & 3)) << 4);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) b_addr[ks] = lbase + BOFF + (wn * 64 + frow) * SEGB + (((ks * 2 + fh) ^ fx) << 4);
  const unsigned zaddr = lbase + ZOFF;
  // bit i: kw = 0 readable for the lane's row of block i; bit 8 + i: kw = 2 (kw = 1 always is)
  unsigned okm = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int m = m0 + wm * 128 + i * 32 + frow;
    m = m < p.mtot ? m : p.mtot - 1;
    const int y = p.rowinfo[m].y;
    okm |= (unsigned)((y >> 3) & 1) << i;
    okm |= (unsigned)((y >> 5) & 1) << (8 + i);
  }

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 fa[2][TM], fb[2][TN];
#define RS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RS_WAIT_FRAGS                                                                                                         \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                         \
               : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]), \
                 "+v"(fa[1][3]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]))
#define RS_BARRIER                          \
  __builtin_amdgcn_sched_barrier(0);        \
  __builtin_amdgcn_s_barrier();             \
  __builtin_amdgcn_sched_barrier(0)
  // LOAD slot: the 12 fragment reads of (group buffer gb, segment sg, tap column t) and B (stage, sg)
  auto load_frags = [&](int gb, int sg, int t, int stage) {
    const unsigned abase = (unsigned)((gb * 2 + sg) * ASEG), bbase = (unsigned)((stage * 2 + sg) * OPSEG);
    const unsigned ok = t == 1 ? 15u : (t == 0 ? okm : okm >> 8);   // (the selects cost 0.7 % of the launch: a timing build without them)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned a0 = a_rel[t][ks] + abase;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const unsigned an = a0 + i * 2048;
        const unsigned ai = ((ok >> i) & 1u) ? an : zaddr + (an & 255u);   // zeros from the SAME banks: a masked lane adds no conflict
        RS_READ(fa[ks][i], ai, 0);
      }
      const unsigned bk = b_addr[ks] + bbase;
      RS_READ(fb[ks][0], bk, 0);
      RS_READ(fb[ks][1], bk, 2048);
    }
  };
  // COMPUTE slot: 16 MFMAs; the span piece (if any) goes out behind the 3rd, the two B pieces behind the 7th and 11th
  auto compute = [&](bool doa, int agb, int asg, int aj, bool dob, int bstage, int bsg) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = mfma_32x32x16(__builtin_bit_cast(bf16x8_t, fb[ks][j]), __builtin_bit_cast(bf16x8_t, fa[ks][i]), acc[i][j]);
          const int n = (ks * TM + i) * TN + j;
          if (n == 2 || n == 6 || n == 10) {   // (the span piece LAST instead of first: same time, measured)
            __builtin_amdgcn_sched_barrier(0);
            if (n == 2) { if (doa) issue_a(agb, asg, aj); }
            else if (dob) issue_b(bstage, bsg, n == 6 ? 0 : 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
  };

  // prologue: the spans of group 0 (buffer 0), B segments 0, 1 (chunk 0) and 2 (chunk 1)
  group_next();
#pragma unroll
  for (int sg = 0; sg < 2; ++sg)
#pragma unroll
    for (int j = 0; j < 2; ++j) { prep_a(j, sg); issue_a(0, sg, j); }
  if (wid < 2) { prep_a(2, wid); issue_a(0, wid, 2); }
  cursor_next();
#pragma unroll
  for (int sg = 0; sg < 2; ++sg) {
    prep_b(sg);
    issue_b(0, sg, 0);
    issue_b(0, sg, 1);
  }
  cursor_next();
  prep_b(0);
  issue_b(1, 0, 0);
  issue_b(1, 0, 1);
  vm_wait<0>();
  RS_BARRIER;

  // One group = 3 chunks (kw = 0, 1, 2) x 2 segments = slots p = 0..5.  Pieces a wave issues in COMPUTE(p) of a group that has a
  // successor: p 0..3: 1 span piece + 2 B pieces, p 4, 5: 2 B pieces (+ the tail piece FIRST in p 4 on waves 0 / 1 - not counted: the
  // waits below are then one piece stricter for those waves).  Waves 0-3 wait at the end of COMPUTE(p) for everything but the pieces of
  // COMPUTE(p) and COMPUTE(p-1): vmcnt 5, 6, 6, 6, 5, 4; waves 4-7 (one slot behind) at the end of LOAD(p) for everything but COMPUTE(p-1)'s:
  // 2, 3, 3, 3, 3, 2.  The last group issues no span pieces and runs the ping-pong kernel's end game with its B-only counts.
#define RS_SLOT_G0(P, KW, SG, MOREG, WAITN)                                                                   \
  {                                                                                                           \
    load_frags(gb, SG, KW, st);                                                                               \
    if (SG == 0) { prep_b(1); } else { cursor_next(); prep_b(0); }                                            \
    if (MOREG) { if (P < 4) prep_a(P & 1, P >> 1); else if (P == 4 && wid < 2) prep_a(2, wid); }              \
    RS_WAIT_FRAGS;                                                                                            \
    RS_BARRIER;                                                                                               \
    compute(MOREG && (P < 4 || (P == 4 && wid < 2)), gb ^ 1, P < 4 ? (P >> 1) : wid, P < 4 ? (P & 1) : 2, true, SG == 0 ? (st ^ 1) : st, SG == 0 ? 1 : 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    vm_wait<WAITN>();                                                                                         \
    RS_BARRIER;                                                                                               \
  }
#define RS_SLOT_G1(P, KW, SG, MOREG, WAITN)                                                                   \
  {                                                                                                           \
    load_frags(gb, SG, KW, st);                                                                               \
    if (SG == 0) { prep_b(1); } else { cursor_next(); prep_b(0); }                                            \
    if (MOREG) { if (P < 4) prep_a(P & 1, P >> 1); }                                                          \
    RS_WAIT_FRAGS;                                                                                            \
    vm_wait<WAITN>();                                                                                         \
    RS_BARRIER;                                                                                               \
    compute(MOREG && P < 4, gb ^ 1, P >> 1, P & 1, true, SG == 0 ? (st ^ 1) : st, SG == 0 ? 1 : 0);            \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
    RS_BARRIER;                                                                                               \
  }
  if (wm == 0) {
    int g = 0;
    for (; g + 1 < ngroups; ++g) {
      const int gb = g & 1;
      int st = g & 1;                 // chunk 3g + kw: stage (3g + kw) & 1 = (g + kw) & 1
      group_next();
      RS_SLOT_G0(0, 0, 0, true, 5) RS_SLOT_G0(1, 0, 1, true, 6)
      st ^= 1;
      RS_SLOT_G0(2, 1, 0, true, 6) RS_SLOT_G0(3, 1, 1, true, 6)
      st ^= 1;
      RS_SLOT_G0(4, 2, 0, true, 5) RS_SLOT_G0(5, 2, 1, true, 4)
    }
    {  // last group: chunks nchunks-3 .. nchunks-1
      const int gb = g & 1;
      int st = g & 1;
      RS_SLOT_G0(0, 0, 0, false, 4) RS_SLOT_G0(1, 0, 1, false, 4)
      st ^= 1;
      // chunk nchunks-2: its first COMPUTE still sends (last chunk, segment 1), its second has nothing left to send
      load_frags(gb, 0, 1, st); prep_b(1); RS_WAIT_FRAGS; RS_BARRIER;
      compute(false, 0, 0, 0, true, st ^ 1, 1); __builtin_amdgcn_sched_barrier(0); vm_wait<4>(); RS_BARRIER;
      load_frags(gb, 1, 1, st); RS_WAIT_FRAGS; RS_BARRIER;
      compute(false, 0, 0, 0, false, 0, 0); __builtin_amdgcn_sched_barrier(0); vm_wait<2>(); RS_BARRIER;
      st ^= 1;
      load_frags(gb, 0, 2, st); RS_WAIT_FRAGS; RS_BARRIER;
      compute(false, 0, 0, 0, false, 0, 0); __builtin_amdgcn_sched_barrier(0); vm_wait<0>(); RS_BARRIER;
      load_frags(gb, 1, 2, st); RS_WAIT_FRAGS; RS_BARRIER;
      compute(false, 0, 0, 0, false, 0, 0); __builtin_amdgcn_sched_barrier(0); RS_BARRIER;
    }
    RS_BARRIER;
  } else {
    RS_BARRIER;
    int g = 0;
    for (; g + 1 < ngroups; ++g) {
      const int gb = g & 1;
      int st = g & 1;
      group_next();
      RS_SLOT_G1(0, 0, 0, true, 2) RS_SLOT_G1(1, 0, 1, true, 3)
      st ^= 1;
      RS_SLOT_G1(2, 1, 0, true, 3) RS_SLOT_G1(3, 1, 1, true, 3)
      st ^= 1;
      RS_SLOT_G1(4, 2, 0, true, 3) RS_SLOT_G1(5, 2, 1, true, 2)
    }
    {
      const int gb = g & 1;
      int st = g & 1;
      RS_SLOT_G1(0, 0, 0, false, 2) RS_SLOT_G1(1, 0, 1, false, 2)
      st ^= 1;
      load_frags(gb, 0, 1, st); prep_b(1); RS_WAIT_FRAGS; vm_wait<2>(); RS_BARRIER;
      compute(false, 0, 0, 0, true, st ^ 1, 1); __builtin_amdgcn_sched_barrier(0); RS_BARRIER;
      load_frags(gb, 1, 1, st); RS_WAIT_FRAGS; vm_wait<2>(); RS_BARRIER;
      compute(false, 0, 0, 0, false, 0, 0); __builtin_amdgcn_sched_barrier(0); RS_BARRIER;
      st ^= 1;
      load_frags(gb, 0, 2, st); RS_WAIT_FRAGS; vm_wait<0>(); RS_BARRIER;
      compute(false, 0, 0, 0, false, 0, 0); __builtin_amdgcn_sched_barrier(0); RS_BARRIER;
      load_frags(gb, 1, 2, st); RS_WAIT_FRAGS; RS_BARRIER;
      compute(false, 0, 0, 0, false, 0, 0); __builtin_amdgcn_sched_barrier(0); RS_BARRIER;
    }
  }
#undef RS_SLOT_G0
#undef RS_SLOT_G1
#undef RS_READ
#undef RS_WAIT_FRAGS
#undef RS_BARRIER
  __syncthreads();

  float* patch = (float*)smem + wid * (32 * (TN * 32 + 4));
  epilogue_rows<TN, TO, true, true, GNB>(*(const f32x16(*)[2][TN]) & acc[0], patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual,
                        p.relu, p.accumulate, m0 + wm * 128, n0 + wn * 64, p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits);
  epilogue_rows<TN, TO, true, true, GNB>(*(const f32x16(*)[2][TN]) & acc[2], patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual,
                        p.relu, p.accumulate, m0 + wm * 128 + 64, n0 + wn * 64, p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits);
  __syncthreads();   // persistent grid: every wave is done with its epilogue patch before the next tile's first DMA pieces land there
  }
  if (clk_on && threadIdx.x == 0) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 v = {__builtin_amdgcn_s_memtime() - clk_c0, __builtin_amdgcn_s_memrealtime() - clk_r0};
    *(u64x2*)g_pp_clock = v;
  }
}

// A/B switches for bench runs and tests, read ONCE per process (never on the launch path): UTV2_W8=0 keeps every forward / dgrad
// row on the 128 x 128 kernel, UTV2_WGRAD_W8=0 keeps every wgrad on the 128 x 128 kernel.
static bool env_flag_on(const char* name) {
  const char* v = getenv(name);
  return !(v && v[0] == '0');
}
static const bool g_use_w8 = env_flag_on("UTV2_W8");
static const bool g_use_wgrad_w8 = env_flag_on("UTV2_WGRAD_W8");
static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
static const int g_epi_general = env_int("UTV2_EPI_PLAIN", 1) ? 0 : 2;   // OR-ed into ConvArgs16::relu (see epilogue_rows)
static const int g_use_pp = env_int("UTV2_PP", 2) == 1 ? 1 : 2;  // 256 x 256 forward tile, ping-pong schedule: 2 = persistent grid of 256 workgroups,
                                                                 // 1 = one tile per workgroup

template <int BN, bool ML>
static void launch_igemm16(const ConvArgs16& a, int tiles, int x_dtype, int y_dtype, hipStream_t stream) {
  const dim3 g(tiles), b(256);
  // v2 kernels: bf16 input, 32-bit element offsets, plain (non-dilated) gather, <= 16 taps
  const int64_t xelems = ML ? (int64_t)a.M * a.xs : (int64_t)a.N * a.H * a.W * a.xs;
  if (x_dtype == UTV2_BF16 && a.C % 32 == 0 && a.in_dil == 1 && a.KH * a.KW <= 16 && xelems < (1ll << 31) &&
      (int64_t)a.K * a.Kred < (1ll << 31)) {
    // BK = 64 holds 2 workgroups per CU (64 KB LDS) against 4 for BK = 32: worth it for long K loops (MFMA-bound 3x3 /
    // wide 1x1 layers) unless the grid is a little over one 512-slot round (tail), not for short HBM-bound ones.
    const bool tail = tiles > 512 && tiles <= 768;
    const bool deep = a.C % 64 == 0 && a.Kred >= 1024 && (!tail || a.bits.gnb_part);   // (the GroupNorm-backward partials exist in the BK = 64 form)
    // 3x3 stride-1 layers with the per-row geometry table: the row-span form (a third of the im2col DMA; bit-identical results)
    // ("same" geometry only: the column neighbour of an output pixel must be the centre pixel of the neighbouring output row index)
    static const bool use_rs = env_int("UTV2_PP_RS", 1) != 0;
    const bool rs_ok = use_rs && a.rowinfo && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && (ML || (a.OH == a.H && a.OW == a.W));
    const bool gnb = ML && BN == 128 && a.bits.gnb_part != nullptr;   // GroupNorm-backward partials: row-span and 128 x 128 BK = 64 kernels
    const bool w8 = g_use_w8 && (!gnb || rs_ok);
    if (w8 && BN == 128 && a.xs >= a.groups * a.C && (a.xs & 7) == 0 && a.C % 64 == 0 && a.Kred >= 1024 && a.K >= 256 && (a.K & 3) == 0 &&
        a.m_begin == 0 && (a.groups == 1 || (a.K / a.groups) % 256 == 0)) {
      // Whole rounds of 256 tiles (one per CU) always pay.  The rest: a partial round costs one 256-tile time (~76 us on the tower
      // shape) whatever its fill, the 128 x 128 kernel ~35-46 us per round of 512 of its tiles - so the big tile also takes the rest
      // (including the partial last row tile, masked in the kernel) when it is more than half a round, else the small kernel does.
      const int tilesN = cdiv(a.K, 256), tiles_m = a.M / 256, tiles_all = cdiv(a.M, 256) * tilesN;
      int main_tiles = (tiles_m * tilesN / 256) * 256;
      if (tiles_all - main_tiles > 128) main_tiles = tiles_all;
      const int main_m = main_tiles == tiles_all ? cdiv(a.M, 256) : main_tiles / tilesN;
      if (main_m > 0) {
        ConvArgs16 m = a;
        m.M = main_tiles == tiles_all ? a.M : main_m * 256;
        const int smem = 2 * (256 + 256) * 128;
        static LdsOptIn lds_opt_in;   // per-device function attribute (thread-safe; the only write-once state of this entry)
        lds_opt_in({(const void*)conv_igemm_bf16_pp<ML, h16_t>, (const void*)conv_igemm_bf16_pp<ML, float>}, smem);
        {
          int grid = main_m * tilesN;
          static const int pp_wgs = [] { int v = env_int("UTV2_PP_WGS", 256); return v < 8 ? 8 : (v > 256 ? 256 : v / 8 * 8); }();   // A/B: CUs the persistent grid takes
          if (g_use_pp == 2 && grid > 256) { m.ntiles = grid; grid = pp_wgs; }
          if (rs_ok) {
            const int smem_rs = 4 * 272 * 64 + 4 * 256 * 64 + 256;
            static LdsOptIn rs_opt_in;
            rs_opt_in({(const void*)conv_igemm_bf16_rs<ML, h16_t>, (const void*)conv_igemm_bf16_rs<ML, float>,
                       (const void*)conv_igemm_bf16_rs<ML, h16_t, ML>}, smem_rs);
            m.mtot = a.M;
            if (gnb) hipLaunchKernelGGL((conv_igemm_bf16_rs<ML, h16_t, ML>), dim3(grid), dim3(512), smem_rs, stream, m);
            else if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16_rs<ML, h16_t>), dim3(grid), dim3(512), smem_rs, stream, m);
            else hipLaunchKernelGGL((conv_igemm_bf16_rs<ML, float>), dim3(grid), dim3(512), smem_rs, stream, m);
          } else if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16_pp<ML, h16_t>), dim3(grid), dim3(512), smem, stream, m);
          else hipLaunchKernelGGL((conv_igemm_bf16_pp<ML, float>), dim3(grid), dim3(512), smem, stream, m);
        }
        if (m.M == a.M) return;
        ConvArgs16 r = a;
        r.m_begin = m.M;
        launch_igemm16<BN, ML>(r, cdiv(a.M - m.M, 128) * cdiv(a.K, 128), x_dtype, y_dtype, stream);
        return;
      }
    }
    if constexpr (BN == 64) {
      // 256 x 64 tile (TALL) for long matrices whose K loop is worth it: 3x3 / 7-tap layers (the HBM-bound 1x1 layers stay on 128 x 64)
      static const bool use_tall = env_int("UTV2_CONV_TALL64", 1) != 0;
      if (use_tall && a.KH * a.KW > 1 && a.M - a.m_begin >= 65536 && a.K > 32) {
        const dim3 gt(cdiv(a.M - a.m_begin, 256) * cdiv(a.K, 64));
        if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16_v2<64, ML, 32, h16_t, true>), gt, b, 0, stream, a);
        else hipLaunchKernelGGL((conv_igemm_bf16_v2<64, ML, 32, float, true>), gt, b, 0, stream, a);
        return;
      }
    }
    if constexpr (BN == 96) {   // 256 x 96 tile: BK = 32 only (a BK = 64 stage would need 88 KB of LDS: one workgroup per CU)
      if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16_v2<96, ML, 32, h16_t>), g, b, 0, stream, a);
      else hipLaunchKernelGGL((conv_igemm_bf16_v2<96, ML, 32, float>), g, b, 0, stream, a);
    } else if (deep) {
      if (gnb) hipLaunchKernelGGL((conv_igemm_bf16_v2<BN, ML, 64, h16_t, false, (ML && BN == 128)>), g, b, 0, stream, a);
      else if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16_v2<BN, ML, 64, h16_t>), g, b, 0, stream, a);
      else hipLaunchKernelGGL((conv_igemm_bf16_v2<BN, ML, 64, float>), g, b, 0, stream, a);
    } else {
      if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16_v2<BN, ML, 32, h16_t>), g, b, 0, stream, a);
      else hipLaunchKernelGGL((conv_igemm_bf16_v2<BN, ML, 32, float>), g, b, 0, stream, a);
    }
    return;
  }
  if constexpr (BN != 96) {   // (the callers send only v2-eligible problems to the 96-wide tile: n96_eligible)
    if (x_dtype == UTV2_BF16) {
      if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16<BN, ML, h16_t, h16_t>), g, b, 0, stream, a);
      else hipLaunchKernelGGL((conv_igemm_bf16<BN, ML, h16_t, float>), g, b, 0, stream, a);
    } else {
      if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16<BN, ML, float, h16_t>), g, b, 0, stream, a);
      else hipLaunchKernelGGL((conv_igemm_bf16<BN, ML, float, float>), g, b, 0, stream, a);
    }
  }
}

// the 256 x 96 tile (conv_igemm_bf16_v2<96>): level-first 16-bit inputs, 64 < K <= 96 output channels, everything the v2 kernels need
static const bool g_use_n96 = env_int("UTV2_CONV_N96", 1) != 0;
static bool n96_eligible(const ConvArgs16& a, int x_dtype) {
  return g_use_n96 && x_dtype == UTV2_BF16 && a.K > 64 && a.K <= 96 && (a.K & 3) == 0 && a.C % 32 == 0 && a.in_dil == 1 && a.KH * a.KW <= 16 &&
         a.groups == 1 && !a.gn_part && a.M >= 8192 && (int64_t)a.M * a.xs < (1ll << 31) && (int64_t)a.K * a.Kred < (1ll << 31);
}

static inline bool bad_dtype(int d) { return d != UTV2_F32 && d != UTV2_BF16; }

extern "C" {

// Measurement aid (synchronises the device; never on the training path): shader clock in GHz that workgroup 0 of the LAST persistent-grid
// multi-level (FCOS tower / RPN head) launch of the 256-tile forward / dgrad kernel ran at (s_memtime ticks per 10 ns s_memrealtime tick over the workgroup's lifetime), and
// that lifetime in microseconds.  0 / 0 when no such launch has run yet.
int utv2_conv_clock_probe(double* ghz, double* lifetime_us) {
  if (!ghz || !lifetime_us) return UTV2_EARG;
  unsigned long long v[2] = {0, 0};
  hipError_t e = hipMemcpyFromSymbol(v, HIP_SYMBOL(g_pp_clock), sizeof(v));
  if (e != hipSuccess) return -(int)e;
  *ghz = v[1] ? (double)v[0] / (10.0 * (double)v[1]) : 0.0;
  *lifetime_us = (double)v[1] * 0.01;
  // launches of two streams (teacher / student) both write the probe: each value pair is one launch's (single 16-byte store), but a
  // pair outside what this chip can clock (0.3 .. 2.6 GHz) is reported as "no reading" rather than as a number
  if (*ghz < 0.3 || *ghz > 2.6) { *ghz = 0.0; *lifetime_us = 0.0; }
  return UTV2_OK;
}

// 1 if the bf16 MFMA kernel supports this conv (C % 8 == 0), else the caller uses the fp32 kernel
int utv2_conv2d_bf16_supported(int C, int KH, int KW) { return (C % 8 == 0) ? 1 : 0; }

// w16: bf16 [K][KH*KW*C].  x is `x_dtype`, y and residual are `y_dtype` (UTV2_F32 / UTV2_BF16).  Otherwise the
// contract of utv2_conv2d_nhwc_fwd (also serves as dgrad).
// rowinfo (optional): the per-output-pixel geometry table of utv2_conv2d_wgrad_bf16 for this conv (device int32[N*OH*OW][2] =
// {input pixel index of tap (0,0), (W << 16) | tap-validity mask}; in_dil == 1 only): tile prologues load it instead of decoding it
int utv2_conv2d_nhwc_fwd_bf16_ri(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                                 const float* bias, const void* residual, const void* mask, const void* post_mask, int N, int H, int W, int C,
                                 int K, int KH, int KW, int stride, int pad, int in_dil, int OH, int OW, int relu, int accumulate,
                                 const int* rowinfo, hipStream_t stream);

int utv2_conv2d_nhwc_fwd_bf16(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                              const float* bias, const void* residual, const void* mask, const void* post_mask, int N, int H, int W, int C,
                              int K, int KH,
                              int KW, int stride, int pad, int in_dil, int OH, int OW, int relu, int accumulate, hipStream_t stream) {
  return utv2_conv2d_nhwc_fwd_bf16_ri(x, x_dtype, w16, y, y_dtype, scale, bias, residual, mask, post_mask, N, H, W, C, K, KH, KW, stride, pad,
                                      in_dil, OH, OW, relu, accumulate, nullptr, stream);
}

static int conv2d_nhwc_fwd_bf16_impl(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                                     const float* bias, const void* residual, const void* mask, const void* post_mask, int N, int H, int W, int C,
                                     int K, int KH, int KW, int stride, int pad, int in_dil, int OH, int OW, int relu, int accumulate,
                                     const int* rowinfo, EpiBits bits, hipStream_t stream);

int utv2_conv2d_nhwc_fwd_bf16_ri(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                                 const float* bias, const void* residual, const void* mask, const void* post_mask, int N, int H, int W, int C,
                                 int K, int KH, int KW, int stride, int pad, int in_dil, int OH, int OW, int relu, int accumulate,
                                 const int* rowinfo, hipStream_t stream) {
  return conv2d_nhwc_fwd_bf16_impl(x, x_dtype, w16, y, y_dtype, scale, bias, residual, mask, post_mask, N, H, W, C, K, KH, KW, stride, pad,
                                   in_dil, OH, OW, relu, accumulate, rowinfo, EpiBits{nullptr, nullptr, nullptr, nullptr, nullptr}, stream);
}

// The same with ReLU masks as BIT planes (16-bit y, K % 8 == 0; uint8 [N*OH*OW][K / 8], bit q of byte c = channel 8c + q):
//   relu_bits (optional, written): bit = the stored output is > 0 - what the backward of the ReLU needs of it;
//   mask_bits / post_mask_bits (optional, read): take the place of mask / post_mask (do not pass both forms of one mask).
// The dgrad of a bottleneck's convs then reads K / 8 bytes per pixel for a sign instead of the 2 K bytes of the forward activation.
int utv2_conv2d_nhwc_fwd_bf16_bits(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                                   const float* bias, const void* residual, const void* mask, const void* post_mask, int N, int H, int W,
                                   int C, int K, int KH, int KW, int stride, int pad, int in_dil, int OH, int OW, int relu, int accumulate,
                                   const int* rowinfo, void* relu_bits, const void* mask_bits, const void* post_mask_bits,
                                   hipStream_t stream) {
  if ((relu_bits || mask_bits || post_mask_bits) && (y_dtype != UTV2_BF16 || (K & 7))) return UTV2_EARG;
  if ((mask && mask_bits) || (post_mask && post_mask_bits)) return UTV2_EARG;
  return conv2d_nhwc_fwd_bf16_impl(x, x_dtype, w16, y, y_dtype, scale, bias, residual, mask, post_mask, N, H, W, C, K, KH, KW, stride, pad,
                                   in_dil, OH, OW, relu, accumulate, rowinfo,
                                   EpiBits{(unsigned char*)relu_bits, (const unsigned char*)mask_bits, (const unsigned char*)post_mask_bits, nullptr, nullptr},
                                   stream);
}

static int conv2d_nhwc_fwd_bf16_impl(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                                     const float* bias, const void* residual, const void* mask, const void* post_mask, int N, int H, int W, int C,
                                     int K, int KH, int KW, int stride, int pad, int in_dil, int OH, int OW, int relu, int accumulate,
                                     const int* rowinfo, EpiBits bits, hipStream_t stream) {
  if (!x || !w16 || !y || (C % 8) || bad_dtype(x_dtype) || bad_dtype(y_dtype) || (rowinfo && in_dil > 1)) return UTV2_EARG;
  ConvArgs16 a;
  a.ntiles = 0;
  a.bits = bits;
  a.lt.n = 0;
  a.x = x; a.w = (const h16_t*)w16; a.y = y; a.scale = scale; a.bias = bias; a.residual = residual; a.mask = mask; a.post_mask = post_mask;
  a.N = N; a.H = H; a.W = W; a.C = C; a.OH = OH; a.OW = OW; a.K = K; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
  a.in_dil = in_dil < 1 ? 1 : in_dil; a.relu = (relu ? 1 : 0) | g_epi_general; a.accumulate = accumulate; a.Kred = KH * KW * C; a.M = N * OH * OW;
  a.xs = C; a.m_begin = 0; a.groups = 1; a.ldy = K; a.gn_part = nullptr; a.rowinfo = (const int2*)rowinfo;
  const bool small = K <= 64;
  const int tiles = cdiv(a.M, 128) * cdiv(K, small ? 64 : 128);
  if (small) launch_igemm16<64, false>(a, tiles, x_dtype, y_dtype, stream);
  else launch_igemm16<128, false>(a, tiles, x_dtype, y_dtype, stream);
  return utv2_launch_status();
}

// Multi-level k x k 'same' conv over a level-first [P][x_pitch] matrix, optionally GROUPED and on column slices:
//   x: row pitch x_pitch elements, the conv reads channels [g*C, (g+1)*C) of a row for group g (C = input channels PER GROUP);
//   w16: bf16 [K][KH*KW*C]; y (and residual): row pitch y_pitch >= K.  x_pitch != C, y_pitch != K or groups > 1 need bf16 x with
//   C % 32 == 0, K % 4 == 0 and (K / groups) % 128 == 0.
//   gn_part (optional; bf16 y, K % 8 == 0): fp32 [ceil(P / 32)][K / 8][2], per 32-row block and 8-channel group the sum / sum of squares
//   of y as stored - the statistics pass of the GroupNorm that follows (utv2_groupnorm_relu_seg_fwd_p32).
//   rowinfo (optional): device int32[P][2], the per-output-row geometry table of utv2_conv2d_wgrad_bf16 for this conv (same pad and k):
//   the tile prologues load it instead of decoding (level, image, row, column) and the tap bounds per staged row.
static int ml_fwd_g(const void* x, int x_dtype, int x_pitch, const void* w16, void* y, int y_dtype, int y_pitch, const float* scale,
                    const float* bias, const void* residual, int nlev, const int* H_host, const int* W_host, int N, int C, int K, int KH, int KW,
                    int pad, int relu, int accumulate, int groups, float* gn_part, const int* rowinfo, EpiBits bits, hipStream_t stream) {
  if (gn_part && (y_dtype != UTV2_BF16 || (K & 7) || (y_pitch & 7) || x_dtype != UTV2_BF16 || (C % 32) || accumulate)) return UTV2_EARG;
  if (!x || !w16 || !y || nlev < 1 || nlev > CONV_MAX_LEVELS || (C % 8) || N <= 0 || bad_dtype(x_dtype) || bad_dtype(y_dtype) || groups < 1 ||
      K % groups || x_pitch < groups * C || y_pitch < K)
    return UTV2_EARG;
  const bool plain = groups == 1 && x_pitch == C && y_pitch == K && !gn_part && !rowinfo && !bits.mask_bits;
  if (!plain && (x_dtype != UTV2_BF16 || (C % 32) || (K & 3) || KH * KW > 16 || (x_pitch & 7) || (y_pitch & 7) ||
                 (groups > 1 && (K / groups) % 128)))
    return UTV2_EARG;
  ConvArgs16 a;
  a.ntiles = 0;
  a.bits = bits;
  a.M = fill_levels16(a.lt, nlev, N, H_host, W_host);
  if (!plain && ((int64_t)a.M * x_pitch >= (1ll << 31) || (int64_t)K * KH * KW * C >= (1ll << 31))) return UTV2_EARG;
  a.x = x; a.w = (const h16_t*)w16; a.y = y; a.scale = scale; a.bias = bias; a.residual = residual; a.mask = nullptr; a.post_mask = nullptr;
  a.N = N; a.H = 0; a.W = 0; a.C = C; a.OH = 0; a.OW = 0; a.K = K; a.KH = KH; a.KW = KW; a.stride = 1; a.pad = pad; a.in_dil = 1;
  a.relu = (relu ? 1 : 0) | g_epi_general; a.accumulate = accumulate; a.Kred = KH * KW * C; a.xs = x_pitch; a.m_begin = 0; a.groups = groups; a.ldy = y_pitch; a.gn_part = gn_part; a.rowinfo = (const int2*)rowinfo;
  const bool small = K <= 64 && plain;
  const int tiles = cdiv(a.M, 128) * cdiv(K, small ? 64 : 128);
  if (n96_eligible(a, x_dtype)) launch_igemm16<96, true>(a, cdiv(a.M, 256), x_dtype, y_dtype, stream);
  else if (small) launch_igemm16<64, true>(a, tiles, x_dtype, y_dtype, stream);
  else launch_igemm16<128, true>(a, tiles, x_dtype, y_dtype, stream);
  return utv2_launch_status();
}

int utv2_conv2d_ml_fwd_bf16_g(const void* x, int x_dtype, int x_pitch, const void* w16, void* y, int y_dtype, int y_pitch,
                              const float* scale, const float* bias, const void* residual, int nlev, const int* H_host,
                              const int* W_host, int N, int C, int K, int KH, int KW, int pad, int relu, int accumulate, int groups,
                              float* gn_part, const int* rowinfo, hipStream_t stream) {
  return ml_fwd_g(x, x_dtype, x_pitch, w16, y, y_dtype, y_pitch, scale, bias, residual, nlev, H_host, W_host, N, C, K, KH, KW, pad, relu,
                  accumulate, groups, gn_part, rowinfo, EpiBits{nullptr, nullptr, nullptr, nullptr, nullptr}, stream);
}

// The dgrad that produces the gradient of a GroupNorm + ReLU output (FCOS towers, fcos/fcos.py:252-304 in backward): a 16-bit level-first
// conv as utv2_conv2d_ml_fwd_bf16_g (dense y: pitch K) whose epilogue applies the ReLU mask - mask_bits: [P * K / 8] bytes, the plane
// utv2_groupnorm_relu_seg_fwd_p32b wrote - and leaves GroupNorm backward's first reduction: gnb_part fp32 [ceil(P / 64)][K][2] =
// {sum y, sum y * gnb_x} per 64-row block and channel over the rows as stored (gnb_x: [P][K], the GroupNorm's input, y's element type).
// utv2_groupnorm_seg_bwd_p64 finishes the backward from them.  C % 64 == 0, KH * KW * C >= 1024, K % 128 == 0 per group.
int utv2_conv2d_ml_fwd_bf16_gnb(const void* x, int x_pitch, const void* w16, void* y, int nlev, const int* H_host, const int* W_host, int N,
                                int C, int K, int KH, int KW, int pad, int groups, const int* rowinfo, const void* mask_bits,
                                const void* gnb_x, float* gnb_part, hipStream_t stream) {
  if (!mask_bits || !gnb_x || !gnb_part || groups < 1 || K % groups || (C % 64) || KH * KW * C < 1024 || (K / groups) % 128 || (K & 7) ||
      g_epi_general)
    return UTV2_EARG;
  return ml_fwd_g(x, UTV2_BF16, x_pitch, w16, y, UTV2_BF16, K, nullptr, nullptr, nullptr, nlev, H_host, W_host, N, C, K, KH, KW, pad, 0, 0,
                  groups, nullptr, rowinfo, EpiBits{nullptr, (const unsigned char*)mask_bits, nullptr, gnb_x, gnb_part}, stream);
}

int utv2_conv2d_ml_fwd_bf16(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                            const float* bias, const void* residual, int nlev, const int* H_host, const int* W_host, int N,
                            int C, int K, int KH, int KW, int pad, int relu, int accumulate, hipStream_t stream) {
  if (!x || !w16 || !y || nlev < 1 || nlev > CONV_MAX_LEVELS || (C % 8) || N <= 0 || bad_dtype(x_dtype) || bad_dtype(y_dtype))
    return UTV2_EARG;
  ConvArgs16 a;
  a.ntiles = 0;
  a.bits = EpiBits{nullptr, nullptr, nullptr, nullptr, nullptr};
  a.M = fill_levels16(a.lt, nlev, N, H_host, W_host);
  a.x = x; a.w = (const h16_t*)w16; a.y = y; a.scale = scale; a.bias = bias; a.residual = residual; a.mask = nullptr; a.post_mask = nullptr;
  a.N = N; a.H = 0; a.W = 0; a.C = C; a.OH = 0; a.OW = 0; a.K = K; a.KH = KH; a.KW = KW; a.stride = 1; a.pad = pad; a.in_dil = 1;
  a.relu = (relu ? 1 : 0) | g_epi_general; a.accumulate = accumulate; a.Kred = KH * KW * C; a.xs = C; a.m_begin = 0; a.groups = 1; a.ldy = K; a.gn_part = nullptr; a.rowinfo = nullptr;
  const bool small = K <= 64;
  const int tiles = cdiv(a.M, 128) * cdiv(K, small ? 64 : 128);
  if (n96_eligible(a, x_dtype)) launch_igemm16<96, true>(a, cdiv(a.M, 256), x_dtype, y_dtype, stream);
  else if (small) launch_igemm16<64, true>(a, tiles, x_dtype, y_dtype, stream);
  else launch_igemm16<128, true>(a, tiles, x_dtype, y_dtype, stream);
  return utv2_launch_status();
}

// Image stem (7x7 stride 2 pad 3 on the 3-channel image) on bf16 MFMA.  xpad16: bf16 [N][H+6][W+8][4], the normalised NHWC4
// image inside a zero border of 3 pixels (5 on the right), so no tap is ever out of bounds and every read is 16-byte
// aligned.  One kernel row kh of an output pixel reads 8 consecutive input pixels = 32 contiguous bf16 (7 taps x 4
// channels + one zero-weighted pixel): the conv is run as KH = 7, KW = 1, C = 32 with a 4-element pixel pitch.
// w16s: bf16 [K][7][32] (last 4 of each 32 zero).  H, W: the padded image canvas (even).
int utv2_conv2d_stem_fwd_bf16(const void* xpad16, const void* w16s, void* y, int y_dtype, const float* scale, const float* bias,
                              int N, int H, int W, int K, int OH, int OW, int relu, hipStream_t stream) {
  if (!xpad16 || !w16s || !y || N <= 0 || (W & 1) || (K & 3) || bad_dtype(y_dtype) || OH != (H + 6 - 7) / 2 + 1 ||
      OW != (W + 6 - 7) / 2 + 1 || (int64_t)N * (H + 6) * (W + 8) * 4 >= (1ll << 31))
    return UTV2_EARG;
  ConvArgs16 a;
  a.ntiles = 0;
  a.bits = EpiBits{nullptr, nullptr, nullptr, nullptr, nullptr};
  a.lt.n = 0;
  a.x = xpad16; a.w = (const h16_t*)w16s; a.y = y; a.scale = scale; a.bias = bias; a.residual = nullptr; a.mask = nullptr; a.post_mask = nullptr;
  a.N = N; a.H = H + 6; a.W = W + 8; a.C = 32; a.OH = OH; a.OW = OW; a.K = K; a.KH = 7; a.KW = 1; a.stride = 2; a.pad = 0;
  a.in_dil = 1; a.relu = (relu ? 1 : 0) | g_epi_general; a.accumulate = 0; a.Kred = 7 * 32; a.M = N * OH * OW; a.xs = 4; a.m_begin = 0; a.groups = 1; a.ldy = K; a.gn_part = nullptr; a.rowinfo = nullptr;
  const bool small = K <= 64;
  const int tiles = cdiv(a.M, 128) * cdiv(K, small ? 64 : 128);
  const dim3 g(tiles), b(256);
  static const bool use_tall = env_int("UTV2_CONV_TALL64", 1) != 0;
  if (small && use_tall && a.M >= 65536 && K > 32) {
    const dim3 gt(cdiv(a.M, 256));
    if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16_v2<64, false, 32, h16_t, true>), gt, b, 0, stream, a);
    else hipLaunchKernelGGL((conv_igemm_bf16_v2<64, false, 32, float, true>), gt, b, 0, stream, a);
  } else if (small) {
    if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16_v2<64, false, 32, h16_t>), g, b, 0, stream, a);
    else hipLaunchKernelGGL((conv_igemm_bf16_v2<64, false, 32, float>), g, b, 0, stream, a);
  } else {
    if (y_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_igemm_bf16_v2<128, false, 32, h16_t>), g, b, 0, stream, a);
    else hipLaunchKernelGGL((conv_igemm_bf16_v2<128, false, 32, float>), g, b, 0, stream, a);
  }
  return utv2_launch_status();
}

// The per-output-pixel geometry table of the 16-bit convs / weight gradients, built ON THE DEVICE (round 6): until then the host
// built it with torch CPU ops and copied it over - 0.3-1 s of host time and a synchronising copy for every new canvas, which is every
// step of the reference recipes (INPUT.MIN_SIZE_TRAIN (400, 1200) "range": a new padded canvas per batch).
//   rowinfo[m] = {start + n*H*W + (oh*stride - pad)*W + (ow*stride - pad),  (W << 16) | mask of the taps (kh, kw) inside the image}
// for output pixel m = (n, oh, ow) of an [N, OH, OW] output over an [N, H, W] input whose first pixel has index `start`.
__global__ __launch_bounds__(256) void rowinfo_kernel(int2* __restrict__ out, int N, int H, int W, int OH, int OW, int stride, int pad,
                                                      int KH, int KW, long long start) {
  const long long total = (long long)N * OH * OW;
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < total; m += (long long)gridDim.x * 256) {
    const int ow = (int)(m % OW);
    const long long t = m / OW;
    const int oh = (int)(t % OH), n = (int)(t / OH);
    const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
    int mask = 0;
    for (int a = 0; a < KH; ++a)
      for (int b = 0; b < KW; ++b)
        if (ih0 + a >= 0 && ih0 + a < H && iw0 + b >= 0 && iw0 + b < W) mask |= 1 << (a * KW + b);
    out[m] = make_int2((int)(start + (long long)n * H * W + (long long)ih0 * W + iw0), (W << 16) | mask);
  }
}

int utv2_rowinfo_nhwc(int* out, int N, int H, int W, int OH, int OW, int stride, int pad, int KH, int KW, int64_t start,
                      hipStream_t stream) {
  if (!out || N < 1 || H < 1 || W < 1 || OH < 1 || OW < 1 || stride < 1 || pad < 0 || KH < 1 || KW < 1 || KH * KW > 16 || W >= 32768 ||
      start < 0 || start + (int64_t)N * H * W >= (1ll << 31))
    return UTV2_EARG;
  int64_t nb = ((int64_t)N * OH * OW + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(rowinfo_kernel, dim3((int)nb), dim3(256), 0, stream, (int2*)out, N, H, W, OH, OW, stride, pad, KH, KW, (long long)start);
  return utv2_launch_status();
}

// n must be a multiple of 4; dst16: bf16[n]
int utv2_f32_to_bf16(const float* src, void* dst16, int64_t n, hipStream_t stream) {
  if (!src || !dst16 || (n & 3)) return UTV2_EARG;
  if (n == 0) return UTV2_OK;
  size_t nb = ((size_t)n / 4 + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((int)nb), dim3(256), 0, stream, src, (h16_t*)dst16, (size_t)n / 4);
  return utv2_launch_status();
}

int utv2_pad_cols_bf16(const void* src, int src_dtype, void* dst16, int64_t rows, int c, int cpad, hipStream_t stream) {
  if (!src || !dst16 || rows < 0 || c < 8 || (c & 7) || (cpad & 7) || cpad < c || bad_dtype(src_dtype)) return UTV2_EARG;
  if (rows == 0) return UTV2_OK;
  size_t nb = ((size_t)rows * (cpad / 8) + 255) / 256;
  if (nb > 8192) nb = 8192;
  if (src_dtype == UTV2_BF16)
    hipLaunchKernelGGL(pad_cols_bf16_kernel<h16_t>, dim3((int)nb), dim3(256), 0, stream, (const h16_t*)src, (h16_t*)dst16, (size_t)rows, c, cpad);
  else
    hipLaunchKernelGGL(pad_cols_bf16_kernel<float>, dim3((int)nb), dim3(256), 0, stream, (const float*)src, (h16_t*)dst16, (size_t)rows, c, cpad);
  return utv2_launch_status();
}

int utv2_weight_flip_transpose_bf16(const float* w, void* wt16, const float* scale, int K, int KH, int KW, int C,
                                    hipStream_t stream) {
  if (!w || !wt16) return UTV2_EARG;
  const size_t n = (size_t)K * KH * KW * C;
  int nb = cdiv((int64_t)n, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(weight_flip_transpose_bf16_kernel, dim3(nb), dim3(256), 0, stream, w, (h16_t*)wt16, scale, K, KH, KW, C);
  return utv2_launch_status();
}

// table: device array of nlayers records {int64 w_off, dst_off, scale_off; int32 K, KH, KW, C, Kpad, 0} (48 bytes each, see FlipDesc):
// bank[dst_off ..] = bf16 flip/transpose of arena[w_off ..] (* scales[scale_off + co] when scale_off >= 0) for every layer, the
// output-channel axis zero-padded from K to Kpad (the dgrad of a layer whose K is no multiple of 32 runs on a padded gradient).
int utv2_weight_flip_transpose_bf16_batched(const float* arena, const float* scales, void* bank, const void* table, int nlayers,
                                            hipStream_t stream) {
  if (!arena || !bank || !table || nlayers < 1) return UTV2_EARG;
  static_assert(sizeof(FlipDesc) == 48, "table record layout is part of the ABI");
  // blocks of a layer stride over its 32 x 32 tiles: 512 per layer keep the one big layer of a model (the box head's 1024 x 12544 fc1:
  // 12544 tiles; 239 us per step at 128 blocks) from serialising on a few CUs, the surplus blocks of small layers exit at once
  hipLaunchKernelGGL(weight_flip_transpose_bf16_batched_kernel, dim3(512, nlayers), dim3(256), 0, stream, arena, scales, (h16_t*)bank,
                     (const FlipDesc*)table);
  return utv2_launch_status();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// bf16 wgrad:  dW[co][k] = sum_m dY[m][co] * Xcol[m][k]   (bf16 operands, fp32 accumulate)
//
// Both GEMM operands are pixel-major in memory ([m][c], c contiguous) while the MFMA wants 8 consecutive
// REDUCTION elements (pixels m) per lane.  The tiles are therefore staged into LDS exactly as they lie in
// memory - [32 pixels][128 channels] bf16, 16-byte copies, no register shuffling - and the fragments are
// fetched with gfx950's transposing LDS read: ds_read_b64_tr_b16 hands lane i of a 16-lane group the 4-pixel
// column i of the [4 pixels][16 channels] block whose rows the group addresses (measured semantics: result
// element j of lane i = element i%4 of the 8-byte chunk addressed by lane 4j + i/4).  Two such reads form one
// 8-deep MFMA operand; A and B use the same pixel -> k-slot map, which is all the reduction needs.
// LDS rows are 256 B + 64 B pad: the 4 rows x 64 B a 32-lane half reads land on 64 distinct banks.
//
// Pixel geometry comes from a per-output-pixel table (no divisions in the loop, any mix of FPN levels / strides):
//   rowinfo[m] = { anchor = input pixel index of tap (0,0) (may lie outside the image), (W << 16) | tapmask }
// tap (kh,kw) of output pixel m reads input pixel anchor + kh*W + kw iff bit kh*KW+kw of tapmask is set.
struct Wgrad16Args {
  const void* x;
  const void* dy;
  float* ws;
  const int2* rowinfo;
  float* bias_ws;  // optional [splits][K]: per-split column sums of dY (conv bias gradient), fused into the dY staging
  int C, K, KH, KW, Kred, M, splits, chunks_per_split;
  int xs;        // elements between consecutive pixels of x (>= groups * C: x may be a channel slice of a wider matrix)
  int groups;    // grouped conv: dW rows [g*K/groups, (g+1)*K/groups) correlate dY with input channels [g*C, (g+1)*C); C, Kred per group
  int dys;       // elements between consecutive rows of dy (>= K: dy may be the first K columns of a zero-padded matrix)
  int debug;     // timing knobs of probe builds (tools/probe/conv_lockstep.h); 0 in the product
};

__device__ __forceinline__ bf16x4_t lds_read_tr16(const unsigned char* p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(bf16x4_t, v);
}

template <typename TX, typename TDY>
__global__ __launch_bounds__(256) void conv_wgrad_bf16(Wgrad16Args p) {
  constexpr int BK = 32, LDR = 320;           // pixels per chunk; LDS row stride in bytes (128 bf16 + pad)
  constexpr int OPB = BK * LDR;               // one operand tile
  constexpr bool X16 = sizeof(TX) == 2, DY16 = sizeof(TDY) == 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * OPB];  // [buf][A | B]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tilesN = (p.Kred + 127) / 128, tilesM = (p.K + 127) / 128;
  // XCD-aware order: block b runs on XCD b % 8 (observed dispatch rule; speed only).  All tiles of one pixel split
  // are given to ONE XCD so the dY / X chunks they share are fetched into that XCD's L2 once, not 8 times.
  const int tiles = tilesM * tilesN;
  int split, bid;
  if ((p.splits & 7) == 0) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    split = xcd + 8 * (idx / tiles);
    bid = idx % tiles;
  } else {
    split = blockIdx.x / tiles;
    bid = blockIdx.x - split * tiles;
  }
  const int mt = bid / tilesN, nt = bid - mt * tilesN;
  const int i0 = mt * 128, j0 = nt * 128;
  const int goff = p.groups > 1 ? (i0 / (p.K / p.groups)) * p.C : 0;  // first input channel of this tile's group

  // staging: thread -> 8-channel group cg of pixel rows pl0 and pl0 + 16 of BOTH operand tiles
  const int cg = tid & 15, pl0 = tid >> 4;
  const int co8 = i0 + cg * 8;
  const bool aok = co8 < p.K;
  const int kk = j0 + cg * 8;
  const bool bok = kk < p.Kred;
  const int tap = bok ? kk / p.C : 0;
  const int ci = bok ? kk - tap * p.C : 0;
  const int dh = tap / p.KW, dw = tap - dh * p.KW;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int chunk_begin = split * p.chunks_per_split;
  const int total_chunks = (p.M + BK - 1) / BK;
  int chunk_end = chunk_begin + p.chunks_per_split;
  if (chunk_end > total_chunks) chunk_end = total_chunks;

  int2 ri[2];
  bf16x8_t ra[2], rb[2];
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
  const bool do_bias = p.bias_ws != nullptr && nt == 0;

  // Software pipeline (one register set, like the forward kernel): per iteration ch the pieces of chunk ch+1 (loaded
  // during iteration ch-1) go registers -> LDS, each followed by the re-issue of the same piece of chunk ch+2; the pixel
  // geometry (rowinfo) runs one more chunk ahead, so no load waits on another load.
  // pieces q: 0,1 = dY rows pl0, pl0+16;  2,3 = im2col rows pl0, pl0+16
  auto rload = [&](int chunk) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      int m = chunk * BK + pl0 + 16 * r;
      m = m < p.M ? m : p.M - 1;
      ri[r] = p.rowinfo[m];
    }
  };
  auto load8 = [&](const void* base, size_t eoff, bool is16, bool ok) -> bf16x8_t {
    bf16x8_t v;
    if (is16) {
      const h16_t* src = ok ? (const h16_t*)base + eoff : (const h16_t*)g_zero64;
      v = *(const bf16x8_t*)src;
    } else {
      const float* src = ok ? (const float*)base + eoff : (const float*)g_zero64;
      const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (h16_t)v0[e];
        v[4 + e] = (h16_t)v1[e];
      }
    }
    return v;
  };
  auto load_piece = [&](int chunk, int q) {
    const int r = q & 1;
    const int m = chunk * BK + pl0 + 16 * r;
    const bool mok = m < p.M;
    if (q < 2) {
      ra[r] = load8(p.dy, (size_t)m * p.dys + co8, DY16, mok && aok);
    } else {
      const int W = ri[r].y >> 16;
      const bool ok = mok && bok && ((ri[r].y >> tap) & 1);
      rb[r] = load8(p.x, (size_t)(ri[r].x + dh * W + dw) * p.xs + goff + ci, X16, ok);
    }
  };
  auto store_piece = [&](int buf, int q) {
    const int r = q & 1;
    unsigned char* dst = smem + buf * 2 * OPB + (q < 2 ? 0 : OPB) + (pl0 + 16 * r) * LDR + cg * 16;
    if (q < 2) {
      *(bf16x8_t*)dst = ra[r];
      if (do_bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[e] += (float)ra[r][e];
      }
    } else {
      *(bf16x8_t*)dst = rb[r];
    }
  };

  // transposing fragment reads: lane (G = lane >> 4, t = lane & 15) addresses pixel row 8*(G>>1) + (t>>2),
  // channels 16*(G&1) + 4*(t&3) .. +3 of its 32-channel MFMA tile, and receives channel 16*(G&1) + t.
  const int G = lane >> 4, t = lane & 15;
  const int frag_off = (8 * (G >> 1) + (t >> 2)) * LDR + (16 * (G & 1) + 4 * (t & 3)) * 2;
  auto iteration = [&](int ch, int buf, auto do_store, auto do_load) {
    constexpr bool STORE = decltype(do_store)::value, LOAD = decltype(do_load)::value;
    const unsigned char* ab = smem + buf * 2 * OPB + frag_off;
    const unsigned char* bb = ab + OPB;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char* ap = ab + s * 16 * LDR + (wm * 64 + i * 32) * 2;
        const bf16x4_t a0 = lds_read_tr16(ap), a1 = lds_read_tr16(ap + 4 * LDR);
        const unsigned char* bp = bb + s * 16 * LDR + (wn * 64 + i * 32) * 2;
        const bf16x4_t b0 = lds_read_tr16(bp), b1 = lds_read_tr16(bp + 4 * LDR);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[i][e] = a0[e]; a[i][4 + e] = a1[e];
          b[i][e] = b0[e]; b[i][4 + e] = b1[e];
        }
      }
      MFMA_BURST_BEGIN;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16(a[i], b[j], acc[i][j]);
      MFMA_BURST_END;
      // s = 0: the im2col pieces (they consume the rowinfo registers), s = 1: the dY pieces, then the next geometry
#pragma unroll
      for (int q = (s == 0 ? 2 : 0); q < (s == 0 ? 4 : 2); ++q) {
        if constexpr (STORE) store_piece(buf ^ 1, q);
        if constexpr (LOAD) load_piece(ch + 2, q);
      }
      if constexpr (LOAD) {
        if (s == 0) rload(ch + 3);  // clamped inside; consumed by the im2col loads of the NEXT iteration
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;
  if (chunk_begin < chunk_end) {
    rload(chunk_begin);
#pragma unroll
    for (int q = 0; q < 4; ++q) load_piece(chunk_begin, q);
    rload(chunk_begin + 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) store_piece(0, q);
    if (chunk_begin + 1 < chunk_end) {
#pragma unroll
      for (int q = 0; q < 4; ++q) load_piece(chunk_begin + 1, q);
      rload(chunk_begin + 2);
    }
    __syncthreads();
    int ch = chunk_begin;
    for (; ch + 2 < chunk_end; ++ch) iteration(ch, (ch - chunk_begin) & 1, yes{}, yes{});
    if (ch + 1 < chunk_end) { iteration(ch, (ch - chunk_begin) & 1, yes{}, no{}); ++ch; }
    iteration(ch, (ch - chunk_begin) & 1, no{}, no{});
  }
  if (p.bias_ws != nullptr && nt == 0) {  // block-uniform: fixed-order combine of the 16 pixel lanes per channel group
    float* red = (float*)smem;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 128 && i0 + tid < p.K) {  // channel i0 + tid = group tid >> 3, element tid & 7
      float s = 0.f;
      for (int k = 0; k < 16; ++k) s += red[(k * 16 + (tid >> 3)) * 8 + (tid & 7)];
      p.bias_ws[(size_t)split * p.K + i0 + tid] = s;
    }
    __syncthreads();
  }

  const int frow = lane & 31, fh = lane >> 5;
  float* out = p.ws + (size_t)split * p.K * p.Kred;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = j0 + wn * 64 + j * 32 + frow;
    if (k >= p.Kred) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = i0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (co < p.K) out[(size_t)co * p.Kred + k] = acc[i][j][e];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// (the lock-step schedule of this tile, conv_wgrad_bf16_w8, lives in tools/probe/conv_lockstep.h since round 6)
#define WGRAD_W8_BP 64   // pixels per chunk of the work-item arithmetic (the ping-pong kernel runs it as two 32-pixel chunks)
// ---------------------------------------------------------------------------------------------
// conv_wgrad_bf16_w8 on the PING-PONG schedule of conv_igemm_bf16_pp (round 5).  The lock-step kernel above keeps the matrix pipe busy
// 47 % of the cycles (profiles/r05_tower_sq_counters.txt): all eight waves read their fragments, multiply and issue DMA together, so both
// waves of a SIMD sit in their non-matrix part at the same time.  Here a chunk is 32 pixels (two k16 steps = one slot of 16 MFMAs), in
// a ring of four 32 KB stages ([dY 32 px x 512 B | X 32 px x 512 B]); waves 0-3 and 4-7 run LOAD(i) | COMPUTE(i) one slot apart,
// slots separated by s_barrier: LOAD = the 24 transposing fragment reads of chunk i and the sources of the wave's 4 DMA pieces of
// chunk i+3; COMPUTE = 16 MFMAs with the geometry loads of chunk i+5 in front and the 4 bare pieces of chunk i+3 behind the 3rd / 7th /
// 11th / 15th.  Same work items, splits, slabs, accumulation order over the pixels (bit-identical slabs), same LDS row layout and swizzle.
// vmcnt: a COMPUTE slot issues [2 geometry loads, 4 pieces]; waves 0-3 wait at the end of COMPUTE(i) for everything but the 4 pieces of
// COMPUTE(i-1) and COMPUTE(i)'s 6 operations (the geometry of chunk i+4 and the pieces of chunk i+1 are in); waves 4-7 at the start
// of LOAD(i) for everything but the last 10 (geometry of chunk i+3) and at its end for everything but COMPUTE(i-1)'s 6.
__global__ __launch_bounds__(512) void conv_wgrad_bf16_pp(Wgrad16Args p) {
  constexpr int BP = 32, ROWB = 512, OPB = BP * ROWB, STAGE = 2 * OPB;   // 16 KB per operand, 32 KB per stage, 4 stages
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int tilesN = p.Kred >> 8, tiles = (p.K >> 8) * tilesN;
  const int per = gridDim.x >> 3;
  const int wi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (wi >= tiles * p.splits) return;
  const int split = wi / tiles, bid = wi - split * tiles;
  const int mt = bid / tilesN, nt = bid - mt * tilesN;
  const int i0 = mt << 8, j0 = nt << 8;
  const int tap = j0 / p.C, ci0 = j0 - tap * p.C + (p.groups > 1 ? (i0 / (p.K / p.groups)) * p.C : 0);
  const int dh = tap / p.KW, dw = tap - dh * p.KW;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // the split's pixel range in 32-pixel chunks (chunks_per_split counts the lock-step kernel's 64-pixel chunks)
  const int total_chunks = (p.M + BP - 1) / BP;
  const int chunk_begin = split * p.chunks_per_split * 2;
  int chunk_end = chunk_begin + p.chunks_per_split * 2;
  if (chunk_end > total_chunks) chunk_end = total_chunks;
  const int nslots = chunk_end - chunk_begin;
  if (nslots <= 0) {   // (an empty split still owes its slab: zeros)
    const int frow0 = lane & 31, fh0 = lane >> 5;
    float* out0 = p.ws + (size_t)split * p.K * p.Kred;
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e)
          out0[(size_t)(i0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh0) * p.Kred + j0 + wn * 64 + j * 32 + frow0] = 0.f;
    return;
  }

  // DMA role of the lane: piece q (0, 1) of an operand = pixel rows 4*wid + 2q + (lane >> 5) of the chunk, 16 bytes at physical slot lane & 31
  const int hr = lane >> 5, slot = lane & 31;
  int choff[2];
#pragma unroll
  for (int o = 0; o < 2; ++o) choff[o] = (((slot >> 2) ^ ((2 * o + hr) & 3)) << 5) + ((slot & 3) << 3);
  const h16_t* __restrict__ xb = (const h16_t*)p.x;
  const h16_t* __restrict__ dyb = (const h16_t*)p.dy;
  const h16_t* zero = (const h16_t*)g_zero64;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x2 ri[2][2];           // rowinfo of the lane's two im2col rows, two chunks in flight (set = parity of (chunk - chunk_begin))
  const int rowl = 4 * wid + hr;  // + 2q
#define WGP_RLOAD(SET)                                                                             \
  [&](int chunk) {                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                \
      int m = chunk * BP + rowl + 2 * q;                                                           \
      m = m < p.M ? m : p.M - 1;                                                                   \
      const int2* src = p.rowinfo + m;                                                             \
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(ri[SET][q]) : "v"(src));               \
    }                                                                                              \
  }
  auto rload0 = WGP_RLOAD(0);
  auto rload1 = WGP_RLOAD(1);
#undef WGP_RLOAD
  const h16_t* psrc[4];    // sources of the wave's 4 pieces of one chunk: 0, 1 dY rows, 2, 3 im2col rows
#define WGP_PREP(SET)                                                                                              \
  [&](int chunk) {                                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                                \
      const int m = chunk * BP + rowl + 2 * q;                                                                     \
      const bool in = (m < p.M) & (chunk < chunk_end);                                                             \
      const h16_t* a0 = dyb + (unsigned)(m * p.K + i0 + choff[q]);                                                 \
      psrc[q] = in ? a0 : zero;                                                                                    \
      const int W = ri[SET][q].y >> 16;                                                                            \
      const h16_t* s0 = xb + (unsigned)((ri[SET][q].x + dh * W + dw) * p.xs + ci0 + choff[q]);                     \
      psrc[2 + q] = (in & (bool)((ri[SET][q].y >> tap) & 1)) ? s0 : zero;                                          \
    }                                                                                                              \
  }
  auto prep0 = WGP_PREP(0);
  auto prep1 = WGP_PREP(1);
#undef WGP_PREP
  auto issue_piece = [&](int stage, int q4) {  // q4 0, 1: dY pieces, 2, 3: im2col pieces
    unsigned char* dst = smem + stage * STAGE + (q4 < 2 ? 0 : OPB) + (4 * wid + 2 * (q4 & 1)) * ROWB;
    __builtin_amdgcn_global_load_lds((gptr_t)psrc[q4], (lptr_t)dst, 16, 0, 0);
  };

  // transposing fragment reads (the lock-step kernel's addressing on 32-pixel stages)
  const int G = lane >> 4, t = lane & 15, r3 = t >> 2;
  const unsigned lrow = (unsigned)(size_t)(lptr_t)smem + (8 * (G >> 1) + r3) * ROWB + 32 * (G & 1) + 8 * (t & 3);
  unsigned aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = lrow + (((wm * 4 + i) ^ r3) << 6);
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = OPB + lrow + (((wn * 2 + j) ^ r3) << 6);
  typedef h16_t frag_t __attribute__((ext_vector_type(8)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  s16x4 al[2][4], ah[2][4], bl[2][2], bh[2][2];
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define READ_FRAGS(set, S)                                                                         \
  TR_READ(al[set][0], ab[0], (S) * 16 * ROWB); TR_READ(ah[set][0], ab[0], ((S) * 16 + 4) * ROWB);  \
  TR_READ(al[set][1], ab[1], (S) * 16 * ROWB); TR_READ(ah[set][1], ab[1], ((S) * 16 + 4) * ROWB);  \
  TR_READ(al[set][2], ab[2], (S) * 16 * ROWB); TR_READ(ah[set][2], ab[2], ((S) * 16 + 4) * ROWB);  \
  TR_READ(al[set][3], ab[3], (S) * 16 * ROWB); TR_READ(ah[set][3], ab[3], ((S) * 16 + 4) * ROWB);  \
  TR_READ(bl[set][0], bb[0], (S) * 16 * ROWB); TR_READ(bh[set][0], bb[0], ((S) * 16 + 4) * ROWB);  \
  TR_READ(bl[set][1], bb[1], (S) * 16 * ROWB); TR_READ(bh[set][1], bb[1], ((S) * 16 + 4) * ROWB)
#define WAIT_FRAGS2                                                                                                             \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                           \
               : "+v"(al[0][0]), "+v"(al[0][1]), "+v"(al[0][2]), "+v"(al[0][3]), "+v"(ah[0][0]), "+v"(ah[0][1]), "+v"(ah[0][2]),  \
                 "+v"(ah[0][3]), "+v"(bl[0][0]), "+v"(bl[0][1]), "+v"(bh[0][0]), "+v"(bh[0][1]));                                \
  asm volatile(""                                                                                                               \
               : "+v"(al[1][0]), "+v"(al[1][1]), "+v"(al[1][2]), "+v"(al[1][3]), "+v"(ah[1][0]), "+v"(ah[1][1]), "+v"(ah[1][2]),  \
                 "+v"(ah[1][3]), "+v"(bl[1][0]), "+v"(bl[1][1]), "+v"(bh[1][0]), "+v"(bh[1][1]))
  // all but the newest N VMEM operations of the wave have completed; ties the geometry registers so that no use moves above the wait
#define WGP_VMWAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(ri[0][0]), "+v"(ri[0][1]), "+v"(ri[1][0]), "+v"(ri[1][1]) : : "memory")
#define WGP_BARRIER                         \
  __builtin_amdgcn_sched_barrier(0);        \
  __builtin_amdgcn_s_barrier();             \
  __builtin_amdgcn_sched_barrier(0)
  auto load_slot = [&](int stage) {
    unsigned ab[4], bb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ab[i] = aoff[i] + stage * STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) bb[j] = boff[j] + stage * STAGE;
    READ_FRAGS(0, 0);
    READ_FRAGS(1, 1);
  };
  auto compute_slot = [&](int stage) {   // 16 MFMAs (steps 0, 1); the 4 pieces of the chunk three ahead go out behind the 3rd, 7th, 11th, 15th
#pragma unroll
    for (int set = 0; set < 2; ++set) {
      frag_t a[4], b[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x4_t lo = __builtin_bit_cast(bf16x4_t, al[set][i]), hi = __builtin_bit_cast(bf16x4_t, ah[set][i]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[i][e] = lo[e]; a[i][4 + e] = hi[e]; }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x4_t lo = __builtin_bit_cast(bf16x4_t, bl[set][j]), hi = __builtin_bit_cast(bf16x4_t, bh[set][j]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { b[j][e] = lo[e]; b[j][4 + e] = hi[e]; }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = mfma_32x32x16(a[i], b[j], acc[i][j]);
          const int n = (set * 4 + i) * 2 + j;
          if ((n & 3) == 2) {
            __builtin_amdgcn_sched_barrier(0);
            issue_piece(stage, n >> 2);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
  };

  // prologue: chunks 0, 1, 2 of the split (stages 0, 1, 2) and the geometry of chunks 3 (set 1) and 4 (set 0)
  const int cb = chunk_begin;
  rload0(cb);
  rload1(cb + 1);
  WGP_VMWAIT(0);
  prep0(cb);
#pragma unroll
  for (int q = 0; q < 4; ++q) issue_piece(0, q);
  prep1(cb + 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) issue_piece(1, q);
  rload0(cb + 2);
  WGP_VMWAIT(0);
  prep0(cb + 2);
#pragma unroll
  for (int q = 0; q < 4; ++q) issue_piece(2, q);
  rload1(cb + 3);
  rload0(cb + 4);
  WGP_VMWAIT(0);
  WGP_BARRIER;

  // slot i (chunk cb + i, stage i & 3): LOAD reads its fragments and prepares the pieces of chunk cb + i + 3 from geometry set (i + 1) & 1;
  // COMPUTE re-fills that set with chunk cb + i + 5 and sends the pieces into stage (i + 3) & 3
  if (wm == 0) {
    for (int i = 0; i < nslots;) {
      load_slot(i & 3); prep1(cb + i + 3); WAIT_FRAGS2; WGP_BARRIER;
      rload1(cb + i + 5); compute_slot((i + 3) & 3); __builtin_amdgcn_sched_barrier(0); WGP_VMWAIT(10); WGP_BARRIER;
      if (++i >= nslots) break;
      load_slot(i & 3); prep0(cb + i + 3); WAIT_FRAGS2; WGP_BARRIER;
      rload0(cb + i + 5); compute_slot((i + 3) & 3); __builtin_amdgcn_sched_barrier(0); WGP_VMWAIT(10); WGP_BARRIER;
      ++i;
    }
    WGP_BARRIER;
  } else {
    WGP_BARRIER;
    for (int i = 0; i < nslots;) {
      WGP_VMWAIT(10); load_slot(i & 3); prep1(cb + i + 3); WAIT_FRAGS2; WGP_VMWAIT(6); WGP_BARRIER;
      rload1(cb + i + 5); compute_slot((i + 3) & 3); __builtin_amdgcn_sched_barrier(0); WGP_BARRIER;
      if (++i >= nslots) break;
      WGP_VMWAIT(10); load_slot(i & 3); prep0(cb + i + 3); WAIT_FRAGS2; WGP_VMWAIT(6); WGP_BARRIER;
      rload0(cb + i + 5); compute_slot((i + 3) & 3); __builtin_amdgcn_sched_barrier(0); WGP_BARRIER;
      ++i;
    }
  }
  WGP_VMWAIT(0);   // the trailing (zero-page) pieces must not land in LDS after the workgroup has gone
#undef TR_READ
#undef READ_FRAGS
#undef WAIT_FRAGS2
#undef WGP_BARRIER
#undef WGP_VMWAIT

  const int frow = lane & 31, fh = lane >> 5;
  float* out = p.ws + (size_t)split * p.K * p.Kred;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = j0 + wn * 64 + j * 32 + frow;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = i0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        out[(size_t)co * p.Kred + k] = acc[i][j][e];
      }
  }
}

// Column sums of a bf16 [M][K] matrix (the bias gradient next to conv_wgrad_bf16_w8, which never holds dY in registers):
// block b sums its contiguous row range per channel -> partial[b][K]; fixed order everywhere.  K % 8 == 0, K <= 2048.
// Four row loads in flight per thread, ~4 blocks per CU (a single dependent load per thread streamed at 3 TB/s).
__global__ __launch_bounds__(256) void colsum_bf16_partial(const h16_t* __restrict__ g, float* __restrict__ partial, int M, int K,
                                                           int rows_per_block) {
  __shared__ float red[256 * 8];
  const int cpr = K >> 3;                    // 16-byte groups per row
  const int rpp = 256 / cpr;                 // rows per pass (K <= 2048)
  const int cg = threadIdx.x % cpr, rl = threadIdx.x / cpr;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float s[4][8];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) s[u][e] = 0.f;
  if (rl < rpp) {
    int r = r0 + rl;
    for (; r + 3 * rpp < r1; r += 4 * rpp) {
      bf16x8_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const bf16x8_t*)(g + (size_t)(r + u * rpp) * K + cg * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[u][e] += (float)v[u][e];
    }
    for (; r < r1; r += rpp) {
      const bf16x8_t v = *(const bf16x8_t*)(g + (size_t)r * K + cg * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[0][e] += (float)v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = (s[0][e] + s[1][e]) + (s[2][e] + s[3][e]);
  __syncthreads();
  for (int c = threadIdx.x; c < K; c += 256) {
    float a = 0.f;
    for (int k = 0; k < rpp; ++k) a += red[(k * cpr + (c >> 3)) * 8 + (c & 7)];
    partial[(size_t)blockIdx.x * K + c] = a;
  }
}

// db[c] (+)= rowscale[c] * sum_b partial[b][c]: 32 channels x 8 row parts per block, four loads in flight, fixed combine order
__global__ __launch_bounds__(256) void colsum_final_f32(const float* __restrict__ partial, float* __restrict__ db, int nb, int K,
                                                        int accumulate, const float* __restrict__ rowscale) {
  __shared__ float red[256];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), part = threadIdx.x >> 5;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < K) {
    int b = part;
    for (; b + 24 < nb; b += 32) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += partial[(size_t)(b + 8 * u) * K + c];
    }
    for (; b < nb; b += 8) s[0] += partial[(size_t)b * K + c];
  }
  red[threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (part == 0 && c < K) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v += red[threadIdx.x + 32 * k];
    if (rowscale) v *= rowscale[c];
    db[c] = accumulate ? db[c] + v : v;
  }
}

// dst[i] (+)= rowscale[i / rowlen] * sum_k ws[k][i]   (fixed order; rowscale optional: the folded FrozenBN multiplier of
// the output channel, applied here instead of to the output gradient).  Four elements per thread (16-byte accesses),
// four independent partial sums over k mod 4 combined in a fixed order: deterministic, and 4 loads in flight per thread.
__global__ __launch_bounds__(256) void reduce_slabs16_f32(const float* __restrict__ ws, float* __restrict__ dst, size_t n, int splits,
                                                          int accumulate, const float* __restrict__ rowscale, int rowlen) {
  const size_t n4 = n >> 2;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    f32x4 s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= splits; k += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += ((const f32x4*)(ws + (size_t)(k + u) * n))[i];
    }
    for (; k < splits; ++k) s[0] += ((const f32x4*)(ws + (size_t)k * n))[i];
    f32x4 t = (s[0] + s[1]) + (s[2] + s[3]);
    if (rowscale) {
      const float r = rowscale[(i * 4) / rowlen];  // rowlen % 4 == 0 (or 1 with n % 4 == 0 handled by the tail kernel)
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] *= r;
    }
    if (accumulate) t += ((const f32x4*)dst)[i];
    ((f32x4*)dst)[i] = t;
  }
}

// scalar form (bias slabs: n = K, one "row" per element)
__global__ void reduce_slabs16_scalar_f32(const float* __restrict__ ws, float* __restrict__ dst, size_t n, int splits, int accumulate,
                                          const float* __restrict__ rowscale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += ws[(size_t)k * n + i];
  if (rowscale) s *= rowscale[i];
  dst[i] = accumulate ? dst[i] + s : s;
}

// ---------------------------------------------------------------------------------------------
// The split-K tails of SEVERAL weight-gradient launches as ONE launch (round 6).  A step has ~56 weight-gradient launches, each followed
// by its slab reduction (+ a bias reduction): 60-70 dispatches of 5-25 us that sit in their own launch gaps on the weight-gradient
// lanes.  With a pending table (utv2_conv2d_wgrad_bf16_d) a launch only RECORDS its tail - {slabs, destination, splits, scale} - and
// utv2_wgrad_fold_flush runs all recorded tails in one kernel: workgroups are dealt to the entries in proportion to their size, and every
// entry is reduced with exactly the arithmetic (order of the partial sums included) of the kernel it replaces:
//   kind 0 = reduce_slabs16_f32, kind 1 = reduce_slabs16_scalar_f32, kind 2 = colsum_final_f32  ->  bit-identical gradients.
#define SLAB_FOLD_MAX 24
struct SlabFold {
  const float* ws;          // kind 0 / 1: [splits][n] slabs; kind 2: [splits = row blocks][n = K] partial column sums
  float* dst;
  const float* rowscale;
  unsigned long long n;
  int splits, accumulate, rowlen, kind;
  int blk0, nblk;           // this entry's workgroups: [blk0, blk0 + nblk)
};
struct SlabFoldTable {
  int count, total_blocks;
  SlabFold e[SLAB_FOLD_MAX];
};

__global__ __launch_bounds__(256) void reduce_slabs_table(SlabFoldTable t) {
  __shared__ float red[256];
  int ei = 0;
  for (int i = 1; i < t.count; ++i)
    if ((int)blockIdx.x >= t.e[i].blk0) ei = i;
  const SlabFold f = t.e[ei];
  const int lb = (int)blockIdx.x - f.blk0;
  if (f.kind == 0) {          // == reduce_slabs16_f32
    const size_t n4 = f.n >> 2;
    const size_t stride = (size_t)f.nblk * 256;
    for (size_t i = (size_t)lb * 256 + threadIdx.x; i < n4; i += stride) {
      f32x4 s[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      int k = 0;
      for (; k + 4 <= f.splits; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += ((const f32x4*)(f.ws + (size_t)(k + u) * f.n))[i];
      }
      for (; k < f.splits; ++k) s[0] += ((const f32x4*)(f.ws + (size_t)k * f.n))[i];
      f32x4 v = (s[0] + s[1]) + (s[2] + s[3]);
      if (f.rowscale) {
        const float r = f.rowscale[(i * 4) / f.rowlen];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= r;
      }
      if (f.accumulate) v += ((const f32x4*)f.dst)[i];
      ((f32x4*)f.dst)[i] = v;
    }
  } else if (f.kind == 1) {   // == reduce_slabs16_scalar_f32
    for (size_t i = (size_t)lb * 256 + threadIdx.x; i < f.n; i += (size_t)f.nblk * 256) {
      float s = 0.f;
      for (int k = 0; k < f.splits; ++k) s += f.ws[(size_t)k * f.n + i];
      if (f.rowscale) s *= f.rowscale[i];
      f.dst[i] = f.accumulate ? f.dst[i] + s : s;
    }
  } else {                    // == colsum_final_f32 (one workgroup per 32 channels; the branch is workgroup-uniform)
    const int K = (int)f.n, nb = f.splits;
    const int c = lb * 32 + (threadIdx.x & 31), part = threadIdx.x >> 5;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < K) {
      int b = part;
      for (; b + 24 < nb; b += 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] += f.ws[(size_t)(b + 8 * u) * K + c];
      }
      for (; b < nb; b += 8) s[0] += f.ws[(size_t)b * K + c];
    }
    red[threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (part == 0 && c < K) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) v += red[threadIdx.x + 32 * k];
      if (f.rowscale) v *= f.rowscale[c];
      f.dst[c] = f.accumulate ? f.dst[c] + v : v;
    }
  }
}

static int slab_fold_add(SlabFoldTable* t, int kind, const float* ws, float* dst, const float* rowscale, size_t n, int splits, int accumulate,
                         int rowlen) {
  if (t->count < 0 || t->count >= SLAB_FOLD_MAX) return UTV2_EARG;
  for (int i = 0; i < t->count; ++i)   // two tails of one table run concurrently: they must not write the same gradient
    if (t->e[i].dst == dst) return UTV2_EARG;
  SlabFold& f = t->e[t->count];
  f.kind = kind; f.ws = ws; f.dst = dst; f.rowscale = rowscale; f.n = n; f.splits = splits; f.accumulate = accumulate; f.rowlen = rowlen;
  int nb;
  if (kind == 0) { nb = cdiv((int64_t)n / 4, 256); if (nb > 2048) nb = 2048; }
  else if (kind == 1) nb = cdiv((int64_t)n, 256);
  else nb = cdiv((int64_t)n, 32);
  if (nb < 1) nb = 1;
  f.blk0 = t->count == 0 ? 0 : t->e[t->count - 1].blk0 + t->e[t->count - 1].nblk;
  f.nblk = nb;
  t->total_blocks = f.blk0 + nb;
  ++t->count;
  return UTV2_OK;
}

extern "C" {

static int wgrad16_small_splits(int M, int K, int Kred) {
  const int chunks = cdiv(M, 32);
  const int tiles = cdiv(K, 128) * cdiv(Kred, 128);
  // ~2 workgroups per CU: every split costs a K x Kred fp32 slab written and re-read, which rivals the operand traffic
  // of the HBM-bound layers; ~4 per CU only when a workgroup's share of the MFMA work is large (load balance wins)
  const double flop = 2.0 * M * K * Kred;
  // (round 6 A/B of this budget at 75 % / 50 % - slab bytes = workgroups x 64 KB, ~3 GB written + re-read per step: FCOS 2+2 +0.5 %,
  // FCOS 4+4 -0.6 / -1.0 %, Faster-RCNN 2+2 -1.3 %, profiles/r06_wgs_pct_ab.txt: the load balance is worth the slabs)
  int splits = cdiv(flop > 512 * 2.0e8 ? 1024 : 512, tiles);
  const int max_by_chunks = chunks / 8 > 0 ? chunks / 8 : 1;
  if (splits > max_by_chunks) splits = max_by_chunks;
  if (splits < 1) splits = 1;
  if (splits > 256) splits = 256;
  if (splits >= 8) splits = (splits + 7) / 8 * 8 <= max_by_chunks ? (splits + 7) / 8 * 8 : splits / 8 * 8;  // one split group per XCD
  return splits;
}

// the 256 x 256-tile kernel: one workgroup per CU in ONE round - as many pixel splits as fit 256 work items, at least 8 chunks each
#define WGRAD_W8_MAX_NB 1024  // row blocks of the bias column-sum pass
// Workgroups (= CUs: one 8-wave, 512-VGPR workgroup fills a CU) a launch may occupy.  Fewer than 256 leaves CUs free for the main
// stream while the weight gradients run on their side stream (UTV2_WGRAD_W8_WGS, a multiple of 8, read once).
static int wgrad16_w8_budget() {
  static const int w = [] {
    int v = env_int("UTV2_WGRAD_W8_WGS", 256);
    v = (v / 8) * 8;
    return v < 8 ? 8 : (v > 256 ? 256 : v);
  }();
  return w;
}
static int wgrad16_w8_splits(int M, int K, int Kred) {
  const int tiles = (K >> 8) * (Kred >> 8), chunks = cdiv(M, WGRAD_W8_BP);
  int splits = wgrad16_w8_budget() / tiles;
  if (splits > chunks / 8) splits = chunks / 8;
  return splits < 1 ? 1 : splits;
}
static bool wgrad16_w8_shape_ok(int M, int C, int K, int KH, int KW, int xs, int groups) {
  const int Kred = KH * KW * C, tiles = (K >> 8) * (Kred >> 8), chunks = cdiv(M, WGRAD_W8_BP);
  if ((K & 255) || (C & 255) || KH * KW > 16 || K > 2048 || tiles > wgrad16_w8_budget() || chunks < 16 || (int64_t)M * K >= (1ll << 31) ||
      (int64_t)M * xs >= (1ll << 31) || (xs & 7) || ((K / groups) & 255))
    return false;
  // 1x1 layers are HBM-bound: every pixel split costs a K x Kred fp32 slab written and re-read, and this kernel needs 256 / tiles of
  // them to fill the chip - it pays there only when a split still covers >= 16 chunks (measured: 256 -> 512 stride 2 at M = 134 400
  // 424 -> 493 TF, the M = 33 600 / 8 400 layers 10-20 % slower)
  if (KH * KW == 1 && chunks < 16 * (256 / tiles)) return false;
  return true;
}

int utv2_conv2d_wgrad_bf16_splits(int M, int K, int Kred) { return wgrad16_small_splits(M, K, Kred); }

// large enough for either kernel (which one runs also depends on the element types)
// db[c] (+)= sum_b partial[b][c] for a [nb][K] matrix of per-block column sums (e.g. utv2_groupnorm_relu_seg_bwd_colsum's): 32 channels x
// 8 row parts per workgroup, fixed combine order
int utv2_colsum_partials(const float* partial, float* db, int nb, int K, int accumulate, hipStream_t stream) {
  if (!partial || !db || nb < 1 || K < 1) return UTV2_EARG;
  hipLaunchKernelGGL(colsum_final_f32, dim3(cdiv(K, 32)), dim3(256), 0, stream, partial, db, nb, K, accumulate, (const float*)nullptr);
  return utv2_launch_status();
}

int64_t utv2_conv2d_wgrad_bf16_workspace_floats(int M, int K, int Kred) {
  int64_t n = (int64_t)wgrad16_small_splits(M, K, Kred) * ((int64_t)K * Kred + K);
  if ((K & 255) == 0 && (Kred & 255) == 0) {
    const int64_t n8 = (int64_t)wgrad16_w8_splits(M, K, Kred) * K * Kred + (int64_t)WGRAD_W8_MAX_NB * K;
    if (n8 > n) n = n8;
  }
  return n;
}

// rowinfo: device int32[M][2] = {anchor input pixel, (W << 16) | tapmask} for every OUTPUT pixel m (built once per
// geometry by the host).  x is `x_dtype` with pixel pitch x_pitch elements, dy is `dy_dtype` [M][K].  C % 8 == 0, K % 8 == 0,
// KH*KW <= 16.  dw [K][KH*KW*C] (+)= rowscale[co] * result; db (optional, [K]) (+)= rowscale[co] * column sums of dy; rowscale optional.
// groups > 1: grouped conv - output channels [g*K/groups, ...) correlate with input channels [g*C, (g+1)*C) (C per group),
// (K / groups) % 128 == 0, x_pitch >= groups * C.
static int wgrad_bf16_impl(const void* x, int x_dtype, int x_pitch, const void* dy, int dy_dtype, int dy_pitch, float* dw, float* db,
                           float* ws, const int* rowinfo, const float* rowscale, int M, int C, int K, int KH, int KW, int accumulate,
                           int groups, hipStream_t stream, SlabFoldTable* pending = nullptr);

int utv2_conv2d_wgrad_bf16(const void* x, int x_dtype, const void* dy, int dy_dtype, float* dw, float* db, float* ws,
                           const int* rowinfo, const float* rowscale, int M, int C, int K, int KH, int KW, int accumulate,
                           hipStream_t stream) {
  return wgrad_bf16_impl(x, x_dtype, C, dy, dy_dtype, K, dw, db, ws, rowinfo, rowscale, M, C, K, KH, KW, accumulate, 1, stream);
}

int utv2_conv2d_wgrad_bf16_g(const void* x, int x_dtype, int x_pitch, const void* dy, int dy_dtype, int dy_pitch, float* dw, float* db,
                             float* ws, const int* rowinfo, const float* rowscale, int M, int C, int K, int KH, int KW, int accumulate,
                             int groups, hipStream_t stream) {
  return wgrad_bf16_impl(x, x_dtype, x_pitch, dy, dy_dtype, dy_pitch, dw, db, ws, rowinfo, rowscale, M, C, K, KH, KW, accumulate, groups,
                         stream);
}

// The same with the split-K tail RECORDED in a caller-owned pending table instead of launched (see reduce_slabs_table): dw / db hold the
// result only after utv2_wgrad_fold_flush(pending, stream) on the same stream; ws must stay untouched until then (a separate workspace per
// recorded launch), and two recorded launches of one table must not share dw or db (the call refuses: flush first).  Room for 8 launches.
int utv2_conv2d_wgrad_bf16_d(const void* x, int x_dtype, int x_pitch, const void* dy, int dy_dtype, int dy_pitch, float* dw, float* db,
                             float* ws, const int* rowinfo, const float* rowscale, int M, int C, int K, int KH, int KW, int accumulate,
                             int groups, void* pending, hipStream_t stream) {
  if (!pending) return UTV2_EARG;
  SlabFoldTable* t = (SlabFoldTable*)pending;
  if (t->count < 0 || t->count + 3 > SLAB_FOLD_MAX) return UTV2_EARG;
  return wgrad_bf16_impl(x, x_dtype, x_pitch, dy, dy_dtype, dy_pitch, dw, db, ws, rowinfo, rowscale, M, C, K, KH, KW, accumulate, groups,
                         stream, t);
}

int64_t utv2_wgrad_fold_table_bytes(void) { return (int64_t)sizeof(SlabFoldTable); }

int utv2_wgrad_fold_pending(const void* pending) { return pending ? ((const SlabFoldTable*)pending)->count : -1; }

int utv2_wgrad_fold_flush(void* pending, hipStream_t stream) {
  if (!pending) return UTV2_EARG;
  SlabFoldTable* t = (SlabFoldTable*)pending;
  if (t->count < 0 || t->count > SLAB_FOLD_MAX) return UTV2_EARG;
  if (t->count == 0) return UTV2_OK;
  hipLaunchKernelGGL(reduce_slabs_table, dim3(t->total_blocks), dim3(256), 0, stream, *t);
  t->count = 0;
  t->total_blocks = 0;
  return utv2_launch_status();
}

static int wgrad_bf16_impl(const void* x, int x_dtype, int x_pitch, const void* dy, int dy_dtype, int dy_pitch, float* dw, float* db,
                           float* ws, const int* rowinfo, const float* rowscale, int M, int C, int K, int KH, int KW, int accumulate,
                           int groups, hipStream_t stream, SlabFoldTable* pending) {
  if (!x || !dy || !dw || !ws || !rowinfo || (C & 7) || (K & 7) || M <= 0 || KH * KW > 16 || bad_dtype(x_dtype) ||
      bad_dtype(dy_dtype) || groups < 1 || K % groups || x_pitch < groups * C || (groups > 1 && (K / groups) % 128) ||
      (x_pitch != C && (x_pitch & (x_dtype == UTV2_BF16 ? 7 : 3))) || dy_pitch < K || (dy_pitch != K && (dy_pitch & (dy_dtype == UTV2_BF16 ? 7 : 3))))
    return UTV2_EARG;
  Wgrad16Args a;
  a.x = x; a.dy = dy; a.ws = ws; a.rowinfo = (const int2*)rowinfo;
  a.C = C; a.K = K; a.KH = KH; a.KW = KW; a.Kred = KH * KW * C; a.M = M; a.debug = 0; a.xs = x_pitch; a.groups = groups; a.dys = dy_pitch;
  if (g_use_wgrad_w8 && x_dtype == UTV2_BF16 && dy_dtype == UTV2_BF16 && dy_pitch == K && wgrad16_w8_shape_ok(M, C, K, KH, KW, x_pitch, groups)) {
    a.splits = wgrad16_w8_splits(M, K, a.Kred);
    a.chunks_per_split = cdiv(cdiv(M, WGRAD_W8_BP), a.splits);
    a.bias_ws = nullptr;
    const size_t n = (size_t)K * a.Kred;
    const int smem = 2 * 2 * WGRAD_W8_BP * 512;
    static LdsOptIn pp_opt_in;
    pp_opt_in({(const void*)conv_wgrad_bf16_pp}, smem);
    hipLaunchKernelGGL(conv_wgrad_bf16_pp, dim3(wgrad16_w8_budget()), dim3(512), smem, stream, a);
    int rb = cdiv((int64_t)n / 4, 256);
    if (rb > 8192) rb = 8192;
    if (pending) {
      if (int rc = slab_fold_add(pending, 0, ws, dw, rowscale, n, a.splits, accumulate, a.Kred)) return rc;
    } else
      hipLaunchKernelGGL(reduce_slabs16_f32, dim3(rb), dim3(256), 0, stream, (const float*)ws, dw, n, a.splits, accumulate, rowscale,
                         a.Kred);
    if (db) {
      float* part = ws + (size_t)a.splits * n;
      int nb = cdiv(M, 64);
      if (nb > WGRAD_W8_MAX_NB) nb = WGRAD_W8_MAX_NB;
      const int rows = cdiv(M, nb);
      nb = cdiv(M, rows);
      hipLaunchKernelGGL(colsum_bf16_partial, dim3(nb), dim3(256), 0, stream, (const h16_t*)dy, part, M, K, rows);
      if (pending) {
        if (int rc = slab_fold_add(pending, 2, part, db, rowscale, (size_t)K, nb, accumulate, 1)) return rc;
      } else
        hipLaunchKernelGGL(colsum_final_f32, dim3(cdiv(K, 32)), dim3(256), 0, stream, (const float*)part, db, nb, K, accumulate, rowscale);
    }
    return utv2_launch_status();
  }
  a.splits = utv2_conv2d_wgrad_bf16_splits(M, K, a.Kred);
  a.chunks_per_split = cdiv(cdiv(M, 32), a.splits);
  const size_t n = (size_t)K * a.Kred;
  a.bias_ws = db ? ws + (size_t)a.splits * n : nullptr;   // bias slabs sit behind the weight slabs
  const int tiles = cdiv(K, 128) * cdiv(a.Kred, 128);
  const dim3 g(tiles * a.splits), b(256);
  if (x_dtype == UTV2_BF16) {
    if (dy_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_wgrad_bf16<h16_t, h16_t>), g, b, 0, stream, a);
    else hipLaunchKernelGGL((conv_wgrad_bf16<h16_t, float>), g, b, 0, stream, a);
  } else {
    if (dy_dtype == UTV2_BF16) hipLaunchKernelGGL((conv_wgrad_bf16<float, h16_t>), g, b, 0, stream, a);
    else hipLaunchKernelGGL((conv_wgrad_bf16<float, float>), g, b, 0, stream, a);
  }
  int rb = cdiv((int64_t)n / 4, 256);   // n = K * Kred, both multiples of 8
  if (rb > 8192) rb = 8192;
  if (pending) {
    if (int rc = slab_fold_add(pending, 0, ws, dw, rowscale, n, a.splits, accumulate, a.Kred)) return rc;
    if (db)
      if (int rc = slab_fold_add(pending, 1, a.bias_ws, db, rowscale, (size_t)K, a.splits, accumulate, 1)) return rc;
    return utv2_launch_status();
  }
  hipLaunchKernelGGL(reduce_slabs16_f32, dim3(rb), dim3(256), 0, stream, (const float*)ws, dw, n, a.splits, accumulate, rowscale,
                       a.Kred);
  if (db)
    hipLaunchKernelGGL(reduce_slabs16_scalar_f32, dim3(cdiv(K, 256)), dim3(256), 0, stream, (const float*)a.bias_ws, db, (size_t)K,
                       a.splits, accumulate, rowscale);
  return utv2_launch_status();
}

}  // extern "C"
