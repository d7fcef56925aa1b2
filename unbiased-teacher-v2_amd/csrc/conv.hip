// NHWC implicit-GEMM convolution for gfx950 on the f32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32 FMA chain, 157 TF chip peak).
//
// Replaces, for the UTv2 hot path, the cuDNN/MIOpen conv2d forward / dgrad /
// wgrad reached from Detectron2's ResNet/FPN and from
// ubteacher/modeling/fcos/fcos.py:252-304, ubteacher/modeling/backbone/fpn.py:21-22
// (see SURVEY.md 2.1).  Layouts (all fp32, all device pointers):
//   activations  [N][H][W][C]            (channels innermost)
//   weights      [Cout][KH][KW][Cin]     (one GEMM row per output channel)
// GEMM view (forward):  Y[m][co] = sum_k A[m][k] * Wt[co][k],
//   m = (n, oh, ow) row-major, k = (kh, kw, ci) row-major.
// Block = 256 threads = 4 waves (2x2), tile 128 x BN x 16, each wave owns a
// 64 x BN/2 sub-tile as TM x TN 32x32 MFMA accumulators.  Operands are staged
// global -> registers -> LDS ([row][k], 80-byte rows: conflict-free for the
// ds_read_b128 fragment reads and 16-byte aligned for the ds_write_b128 stores),
// double-buffered so the next chunk's global loads are in flight during the MFMAs.
// dgrad is the same kernel run on dY with the flipped/transposed weight image and
// an input-dilation predicate (for strided forward convs).
#include "common.h"

#define CONV_MAX_LEVELS 8
// Multi-level ("ragged") activations: rows [start[l], start[l+1]) hold N images of H[l] x W[l] pixels
// (level-first layout of DESIGN.md section 2), so ONE launch covers every FPN level of a shared head.
struct LevelTab {
  int n;                         // 0 = single dense NHWC tensor
  int start[CONV_MAX_LEVELS + 1];
  int H[CONV_MAX_LEVELS], W[CONV_MAX_LEVELS];
};

struct ConvArgs {
  LevelTab lt;
  const float* x;
  const float* w;
  float* y;
  const float* scale;     // per-Cout multiplier (folded FrozenBN) or null
  const float* bias;      // per-Cout addend or null
  const float* residual;  // same shape as y, or null
  int N, H, W, C;         // input tensor
  int OH, OW, K;          // output spatial, output channels
  int KH, KW, stride, pad, in_dil;
  int relu;
  int Kred;               // reduction length of one weight row (may be padded, MODE 1)
  int M;                  // N*OH*OW
  int accumulate;         // y += result (used by dgrad into an existing gradient)
  int y_bf16;             // store y as bf16 (the image stem under AMP: the bf16 activation pipeline starts at its output)
};

// MODE 0: C % 16 == 0 (a 16-wide k chunk never straddles a filter tap)
// MODE 1: C == 4 (stem on the NHWC4-padded image): each float4 is one tap
// MODE 2: generic scalar gather (any C), weight rows unaligned
__device__ __forceinline__ void ml_decode(const LevelTab& lt, int m, int& pixbase, int& H, int& W, int& oh, int& ow) {
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

template <int BN, int MODE, bool ML = false>
__global__ __launch_bounds__(256) void conv_igemm_f32(ConvArgs p) {
  constexpr int BM = 128, BK = 16, LDK = 20;
  constexpr int TM = 2, TN = BN / 64;
  constexpr int BROWS = BN / 64;  // B rows per thread
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDK];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // XCD-aware tile order: each XCD (bid % 8) walks a contiguous range of tiles so
  // neighbouring M tiles (shared halo rows, same weights) hit one L2.
  const int tilesN = (p.K + BN - 1) / BN;
  const int nwg = gridDim.x;
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mt = tile / tilesN, nt = tile - mt * tilesN;
  const int m0 = mt * BM, n0 = nt * BN;

  const int lrow = tid >> 2, lk4 = tid & 3;

  // per-thread A rows
  int ih0[2], iw0[2];
  long long pixbase[2];
  int Hr[2], Wr[2];
  bool mvalid[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int m = m0 + lrow + 64 * r;
    mvalid[r] = m < p.M;
    const int mm = mvalid[r] ? m : 0;
    if constexpr (ML) {
      int pb, oh, ow;
      ml_decode(p.lt, mm, pb, Hr[r], Wr[r], oh, ow);
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
  const float* wrow[BROWS];
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

  const int nchunks = (p.Kred + BK - 1) / BK;
  int kh = 0, kw = 0, c0 = 0;  // MODE 0 chunk state

  f32x4 ra[2], rb[BROWS];

  auto gload = [&](int kc) {
    if constexpr (MODE == 0) {
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
        ok = ok && (unsigned)ih < (unsigned)Hr[r] && (unsigned)iw < (unsigned)Wr[r];
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *(const f32x4*)(p.x + (size_t)(pixbase[r] + (long long)ih * Wr[r] + iw) * p.C + c0 + lk4 * 4);
        ra[r] = v;
      }
#pragma unroll
      for (int r = 0; r < BROWS; ++r) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (bvalid[r]) v = *(const f32x4*)(wrow[r] + (kh * p.KW + kw) * p.C + c0 + lk4 * 4);
        rb[r] = v;
      }
      // taps innermost: the KH*KW shifted reads of one channel slab stay L1/L2 resident
      if (++kw == p.KW) {
        kw = 0;
        if (++kh == p.KH) { kh = 0; c0 += BK; }
      }
    } else if constexpr (MODE == 1) {
      const int tap = kc * 4 + lk4;
      const int tkh = tap / p.KW, tkw = tap - tkh * p.KW;
      const bool tapok = tap < p.KH * p.KW;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int ih = ih0[r] + tkh, iw = iw0[r] + tkw;
        const bool ok = tapok && mvalid[r] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *(const f32x4*)(p.x + (size_t)(pixbase[r] + (long long)ih * p.W + iw) * 4);
        ra[r] = v;
      }
#pragma unroll
      for (int r = 0; r < BROWS; ++r) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (bvalid[r]) v = *(const f32x4*)(wrow[r] + kc * BK + lk4 * 4);
        rb[r] = v;
      }
    } else {
      const int kbase = kc * BK + lk4 * 4;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = kbase + e;
          if (k < p.Kred && mvalid[r]) {
            const int tap = k / p.C, ci = k - tap * p.C;
            const int tkh = tap / p.KW, tkw = tap - tkh * p.KW;
            int ihn = ih0[r] + tkh, iwn = iw0[r] + tkw;
            bool ok = true;
            int ih = ihn, iw = iwn;
            if (p.in_dil > 1) {
              ok = (ihn % p.in_dil == 0) && (iwn % p.in_dil == 0);
              ih = ihn / p.in_dil;
              iw = iwn / p.in_dil;
            }
            ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            if (ok) v[e] = p.x[(size_t)(pixbase[r] + (long long)ih * p.W + iw) * p.C + ci];
          }
        }
        ra[r] = v;
      }
#pragma unroll
      for (int r = 0; r < BROWS; ++r) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = kbase + e;
          if (bvalid[r] && k < p.Kred) v[e] = wrow[r][k];
        }
        rb[r] = v;
      }
    }
  };

  auto lds_store = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
      *(f32x4*)(As + buf * BM * LDK + (lrow + 64 * r) * LDK + lk4 * 4) = ra[r];
#pragma unroll
    for (int r = 0; r < BROWS; ++r)
      *(f32x4*)(Bs + buf * BN * LDK + (lrow + 64 * r) * LDK + lk4 * 4) = rb[r];
  };

  gload(0);
  lds_store(0);
  __syncthreads();

  const int frow = lane & 31, fh = lane >> 5;
  for (int kc = 0; kc < nchunks; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nchunks) gload(kc + 1);

    f32x4 a[TM][2], b[TN][2];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float* ap = As + buf * BM * LDK + (wm * 64 + i * 32 + frow) * LDK + fh * 8;
      a[i][0] = *(const f32x4*)ap;
      a[i][1] = *(const f32x4*)(ap + 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float* bp = Bs + buf * BN * LDK + (wn * (BN / 2) + j * 32 + frow) * LDK + fh * 8;
      b[j][0] = *(const f32x4*)bp;
      b[j][1] = *(const f32x4*)(bp + 4);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s >> 2][s & 3], b[j][s >> 2][s & 3], acc[i][j], 0, 0, 0);

    if (kc + 1 < nchunks) lds_store(buf ^ 1);
    __syncthreads();
  }

  if ((p.K & 3) == 0) {
    // all MFMAs retired and every wave is past the last barrier of the K loop: the staging LDS is free
    float* patch = (float*)smem + wid * (32 * ((BN / 2) + 4));
    if constexpr (MODE == 1) {
      if (p.y_bf16) {  // no residual / accumulate on this path
        epilogue_rows<TN, h16_t>(acc, patch, lane, (h16_t*)p.y, p.scale, p.bias, (const h16_t*)nullptr, p.relu, 0, m0 + wm * 64,
                                  n0 + wn * (BN / 2), p.M, p.K);
        return;
      }
    }
    epilogue_rows<TN>(acc, patch, lane, p.y, p.scale, p.bias, p.residual, p.relu, p.accumulate, m0 + wm * 64,
                      n0 + wn * (BN / 2), p.M, p.K);
    return;
  }
  // epilogue: D[row][col]: col = lane&31 (cout), row = (e&3) + 8*(e>>2) + 4*(lane>>5)
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
        const size_t off = (size_t)m * p.K + co;
        float v = acc[i][j][e] * sc + bi;
        if (p.residual) v += p.residual[off];
        if (p.relu) v = fmaxf(v, 0.f);
        if (p.accumulate) v += p.y[off];
        p.y[off] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// wgrad:  dW[co][tap][ci] = sum_m dY[m][co] * X[n, ih(m,tap), iw(m,tap), ci]
// GEMM rows = co, cols = k = (tap, ci), reduction over output pixels m, split into `splits`
// contiguous pixel ranges; every split writes its own [K][Kred] slab (deterministic), a second
// kernel sums the slabs into (accumulates onto) the gradient.
struct WgradArgs {
  LevelTab lt;
  const float* x;   // forward input  [N][H][W][C]
  const float* dy;  // output grad    [N][OH][OW][K]
  float* ws;        // [splits][K][Kred]
  int N, H, W, C, OH, OW, K, KH, KW, stride, pad;
  int Kred, M, splits, chunks_per_split;
};

template <int VEC, bool ML = false>
__global__ __launch_bounds__(256) void conv_wgrad_f32(WgradArgs p) {
  constexpr int BM = 128, BN = 128, BK = 16;
  constexpr int TM = 2, TN = 2;
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * (BM + BN)];
  float* As = smem;                    // [2][BK][BM]  dY chunk
  float* Bs = smem + 2 * BK * BM;      // [2][BK][BN]  im2col chunk

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int tilesN = (p.Kred + BN - 1) / BN;
  const int tilesM = (p.K + BM - 1) / BM;
  int bid = blockIdx.x;
  const int split = bid / (tilesM * tilesN);
  bid -= split * tilesM * tilesN;
  const int mt = bid / tilesN, nt = bid - mt * tilesN;
  const int i0 = mt * BM, j0 = nt * BN;

  const int lp = tid >> 5;   // pixel row within pass: 0..7 (+8 second pass)
  const int c4 = tid & 31;   // float4 column

  // this thread's fixed (tap, ci) for the im2col operand
  const int kidx = j0 + c4 * 4;
  int tkh[4], tkw[4], tci[4];
  bool kok[4];
  if constexpr (VEC) {
    kok[0] = kidx < p.Kred;
    const int kk = kok[0] ? kidx : 0;
    const int tap = kk / p.C;
    tci[0] = kk - tap * p.C;
    tkh[0] = tap / p.KW;
    tkw[0] = tap - tkh[0] * p.KW;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kok[e] = kidx + e < p.Kred;
      const int kk = kok[e] ? kidx + e : 0;
      const int tap = kk / p.C;
      tci[e] = kk - tap * p.C;
      tkh[e] = tap / p.KW;
      tkw[e] = tap - tkh[e] * p.KW;
    }
  }
  const int coidx = i0 + c4 * 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int chunk_begin = split * p.chunks_per_split;
  const int total_chunks = (p.M + BK - 1) / BK;
  int chunk_end = chunk_begin + p.chunks_per_split;
  if (chunk_end > total_chunks) chunk_end = total_chunks;

  f32x4 ra[2], rb[2];
  const int hw = p.OH * p.OW;

  auto gload = [&](int chunk) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int m = chunk * BK + lp + 8 * r;
      const bool mv = m < p.M;
      const int mm = mv ? m : 0;
      int n, oh, ow, Hh = p.H, Ww = p.W;
      long long pb;
      if constexpr (ML) {
        int pbi;
        ml_decode(p.lt, mm, pbi, Hh, Ww, oh, ow);
        pb = pbi;
        n = 0;
      } else {
        n = mm / hw;
        const int rem = mm - n * hw;
        oh = rem / p.OW;
        ow = rem - oh * p.OW;
        pb = (long long)n * p.H * p.W;
      }
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if constexpr (VEC) {
        if (mv && coidx < p.K) va = *(const f32x4*)(p.dy + (size_t)mm * p.K + coidx);
        const int ih = oh * p.stride - p.pad + tkh[0], iw = ow * p.stride - p.pad + tkw[0];
        if (mv && kok[0] && (unsigned)ih < (unsigned)Hh && (unsigned)iw < (unsigned)Ww)
          vb = *(const f32x4*)(p.x + (size_t)(pb + (long long)ih * Ww + iw) * p.C + tci[0]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (mv && coidx + e < p.K) va[e] = p.dy[(size_t)mm * p.K + coidx + e];
          const int ih = oh * p.stride - p.pad + tkh[e], iw = ow * p.stride - p.pad + tkw[e];
          if (mv && kok[e] && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W)
            vb[e] = p.x[((size_t)((long long)n * p.H + ih) * p.W + iw) * p.C + tci[e]];
        }
      }
      ra[r] = va;
      rb[r] = vb;
    }
  };
  auto lds_store = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      *(f32x4*)(As + buf * BK * BM + (lp + 8 * r) * BM + c4 * 4) = ra[r];
      *(f32x4*)(Bs + buf * BK * BN + (lp + 8 * r) * BN + c4 * 4) = rb[r];
    }
  };

  const int frow = lane & 31, fh = lane >> 5;
  if (chunk_begin < chunk_end) {
    gload(chunk_begin);
    lds_store(0);
    __syncthreads();
    for (int ch = chunk_begin; ch < chunk_end; ++ch) {
      const int buf = (ch - chunk_begin) & 1;
      if (ch + 1 < chunk_end) gload(ch + 1);
      const float* ab = As + buf * BK * BM + wm * 64 + frow;
      const float* bb = Bs + buf * BK * BN + wn * 64 + frow;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = ab[(2 * s + fh) * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = bb[(2 * s + fh) * BN + j * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (ch + 1 < chunk_end) lds_store(buf ^ 1);
      __syncthreads();
    }
  }

  float* out = p.ws + (size_t)split * p.K * p.Kred;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int k = j0 + wn * 64 + j * 32 + frow;
    if (k >= p.Kred) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = i0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (co < p.K) out[(size_t)co * p.Kred + k] = acc[i][j][e];
      }
  }
}

// dst[i] (+)= sum_s ws[s][i]
__global__ void reduce_slabs_f32(const float* __restrict__ ws, float* __restrict__ dst, size_t n, int splits,
                                 int accumulate) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += ws[(size_t)k * n + i];
    dst[i] = accumulate ? dst[i] + s : s;
  }
}

// column sums of a [M][C] matrix into partial[P][C]; rows split over gridDim.y
__global__ __launch_bounds__(256) void colsum_partial_f32(const float* __restrict__ g, float* __restrict__ partial, int M,
                                                        int C) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const int P = gridDim.y;
  const int rows_per = (M + P - 1) / P;
  const int r0 = blockIdx.y * rows_per;
  int r1 = r0 + rows_per;
  if (r1 > M) r1 = M;
  float s = 0.f;
  if (c < C)
    for (int r = r0 + rl; r < r1; r += 4) s += g[(size_t)r * C + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// vectorised column sums: thread t owns channel quad t % C4 and row lane t / C4; rows [r0, r0+rows) per block
__global__ __launch_bounds__(256) void colsum_partial_vec_f32(const float* __restrict__ g, float* __restrict__ partial, int M,
                                                            int C, int rows_per_block) {
  __shared__ f32x4 red[256];
  const int C4 = C >> 2;
  const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4, RL = 256 / C4;
  const int r0 = blockIdx.x * rows_per_block;
  int r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + rl; r < r1; r += RL) acc += ((const f32x4*)g)[(size_t)r * C4 + c4];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < RL; ++k) acc += red[k * C4 + c4];
    ((f32x4*)partial)[(size_t)blockIdx.x * C4 + c4] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// weight image for dgrad:  wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci]
__global__ void weight_flip_transpose_f32(const float* __restrict__ w, float* __restrict__ wt, int K, int KH, int KW,
                                          int C) {
  const size_t n = (size_t)K * KH * KW * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    // i indexes wt: [ci][kh'][kw'][co]
    size_t t = i;
    const int co = (int)(t % K); t /= K;
    const int kwp = (int)(t % KW); t /= KW;
    const int khp = (int)(t % KH); t /= KH;
    const int ci = (int)t;
    wt[i] = w[(((size_t)co * KH + (KH - 1 - khp)) * KW + (KW - 1 - kwp)) * C + ci];
  }
}

extern "C" int utv2_conv2d_wgrad_splits(int N, int OH, int OW, int K, int Kred);

static int fill_levels(LevelTab& lt, int nlev, int N, const int* H, const int* W) {
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

extern "C" {

// Multi-level "same" convolution (stride 1, pad (k-1)/2) on a level-first [P][C] buffer: one launch for all
// FPN levels of a shared head (towers, prediction convs, RPN conv).  H_host/W_host: host int[nlev].
// Also serves as its own dgrad (pass the flipped/transposed weight image, pad = k-1-pad).
int utv2_conv2d_ml_fwd(const float* x, const float* w, float* y, const float* scale, const float* bias,
                       const float* residual, int nlev, const int* H_host, const int* W_host, int N, int C, int K, int KH,
                       int KW, int pad, int relu, int accumulate, hipStream_t stream) {
  if (!x || !w || !y || nlev < 1 || nlev > CONV_MAX_LEVELS || (C % 16) || N <= 0) return UTV2_EARG;
  ConvArgs a;
  a.M = fill_levels(a.lt, nlev, N, H_host, W_host);
  a.x = x; a.w = w; a.y = y; a.scale = scale; a.bias = bias; a.residual = residual;
  a.N = N; a.H = 0; a.W = 0; a.C = C; a.OH = 0; a.OW = 0; a.K = K; a.KH = KH; a.KW = KW;
  a.stride = 1; a.pad = pad; a.in_dil = 1; a.relu = relu; a.accumulate = accumulate; a.y_bf16 = 0;
  a.Kred = KH * KW * C;
  const bool small = K <= 64;
  const int tiles = cdiv(a.M, 128) * cdiv(K, small ? 64 : 128);
  if (small) hipLaunchKernelGGL((conv_igemm_f32<64, 0, true>), dim3(tiles), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv_igemm_f32<128, 0, true>), dim3(tiles), dim3(256), 0, stream, a);
  return utv2_launch_status();
}

int utv2_conv2d_ml_wgrad(const float* x, const float* dy, float* dw, float* ws, int nlev, const int* H_host,
                         const int* W_host, int N, int C, int K, int KH, int KW, int pad, int accumulate,
                         hipStream_t stream) {
  if (!x || !dy || !dw || !ws || nlev < 1 || nlev > CONV_MAX_LEVELS || (C & 3) || (K & 3)) return UTV2_EARG;
  WgradArgs a;
  a.M = fill_levels(a.lt, nlev, N, H_host, W_host);
  a.x = x; a.dy = dy; a.ws = ws;
  a.N = N; a.H = 0; a.W = 0; a.C = C; a.OH = 1; a.OW = a.M; a.K = K; a.KH = KH; a.KW = KW; a.stride = 1; a.pad = pad;
  a.Kred = KH * KW * C;
  a.splits = utv2_conv2d_wgrad_splits(1, 1, a.M, K, a.Kred);
  const int chunks = cdiv(a.M, 16);
  a.chunks_per_split = cdiv(chunks, a.splits);
  const int tiles = cdiv(K, 128) * cdiv(a.Kred, 128);
  hipLaunchKernelGGL((conv_wgrad_f32<1, true>), dim3(tiles * a.splits), dim3(256), 0, stream, a);
  const size_t n = (size_t)K * a.Kred;
  int rb = cdiv((int64_t)n, 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(reduce_slabs_f32, dim3(rb), dim3(256), 0, stream, (const float*)ws, dw, n, a.splits, accumulate);
  return utv2_launch_status();
}

// Forward / dgrad implicit GEMM.  x:[N,H,W,C] w:[K][Kred] y:[N,OH,OW,K].
// y = relu?( conv(x,w) * scale[co] + bias[co] + residual ) (+ y if accumulate)
// in_dil > 1 turns the gather into the transposed-conv (dgrad of a strided conv) form.
int utv2_conv2d_nhwc_fwd(const float* x, const float* w, float* y, const float* scale, const float* bias,
                         const float* residual, int N, int H, int W, int C, int K, int KH, int KW, int stride, int pad,
                         int in_dil, int OH, int OW, int relu, int accumulate, int Kred, hipStream_t stream) {
  if (!x || !w || !y || N <= 0 || C <= 0 || K <= 0) return UTV2_EARG;
  ConvArgs a;
  a.lt.n = 0;
  a.x = x; a.w = w; a.y = y; a.scale = scale; a.bias = bias; a.residual = residual;
  a.N = N; a.H = H; a.W = W; a.C = C; a.OH = OH; a.OW = OW; a.K = K; a.KH = KH; a.KW = KW;
  a.stride = stride; a.pad = pad; a.in_dil = in_dil < 1 ? 1 : in_dil; a.relu = relu; a.accumulate = accumulate;
  a.M = N * OH * OW;
  a.y_bf16 = 0;
  const int kred_nat = KH * KW * C;
  int mode;
  if (C % 16 == 0 && Kred == kred_nat) mode = 0;
  else if (C == 4 && Kred % 16 == 0 && Kred >= kred_nat && a.in_dil == 1) mode = 1;
  else { mode = 2; if (Kred != kred_nat) return UTV2_EARG; }
  a.Kred = Kred;
  const bool small = K <= 64;
  const int BN = small ? 64 : 128;
  const int tiles = cdiv(a.M, 128) * cdiv(K, BN);
  dim3 grid(tiles), block(256);
  if (mode == 0) {
    if (small) hipLaunchKernelGGL((conv_igemm_f32<64, 0>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv_igemm_f32<128, 0>), grid, block, 0, stream, a);
  } else if (mode == 1) {
    if (small) hipLaunchKernelGGL((conv_igemm_f32<64, 1>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv_igemm_f32<128, 1>), grid, block, 0, stream, a);
  } else {
    if (small) hipLaunchKernelGGL((conv_igemm_f32<64, 2>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv_igemm_f32<128, 2>), grid, block, 0, stream, a);
  }
  return utv2_launch_status();
}

// The image stem (C == 4: zero-padded NHWC4 image, weight rows padded to Kred % 16 == 0) with a selectable output
// element type: under AMP it writes bf16, so the 7x7 conv's large output is stored and re-read at half the bytes.
int utv2_conv2d_stem_fwd(const float* x, const float* w, void* y, int y_dtype, const float* scale, const float* bias, int N,
                         int H, int W, int K, int KH, int KW, int stride, int pad, int OH, int OW, int relu, int Kred,
                         hipStream_t stream) {
  if (!x || !w || !y || N <= 0 || K <= 0 || (K & 3) || (Kred & 15) || Kred < KH * KW * 4 ||
      (y_dtype != UTV2_F32 && y_dtype != UTV2_BF16))
    return UTV2_EARG;
  ConvArgs a;
  a.lt.n = 0;
  a.x = x; a.w = w; a.y = (float*)y; a.scale = scale; a.bias = bias; a.residual = nullptr;
  a.N = N; a.H = H; a.W = W; a.C = 4; a.OH = OH; a.OW = OW; a.K = K; a.KH = KH; a.KW = KW;
  a.stride = stride; a.pad = pad; a.in_dil = 1; a.relu = relu; a.accumulate = 0;
  a.M = N * OH * OW;
  a.y_bf16 = y_dtype == UTV2_BF16;
  a.Kred = Kred;
  const bool small = K <= 64;
  const int tiles = cdiv(a.M, 128) * cdiv(K, small ? 64 : 128);
  if (small) hipLaunchKernelGGL((conv_igemm_f32<64, 1>), dim3(tiles), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv_igemm_f32<128, 1>), dim3(tiles), dim3(256), 0, stream, a);
  return utv2_launch_status();
}

// number of pixel-range splits the wgrad launch will use, and the workspace floats it needs
int utv2_conv2d_wgrad_splits(int N, int OH, int OW, int K, int Kred) {
  const int M = N * OH * OW;
  const int chunks = cdiv(M, 16);
  const int tiles = cdiv(K, 128) * cdiv(Kred, 128);
  int splits = cdiv(1024, tiles);           // aim for ~4 blocks per CU
  const int max_by_chunks = chunks / 8 > 0 ? chunks / 8 : 1;  // >= 8 chunks of work per split
  if (splits > max_by_chunks) splits = max_by_chunks;
  if (splits < 1) splits = 1;
  if (splits > 256) splits = 256;
  return splits;
}

int64_t utv2_conv2d_wgrad_workspace_floats(int N, int OH, int OW, int K, int Kred) {
  return (int64_t)utv2_conv2d_wgrad_splits(N, OH, OW, K, Kred) * K * Kred;
}

// dw[K][KH*KW*C] (+)= wgrad(x, dy).  ws: workspace of utv2_conv2d_wgrad_workspace_floats floats.
int utv2_conv2d_nhwc_wgrad(const float* x, const float* dy, float* dw, float* ws, int N, int H, int W, int C, int K,
                           int KH, int KW, int stride, int pad, int OH, int OW, int accumulate, hipStream_t stream) {
  if (!x || !dy || !dw || !ws) return UTV2_EARG;
  WgradArgs a;
  a.lt.n = 0;
  a.x = x; a.dy = dy; a.ws = ws;
  a.N = N; a.H = H; a.W = W; a.C = C; a.OH = OH; a.OW = OW; a.K = K; a.KH = KH; a.KW = KW;
  a.stride = stride; a.pad = pad;
  a.Kred = KH * KW * C;
  a.M = N * OH * OW;
  a.splits = utv2_conv2d_wgrad_splits(N, OH, OW, K, a.Kred);
  const int chunks = cdiv(a.M, 16);
  a.chunks_per_split = cdiv(chunks, a.splits);
  const int tiles = cdiv(K, 128) * cdiv(a.Kred, 128);
  dim3 grid(tiles * a.splits), block(256);
  const bool vec = (C % 4 == 0) && (K % 4 == 0);
  if (vec) hipLaunchKernelGGL((conv_wgrad_f32<1>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((conv_wgrad_f32<0>), grid, block, 0, stream, a);
  const size_t n = (size_t)K * a.Kred;
  int rb = cdiv((int64_t)n, 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(reduce_slabs_f32, dim3(rb), dim3(256), 0, stream, (const float*)ws, dw, n, a.splits, accumulate);
  return utv2_launch_status();
}

// db[C] (+)= column sums of g[M][C].  ws: >= 1024*C floats.
int utv2_colsum(const float* g, float* db, float* ws, int M, int C, int accumulate, hipStream_t stream) {
  if (!g || !db || !ws) return UTV2_EARG;
  if ((C & 3) == 0 && C / 4 <= 256 && 256 % (C / 4) == 0 && M >= 1024) {
    int rows = 256;
    while (cdiv(M, rows) > 1024) rows *= 2;
    const int nb = cdiv(M, rows);
    hipLaunchKernelGGL(colsum_partial_vec_f32, dim3(nb), dim3(256), 0, stream, g, ws, M, C, rows);
    hipLaunchKernelGGL(reduce_slabs_f32, dim3(cdiv(C, 256)), dim3(256), 0, stream, (const float*)ws, db, (size_t)C, nb, accumulate);
    return utv2_launch_status();
  }
  int P = cdiv(M, 512);
  if (P > 64) P = 64;
  if (P < 1) P = 1;
  hipLaunchKernelGGL(colsum_partial_f32, dim3(cdiv(C, 64), P), dim3(256), 0, stream, g, ws, M, C);
  hipLaunchKernelGGL(reduce_slabs_f32, dim3(cdiv(C, 256)), dim3(256), 0, stream, (const float*)ws, db, (size_t)C, P,
                     accumulate);
  return utv2_launch_status();
}

int utv2_weight_flip_transpose(const float* w, float* wt, int K, int KH, int KW, int C, hipStream_t stream) {
  if (!w || !wt) return UTV2_EARG;
  const size_t n = (size_t)K * KH * KW * C;
  int nb = cdiv((int64_t)n, 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(weight_flip_transpose_f32, dim3(nb), dim3(256), 0, stream, w, wt, K, KH, KW, C);
  return utv2_launch_status();
}

}  // extern "C"
