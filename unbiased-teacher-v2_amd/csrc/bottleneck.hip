// One frozen ResNet bottleneck (D2 BottleneckBlock with FrozenBN folded to scale / shift; reference backbone: the R-50 built through
// build_fcos_resnet_fpn_backbone, ubteacher/modeling/backbone/fpn.py:21-22; res2 under MODEL.BACKBONE.FREEZE_AT 2) as ONE kernel:
//
//     identity blocks (res2 blocks 1-2):  y = relu( conv3(relu(conv2(relu(conv1(x) s1 + b1)) s2 + b2)) s3 + b3 + x )            x: [N, H, W, 256]
//     the stage's first block (res2 block 0, stride 1):  ... + h16(shortcut_1x1(x) ss + bs) instead of + x                     x: [N, H, W, 64]
//     y: [N, H, W, 256], 16-bit NHWC; conv1 1x1 -> 64, conv2 3x3 pad 1 -> 64, conv3 1x1 -> 256
//
// The convs of such a block run HBM-bound in 16-bit: unfused they move x (read), c1 (write + read), c2 (write + read), x again
// (residual, or the shortcut conv's read and its 256-channel output written + read) and y; fused, the 64-channel intermediates live in LDS
// and the block moves x once (plus the 3x3 halo, mostly L2 hits between neighbouring tiles of one XCD), the residual (L2) and y.  No
// backward exists for a frozen stage, so nothing has to be kept.
//
// Workgroup = 4 waves = one 8 x 16 output tile of one image.  All GEMMs are computed TRANSPOSED (A = weights [co][k], B = pixels
// [px][k], C[co][px]): a lane then holds 4 consecutive channels of one pixel per accumulator quad and intermediates go to LDS with
// 8-byte accesses.
//   phase 1  conv1 on the tile's 10 x 18 halo (180 pixels, padded to 192 = 6 pixel blocks): K = CIN in 32-channel chunks, x and w1
//            chunks through a register ring (three chunks ahead) and a double-buffered LDS stage; result c1 [192 px][64] 16-bit in LDS,
//            ZERO outside the image (the 3x3's zero padding applies to c1).  12 MFMA blocks (6 pixel x 2 channel) = 3 per wave.
//   phase 2  conv2 from c1 in LDS (implicit im2col = shifted row addresses), w2 streamed per tap (8 KB, register ring + double buffer);
//            wave w owns pixel block w, both channel blocks; result c2 [128 px][64] in LDS.
//   phase 3  conv3 (+ the shortcut conv on the tile's own pixels of x) in four groups of 64 output channels: that quarter of w3 (and of
//            the shortcut weight) in LDS, the wave's c2 (and x) fragments in registers; value = acc * s + b (+ the shortcut rounded to
//            16 bits, as the unfused chain stores it) bounces through a wave-private fp32 LDS patch so that the residual is read and y
//            written in full 128-byte row pieces.
// LDS: 43-45 KB (regions reused phase by phase, see the map in the kernel) -> three workgroups per CU, which overlap each other's load and
// MFMA phases.  All LDS rows are XOR-swizzled on their 16-byte slots (64-byte rows: slot ^ (row >> 2 & 3); 128-byte rows: slot ^ (row & 7))
// so the 32-row fragment reads are conflict-free.  blockIdx -> tile is XCD-aware (contiguous tile ranges per XCD: neighbouring tiles
// share their halo through that XCD's L2).
#include "common.h"

#define BT_TH 8
#define BT_TW 16
#define BT_HW 18
#define BT_HALO 180
#define BT_HP 192
#define BT_COUT 256
#define BT_MID 64
#define BT_STAGE 16384
#define BT_C1 (BT_HP * 128)
#define BT_PRM 1280
#define BT_LDS (BT_STAGE + BT_C1 + BT_PRM * 4)

struct BtArgs {
  const h16_t* x;
  h16_t* y;
  const h16_t *w1, *w2, *w3, *wsc;               // [64][CIN], [64][3*3*64] (tap-major, channel-minor), [256][64], shortcut [256][CIN] or null
  const float *s1, *b1, *s2, *b2, *s3, *b3, *ssc, *bsc;   // folded FrozenBN scale / shift per output channel
  int N, H, W, tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ bf16x8_t lds_frag(const unsigned char* smem, int off) { return *(const bf16x8_t*)(smem + off); }

// CIN = channels of x (256: identity block, SC false; 64: the stage's first block, SC true = 1x1 shortcut conv instead of + x)
template <int CIN, bool SC>
__global__ __launch_bounds__(256, 3) void bottleneck_fused(BtArgs a) {
  constexpr int NCH = CIN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS map: R0 = [0, 16K): phase-1 buffer 0, the w2 tap buffers, then c2, then the bounce patches;
  // R1 = [16K, 40K): phase-1 buffer 1 (first 16 KB), then c1, then the resident quarter of w3 (8 KB) and of the shortcut weight (8 KB);
  // [40K, 45K): scale / shift vectors
  unsigned char* stageA = smem;
  unsigned char* c1 = smem + BT_STAGE;
  unsigned char* stageB = smem + BT_STAGE;                  // phase-1 buffer 1 lies where c1 will be written after the last chunk
  unsigned char* c2 = smem;
  unsigned char* w3l = smem + BT_STAGE;
  float* prm = (float*)(smem + BT_STAGE + BT_C1);           // s1 b1 s2 b2 (64 each) s3 b3 ssc bsc (256 each)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, fh = lane >> 5;

  // XCD-aware bijective remap (blockIdx round-robins over the 8 XCDs): every XCD walks a contiguous range of tiles
  int t;
  {
    const int bid = blockIdx.x, q = a.ntiles >> 3, r = a.ntiles & 7, xcd = bid & 7, idx = bid >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int per = a.tiles_x * a.tiles_y;
  const int n = t / per, rem = t - n * per, ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
  const int y0 = ty * BT_TH, x0 = tx * BT_TW;
  const h16_t* ximg = a.x + (size_t)n * a.H * a.W * CIN;

  // ---------------- phase 1: c1 = relu(conv1(x) * s1 + b1) on the halo ----------------
  // staging duties of this thread: three 16-byte pieces of the x chunk (192 px x 4 slots), one of the w1 chunk (64 co x 4 slots).
  // Chunk c travels global -> register ring slot c % 3 (issued three chunks ahead) -> LDS buffer c & 1 (one chunk ahead of the MFMAs).
  const h16_t* xsrc[3];
  int xdst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int q = tid + 256 * i, px = q >> 2, slot = q & 3;
    const int hy = px / BT_HW, hx = px - hy * BT_HW, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = px < BT_HALO && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    xsrc[i] = ok ? ximg + ((size_t)gy * a.W + gx) * CIN + slot * 8 : nullptr;
    xdst[i] = px * 64 + ((slot ^ ((px >> 2) & 3)) << 4);
  }
  const int wco = tid >> 2, wslot = tid & 3;
  const h16_t* wsrc = a.w1 + wco * CIN + wslot * 8;
  const int wdst = BT_HP * 64 + wco * 64 + ((wslot ^ ((wco >> 2) & 3)) << 4);

  bf16x8_t xr[3][3], wr[3];
  const bf16x8_t zero8 = {};
#pragma unroll
  for (int c = 0; c < (NCH < 3 ? NCH : 3); ++c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) xr[c][i] = xsrc[i] ? *(const bf16x8_t*)(xsrc[i] + c * 32) : zero8;
    wr[c] = *(const bf16x8_t*)(wsrc + c * 32);
  }
  // folded FrozenBN parameters -> LDS
  {
    const float* srcs[4] = {a.s1, a.b1, a.s2, a.b2};
    prm[tid] = srcs[tid >> 6][tid & 63];
    prm[256 + tid] = a.s3[tid];
    prm[512 + tid] = a.b3[tid];
    if constexpr (SC) {
      prm[768 + tid] = a.ssc[tid];
      prm[1024 + tid] = a.bsc[tid];
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) *(bf16x8_t*)(stageA + xdst[i]) = xr[0][i];
  *(bf16x8_t*)(stageA + wdst) = wr[0];
  __syncthreads();

  f32x16 acc1[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc1[i][e] = 0.f;
  const int pbA = wave, pbB = 4 + (wave >> 1), cbB = wave & 1;   // blocks (pbA, 0), (pbA, 1), (pbB, cbB)
  const int rowA = pbA * 32 + l31, rowB = pbB * 32 + l31;
  const int w0row = l31, w1row = 32 + l31;
  // w2 taps: two pieces per thread of a [64 co][64 ci] tap image (128-byte rows), register ring of three taps
  const int t2co[2] = {tid >> 3, (tid + 256) >> 3}, t2slot = tid & 7;
  bf16x8_t tr[3][2];
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) {
    unsigned char* cur = (kc & 1) ? stageB : stageA;
    unsigned char* nxt = (kc & 1) ? stageA : stageB;
    if (kc + 3 < NCH) {          // ring slot kc % 3 went to LDS one iteration ago
#pragma unroll
      for (int i = 0; i < 3; ++i) xr[kc % 3][i] = xsrc[i] ? *(const bf16x8_t*)(xsrc[i] + (kc + 3) * 32) : zero8;
      wr[kc % 3] = *(const bf16x8_t*)(wsrc + (kc + 3) * 32);
    }
    // the first three w2 taps take the load slots at the tail of the x stream (all three at once when the stream is short)
#pragma unroll
    for (int tp = 0; tp < 3; ++tp)
      if ((NCH >= 3 && kc == NCH - 3 + tp) || (NCH < 3 && kc == NCH - 1)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) tr[tp][i] = *(const bf16x8_t*)(a.w2 + t2co[i] * 576 + tp * 64 + t2slot * 8);
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int slot = ks * 2 + fh;
      const bf16x8_t xa = lds_frag(cur, rowA * 64 + ((slot ^ ((rowA >> 2) & 3)) << 4));
      const bf16x8_t xb = lds_frag(cur, rowB * 64 + ((slot ^ ((rowB >> 2) & 3)) << 4));
      const bf16x8_t wa = lds_frag(cur, BT_HP * 64 + w0row * 64 + ((slot ^ ((w0row >> 2) & 3)) << 4));
      const bf16x8_t wb = lds_frag(cur, BT_HP * 64 + w1row * 64 + ((slot ^ ((w1row >> 2) & 3)) << 4));
      acc1[0] = mfma_32x32x16(wa, xa, acc1[0]);
      acc1[1] = mfma_32x32x16(wb, xa, acc1[1]);
      acc1[2] = mfma_32x32x16(cbB ? wb : wa, xb, acc1[2]);
    }
    if (kc + 1 < NCH) {
#pragma unroll
      for (int i = 0; i < 3; ++i) *(bf16x8_t*)(nxt + xdst[i]) = xr[(kc + 1) % 3][i];
      *(bf16x8_t*)(nxt + wdst) = wr[(kc + 1) % 3];
    }
    __syncthreads();
  }
  // (the last chunk's buffer is dead after that barrier: c1 may overwrite phase-1 buffer 1; R0 has been idle for a whole chunk when
  // NCH > 1 and is free after the barrier either way)

  // c1 epilogue: value = acc * s + b, ReLU, zero outside the image, round, 8-byte LDS stores
  {
    const int pbs[3] = {pbA, pbA, pbB}, cbs[3] = {0, 1, cbB};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int px = pbs[i] * 32 + l31;
      const int hy = px / BT_HW, hx = px - hy * BT_HW, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
      const bool ok = px < BT_HALO && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = cbs[i] * 32 + 8 * q + 4 * fh;
        const f32x4 sc = *(const f32x4*)(prm + ch), bi = *(const f32x4*)(prm + 64 + ch);
        bf16x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = acc1[i][4 * q + j] * sc[j] + bi[j];
          v = ok ? fmaxf(v, 0.f) : 0.f;
          o[j] = (h16_t)v;
        }
        *(bf16x4_t*)(c1 + px * 128 + (((cbs[i] * 4 + q) ^ (px & 7)) << 4) + fh * 8) = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) *(bf16x8_t*)(stageA + t2co[i] * 128 + ((t2slot ^ (t2co[i] & 7)) << 4)) = tr[0][i];
  __syncthreads();

  // ---------------- phase 2: c2 = relu(conv2(c1) * s2 + b2) ----------------
  f32x16 acc2[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
  const int p = wave * 32 + l31, pr = p >> 4, pc = p & 15;   // this lane's output pixel of the tile
  // a quarter of w3 = rows 64 g .. 64 g + 63 = [64][64] (128-byte rows): two pieces per thread; the same of the shortcut weight
  const int qco[2] = {tid >> 3, (tid + 256) >> 3}, qslot = tid & 7;
  bf16x8_t wq[2], sq[2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    unsigned char* cur = stageA + (tap & 1) * 8192;
    unsigned char* nxt = stageA + ((tap + 1) & 1) * 8192;
    if (tap + 3 < 9) {
#pragma unroll
      for (int i = 0; i < 2; ++i) tr[tap % 3][i] = *(const bf16x8_t*)(a.w2 + t2co[i] * 576 + (tap + 3) * 64 + t2slot * 8);
    } else if (tap == 6) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wq[i] = *(const bf16x8_t*)(a.w3 + qco[i] * BT_MID + qslot * 8);
        if constexpr (SC) sq[i] = *(const bf16x8_t*)(a.wsc + qco[i] * CIN + qslot * 8);
      }
    }
    const int dy = tap / 3, dx = tap - dy * 3;
    const int h = (pr + dy) * BT_HW + pc + dx;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int slot = ks * 2 + fh;
      const bf16x8_t xf = lds_frag(c1, h * 128 + ((slot ^ (h & 7)) << 4));
      const bf16x8_t wa = lds_frag(cur, w0row * 128 + ((slot ^ (w0row & 7)) << 4));
      const bf16x8_t wb = lds_frag(cur, w1row * 128 + ((slot ^ (w1row & 7)) << 4));
      acc2[0] = mfma_32x32x16(wa, xf, acc2[0]);
      acc2[1] = mfma_32x32x16(wb, xf, acc2[1]);
    }
    if (tap + 1 < 9) {
#pragma unroll
      for (int i = 0; i < 2; ++i) *(bf16x8_t*)(nxt + t2co[i] * 128 + ((t2slot ^ (t2co[i] & 7)) << 4)) = tr[(tap + 1) % 3][i];
    }
    __syncthreads();
  }
  // c1 and the tap buffers are dead from here on: c2 goes to R0, the first quarter of w3 (and of the shortcut weight) to R1; once every
  // wave holds its c2 fragments R0 carries the wave-private fp32 bounce patches (4 x 4 KB).
  // Bounce patch rows <-> pixels the wave stores in the coalesced pass: pass i, lane -> pixel (8 i + lane / 8), channels 8 (lane % 8) .. +8
  const int sub = lane >> 3, cv = lane & 7;
  int opix[4];                   // element offset of (the pixel, channel 8 cv) in y (and in x for the identity residual), -1 outside the image
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pp = wave * 32 + 8 * i + sub, gy = y0 + (pp >> 4), gx = x0 + (pp & 15);
    opix[i] = (gy < a.H && gx < a.W) ? ((gy * a.W + gx) * BT_COUT + cv * 8) : -1;
  }
  h16_t* yout = a.y + (size_t)n * a.H * a.W * BT_COUT;
  bf16x8_t res[2][4];            // identity: the residual of channel group g lives in res[g & 1], fetched one group ahead
  bf16x8_t xf3[4];               // shortcut: this lane's own pixel of x as conv fragments (K = 64)
  if constexpr (!SC) {
#pragma unroll
    for (int i = 0; i < 4; ++i) res[0][i] = opix[i] >= 0 ? *(const bf16x8_t*)(ximg + opix[i]) : zero8;
  } else {
    const int gy = y0 + pr, gx = x0 + pc;
    const bool in = gy < a.H && gx < a.W;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      xf3[ks] = in ? *(const bf16x8_t*)(ximg + ((size_t)gy * a.W + gx) * CIN + ks * 16 + fh * 8) : zero8;
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = cb * 32 + 8 * q + 4 * fh;
      const f32x4 sc = *(const f32x4*)(prm + 128 + ch), bi = *(const f32x4*)(prm + 192 + ch);
      bf16x4_t o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (h16_t)fmaxf(acc2[cb][4 * q + j] * sc[j] + bi[j], 0.f);
      *(bf16x4_t*)(c2 + p * 128 + (((cb * 4 + q) ^ (p & 7)) << 4) + fh * 8) = o;
    }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    *(bf16x8_t*)(w3l + qco[i] * 128 + ((qslot ^ (qco[i] & 7)) << 4)) = wq[i];
    if constexpr (SC) *(bf16x8_t*)(w3l + 8192 + qco[i] * 128 + ((qslot ^ (qco[i] & 7)) << 4)) = sq[i];
  }
  __syncthreads();

  // ---------------- phase 3: y = relu(conv3(c2) * s3 + b3 + residual) ----------------
  bf16x8_t cf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) cf[ks] = lds_frag(c2, p * 128 + (((ks * 2 + fh) ^ (p & 7)) << 4));
  __syncthreads();               // every wave holds its c2 fragments: R0 now carries the patches
  float* patch = (float*)smem + wave * (16 * 64);   // wave-private fp32 [16 px][64 ch], 16-byte quads XOR-swizzled by the row
#pragma unroll
  for (int g = 0; g < 4; ++g) {  // channel group g: output channels 64 g .. 64 g + 63
    if (g > 0) {                 // the next quarter of the weights
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *(bf16x8_t*)(w3l + qco[i] * 128 + ((qslot ^ (qco[i] & 7)) << 4)) = wq[i];
        if constexpr (SC) *(bf16x8_t*)(w3l + 8192 + qco[i] * 128 + ((qslot ^ (qco[i] & 7)) << 4)) = sq[i];
      }
      __syncthreads();
    }
    if (g + 1 < 4) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        wq[i] = *(const bf16x8_t*)(a.w3 + ((g + 1) * 64 + qco[i]) * BT_MID + qslot * 8);
        if constexpr (SC) sq[i] = *(const bf16x8_t*)(a.wsc + ((g + 1) * 64 + qco[i]) * CIN + qslot * 8);
      }
      if constexpr (!SC) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          res[(g + 1) & 1][i] = opix[i] >= 0 ? *(const bf16x8_t*)(ximg + opix[i] + (g + 1) * 64) : zero8;
      }
    }
    f32x16 acc3[2], accs[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc3[u][e] = 0.f;
        if constexpr (SC) accs[u][e] = 0.f;
      }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row = u * 32 + l31;
        const bf16x8_t wf = lds_frag(w3l, row * 128 + (((ks * 2 + fh) ^ (row & 7)) << 4));
        acc3[u] = mfma_32x32x16(wf, cf[ks], acc3[u]);
        if constexpr (SC) {
          const bf16x8_t sf = lds_frag(w3l + 8192, row * 128 + (((ks * 2 + fh) ^ (row & 7)) << 4));
          accs[u] = mfma_32x32x16(sf, xf3[ks], accs[u]);
        }
      }
    }
    if constexpr (SC) {
      // value = conv3 * s3 + b3 + h16(shortcut * ssc + bsc) in the accumulator layout -> fp32 patch, 16 pixels per round -> ReLU, round,
      // full 128-byte row pieces of y (the second set of accumulators leaves no registers for the coalesced-layout form below)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if ((l31 >> 4) == hh) {
          const int prow = l31 & 15;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int cl = u * 32 + 8 * q + 4 * fh, ch = g * 64 + cl;
              const f32x4 sc = *(const f32x4*)(prm + 256 + ch), bi = *(const f32x4*)(prm + 512 + ch);
              const f32x4 s2c = *(const f32x4*)(prm + 768 + ch), b2c = *(const f32x4*)(prm + 1024 + ch);
              f32x4 v;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = acc3[u][4 * q + j] * sc[j] + bi[j] + (float)(h16_t)(accs[u][4 * q + j] * s2c[j] + b2c[j]);
              *(f32x4*)(patch + prow * 64 + (((cl >> 2) ^ prow) << 2)) = v;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed (wave-private patch)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const int i = hh * 2 + i2, prow = 8 * i2 + sub;
          const f32x4 v0 = *(const f32x4*)(patch + prow * 64 + (((2 * cv) ^ prow) << 2)), v1 = *(const f32x4*)(patch + prow * 64 + (((2 * cv + 1) ^ prow) << 2));
          bf16x8_t o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = (h16_t)fmaxf(v0[j], 0.f);
            o[4 + j] = (h16_t)fmaxf(v1[j], 0.f);
          }
          if (opix[i] >= 0) *(bf16x8_t*)(yout + opix[i] + g * 64) = o;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();      // the patch is rewritten by the next round
      }
    } else {
      // The raw accumulators bounce through the fp32 patch, 16 pixels per round; scale / shift, the residual, ReLU and the rounding run
      // in the coalesced layout: lane -> 8 consecutive channels of one pixel, full 128-byte row pieces of x / y.
      const f32x4 s3a = *(const f32x4*)(prm + 256 + g * 64 + cv * 8), s3b = *(const f32x4*)(prm + 256 + g * 64 + cv * 8 + 4);
      const f32x4 b3a = *(const f32x4*)(prm + 512 + g * 64 + cv * 8), b3b = *(const f32x4*)(prm + 512 + g * 64 + cv * 8 + 4);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if ((l31 >> 4) == hh) {
          const int prow = l31 & 15;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int cl = u * 32 + 8 * q + 4 * fh;
              const f32x4 v = {acc3[u][4 * q], acc3[u][4 * q + 1], acc3[u][4 * q + 2], acc3[u][4 * q + 3]};
              *(f32x4*)(patch + prow * 64 + (((cl >> 2) ^ prow) << 2)) = v;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes have landed (wave-private patch)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
          const int i = hh * 2 + i2, prow = 8 * i2 + sub;
          const f32x4 v0 = *(const f32x4*)(patch + prow * 64 + (((2 * cv) ^ prow) << 2)), v1 = *(const f32x4*)(patch + prow * 64 + (((2 * cv + 1) ^ prow) << 2));
          bf16x8_t o;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = (h16_t)fmaxf(v0[j] * s3a[j] + b3a[j] + (float)res[g & 1][i][j], 0.f);
            o[4 + j] = (h16_t)fmaxf(v1[j] * s3b[j] + b3b[j] + (float)res[g & 1][i][4 + j], 0.f);
          }
          if (opix[i] >= 0) *(bf16x8_t*)(yout + opix[i] + g * 64) = o;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();      // the patch is rewritten by the next round
      }
    }
  }
}

extern "C" {

// One frozen bottleneck with 64 mid and 256 output channels, 3x3 pad 1, all strides 1, 16-bit NHWC (ResNet-50 res2).
// x: [N, H, W, C], y: [N, H, W, 256] of this library's 16-bit type (y != x); w1 [64][C], w2 [64][3][3][64], w3 [256][64] 16-bit;
// s* / b*: fp32 scale and shift of the folded FrozenBN per output channel.  value = acc * s + b, ReLU after conv1 / conv2.
// wsc == NULL (C = 256): identity block, conv3's value + x, ReLU.  wsc [256][C] (C = 64) with ssc / bsc: the stage's first block,
// conv3's value + h16(shortcut(x) * ssc + bsc), ReLU.
int utv2_bottleneck_fwd_bf16(const void* x, void* y, const void* w1, const void* w2, const void* w3, const void* wsc, const float* s1,
                             const float* b1, const float* s2, const float* b2, const float* s3, const float* b3, const float* ssc,
                             const float* bsc, int N, int H, int W, int C, int MID, hipStream_t stream) {
  if (!x || !y || x == y || !w1 || !w2 || !w3 || !s1 || !b1 || !s2 || !b2 || !s3 || !b3 || N < 1 || H < 1 || W < 1) return UTV2_EARG;
  if (MID != BT_MID || !((C == 256 && !wsc) || (C == 64 && wsc && ssc && bsc))) return UTV2_EARG;
  if ((long long)H * W * BT_COUT > 0x7fffffffll) return UTV2_EARG;
  BtArgs a;
  a.x = (const h16_t*)x; a.y = (h16_t*)y; a.w1 = (const h16_t*)w1; a.w2 = (const h16_t*)w2; a.w3 = (const h16_t*)w3; a.wsc = (const h16_t*)wsc;
  a.s1 = s1; a.b1 = b1; a.s2 = s2; a.b2 = b2; a.s3 = s3; a.b3 = b3; a.ssc = ssc; a.bsc = bsc;
  a.N = N; a.H = H; a.W = W;
  a.tiles_x = cdiv(W, BT_TW); a.tiles_y = cdiv(H, BT_TH);
  const long long nt = (long long)N * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffff) return UTV2_EARG;
  a.ntiles = (int)nt;
  if (wsc) hipLaunchKernelGGL((bottleneck_fused<64, true>), dim3((unsigned)a.ntiles), dim3(256), BT_LDS, stream, a);
  else hipLaunchKernelGGL((bottleneck_fused<256, false>), dim3((unsigned)a.ntiles), dim3(256), BT_LDS, stream, a);
  return utv2_launch_status();
}

// 1 when utv2_bottleneck_fwd_bf16 takes this block shape (C input channels, MID mid channels, with / without a shortcut conv)
int utv2_bottleneck_supported(int C, int MID, int has_shortcut) {
  return MID == BT_MID && ((C == 256 && !has_shortcut) || (C == 64 && has_shortcut));
}

}  // extern "C"
