// The frozen ResNet stem as ONE kernel: D2 BasicStem = conv 7x7 stride 2 pad 3 (3 -> 64) + FrozenBN + ReLU, then max_pool2d 3x3 stride 2
// pad 1 (reference backbone: the R-50 built through build_fcos_resnet_fpn_backbone, ubteacher/modeling/backbone/fpn.py:21-22;
// MODEL.BACKBONE.FREEZE_AT >= 1).  Unfused the conv writes its [N, H/2, W/2, 64] output (413 MB for 12 images of 800 x 1344) and the
// pool reads it back; here the conv outputs of a pooled tile live in LDS and only the pooled [N, H/4, W/4, 64] tensor is written.
//
// Persistent workgroups (two per CU) of 4 waves: the weight image [64][7 x 32] (7 taps x 4 channels + 4 zeros per kernel row, the
// layout of utv2_conv2d_stem_fwd_bf16) is loaded into LDS ONCE; a workgroup then walks pooled tiles of 4 x 16 pixels = 9 x 33 conv
// outputs (297, padded to 10 blocks of 32).  The GEMM is transposed (A = weights [co][k], B = conv pixels [px][k], C[co][px]):
//   B fragments come STRAIGHT from the zero-bordered NHWC4 image (utv2_preprocess_image_bf16pad): 8 consecutive k = 2 pixels x 4 channels
//   of one image row = 16 aligned bytes, and the 32 lanes of a block read a contiguous 512-byte run of it (conv pixels along x are 2
//   image pixels apart) - no im2col staging at all; a register ring keeps three k-steps of them in flight;
//   20 MFMA blocks (10 pixel x 2 channel) = 5 per wave, K = 224 = 14 steps;
//   value = acc * scale + shift, ReLU, ZERO outside the conv output (every pool window holds a real non-negative value, so zero is as
//   good as the pool's -inf padding), rounded to 16 bits as the unfused conv stores it -> LDS [320 px][64 ch];
//   pool: one thread per (pooled pixel, 8-channel group), nine 16-byte LDS reads, one 16-byte store of a full 128-byte output row piece.
#include "common.h"

#define SP_PH 4
#define SP_PW 16
#define SP_CH (2 * SP_PH + 1)    // 9 conv rows
#define SP_CW (2 * SP_PW + 1)    // 33 conv columns
#define SP_NPX (SP_CH * SP_CW)   // 297
#define SP_NPB 10                // pixel blocks of 32
#define SP_WROW 464              // bytes per weight row in LDS: 448 + 16 pad (16-byte slot index 29 r mod 16: conflict-free fragment reads)
#define SP_WBYTES (64 * SP_WROW)
#define SP_TILE (SP_NPB * 32 * 128)
#define SP_LDS (SP_WBYTES + SP_TILE + 512)

struct SpArgs {
  const h16_t* xpad;   // [N][H + 6][W + 8][4]
  const h16_t* w;      // [64][224]
  h16_t* y;            // [N][PH][PW][64]
  const float *scale, *shift;
  int N, H, W, OH, OW, PH, PW, tiles_x, tiles_y, ntiles;
};

__global__ __launch_bounds__(256, 2) void stem_pool_fused(SpArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;
  unsigned char* tile = smem + SP_WBYTES;
  float* prm = (float*)(smem + SP_WBYTES + SP_TILE);   // scale[64] shift[64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, fh = lane >> 5;

  // weights -> LDS, once: 64 rows x 28 pieces of 16 bytes
  for (int q = tid; q < 64 * 28; q += 256) {
    const int co = q / 28, s = q - co * 28;
    *(bf16x8_t*)(wl + co * SP_WROW + s * 16) = *(const bf16x8_t*)(a.w + co * 224 + s * 8);
  }
  if (tid < 64) {
    prm[tid] = a.scale[tid];
    prm[64 + tid] = a.shift[tid];
  }
  __syncthreads();

  // this wave's five MFMA blocks b = 5 wave + j: pixel block b >> 1, channel block b & 1; three distinct pixel blocks pb0 .. pb0 + 2
  const int pb0 = (wave * 5) >> 1;
  const bool odd = wave & 1;
  const int rowpitch = (a.W + 8) * 4;                  // elements per image row
  const int per = a.tiles_x * a.tiles_y;

  for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
    const int n = t / per, rem = t - n * per, ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int cy0 = 2 * ty * SP_PH - 1, cx0 = 2 * tx * SP_PW - 1;    // conv coordinates of the tile's first output
    const h16_t* img = a.xpad + (size_t)n * (a.H + 6) * rowpitch;
    // B-fragment base addresses of this lane's conv pixel in each of the wave's three pixel blocks (tap (0, 2 fh))
    const h16_t* bsrc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int m = (pb0 + i) * 32 + l31, r = m / SP_CW, c = m - r * SP_CW;
      int cy = cy0 + r, cx = cx0 + c;
      const bool ok = m < SP_NPX && cy >= 0 && cy < a.OH && cx >= 0 && cx < a.OW;
      cy = ok ? cy : 0;
      cx = ok ? cx : 0;
      bsrc[i] = img + (size_t)(2 * cy) * rowpitch + (2 * cx + 2 * fh) * 4;
    }
    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8_t br[3][3];                                 // ring: k-step s in slot s % 3
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int i = 0; i < 3; ++i) br[s][i] = *(const bf16x8_t*)(bsrc[i] + (s >> 1) * rowpitch + (s & 1) * 16);
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      bf16x8_t xb[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) xb[i] = br[s % 3][i];
      if (s + 3 < 14) {
#pragma unroll
        for (int i = 0; i < 3; ++i) br[s % 3][i] = *(const bf16x8_t*)(bsrc[i] + ((s + 3) >> 1) * rowpitch + ((s + 3) & 1) * 16);
      }
      const bf16x8_t wa = *(const bf16x8_t*)(wl + l31 * SP_WROW + s * 32 + fh * 16);
      const bf16x8_t wb = *(const bf16x8_t*)(wl + (32 + l31) * SP_WROW + s * 32 + fh * 16);
      if (!odd) {   // blocks (p0,c0) (p0,c1) (p1,c0) (p1,c1) (p2,c0)
        acc[0] = mfma_32x32x16(wa, xb[0], acc[0]);
        acc[1] = mfma_32x32x16(wb, xb[0], acc[1]);
        acc[2] = mfma_32x32x16(wa, xb[1], acc[2]);
        acc[3] = mfma_32x32x16(wb, xb[1], acc[3]);
        acc[4] = mfma_32x32x16(wa, xb[2], acc[4]);
      } else {      // blocks (p0,c1) (p1,c0) (p1,c1) (p2,c0) (p2,c1)
        acc[0] = mfma_32x32x16(wb, xb[0], acc[0]);
        acc[1] = mfma_32x32x16(wa, xb[1], acc[1]);
        acc[2] = mfma_32x32x16(wb, xb[1], acc[2]);
        acc[3] = mfma_32x32x16(wa, xb[2], acc[3]);
        acc[4] = mfma_32x32x16(wb, xb[2], acc[4]);
      }
    }
    // conv epilogue -> LDS [px][64] 16-bit (128-byte rows, 16-byte slots XOR-swizzled by the row)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int b = wave * 5 + j, pb = b >> 1, cb = b & 1;
      const int m = pb * 32 + l31, r = m / SP_CW, c = m - r * SP_CW, cy = cy0 + r, cx = cx0 + c;
      const bool ok = m < SP_NPX && cy >= 0 && cy < a.OH && cx >= 0 && cx < a.OW;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ch = cb * 32 + 8 * q + 4 * fh;
        const f32x4 sc = *(const f32x4*)(prm + ch), bi = *(const f32x4*)(prm + 64 + ch);
        bf16x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[j][4 * q + e] * sc[e] + bi[e];
          o[e] = (h16_t)(ok ? fmaxf(v, 0.f) : 0.f);
        }
        *(bf16x4_t*)(tile + m * 128 + (((cb * 4 + q) ^ (m & 7)) << 4) + fh * 8) = o;
      }
    }
    __syncthreads();
    // 3x3 stride-2 max over the tile: pooled (pr, pc) <- conv rows 2 pr .. 2 pr + 2, columns 2 pc .. 2 pc + 2 of the tile
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + 256 * it, pp = item >> 3, cg = item & 7, pr = pp >> 4, pc = pp & 15;
      float mx[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) mx[e] = 0.f;
#pragma unroll
      for (int dr = 0; dr < 3; ++dr)
#pragma unroll
        for (int dc = 0; dc < 3; ++dc) {
          const int m = (2 * pr + dr) * SP_CW + 2 * pc + dc;
          const bf16x8_t v = *(const bf16x8_t*)(tile + m * 128 + ((cg ^ (m & 7)) << 4));
#pragma unroll
          for (int e = 0; e < 8; ++e) mx[e] = fmaxf(mx[e], (float)v[e]);
        }
      const int py = ty * SP_PH + pr, px = tx * SP_PW + pc;
      if (py < a.PH && px < a.PW) {
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (h16_t)mx[e];
        *(bf16x8_t*)(a.y + (((size_t)n * a.PH + py) * a.PW + px) * 64 + cg * 8) = o;
      }
    }
    __syncthreads();   // the next tile's epilogue overwrites the conv tile
  }
}

extern "C" {

// D2 BasicStem (conv 7x7 s2 p3, 3 -> 64, FrozenBN as scale / shift, ReLU) + max_pool2d(3, 2, 1) in one launch.  xpad16: [N][H+6][W+8][4]
// of the library's 16-bit type (utv2_preprocess_image_bf16pad), w16s: [64][7][32] (the weight image of utv2_conv2d_stem_fwd_bf16);
// y: [N][PH][PW][64] 16-bit with OH = (H - 1) / 2 + 1, PH = (OH - 1) / 2 + 1 (same for W).  Values equal relu-conv rounded to 16 bits,
// then the exact max.
int utv2_stem_pool_fwd_bf16(const void* xpad16, const void* w16s, void* y, const float* scale, const float* shift, int N, int H, int W,
                            int K, hipStream_t stream) {
  if (!xpad16 || !w16s || !y || !scale || !shift || N < 1 || H < 1 || W < 1 || K != 64 || (W & 1)) return UTV2_EARG;
  SpArgs a;
  a.xpad = (const h16_t*)xpad16; a.w = (const h16_t*)w16s; a.y = (h16_t*)y; a.scale = scale; a.shift = shift;
  a.N = N; a.H = H; a.W = W;
  a.OH = (H - 1) / 2 + 1; a.OW = (W - 1) / 2 + 1;
  a.PH = (a.OH - 1) / 2 + 1; a.PW = (a.OW - 1) / 2 + 1;
  a.tiles_x = cdiv(a.PW, SP_PW); a.tiles_y = cdiv(a.PH, SP_PH);
  const long long nt = (long long)N * a.tiles_x * a.tiles_y;
  if (nt > 0x7fffffff) return UTV2_EARG;
  a.ntiles = (int)nt;
  static LdsOptIn lds_opt_in;
  lds_opt_in({(const void*)stem_pool_fused}, SP_LDS);
  const int grid = a.ntiles < 512 ? a.ntiles : 512;
  hipLaunchKernelGGL(stem_pool_fused, dim3((unsigned)grid), dim3(256), SP_LDS, stream, a);
  return utv2_launch_status();
}

}  // extern "C"
