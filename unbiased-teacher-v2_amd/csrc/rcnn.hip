// Faster-RCNN specific kernels of the UTv2 step (gfx950): batched box<->gt matching (IoU max /
// argmax + low-quality pass), multi-level RoIAlign (aligned, adaptive sampling) NHWC fwd/bwd,
// softmax focal loss fwd/bwd.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// Matching (D2 pairwise_iou + Matcher [D2-recall]; called from proposal_generator/rpn.py:112-148 and
// roi_heads/roi_heads.py:213-231).  One thread per candidate box, the image's valid gts in LDS.
//   max_iou[n][p]  = max_g IoU(gt[n][g], box[p])   (-1 when the image has no valid gt)
//   arg[n][p]      = first g (compacted order) attaining it
//   gt_max[n][g]   = max_p IoU (as float bits, via atomicMax; IoU >= 0 so bit order == value order)
#define MB_MAXG 256
__device__ __forceinline__ float iou_d2(const float4& a, const float4& b) {
  const float aa = (a.z - a.x) * (a.w - a.y), ab = (b.z - b.x) * (b.w - b.y);
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
  const float h = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float inter = w * h;
  return inter > 0.f ? inter / (aa + ab - inter) : 0.f;
}

// valid gts of image n, compacted in their original order (256 threads, G <= MB_MAXG): sidx[k] = k-th valid slot; returns their number
__device__ __forceinline__ int compact_valid_gts(const unsigned char* __restrict__ gt_valid, int n, int G, int* sidx, int* wcnt) {
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  const bool v = t < G && gt_valid[n * G + t];
  const unsigned long long b = __ballot(v);
  if (lane == 0) wcnt[w] = __popcll(b);
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += wcnt[i];
  if (v) sidx[base + __popcll(b & ((1ull << lane) - 1ull))] = t;
  const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(256) void match_boxes_kernel(const float* __restrict__ boxes, long long box_img_stride, int P,
                                                        const float* __restrict__ gt_boxes, const unsigned char* __restrict__ gt_valid,
                                                        int G, float* __restrict__ max_iou, int* __restrict__ arg,
                                                        unsigned* __restrict__ gt_max_bits) {
  __shared__ float4 sg[MB_MAXG];
  __shared__ int sidx[MB_MAXG];
  __shared__ unsigned smax[MB_MAXG];
  __shared__ int wcnt[4];
  const int n = blockIdx.y;
  const int Gv = compact_valid_gts(gt_valid, n, G, sidx, wcnt);
  for (int k = threadIdx.x; k < Gv; k += blockDim.x) {
    sg[k] = ((const float4*)gt_boxes)[(size_t)n * G + sidx[k]];
    smax[k] = 0u;
  }
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) {
    const float4 b = *(const float4*)(boxes + (size_t)n * box_img_stride + (size_t)p * 4);
    float best = -1.f;
    int bi = 0;
    for (int k = 0; k < Gv; ++k) {
      const float v = iou_d2(sg[k], b);
      if (v > best) { best = v; bi = k; }
      // most pairs are disjoint and the maximum settles quickly: the LDS atomic only when this pair can still raise it
      if (gt_max_bits && v > 0.f && __float_as_uint(v) > ((volatile unsigned*)smax)[k]) atomicMax(&smax[k], __float_as_uint(v));
    }
    max_iou[(size_t)n * P + p] = best;
    arg[(size_t)n * P + p] = Gv > 0 ? sidx[bi] : 0;
  }
  __syncthreads();
  if (gt_max_bits)
    for (int k = threadIdx.x; k < Gv; k += blockDim.x) atomicMax(&gt_max_bits[(size_t)n * G + sidx[k]], smax[k]);
}

// lowq[n][p] = 1 if IoU(gt g, box p) == gt_max[n][g] for some valid g  (Matcher.set_low_quality_matches_)
__global__ __launch_bounds__(256) void match_lowq_kernel(const float* __restrict__ boxes, long long box_img_stride, int P,
                                                       const float* __restrict__ gt_boxes, const unsigned char* __restrict__ gt_valid,
                                                       int G, const unsigned* __restrict__ gt_max_bits,
                                                       unsigned char* __restrict__ lowq) {
  __shared__ float4 sg[MB_MAXG];
  __shared__ float smax[MB_MAXG];
  __shared__ int sidx[MB_MAXG];
  __shared__ int wcnt[4];
  const int n = blockIdx.y;
  const int scount = compact_valid_gts(gt_valid, n, G, sidx, wcnt);
  for (int k = threadIdx.x; k < scount; k += blockDim.x) {
    sg[k] = ((const float4*)gt_boxes)[(size_t)n * G + sidx[k]];
    smax[k] = __uint_as_float(gt_max_bits[(size_t)n * G + sidx[k]]);
  }
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float4 b = *(const float4*)(boxes + (size_t)n * box_img_stride + (size_t)p * 4);
  unsigned char f = 0;
  for (int k = 0; k < scount; ++k)
    if (iou_d2(sg[k], b) == smax[k]) f = 1;
  lowq[(size_t)n * P + p] = f;
}

static bool env_on(const char* name) {   // A/B switches, read once per process: on unless set to "0"
  const char* v = getenv(name);
  return !(v && v[0] == '0');
}

// ---------------------------------------------------------------------------------------------
// Multi-level RoIAlign, aligned=True, sampling_ratio=0 (torchvision.ops.roi_align via D2 ROIPooler,
// roi_heads/roi_heads.py:28-45,118 [D2-recall]).  Features NHWC per level, output [R][PH][PW][C].
// Level: clamp(floor(canonical_level + log2(sqrt(area)/canonical_size + 1e-8)), min_level, max_level).
struct RoiLevels {
  const void* feat[4];   // element type = the launch's T
  float* dfeat[4];
  int H[4], W[4];
  float scale[4];
  int num_levels, min_level;
};

__device__ __forceinline__ int roi_level(const float4& b, const RoiLevels& L) {
  const float area = (b.z - b.x) * (b.w - b.y);
  int lvl = (int)floorf(4.f + log2f(sqrtf(area) / 224.f + 1e-8f));
  lvl = min(max(lvl, L.min_level), L.min_level + L.num_levels - 1);
  return lvl - L.min_level;
}

// The gh x gw bilinear samples of a bin are SEPARABLE: sample (iy, ix) puts weight wy(iy, Y) * wx(ix, X) on pixel (Y, X) and is
// skipped iff its y or its x is out of range, so the total weight of a pixel is WY[Y] * WX[X] with WY / WX summed per axis.
// A bin therefore touches (rows x cols) pixels once - (gh+1)(gw+1) loads / atomics instead of 4*gh*gw.
#define ROI_MAXT 18  // taps per axis kept in registers; larger bins (never with the D2 level assignment) take the slow path
struct AxisTaps {
  int lo, n;            // first pixel index, number of pixels
  float w[ROI_MAXT];
};

__device__ __forceinline__ bool axis_taps(float start, float bin, int g, int size, AxisTaps& t) {
  // accumulate the weights of the g samples of one axis onto pixel indices; returns false if they do not fit
  int lo = 1 << 30, hi = -1;
  for (int i = 0; i < g; ++i) {
    float v = start + ((float)i + 0.5f) * bin / (float)g;
    if (v < -1.f || v > (float)size) continue;
    if (v <= 0.f) v = 0.f;
    int l = (int)v;
    if (l >= size - 1) l = size - 1;
    const int h = l >= size - 1 ? l : l + 1;
    lo = min(lo, l);
    hi = max(hi, h);
  }
  t.lo = lo;
  t.n = hi >= lo ? hi - lo + 1 : 0;
  if (t.n > ROI_MAXT) return false;
#pragma unroll
  for (int k = 0; k < ROI_MAXT; ++k) t.w[k] = 0.f;
  for (int i = 0; i < g; ++i) {
    float v = start + ((float)i + 0.5f) * bin / (float)g;
    if (v < -1.f || v > (float)size) continue;
    if (v <= 0.f) v = 0.f;
    int l = (int)v, h;
    if (l >= size - 1) { h = l = size - 1; v = (float)l; } else h = l + 1;
    const float fr = v - (float)l;
#pragma unroll
    for (int k = 0; k < ROI_MAXT; ++k) {   // register array: no dynamic indexing
      if (k == l - lo) t.w[k] += 1.f - fr;
      if (k == h - lo) t.w[k] += fr;
    }
  }
  return true;
}

template <bool BWD, typename T>
__global__ __launch_bounds__(64) void roi_align_kernel(RoiLevels L, const float* __restrict__ rois, const int* __restrict__ roi_batch,
                                                     const unsigned char* __restrict__ roi_valid, int C, int PH, int PW,
                                                     T* __restrict__ out /* fwd: y, bwd: dy */) {
  // one block per (roi, bin); 64 lanes x 4 channels cover up to 256 channels per pass
  const int bin = blockIdx.x % (PH * PW);
  const int r = blockIdx.x / (PH * PW);
  const int ph = bin / PW, pw = bin % PW;
  const int C4 = C >> 2;
  T* o = out + ((size_t)r * PH * PW + bin) * C;
  const bool ok = roi_valid ? roi_valid[r] != 0 : true;
  if (!ok) {
    if (!BWD)
      for (int c = threadIdx.x; c < C4; c += 64) st4(o, c, f32x4{0.f, 0.f, 0.f, 0.f});
    return;
  }
  const float4 b = ((const float4*)rois)[r];
  const int li = roi_level(b, L);
  const int H = L.H[li], W = L.W[li];
  const float sc = L.scale[li];
  const float x1 = b.x * sc - 0.5f, y1 = b.y * sc - 0.5f, x2 = b.z * sc - 0.5f, y2 = b.w * sc - 0.5f;
  const float rw = x2 - x1, rh = y2 - y1;
  const float bw = rw / (float)PW, bh = rh / (float)PH;
  const int gh = (int)ceilf(rh / (float)PH), gw = (int)ceilf(rw / (float)PW);
  const float cnt = fmaxf((float)(gh * gw), 1.f);
  const int n = roi_batch[r];
  const T* f = (const T*)L.feat[li] + (size_t)n * H * W * C;
  float* df = BWD ? L.dfeat[li] + (size_t)n * H * W * C : nullptr;
  AxisTaps ty, tx;
  const bool fits = axis_taps(y1 + ph * bh, bh, gh, H, ty) && axis_taps(x1 + pw * bw, bw, gw, W, tx);
  if (fits) {
    for (int c = threadIdx.x; c < C4; c += 64) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (BWD) {
        g = ld4((const T*)o, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] /= cnt;
      }
#pragma unroll 1
      for (int a = 0; a < ty.n; ++a) {
        float wy = 0.f;
#pragma unroll
        for (int k = 0; k < ROI_MAXT; ++k) wy = k == a ? ty.w[k] : wy;
        if (wy == 0.f) continue;
#pragma unroll 1
        for (int bq = 0; bq < tx.n; ++bq) {
          float wx = 0.f;
#pragma unroll
          for (int k = 0; k < ROI_MAXT; ++k) wx = k == bq ? tx.w[k] : wx;
          const float w = wy * wx;
          if (w == 0.f) continue;
          const size_t off = ((size_t)(ty.lo + a) * W + (tx.lo + bq)) * C4 + c;
          if (!BWD) {
            const f32x4 v = ld4(f, off);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += w * v[e];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(df + off * 4 + e, g[e] * w);
          }
        }
      }
      if (!BWD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] /= cnt;
        st4(o, c, acc);
      }
    }
    return;
  }
  // generic path: every sample, four taps each
  for (int c = threadIdx.x; c < C4; c += 64) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    if (BWD) {
      g = ld4((const T*)o, c);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] /= cnt;
    }
    for (int iy = 0; iy < gh; ++iy) {
      float y = y1 + ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = x1 + pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
        float yy = y;
        if (yy < -1.f || yy > (float)H || x < -1.f || x > (float)W) continue;
        if (yy <= 0.f) yy = 0.f;
        if (x <= 0.f) x = 0.f;
        int yl = (int)yy, xl = (int)x, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
        const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const size_t o1 = ((size_t)yl * W + xl) * C4 + c, o2 = ((size_t)yl * W + xh) * C4 + c;
        const size_t o3 = ((size_t)yh * W + xl) * C4 + c, o4 = ((size_t)yh * W + xh) * C4 + c;
        if (!BWD) {
          const f32x4 v1 = ld4(f, o1), v2 = ld4(f, o2), v3 = ld4(f, o3), v4 = ld4(f, o4);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] += w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            atomicAdd(df + o1 * 4 + e, g[e] * w1);
            atomicAdd(df + o2 * 4 + e, g[e] * w2);
            atomicAdd(df + o3 * 4 + e, g[e] * w3);
            atomicAdd(df + o4 * 4 + e, g[e] * w4);
          }
        }
      }
    }
    if (!BWD) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] /= cnt;
      st4(o, c, acc);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// RoIAlign forward, one 256-thread workgroup per ROI (PH, PW <= 7).  roi_align_kernel gives every (roi, bin) its own wave, and every
// lane of it rebuilds both axes' tap tables in registers (18-entry arrays read through compare / select chains: the kernel was bound by
// those, 470 us for 150 MB of output with nothing else in flight).  Here 14 threads build the 7 + 7 per-axis tables of the ROI ONCE into
// LDS (same axis_taps), then each thread takes (bin, 16-byte channel group) items: <= 3 x 3 taps of one 16-byte load each for the ROI
// sizes the level assignment produces.  Same per-channel arithmetic in the same order as roi_align_kernel (identical up to the FMA
// contraction of the sample coordinates: tests/test_rcnn_kernels_gpu.py).
template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_roi_kernel(RoiLevels L, const float* __restrict__ rois, const int* __restrict__ roi_batch,
                                                              const unsigned char* __restrict__ roi_valid, int C, int PH, int PW,
                                                              T* __restrict__ out) {
  constexpr int V = 16 / (int)sizeof(T);          // channels per 16-byte access
  __shared__ float wt[2][7][ROI_MAXT];
  __shared__ int lo_[2][7], n_[2][7];
  __shared__ int all_fit;
  const int r = blockIdx.x, tid = threadIdx.x;
  const int CQ = C / V, items = PH * PW * CQ;
  T* o = out + (size_t)r * PH * PW * C;
  if (roi_valid && roi_valid[r] == 0) {
    for (int it = tid; it < items; it += 256) {
      uint4 z = {0, 0, 0, 0};
      ((uint4*)o)[it] = z;
    }
    return;
  }
  const float4 b = ((const float4*)rois)[r];
  const int li = roi_level(b, L);
  const int H = L.H[li], W = L.W[li];
  const float sc = L.scale[li];
  const float x1 = b.x * sc - 0.5f, y1 = b.y * sc - 0.5f, x2 = b.z * sc - 0.5f, y2 = b.w * sc - 0.5f;
  const float rw = x2 - x1, rh = y2 - y1;
  const float bw = rw / (float)PW, bh = rh / (float)PH;
  const int gh = (int)ceilf(rh / (float)PH), gw = (int)ceilf(rw / (float)PW);
  const float cnt = fmaxf((float)(gh * gw), 1.f);
  const T* f = (const T*)L.feat[li] + (size_t)roi_batch[r] * H * W * C;
  if (tid == 0) all_fit = 1;
  __syncthreads();
  if (tid < PH + PW) {
    const bool isx = tid >= PH;
    const int k = isx ? tid - PH : tid;
    AxisTaps t;
    const bool fits = isx ? axis_taps(x1 + k * bw, bw, gw, W, t) : axis_taps(y1 + k * bh, bh, gh, H, t);
    if (!fits) all_fit = 0;     // benign race: every writer stores 0
    else {
      lo_[isx][k] = t.lo;
      n_[isx][k] = t.n;
#pragma unroll
      for (int q = 0; q < ROI_MAXT; ++q) wt[isx][k][q] = t.w[q];
    }
  }
  __syncthreads();
  if (all_fit) {
    for (int it = tid; it < items; it += 256) {
      const int bin = it / CQ, cq = it - bin * CQ;
      const int ph = bin / PW, pw = bin - ph * PW;
      float acc[V];
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = 0.f;
      const int ylo = lo_[0][ph], yn = n_[0][ph], xlo = lo_[1][pw], xn = n_[1][pw];
      for (int a = 0; a < yn; ++a) {
        const float wy = wt[0][ph][a];
        if (wy == 0.f) continue;
        for (int bq = 0; bq < xn; ++bq) {
          const float w = wy * wt[1][pw][bq];
          if (w == 0.f) continue;
          const T* src = f + ((size_t)(ylo + a) * W + (xlo + bq)) * C + cq * V;
          if constexpr (sizeof(T) == 2) {
            const bf16x8_t v = *(const bf16x8_t*)src;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += w * (float)v[e];
          } else {
            const f32x4 v = *(const f32x4*)src;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += w * v[e];
          }
        }
      }
      if constexpr (sizeof(T) == 2) {
        bf16x8_t ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (h16_t)(acc[e] / cnt);
        *(bf16x8_t*)(o + (size_t)bin * C + cq * V) = ov;
      } else {
        f32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = acc[e] / cnt;
        *(f32x4*)(o + (size_t)bin * C + cq * V) = ov;
      }
    }
    return;
  }
  // a bin wider than ROI_MAXT pixels (never with the D2 level assignment): every sample, four taps each - roi_align_kernel's generic path
  const int C4 = C >> 2;
  for (int it = tid; it < PH * PW * C4; it += 256) {
    const int bin = it / C4, c = it - bin * C4;
    const int ph = bin / PW, pw = bin - ph * PW;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int iy = 0; iy < gh; ++iy) {
      const float y = y1 + ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        float x = x1 + pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
        float yy = y;
        if (yy < -1.f || yy > (float)H || x < -1.f || x > (float)W) continue;
        if (yy <= 0.f) yy = 0.f;
        if (x <= 0.f) x = 0.f;
        int yl = (int)yy, xl = (int)x, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
        const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const f32x4 v1 = ld4(f, ((size_t)yl * W + xl) * C4 + c), v2 = ld4(f, ((size_t)yl * W + xh) * C4 + c);
        const f32x4 v3 = ld4(f, ((size_t)yh * W + xl) * C4 + c), v4 = ld4(f, ((size_t)yh * W + xh) * C4 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] /= cnt;
    st4(o + (size_t)bin * C, c, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// RoIAlign backward as a GATHER over output tiles: deterministic (no atomics), every gradient pixel written exactly once in its
// final element type (no zero-fill, no fp32 staging buffer, no conversion pass).
// One 256-thread workgroup owns an 8 x 8 pixel tile of one (level, image) gradient map with all C <= 256 channels (lane -> 4 channels,
// wave w -> tile rows 2w, 2w+1).  It scans the image's ROIs (they are contiguous: rois_per_image each) 256 at a time, keeps - in ROI
// order - those on its level whose sample footprint meets the tile, and for each of them builds the separable weight tables
//   WY[ph][ty] = sum over the gh samples of bin row ph of the bilinear weight they put on tile row ty   (WX alike)
// (the forward's sampling rule, sample by sample: skipped outside [-1, size], clamped at 0, top pixel clamp) and accumulates
//   g[ty][tx][c] += WY[ph][ty] * WX[pw][tx] / (gh * gw) * dy[roi][ph][pw][c]
// in fixed (roi, ph, pw) order.  The atomic scatter it replaces was the largest kernel of the Faster-RCNN step (6 ms of 52).
#define RB_T 8
struct RoiBwdArgs {
  void* dfeat[4];       // gradient maps [N][H][W][C], element type TO
  int H[4], W[4];
  float scale[4];
  int tile_start[5];    // first block of each level (tiles_y * tiles_x * N blocks per level)
  int num_levels, min_level, N, P, C, PH, PW;
};

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void roi_align_bwd_tiled(RoiBwdArgs a, const float* __restrict__ rois, const unsigned char* __restrict__ roi_valid,
                                                          const TI* __restrict__ dy) {
  __shared__ int list[256];
  __shared__ int wcount[4];
  __shared__ float wy[7 * RB_T], wx[7 * RB_T];   // PH, PW <= 7
  __shared__ float geo[8];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int li = 0;
#pragma unroll
  for (int l = 1; l < 4; ++l)
    if (l < a.num_levels && (int)blockIdx.x >= a.tile_start[l]) li = l;
  const int H = a.H[li], W = a.W[li];
  const int tx_n = (W + RB_T - 1) / RB_T, ty_n = (H + RB_T - 1) / RB_T;
  int b = blockIdx.x - a.tile_start[li];
  const int n = b / (tx_n * ty_n);
  b -= n * tx_n * ty_n;
  const int ty0 = (b / tx_n) * RB_T, tx0 = (b % tx_n) * RB_T;
  const float sc = a.scale[li];
  const int C4 = a.C >> 2;
  const bool cok = lane < C4;

  f32x4 acc[2][RB_T];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int x = 0; x < RB_T; ++x) acc[r][x] = f32x4{0.f, 0.f, 0.f, 0.f};

  RoiLevels LV;   // only for roi_level()
  LV.num_levels = a.num_levels;
  LV.min_level = a.min_level;
  for (int base = 0; base < a.P; base += 256) {
    // ---- which of these 256 ROIs touch the tile (kept in ROI order) ----
    const int slot = base + tid;
    bool hit = false;
    if (slot < a.P) {
      const int r = n * a.P + slot;
      if (!roi_valid || roi_valid[r]) {
        const float4 bx = ((const float4*)rois)[r];
        if (roi_level(bx, LV) == li) {
          const float y1 = bx.y * sc - 0.5f, y2 = bx.w * sc - 0.5f, x1 = bx.x * sc - 0.5f, x2 = bx.z * sc - 0.5f;
          // pixels a sample in [lo, hi] can put weight on: floor(max(lo, 0)) .. floor(hi) + 1 (a superset is harmless)
          const int ylo = (int)floorf(fmaxf(y1, 0.f)), yhi = (int)floorf(fmaxf(y2, 0.f)) + 1;
          const int xlo = (int)floorf(fmaxf(x1, 0.f)), xhi = (int)floorf(fmaxf(x2, 0.f)) + 1;
          hit = ylo <= ty0 + RB_T - 1 && yhi >= ty0 && xlo <= tx0 + RB_T - 1 && xhi >= tx0;
        }
      }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wcount[wid] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wid) off += wcount[w];
      total += wcount[w];
    }
    if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = slot;
    __syncthreads();
    // ---- accumulate them one by one ----
    for (int k = 0; k < total; ++k) {
      const int r = n * a.P + list[k];
      const float4 bx = ((const float4*)rois)[r];
      const float x1 = bx.x * sc - 0.5f, y1 = bx.y * sc - 0.5f, x2 = bx.z * sc - 0.5f, y2 = bx.w * sc - 0.5f;
      const float rw = x2 - x1, rh = y2 - y1;
      const float bw = rw / (float)a.PW, bh = rh / (float)a.PH;
      const int gh = (int)ceilf(rh / (float)a.PH), gw = (int)ceilf(rw / (float)a.PW);
      // weight tables: thread (axis, p, t) sums the taps of the bin's samples on tile row / column t
      if (tid < 2 * 7 * RB_T) {
        const int axis = tid / (7 * RB_T), rem = tid - axis * 7 * RB_T, pb = rem / RB_T, t = rem - pb * RB_T;
        const int nb = axis ? a.PW : a.PH;
        float wsum = 0.f;
        if (pb < nb) {
          const float start = (axis ? x1 : y1) + pb * (axis ? bw : bh), bin = axis ? bw : bh;
          const int g = axis ? gw : gh, size = axis ? W : H, pix = (axis ? tx0 : ty0) + t;
          for (int i = 0; i < g; ++i) {
            float v = start + ((float)i + 0.5f) * bin / (float)g;
            if (v < -1.f || v > (float)size) continue;
            if (v <= 0.f) v = 0.f;
            int l = (int)v, h;
            if (l >= size - 1) { h = l = size - 1; v = (float)l; } else h = l + 1;
            const float fr = v - (float)l;
            if (l == pix) wsum += 1.f - fr;
            if (h == pix) wsum += fr;
          }
        }
        (axis ? wx : wy)[pb * RB_T + t] = wsum;
      }
      __syncthreads();
      if (cok) {
        const float inv = 1.f / fmaxf((float)(gh * gw), 1.f);
        const TI* g0 = dy + (size_t)r * a.PH * a.PW * a.C;
        for (int ph = 0; ph < a.PH; ++ph) {
          const float w0 = wy[ph * RB_T + 2 * wid], w1 = wy[ph * RB_T + 2 * wid + 1];
          if (w0 == 0.f && w1 == 0.f) continue;              // wave-uniform
          for (int pw = 0; pw < a.PW; ++pw) {
            float wxs[RB_T];
            bool any = false;
#pragma unroll
            for (int x = 0; x < RB_T; ++x) { wxs[x] = wx[pw * RB_T + x]; any = any || wxs[x] != 0.f; }
            if (!any) continue;                               // uniform
            f32x4 g = ld4(g0 + (size_t)(ph * a.PW + pw) * a.C, lane);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] *= inv;
#pragma unroll
            for (int x = 0; x < RB_T; ++x) {
              const float c0 = w0 * wxs[x], c1 = w1 * wxs[x];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                acc[0][x][e] += c0 * g[e];
                acc[1][x][e] += c1 * g[e];
              }
            }
          }
        }
      }
      __syncthreads();
    }
  }
  if (cok) {
    TO* out = (TO*)a.dfeat[li] + (size_t)n * H * W * a.C;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int y = ty0 + 2 * wid + r;
      if (y >= H) continue;
#pragma unroll
      for (int x = 0; x < RB_T; ++x) {
        if (tx0 + x >= W) continue;
        st4(out + ((size_t)y * W + tx0 + x) * a.C, lane, acc[r][x]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Softmax focal loss of the ROI head (roi_heads/fast_rcnn.py:925-936 + FocalLoss :1405-1429):
//   CE = logsumexp(x) - x[t];  p = exp(-CE);  loss = (1-p)^gamma * CE      (gamma 1.5), summed.
// Rows with target < 0 are skipped.  One wave per row.
__global__ __launch_bounds__(256) void softmax_focal_fwd_kernel(const float* __restrict__ logits, const int* __restrict__ target, int R, int C,
                                                              float gamma, float* __restrict__ partial) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float acc = 0.f;
  for (int r = blockIdx.x * 4 + wid; r < R; r += gridDim.x * 4) {
    const int t = target[r];
    if (t < 0) continue;
    const float* x = logits + (size_t)r * C;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
    m = wave_reduce_max(m);
    m = __shfl(m, 0, 64);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
    s = wave_reduce_sum(s);
    if (lane == 0) {
      const float ce = logf(s) + m - x[t];
      const float p = expf(-ce);
      acc += powf(1.f - p, gamma) * ce;
    }
  }
  if (lane == 0) red[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// dlogits[r][c] = coef * dL/dCE * (softmax_c - [c==t]);  dL/dCE = (1-p)^g + g (1-p)^(g-1) p CE
__global__ __launch_bounds__(256) void softmax_focal_bwd_kernel(const float* __restrict__ logits, const int* __restrict__ target, int R, int C,
                                                              float gamma, const float* __restrict__ coef, float* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const float k = coef[0];
  for (int r = blockIdx.x * 4 + wid; r < R; r += gridDim.x * 4) {
    const int t = target[r];
    const float* x = logits + (size_t)r * C;
    float* d = dlogits + (size_t)r * C;
    if (t < 0) {
      for (int c = lane; c < C; c += 64) d[c] = 0.f;
      continue;
    }
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
    m = wave_reduce_max(m);
    m = __shfl(m, 0, 64);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
    s = wave_reduce_sum(s);
    s = __shfl(s, 0, 64);
    const float ce = logf(s) + m - x[t];
    const float p = expf(-ce);
    const float om = 1.f - p;
    const float dce = powf(om, gamma) + (om > 0.f ? gamma * powf(om, gamma - 1.f) * p * ce : 0.f);
    for (int c = lane; c < C; c += 64) {
      const float sm = expf(x[c] - m) / s;
      d[c] = k * dce * (sm - (c == t ? 1.f : 0.f));
    }
  }
}

__global__ void sum_partials_f32(const float* __restrict__ partial, int n, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += (double)partial[i];
    out[0] = (float)s;
  }
}

// ---------------------------------------------------------------------------------------------
// RPN proposal decoding (D2 find_top_rpn_proposals up to the NMS [D2-recall]; proposal_generator/rpn.py runs it through
// detectron2's RPN.predict_proposals).  The head output is level-first: rows (level, image, pixel) x ch floats, the A objectness
// logits first, then the 4A deltas (anchor-major).
#define RPN_MAX_LEVELS 8
struct RpnLevels {
  int n;
  int row0[RPN_MAX_LEVELS];     // first head-output row of the level (N * hw rows per level)
  int hw[RPN_MAX_LEVELS];
  int anchor0[RPN_MAX_LEVELS];  // first anchor of the level in the concatenated anchor table
  int out0[RPN_MAX_LEVELS + 1]; // first candidate slot of the level in the per-image output (k_l = out0[l+1] - out0[l])
};

// sortable keys of every objectness logit, in memory order: descending key order == (logit desc, anchor index asc) inside the
// (level, image) row the logit belongs to; 63-bit non-negative (the select kernel's negative = empty)
__global__ __launch_bounds__(256) void rpn_rank_keys_kernel(RpnLevels L, const float* __restrict__ head, int N, int A, int ch,
                                                          long long total, long long* __restrict__ keys) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int r = (int)(t / A), a = (int)(t - (long long)r * A);
  int l = 0;
#pragma unroll
  for (int i = 1; i < RPN_MAX_LEVELS; ++i)
    if (i < L.n && r >= L.row0[i]) l = i;
  const int p = (r - L.row0[l]) % L.hw[l];
  const int bits = __float_as_int(head[(size_t)r * ch + a]);
  const long long mono = (long long)(bits ^ ((bits >> 31) & 0x7FFFFFFF)) + 2147483648ll;
  keys[t] = mono * 2147483648ll + (2147483647ll - ((long long)p * A + a));
}

// one thread per (image, candidate): anchor index from the selected key, Box2BoxTransform.apply_deltas (separately rounded mul / add
// as the elementwise reference chain), clip to the image, the finite / min-size keep mask
__global__ __launch_bounds__(256) void rpn_decode_kernel(RpnLevels L, const long long* __restrict__ top, int maxk, const float* __restrict__ head,
                                                       const float* __restrict__ anchors, const float* __restrict__ image_hw, int N, int A,
                                                       int ch, float wx, float wy, float ww, float wh, float scale_clamp, float min_size,
                                                       float* __restrict__ boxes, float* __restrict__ scores, int* __restrict__ lvls,
                                                       unsigned char* __restrict__ keep) {
  const int K = L.out0[L.n];
  const int j = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (j >= K) return;
  int l = 0;
#pragma unroll
  for (int i = 1; i < RPN_MAX_LEVELS; ++i)
    if (i < L.n && j >= L.out0[i]) l = i;
  const size_t o = (size_t)n * K + j;
  const long long key = top[((size_t)l * N + n) * maxk + (j - L.out0[l])];
  lvls[o] = l;
  if (key < 0) {
    *(float4*)(boxes + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    scores[o] = 0.f;
    keep[o] = 0;
    return;
  }
  const int idx = (int)(2147483647ll - (key & 2147483647ll));
  const int p = idx / A, a = idx - p * A;
  const float* hrow = head + ((size_t)L.row0[l] + (size_t)n * L.hw[l] + p) * ch;
  const float sc = hrow[a];
  const float4 an = *(const float4*)(anchors + ((size_t)L.anchor0[l] + idx) * 4);
  const float d0 = hrow[A + a * 4], d1 = hrow[A + a * 4 + 1], d2 = hrow[A + a * 4 + 2], d3 = hrow[A + a * 4 + 3];
  const float w = an.z - an.x, h = an.w - an.y;
  const float cx = an.x + 0.5f * w, cy = an.y + 0.5f * h;
  const float dx = d0 / wx, dy = d1 / wy;
  float dw = d2 / ww, dh = d3 / wh;
  dw = dw > scale_clamp ? scale_clamp : dw;   // torch.clamp(max=): NaN stays NaN
  dh = dh > scale_clamp ? scale_clamp : dh;
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
  const bool finite = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2) && isfinite(sc);
  const float H = image_hw[n * 2], W = image_hw[n * 2 + 1];
  x1 = x1 < 0.f ? 0.f : x1; y1 = y1 < 0.f ? 0.f : y1; x2 = x2 < 0.f ? 0.f : x2; y2 = y2 < 0.f ? 0.f : y2;
  x1 = x1 > W ? W : x1; y1 = y1 > H ? H : y1; x2 = x2 > W ? W : x2; y2 = y2 > H ? H : y2;
  *(float4*)(boxes + o * 4) = make_float4(x1, y1, x2, y2);
  scores[o] = finite ? sc : 0.f;
  keep[o] = (finite && (x2 - x1) > min_size && (y2 - y1) > min_size) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------
// RPN losses on the sampled anchors (proposal_generator/rpn.py:153-225): sum of BCE-with-logits over the sampled positives and
// negatives (weighted by the matched pseudo box's score when gt_scores is given - negatives too) and sum of |delta - target|
// over the positives, plus d(sum)/d(logit) per slot and d(sum)/d(delta) per positive slot.  The logits / deltas are read either from
// dense [N][R] / [N][R][4] tensors or straight from the level-first head output (rows (level, image, pixel) x ch: A logits, then 4A
// anchor-major deltas).  One workgroup, fixed summation order: deterministic.
struct RpnLossArgs {
  RpnLevels L;
  int head, N, A, ch, R, npos, nneg, G;
  const float* obj;
  const float* deltas;
  const float* anchors;
  const long long* pos_idx;
  const unsigned char* pos_valid;
  const long long* neg_idx;
  const unsigned char* neg_valid;
  const int* matched;
  const unsigned char* has_gt;
  const float* gt_boxes;
  const float* gt_scores;
  float wx, wy, ww, wh;
};

// element offsets of anchor r's logit and of its first delta for image n
__device__ __forceinline__ void rpn_locate(const RpnLevels& L, int head, int n, int r, int A, int ch, int R, size_t& o_obj, size_t& o_dl) {
  if (!head) {
    o_obj = (size_t)n * R + r;
    o_dl = ((size_t)n * R + r) * 4;
    return;
  }
  int l = 0;
#pragma unroll
  for (int i = 1; i < RPN_MAX_LEVELS; ++i)
    if (i < L.n && r >= L.anchor0[i]) l = i;
  const int q = r - L.anchor0[l], p = q / A, a = q - p * A;
  const size_t row = (size_t)L.row0[l] + (size_t)n * L.hw[l] + p;
  o_obj = row * ch + a;
  o_dl = row * ch + A + a * 4;
}

__global__ __launch_bounds__(256) void rpn_loss_fwd_kernel(RpnLossArgs p, float* __restrict__ sums, float* __restrict__ gobj,
                                                         float* __restrict__ gdl) {
  __shared__ float red[2][256];
  const int S = p.npos + p.nneg;
  float cls = 0.f, loc = 0.f;
  for (int slot = threadIdx.x; slot < p.N * S; slot += blockDim.x) {
    const int n = slot / S, j = slot - n * S;
    const bool pos = j < p.npos;
    const long long idx = pos ? p.pos_idx[(size_t)n * p.npos + j] : p.neg_idx[(size_t)n * p.nneg + (j - p.npos)];
    const bool valid = pos ? p.pos_valid[(size_t)n * p.npos + j] : p.neg_valid[(size_t)n * p.nneg + (j - p.npos)];
    size_t oo, od;
    rpn_locate(p.L, p.head, n, (int)idx, p.A, p.ch, p.R, oo, od);
    const float x = p.obj[oo], t = pos ? 1.f : 0.f;
    const bool hg = p.has_gt[n] != 0;
    const int g = p.matched[(size_t)n * p.R + idx];
    float w = valid ? 1.f : 0.f;
    if (p.gt_scores) w = w * (hg ? p.gt_scores[(size_t)n * p.G + g] : 0.f);
    const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    // (selects, not products: a slot that carries no weight - an empty sampling slot, an image without boxes - contributes nothing even
    // when the logit its index happens to name is not finite; the reference never indexes it)
    if (w != 0.f) cls += bce * w;
    gobj[slot] = w != 0.f ? (1.f / (1.f + expf(-x)) - t) * w : 0.f;
    if (pos) {
      const bool pv = valid && hg;
      float4 a = make_float4(0.f, 0.f, 1.f, 1.f), b = a;
      if (pv) {
        a = *(const float4*)(p.anchors + (size_t)idx * 4);
        b = *(const float4*)(p.gt_boxes + ((size_t)n * p.G + g) * 4);
      }
      const float sw = a.z - a.x, sh = a.w - a.y, sx = a.x + 0.5f * sw, sy = a.y + 0.5f * sh;
      const float tw = b.z - b.x, th = b.w - b.y, tx = b.x + 0.5f * tw, ty = b.y + 0.5f * th;
      const float tgt[4] = {p.wx * (tx - sx) / sw, p.wy * (ty - sy) / sh, p.ww * logf(tw / sw), p.wh * logf(th / sh)};
      const float m = pv ? 1.f : 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = p.deltas[od + c] - tgt[c];
        if (pv) loc += fabsf(d);
        gdl[((size_t)n * p.npos + j) * 4 + c] = (d > 0.f ? 1.f : d < 0.f ? -1.f : 0.f) * m;
      }
    }
  }
  red[0][threadIdx.x] = cls;
  red[1][threadIdx.x] = loc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { sums[0] = red[0][0]; sums[1] = red[1][0]; }
}

// gradient of (gout[0] * cls_sum + gout[1] * loc_sum) scattered to the sampled anchors' logits / deltas (the caller zero-fills the
// gradient buffers; the valid slots of an image are distinct anchors, invalid slots carry no gradient and are skipped)
__global__ __launch_bounds__(256) void rpn_loss_bwd_kernel(RpnLossArgs p, const float* __restrict__ gobj, const float* __restrict__ gdl,
                                                         const float* __restrict__ gout_cls, const float* __restrict__ gout_loc,
                                                         float* __restrict__ grad_obj, float* __restrict__ grad_deltas) {
  const int S = p.npos + p.nneg;
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= p.N * S) return;
  const int n = slot / S, j = slot - n * S;
  const bool pos = j < p.npos;
  const bool valid = pos ? p.pos_valid[(size_t)n * p.npos + j] : p.neg_valid[(size_t)n * p.nneg + (j - p.npos)];
  if (!valid) return;
  const long long idx = pos ? p.pos_idx[(size_t)n * p.npos + j] : p.neg_idx[(size_t)n * p.nneg + (j - p.npos)];
  size_t oo, od;
  rpn_locate(p.L, p.head, n, (int)idx, p.A, p.ch, p.R, oo, od);
  grad_obj[oo] = gobj[slot] * gout_cls[0];
  if (pos) {
    const float g1 = gout_loc[0];
#pragma unroll
    for (int c = 0; c < 4; ++c) grad_deltas[od + c] = gdl[((size_t)n * p.npos + j) * 4 + c] * g1;
  }
}

// ---------------------------------------------------------------------------------------------
// Anchor / proposal labelling and random subsampling (proposal_generator/rpn.py:112-148 -> D2 label_and_sample_anchors +
// subsample_labels; roi_heads/roi_heads.py:141-270 -> D2 add_ground_truth_to_proposals, Matcher, _sample_proposals) without the
// chain of elementwise / topk / gather launches.  Random choice = "the k smallest of one uniform key per slot" (the keys come from
// the caller: device RNG in training, injected in the parity tests); equal keys are taken in slot order.
//
// RPN, R ~ 2e5 anchors per image: rpn_sample_keys_kernel labels every anchor (IoU < lo: negative, >= hi or low-quality match: positive,
// image without gt: all negative) and writes one sortable 63-bit key per anchor into a row of positives and a row of negatives
// (-1 = not a candidate; descending key order == ascending random key, then ascending anchor index) - the input of the exact radix
// select utv2_topk_rows_i64; rpn_sample_unpack_kernel turns the selected keys into the sampler's index / valid arrays.
__global__ __launch_bounds__(256) void rpn_sample_keys_kernel(const float* __restrict__ mx, const unsigned char* __restrict__ lowq,
                                                            const unsigned char* __restrict__ gt_valid, int G,
                                                            const float* __restrict__ keys, int N, int R, float lo, float hi,
                                                            long long* __restrict__ out) {
  const int n = blockIdx.y;
  int any = 0;
  for (int g = threadIdx.x; g < G; g += blockDim.x) any |= gt_valid[(size_t)n * G + g];
  const int has_gt = __syncthreads_or(any);
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const size_t o = (size_t)n * R + r;
  const float v = mx[o];
  int label = v < lo ? 0 : -1;
  if (v >= hi) label = 1;
  if (lowq[o]) label = 1;
  if (!has_gt) label = 0;
  const long long k = ((long long)(0x7fffffff - (__float_as_int(keys[o]) & 0x7fffffff)) << 32) | (long long)(0xffffffffu - (unsigned)r);
  out[o] = label == 1 ? k : -1;
  out[(size_t)N * R + o] = label == 0 ? k : -1;
}

__global__ __launch_bounds__(256) void rpn_sample_unpack_kernel(const long long* __restrict__ top, int k, int N, int npos_max, int nneg_max,
                                                              const unsigned char* __restrict__ gt_valid, int G,
                                                              long long* __restrict__ pos_idx, unsigned char* __restrict__ pos_valid,
                                                              long long* __restrict__ neg_idx, unsigned char* __restrict__ neg_valid,
                                                              unsigned char* __restrict__ has_gt) {
  const int n = blockIdx.x;
  int any = 0, cnt = 0;
  for (int g = threadIdx.x; g < G; g += blockDim.x) any |= gt_valid[(size_t)n * G + g];
  for (int j = threadIdx.x; j < npos_max; j += blockDim.x) {
    const long long key = j < k ? top[(size_t)n * k + j] : -1;
    const bool ok = key >= 0;
    pos_idx[(size_t)n * npos_max + j] = ok ? (long long)(0xffffffffu - (unsigned)(key & 0xffffffffll)) : 0;
    pos_valid[(size_t)n * npos_max + j] = ok;
    cnt += ok;
  }
  any = __syncthreads_or(any);
  __shared__ int npos;
  if (threadIdx.x == 0) npos = 0;
  __syncthreads();
  if (cnt) atomicAdd(&npos, cnt);
  __syncthreads();
  if (threadIdx.x == 0) has_gt[n] = any != 0;
  const int room = nneg_max - npos;   // negatives fill what the positives leave of the per-image batch
  for (int j = threadIdx.x; j < nneg_max; j += blockDim.x) {
    const long long key = j < k ? top[(size_t)(N + n) * k + j] : -1;
    const bool ok = key >= 0 && j < room;
    neg_idx[(size_t)n * nneg_max + j] = key >= 0 ? (long long)(0xffffffffu - (unsigned)(key & 0xffffffffll)) : 0;
    neg_valid[(size_t)n * nneg_max + j] = ok;
  }
}

// ROI heads, P <= 4096 proposals (+ appended gt boxes) per image: one workgroup per image sorts (group, key, slot) in LDS - group 0
// foreground (matched IoU >= thr), 1 background, 2 not a candidate - takes the first <= nfg_max foreground and fills the batch with
// background, and gathers everything the losses need for the B sampled slots (foreground first, each part in ascending key order -
// the order D2's cat + the reference's stable valid-first packing give).
#define ROI_SAMPLE_MAX 4096
__global__ __launch_bounds__(512) void roi_sample_kernel(const float* __restrict__ pb, const unsigned char* __restrict__ pv,
                                                       const float* __restrict__ mx, const int* __restrict__ arg,
                                                       const float* __restrict__ keys, int P, const float* __restrict__ gt_boxes,
                                                       const int* __restrict__ gt_classes, const unsigned char* __restrict__ gt_valid,
                                                       const float* __restrict__ gt_scores, const float* __restrict__ gt_std, int G,
                                                       float iou_thr, int num_classes, int B, int nfg_max, int npow2,
                                                       float* __restrict__ out_prop, long long* __restrict__ out_cls,
                                                       float* __restrict__ out_gtb, unsigned char* __restrict__ out_valid,
                                                       long long* __restrict__ out_idx, float* __restrict__ out_conf,
                                                       float* __restrict__ out_std) {
  __shared__ unsigned long long sk[ROI_SAMPLE_MAX];
  __shared__ int cnt[2];
  const int n = blockIdx.x, t = threadIdx.x;
  int any = 0;
  for (int g = t; g < G; g += blockDim.x) any |= gt_valid[(size_t)n * G + g];
  if (t < 2) cnt[t] = 0;
  const int has_gt = __syncthreads_or(any);
  for (int i = t; i < npow2; i += blockDim.x) {
    unsigned long long grp = 2, kb = 0;
    if (i < P && pv[(size_t)n * P + i]) {
      const bool fgm = has_gt && mx[(size_t)n * P + i] >= iou_thr;
      const int cls = fgm ? gt_classes[(size_t)n * G + arg[(size_t)n * P + i]] : num_classes;
      grp = cls != num_classes ? 0 : 1;
      kb = (unsigned)__float_as_int(keys[(size_t)n * P + i]) & 0x7fffffffu;
      atomicAdd(&cnt[grp], 1);
    }
    sk[i] = (grp << 60) | (kb << 20) | (unsigned long long)i;
  }
  __syncthreads();
  for (int k = 2; k <= npow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int q = t; q < (npow2 >> 1); q += blockDim.x) {
        const int i = 2 * q - (q & (j - 1)), l = i + j;   // i has bit j clear, l = i | j
        const unsigned long long a = sk[i], b = sk[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { sk[i] = b; sk[l] = a; }
      }
      __syncthreads();
    }
  const int nfg = cnt[0], nbg = cnt[1];
  const int fsel = nfg < nfg_max ? nfg : nfg_max;
  const int bsel = nbg < B - fsel ? nbg : B - fsel;
  for (int j = t; j < B; j += blockDim.x) {
    const int src = j < fsel ? j : (j < fsel + bsel ? nfg + (j - fsel) : -1);
    const size_t o = (size_t)n * B + j;
    float4 box = make_float4(0.f, 0.f, 0.f, 0.f), gb = box, gs = box;
    long long cls = -1, idx = 0;
    float conf = 0.f;
    if (src >= 0) {
      const int i = (int)(sk[src] & 0xfffffull);
      idx = i;
      box = *(const float4*)(pb + ((size_t)n * P + i) * 4);
      cls = j < fsel ? (long long)gt_classes[(size_t)n * G + arg[(size_t)n * P + i]] : (long long)num_classes;
      if (has_gt) {
        const size_t g = (size_t)n * G + arg[(size_t)n * P + i];
        gb = *(const float4*)(gt_boxes + g * 4);
        if (gt_scores) conf = gt_scores[g];
        if (gt_std) gs = *(const float4*)(gt_std + g * 4);
      }
    }
    *(float4*)(out_prop + o * 4) = box;
    out_cls[o] = cls;
    *(float4*)(out_gtb + o * 4) = gb;
    out_valid[o] = src >= 0;
    out_idx[o] = idx;
    if (out_conf) out_conf[o] = conf;
    if (out_std) *(float4*)(out_std + o * 4) = gs;
  }
}

// ---------------------------------------------------------------------------------------------
// Box regression losses of the boundary-variance predictor on the sampled ROIs (roi_heads/fast_rcnn.py:938-1090: `box_reg_loss`
// nlloss / smooth_l1(beta 0) and `box_reg_pseudo_loss` tsbetter / smooth_l1), summed, with the derivatives w.r.t. the predicted
// deltas and std logits.  mode 0: L1 + 0.05 * sum(NLL * IoU(gt, decoded box)) with the gradient flowing through the IoU;
// 1: L1; 2: L1 on the boundaries where the teacher is more certain than the student by ts_better and above t_cert; 3: L1.
// Box2BoxXYXYTransform (box_regression.py:11-129): get_deltas divides by (side + 1), apply_deltas multiplies by the side (B3 kept).
// One workgroup, fixed summation order.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void roi_box_loss_kernel(const float* __restrict__ deltas, const float* __restrict__ stdl, long long ld,
                                                         const long long* __restrict__ cls, const float* __restrict__ prop,
                                                         const float* __restrict__ gtb, const float* __restrict__ gstd, int R,
                                                         int num_classes, int mode, float wx, float wy, float clampv, float ts_better,
                                                         float t_cert, float* __restrict__ sum, float* __restrict__ gd,
                                                         float* __restrict__ gs) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const long long c = cls[r];
    const bool fg = c >= 0 && c < num_classes;
    float4 pb = make_float4(0.f, 0.f, 1.f, 1.f), gb = pb;
    if (fg) {
      pb = *(const float4*)(prop + (size_t)r * 4);
      gb = *(const float4*)(gtb + (size_t)r * 4);
    }
    const float d[4] = {deltas[(size_t)r * ld], deltas[(size_t)r * ld + 1], deltas[(size_t)r * ld + 2], deltas[(size_t)r * ld + 3]};
    const float sl[4] = {stdl[(size_t)r * ld], stdl[(size_t)r * ld + 1], stdl[(size_t)r * ld + 2], stdl[(size_t)r * ld + 3]};
    const float sw = pb.z - pb.x + 1.f, sh = pb.w - pb.y + 1.f;
    const float t[4] = {wx * (gb.x - pb.x) / sw, wx * (gb.z - pb.z) / sw, wy * (gb.y - pb.y) / sh, wy * (gb.w - pb.w) / sh};
    float g_d[4] = {0.f, 0.f, 0.f, 0.f}, g_s[4] = {0.f, 0.f, 0.f, 0.f};
    const float fgf = fg ? 1.f : 0.f;
    if (mode == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float ct = 1.f - sigmoidf_(gstd ? gstd[(size_t)r * 4 + k] : 0.f), cs = 1.f - sigmoidf_(sl[k]);
        const float m = (ct > cs + ts_better && ct > t_cert && fg) ? 1.f : 0.f;
        const float df = d[k] - t[k];
        if (m != 0.f) acc += fabsf(df);     // (selected rows only: a non-finite delta on an unselected row must not reach the sum as inf * 0)
        g_d[k] = (df > 0.f ? 1.f : df < 0.f ? -1.f : 0.f) * m;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float df = d[k] - t[k];
        if (fg) acc += fabsf(df);
        g_d[k] = (df > 0.f ? 1.f : df < 0.f ? -1.f : 0.f) * fgf;
      }
      if (mode == 0) {
        const float w = pb.z - pb.x, h = pb.w - pb.y;
        const float wd[4] = {wx, wx, wy, wy};
        const float side[4] = {w, w, h, h};
        float q[4], dq[4];   // clamped delta / weight and its derivative w.r.t. the delta
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float v = d[k] / wd[k];
          q[k] = v > clampv ? clampv : v < -clampv ? -clampv : v;   // NaN stays NaN like torch.clamp
          dq[k] = (v >= -clampv && v <= clampv) ? side[k] / wd[k] : 0.f;
        }
        const float px1 = q[0] * w + pb.x, px2 = q[1] * w + pb.z, py1 = q[2] * h + pb.y, py2 = q[3] * h + pb.w;
        const float a1 = (gb.z - gb.x) * (gb.w - gb.y), a2 = (px2 - px1) * (py2 - py1);
        const float ltx = fmaxf(gb.x, px1), lty = fmaxf(gb.y, py1), rbx = fminf(gb.z, px2), rby = fminf(gb.w, py2);
        const float whx = fmaxf(rbx - ltx, 0.f), why = fmaxf(rby - lty, 0.f);
        const float I = whx * why, U = a1 + a2 - I;
        const float iou = fg ? I / U : 0.f;
        float nll = 0.f, sig[4], sq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          sig[k] = sigmoidf_(sl[k]);
          sq[k] = sig[k] * sig[k];
          const float df = t[k] - d[k];
          nll += df * df / (2.f * sq[k]) + 0.5f * logf(sq[k]);
        }
        nll += 2.f * 1.8378770664093453f;  // 2 log(2 pi)
        // (foreground rows only, as the reference indexes them: on a background row a std logit below -88 makes nll = inf - inf, and
        // NaN * 0 would put a NaN into the loss VALUE - the gradients below were always guarded)
        if (fg) acc += 0.05f * (nll * iou);
        if (fg) {
          // torch.max / torch.min send the gradient to the selected operand (half on a tie), clamp(min=0) passes it where its input >= 0
          const float cx = (rbx - ltx) >= 0.f ? 1.f : 0.f, cy = (rby - lty) >= 0.f ? 1.f : 0.f;
          const float mx1 = px1 > gb.x ? 1.f : px1 == gb.x ? 0.5f : 0.f, mx2 = px2 < gb.z ? 1.f : px2 == gb.z ? 0.5f : 0.f;
          const float my1 = py1 > gb.y ? 1.f : py1 == gb.y ? 0.5f : 0.f, my2 = py2 < gb.w ? 1.f : py2 == gb.w ? 0.5f : 0.f;
          // order of the deltas: x1, x2, y1, y2
          const float dI[4] = {-why * cx * mx1, why * cx * mx2, -whx * cy * my1, whx * cy * my2};
          const float dA[4] = {-(py2 - py1), (py2 - py1), -(px2 - px1), (px2 - px1)};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float diou = (dI[k] * (U + I) - I * dA[k]) / (U * U) * dq[k];
            const float df = d[k] - t[k];
            g_d[k] += 0.05f * (iou * df / sq[k] + nll * diou);
            g_s[k] = 0.05f * iou * (1.f - sig[k]) * (1.f - df * df / sq[k]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      gd[(size_t)r * 4 + k] = g_d[k];
      gs[(size_t)r * 4 + k] = g_s[k];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) sum[0] = red[0];
}

// ---------------------------------------------------------------------------------------------
// The predictor's inference chain (roi_heads/fast_rcnn.py:1094-1125,1162-1225 + D2 fast_rcnn_inference [D2-recall]) around the exact
// top-k and the class-aware NMS: three launches in place of ~55 ATen ops per teacher pass.
//   keys:   one thread per (image, proposal): Box2BoxXYXYTransform.apply_deltas (box_regression.py:88-128: divide by the weight, clamp to
//           +-scale_clamp, scale by the proposal's width / height, add to its corner), clip to the image, and for every foreground
//           class a sortable 63-bit key of its probability - (order-preserving float bits) << 32 | (2^32 - 1 - flat index) - or the
//           key of -1 when the class is not a candidate (probability <= thr, invalid slot, non-finite box or probabilities).
//           Descending key order == (probability desc, flat (proposal, class) index asc): the tie rule of the ATen chain it replaces.
//   gather: the k best keys of an image -> probability (recovered from the key), proposal row, class, decoded box, candidate flag
//   pack:   the NMS survivors -> padded detections (+ the raw std logits of their proposal rows, fast_rcnn.py:1118-1123)
__device__ __forceinline__ long long order_key_f32(float v, unsigned flat) {
  const int i = __float_as_int(v);
  const long long mono = (long long)(i ^ ((i >> 31) & 0x7FFFFFFF));
  return mono * 4294967296ll + (long long)(4294967295u - flat);
}

__global__ __launch_bounds__(256) void roi_infer_keys_kernel(const float* __restrict__ probs, const float* __restrict__ deltas,
                                                           const float* __restrict__ prop, const unsigned char* __restrict__ valid,
                                                           const float* __restrict__ whwh, int N, int P, int K, float wx, float wy,
                                                           float clampv, float thr, float* __restrict__ boxes, long long* __restrict__ keys) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * P) return;
  const int n = (int)(t / P), p = (int)(t - (long long)n * P);
  const float4 b = *(const float4*)(prop + t * 4);
  const float4 d = *(const float4*)(deltas + t * 4);
  const float w = b.z - b.x, h = b.w - b.y;
  auto cl = [clampv](float v) { return v < -clampv ? -clampv : (v > clampv ? clampv : v); };   // a NaN stays a NaN (torch.clamp)
  // ATen divides a tensor by a host scalar as a multiplication by its fp32 reciprocal: the same here, so that the boxes are bit-identical
  // to the op chain this replaces (a true division differs in the last bit for weights like 10)
  const float iwx = 1.f / wx, iwy = 1.f / wy;
  const float dl = cl(d.x * iwx), dr = cl(d.y * iwx), dd = cl(d.z * iwy), du = cl(d.w * iwy);
  float4 o;
  o.x = dl * w + b.x; o.y = dd * h + b.y; o.z = dr * w + b.z; o.w = du * h + b.w;
  bool ok = valid[t] != 0 && isfinite(o.x) && isfinite(o.y) && isfinite(o.z) && isfinite(o.w);
  const float4 lim = *(const float4*)(whwh + n * 4);
  o.x = fminf(fmaxf(o.x, 0.f), lim.x); o.y = fminf(fmaxf(o.y, 0.f), lim.y); o.z = fminf(fmaxf(o.z, 0.f), lim.z); o.w = fminf(fmaxf(o.w, 0.f), lim.w);
  *(float4*)(boxes + t * 4) = o;
  const float* pr = probs + t * (K + 1);
  for (int c = 0; c < K; ++c) ok = ok && isfinite(pr[c]);
  long long* kr = keys + (long long)n * P * K + (long long)p * K;
  for (int c = 0; c < K; ++c) {
    const float v = pr[c];
    kr[c] = order_key_f32((ok && v > thr) ? v : -1.f, (unsigned)(p * K + c));
  }
}

__global__ __launch_bounds__(256) void roi_infer_gather_kernel(const long long* __restrict__ top, const float* __restrict__ boxes, int N,
                                                             int P, int K, int k, float thr, float* __restrict__ sc, long long* __restrict__ rows,
                                                             int* __restrict__ cls, float* __restrict__ cb, unsigned char* __restrict__ valid) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)N * k) return;
  const int n = (int)(t / k);
  const long long key = top[t];
  const unsigned flat = 4294967295u - (unsigned)(key & 0xFFFFFFFFll);
  const int mono = (int)(key >> 32);
  const float v = __int_as_float(mono ^ ((mono >> 31) & 0x7FFFFFFF));
  const unsigned r = flat / (unsigned)K;
  sc[t] = v;
  rows[t] = (long long)r;
  cls[t] = (int)(flat - r * (unsigned)K);
  *(float4*)(cb + t * 4) = *(const float4*)(boxes + ((long long)n * P + r) * 4);
  valid[t] = v > thr ? 1 : 0;
}

__global__ __launch_bounds__(256) void roi_infer_pack_kernel(const int* __restrict__ kidx, const int* __restrict__ cnt, const float* __restrict__ cb,
                                                           const float* __restrict__ sc, const int* __restrict__ cls, const long long* __restrict__ rows,
                                                           const float* __restrict__ stdl, int N, int P, int k, int D, float* __restrict__ oboxes,
                                                           float* __restrict__ oscores, int* __restrict__ ocls, float* __restrict__ ostd,
                                                           long long* __restrict__ orows, unsigned char* __restrict__ ovalid) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * D) return;
  const int n = t / D, d = t - n * D;
  int ix = kidx[t];
  ix = ix < 0 ? 0 : ix;
  const long long src = (long long)n * k + ix;
  *(float4*)(oboxes + (long long)t * 4) = *(const float4*)(cb + src * 4);
  oscores[t] = sc[src];
  ocls[t] = cls[src];
  const long long r = rows[src];
  orows[t] = r;
  *(float4*)(ostd + (long long)t * 4) = *(const float4*)(stdl + ((long long)n * P + r) * 4);
  ovalid[t] = d < cnt[n] ? 1 : 0;
}

// survivors of utv2_nms_batched -> padded (boxes, scores, valid): the gather chain behind the RPN's NMS (D2 find_top_rpn_proposals)
__global__ __launch_bounds__(256) void nms_pack_kernel(const int* __restrict__ kidx, const int* __restrict__ cnt, const float* __restrict__ boxes,
                                                     const float* __restrict__ scores, int N, int M, int D, float* __restrict__ ob,
                                                     float* __restrict__ os, unsigned char* __restrict__ ov) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N * D) return;
  const int n = t / D, d = t - n * D;
  int ix = kidx[t];
  ix = ix < 0 ? 0 : ix;
  const long long src = (long long)n * M + ix;
  *(float4*)(ob + (long long)t * 4) = *(const float4*)(boxes + src * 4);
  os[t] = scores[src];
  ov[t] = d < cnt[n] ? 1 : 0;
}

extern "C" {

int utv2_nms_pack(const int* kidx, const int* cnt, const float* boxes, const float* scores, int N, int M, int D, float* oboxes, float* oscores,
                  unsigned char* ovalid, hipStream_t stream) {
  if (!kidx || !cnt || !boxes || !scores || !oboxes || !oscores || !ovalid || N < 1 || M < 1 || D < 1) return UTV2_EARG;
  hipLaunchKernelGGL(nms_pack_kernel, dim3((unsigned)cdiv((long long)N * D, 256)), dim3(256), 0, stream, kidx, cnt, boxes, scores, N, M, D, oboxes,
                     oscores, ovalid);
  return utv2_launch_status();
}


// boxes: [N][P][4] (box_img_stride = P*4) or shared anchors [P][4] (box_img_stride = 0).
// gt_max_bits (optional, [N][G] uint32, must be zeroed by the caller) receives max-over-boxes IoU bits.
int utv2_match_boxes(const float* boxes, int64_t box_img_stride, int N, int P, const float* gt_boxes,
                     const unsigned char* gt_valid, int G, float* max_iou, int* arg, unsigned* gt_max_bits,
                     hipStream_t stream) {
  if (!boxes || !gt_boxes || !gt_valid || !max_iou || !arg || G < 1 || G > MB_MAXG) return UTV2_EARG;
  if (P == 0) return UTV2_OK;
  hipLaunchKernelGGL(match_boxes_kernel, dim3(cdiv(P, 256), N), dim3(256), 0, stream, boxes, (long long)box_img_stride, P,
                     gt_boxes, gt_valid, G, max_iou, arg, gt_max_bits);
  return utv2_launch_status();
}

int utv2_match_lowq(const float* boxes, int64_t box_img_stride, int N, int P, const float* gt_boxes,
                    const unsigned char* gt_valid, int G, const unsigned* gt_max_bits, unsigned char* lowq,
                    hipStream_t stream) {
  if (!boxes || !gt_boxes || !gt_valid || !gt_max_bits || !lowq || G < 1 || G > MB_MAXG) return UTV2_EARG;
  if (P == 0) return UTV2_OK;
  hipLaunchKernelGGL(match_lowq_kernel, dim3(cdiv(P, 256), N), dim3(256), 0, stream, boxes, (long long)box_img_stride, P,
                     gt_boxes, gt_valid, G, gt_max_bits, lowq);
  return utv2_launch_status();
}

static int fill_levels(RoiLevels& L, int num_levels, int min_level, const void* const* feats, float* const* dfeats,
                       const int* H, const int* W, const float* scales) {
  if (num_levels < 1 || num_levels > 4) return UTV2_EARG;
  L.num_levels = num_levels;
  L.min_level = min_level;
  for (int i = 0; i < 4; ++i) {
    const bool on = i < num_levels;
    L.feat[i] = on && feats ? feats[i] : nullptr;
    L.dfeat[i] = on && dfeats ? dfeats[i] : nullptr;
    L.H[i] = on ? H[i] : 0;
    L.W[i] = on ? W[i] : 0;
    L.scale[i] = on ? scales[i] : 0.f;
  }
  return UTV2_OK;
}

// feats_host: host array of num_levels device pointers (NHWC level features of element type `dtype`, same C).
// rois [R][4] xyxy in image coordinates, roi_batch [R] image index, roi_valid [R] (optional).
// out [R][PH][PW][C] of element type `dtype`.
int utv2_roi_align_fwd(int num_levels, int min_level, const void* const* feats_host, const int* H_host, const int* W_host,
                       const float* scales_host, const float* rois, const int* roi_batch, const unsigned char* roi_valid,
                       int R, int C, int PH, int PW, void* out, int dtype, hipStream_t stream) {
  RoiLevels L;
  if (fill_levels(L, num_levels, min_level, feats_host, nullptr, H_host, W_host, scales_host) != UTV2_OK || (C & 3) || !rois ||
      !roi_batch || !out || (dtype != UTV2_F32 && dtype != UTV2_BF16))
    return UTV2_EARG;
  if (R == 0) return UTV2_OK;
  static const bool per_roi = env_on("UTV2_ROI_FWD_PER_ROI");   // A/B: 0 = one wave per (roi, bin)
  const int V = dtype == UTV2_BF16 ? 8 : 4;
  if (per_roi && PH <= 7 && PW <= 7 && C % V == 0) {
    if (dtype == UTV2_BF16)
      hipLaunchKernelGGL((roi_align_fwd_roi_kernel<h16_t>), dim3(R), dim3(256), 0, stream, L, rois, roi_batch, roi_valid, C, PH, PW, (h16_t*)out);
    else
      hipLaunchKernelGGL((roi_align_fwd_roi_kernel<float>), dim3(R), dim3(256), 0, stream, L, rois, roi_batch, roi_valid, C, PH, PW, (float*)out);
    return utv2_launch_status();
  }
  if (dtype == UTV2_BF16)
    hipLaunchKernelGGL((roi_align_kernel<false, h16_t>), dim3(R * PH * PW), dim3(64), 0, stream, L, rois, roi_batch, roi_valid, C, PH,
                       PW, (h16_t*)out);
  else
    hipLaunchKernelGGL((roi_align_kernel<false, float>), dim3(R * PH * PW), dim3(64), 0, stream, L, rois, roi_batch, roi_valid, C, PH,
                       PW, (float*)out);
  return utv2_launch_status();
}

// dfeats (fp32, += : caller zero-fills) receive the scatter of dy (element type `dtype`) through the same sampling pattern.
int utv2_roi_align_bwd(int num_levels, int min_level, float* const* dfeats_host, const int* H_host, const int* W_host,
                       const float* scales_host, const float* rois, const int* roi_batch, const unsigned char* roi_valid,
                       int R, int C, int PH, int PW, const void* dy, int dtype, hipStream_t stream) {
  RoiLevels L;
  if (fill_levels(L, num_levels, min_level, nullptr, dfeats_host, H_host, W_host, scales_host) != UTV2_OK || (C & 3) || !rois ||
      !roi_batch || !dy || (dtype != UTV2_F32 && dtype != UTV2_BF16))
    return UTV2_EARG;
  if (R == 0) return UTV2_OK;
  if (dtype == UTV2_BF16)
    hipLaunchKernelGGL((roi_align_kernel<true, h16_t>), dim3(R * PH * PW), dim3(64), 0, stream, L, rois, roi_batch, roi_valid, C, PH,
                       PW, (h16_t*)dy);
  else
    hipLaunchKernelGGL((roi_align_kernel<true, float>), dim3(R * PH * PW), dim3(64), 0, stream, L, rois, roi_batch, roi_valid, C, PH,
                       PW, (float*)dy);
  return utv2_launch_status();
}

/* Deterministic gather form (see roi_align_bwd_tiled): the ROIs of image n are rois[n*rois_per_image .. (n+1)*rois_per_image), C <= 256,
 * PH, PW <= 7; every element of every dfeats[l] ([N][H_l][W_l][C], element type out_dtype) is WRITTEN (no zero-fill needed). */
int utv2_roi_align_bwd_tiled(int num_levels, int min_level, void* const* dfeats_host, const int* H_host, const int* W_host,
                             const float* scales_host, const float* rois, const unsigned char* roi_valid, int N, int rois_per_image,
                             int C, int PH, int PW, const void* dy, int dy_dtype, int out_dtype, hipStream_t stream) {
  if (num_levels < 1 || num_levels > 4 || !dfeats_host || !H_host || !W_host || !scales_host || !rois || !dy || (C & 3) || C > 256 ||
      PH < 1 || PH > 7 || PW < 1 || PW > 7 || N < 1 || rois_per_image < 1 || (dy_dtype != UTV2_F32 && dy_dtype != UTV2_BF16) ||
      (out_dtype != UTV2_F32 && out_dtype != UTV2_BF16))
    return UTV2_EARG;
  RoiBwdArgs a;
  int blocks = 0;
  for (int l = 0; l < 4; ++l) {
    a.tile_start[l] = blocks;
    if (l < num_levels) {
      if (!dfeats_host[l]) return UTV2_EARG;
      a.dfeat[l] = dfeats_host[l]; a.H[l] = H_host[l]; a.W[l] = W_host[l]; a.scale[l] = scales_host[l];
      blocks += cdiv(H_host[l], RB_T) * cdiv(W_host[l], RB_T) * N;
    } else {
      a.dfeat[l] = nullptr; a.H[l] = a.W[l] = 1; a.scale[l] = 1.f;
    }
  }
  a.tile_start[4] = blocks;
  a.num_levels = num_levels; a.min_level = min_level; a.N = N; a.P = rois_per_image; a.C = C; a.PH = PH; a.PW = PW;
  const dim3 g(blocks), b(256);
  if (dy_dtype == UTV2_BF16) {
    if (out_dtype == UTV2_BF16) hipLaunchKernelGGL((roi_align_bwd_tiled<h16_t, h16_t>), g, b, 0, stream, a, rois, roi_valid, (const h16_t*)dy);
    else hipLaunchKernelGGL((roi_align_bwd_tiled<h16_t, float>), g, b, 0, stream, a, rois, roi_valid, (const h16_t*)dy);
  } else {
    if (out_dtype == UTV2_BF16) hipLaunchKernelGGL((roi_align_bwd_tiled<float, h16_t>), g, b, 0, stream, a, rois, roi_valid, (const float*)dy);
    else hipLaunchKernelGGL((roi_align_bwd_tiled<float, float>), g, b, 0, stream, a, rois, roi_valid, (const float*)dy);
  }
  return utv2_launch_status();
}

// The scalar tail of the Faster-RCNN UTv2 losses in ONE launch (the FCOS counterpart is utv2_fcos_loss_combine): raw kernel sums of the
// two loss branches -> the eight losses of the trainer's record_dict, their weighted total and d total / d raw sums.
//   loss_cls = focal / Rn, loss_box_reg = w_box * box / Rn   (Rn = number of sampled ROIs = targets >= 0, at least 1: fast_rcnn.py:925-936)
//   loss_rpn_cls = w_rpn_cls * rpn[0] / norm, loss_rpn_loc = w_rpn_loc * rpn[1] / norm   (norm = batch_size_per_image * images, rpn.py:214-224;
//   the RPN weights arrive squared: applied inside losses() and again in forward())
//   total = sum_k wt[k] * loss_k in record_dict order (engine/trainer.py:880-893: pseudo terms x UNSUP_LOSS_WEIGHT, loss_box_reg_pseudo x
//   UNSUP_REG_LOSS_WEIGHT, loss_rpn_loc_pseudo x 0).   rec[9] = {cls, box, rpn_cls, rpn_loc} x {supervised, pseudo}, total;  coef[8].
struct RcnnCombineArgs {
  const float* rpn[2];
  const float* focal[2];
  const float* box[2];
  const int* tgt[2];
  int ntgt[2];
  float rpn_norm[2];
  float w_rpn_cls, w_rpn_loc, w_box;
  float wt[8];
};

__global__ __launch_bounds__(64) void rcnn_loss_combine_kernel(RcnnCombineArgs a, float* __restrict__ rec, float* __restrict__ coef) {
  const int lane = threadIdx.x;
  float rn[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    float c = 0.f;
    for (int i = lane; i < a.ntgt[b]; i += 64) c += a.tgt[b][i] >= 0 ? 1.f : 0.f;   // exact in fp32 (counts < 2^24)
    c = wave_reduce_sum(c);
    rn[b] = fmaxf(__shfl(c, 0, 64), 1.f);
  }
  if (lane != 0) return;
  float total = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const float v[4] = {a.focal[b][0] / rn[b], a.w_box * (a.box[b][0] / rn[b]), a.w_rpn_cls * (a.rpn[b][0] / a.rpn_norm[b]),
                        a.w_rpn_loc * (a.rpn[b][1] / a.rpn_norm[b])};
    const float d[4] = {1.f / rn[b], a.w_box / rn[b], a.w_rpn_cls / a.rpn_norm[b], a.w_rpn_loc / a.rpn_norm[b]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rec[4 * b + k] = v[k];
      coef[4 * b + k] = a.wt[4 * b + k] * d[k];
      total += v[k] * a.wt[4 * b + k];
    }
  }
  rec[8] = total;
}

#define SF_BLOCKS 256
// loss_sum[0] = sum_r (1-p_r)^gamma * CE_r over rows with target >= 0.  ws >= 256 floats.
int utv2_softmax_focal_fwd(const float* logits, const int* target, int R, int C, float gamma, float* loss_sum, float* ws,
                           hipStream_t stream) {
  if (!logits || !target || !loss_sum || !ws) return UTV2_EARG;
  hipLaunchKernelGGL(softmax_focal_fwd_kernel, dim3(SF_BLOCKS), dim3(256), 0, stream, logits, target, R, C, gamma, ws);
  hipLaunchKernelGGL(sum_partials_f32, dim3(1), dim3(64), 0, stream, (const float*)ws, SF_BLOCKS, loss_sum);
  return utv2_launch_status();
}

// rpn_*: device float[2] {sum BCE, sum L1} of utv2_rpn_loss_fwd; focal_* / box_*: device float[1] (utv2_softmax_focal_fwd, utv2_roi_box_loss);
// tgt_*: the int32 class targets of the sampled ROIs (< 0 = empty slot); wt_host: host float[8], the trainer's weight per loss in the
// order {cls, box_reg, rpn_cls, rpn_loc} x {supervised, pseudo}.  rec: device float[9], coef: device float[8].
int utv2_rcnn_loss_combine(const float* rpn_sup, const float* rpn_uns, const float* focal_sup, const float* focal_uns, const float* box_sup,
                           const float* box_uns, const int* tgt_sup, int n_sup, const int* tgt_uns, int n_uns, float rpn_norm_sup,
                           float rpn_norm_uns, float w_rpn_cls, float w_rpn_loc, float w_box, const float* wt_host, float* rec, float* coef,
                           hipStream_t stream) {
  if (!rpn_sup || !rpn_uns || !focal_sup || !focal_uns || !box_sup || !box_uns || !tgt_sup || !tgt_uns || !wt_host || !rec || !coef ||
      n_sup < 0 || n_uns < 0 || !(rpn_norm_sup > 0.f) || !(rpn_norm_uns > 0.f))
    return UTV2_EARG;
  RcnnCombineArgs a;
  a.rpn[0] = rpn_sup; a.rpn[1] = rpn_uns; a.focal[0] = focal_sup; a.focal[1] = focal_uns; a.box[0] = box_sup; a.box[1] = box_uns;
  a.tgt[0] = tgt_sup; a.tgt[1] = tgt_uns; a.ntgt[0] = n_sup; a.ntgt[1] = n_uns; a.rpn_norm[0] = rpn_norm_sup; a.rpn_norm[1] = rpn_norm_uns;
  a.w_rpn_cls = w_rpn_cls; a.w_rpn_loc = w_rpn_loc; a.w_box = w_box;
  for (int k = 0; k < 8; ++k) a.wt[k] = wt_host[k];
  hipLaunchKernelGGL(rcnn_loss_combine_kernel, dim3(1), dim3(64), 0, stream, a, rec, coef);
  return utv2_launch_status();
}

int utv2_softmax_focal_bwd(const float* logits, const int* target, int R, int C, float gamma, const float* coef,
                           float* dlogits, hipStream_t stream) {
  if (!logits || !target || !coef || !dlogits) return UTV2_EARG;
  if (R == 0) return UTV2_OK;
  hipLaunchKernelGGL(softmax_focal_bwd_kernel, dim3(cdiv(R, 4) > 4096 ? 4096 : cdiv(R, 4)), dim3(256), 0, stream, logits, target,
                     R, C, gamma, coef, dlogits);
  return utv2_launch_status();
}

static int fill_rpn_levels(RpnLevels& L, int num_levels, int N, const int* hw_host, const int* k_host, int A) {
  if (num_levels < 1 || num_levels > RPN_MAX_LEVELS || !hw_host) return UTV2_EARG;
  L.n = num_levels;
  int row = 0, anc = 0, out = 0;
  for (int l = 0; l < num_levels; ++l) {
    if (hw_host[l] < 1) return UTV2_EARG;
    L.row0[l] = row; L.hw[l] = hw_host[l]; L.anchor0[l] = anc; L.out0[l] = out;
    row += N * hw_host[l];
    anc += hw_host[l] * A;
    out += k_host ? k_host[l] : 0;
  }
  L.out0[num_levels] = out;
  return UTV2_OK;
}

int utv2_roi_infer_keys(const float* probs, const float* deltas, const float* prop, const unsigned char* valid, const float* whwh, int N, int P,
                        int K, float wx, float wy, float scale_clamp, float thr, float* boxes, int64_t* keys, hipStream_t stream) {
  if (!probs || !deltas || !prop || !valid || !whwh || !boxes || !keys || N < 1 || P < 1 || K < 1 || (int64_t)P * K >= (1ll << 32)) return UTV2_EARG;
  hipLaunchKernelGGL(roi_infer_keys_kernel, dim3((unsigned)cdiv((long long)N * P, 256)), dim3(256), 0, stream, probs, deltas, prop, valid, whwh, N,
                     P, K, wx, wy, scale_clamp, thr, boxes, (long long*)keys);
  return utv2_launch_status();
}

int utv2_roi_infer_gather(const int64_t* top, const float* boxes, int N, int P, int K, int k, float thr, float* sc, int64_t* rows, int* cls,
                          float* cb, unsigned char* valid, hipStream_t stream) {
  if (!top || !boxes || !sc || !rows || !cls || !cb || !valid || N < 1 || P < 1 || K < 1 || k < 1) return UTV2_EARG;
  hipLaunchKernelGGL(roi_infer_gather_kernel, dim3((unsigned)cdiv((long long)N * k, 256)), dim3(256), 0, stream, (const long long*)top, boxes, N, P,
                     K, k, thr, sc, (long long*)rows, cls, cb, valid);
  return utv2_launch_status();
}

int utv2_roi_infer_pack(const int* kidx, const int* cnt, const float* cb, const float* sc, const int* cls, const int64_t* rows, const float* stdl,
                        int N, int P, int k, int D, float* oboxes, float* oscores, int* ocls, float* ostd, int64_t* orows, unsigned char* ovalid,
                        hipStream_t stream) {
  if (!kidx || !cnt || !cb || !sc || !cls || !rows || !stdl || !oboxes || !oscores || !ocls || !ostd || !orows || !ovalid || N < 1 || D < 1)
    return UTV2_EARG;
  hipLaunchKernelGGL(roi_infer_pack_kernel, dim3((unsigned)cdiv((long long)N * D, 256)), dim3(256), 0, stream, kidx, cnt, cb, sc, cls,
                     (const long long*)rows, stdl, N, P, k, D, oboxes, oscores, ocls, ostd, (long long*)orows, ovalid);
  return utv2_launch_status();
}

int utv2_rpn_rank_keys(const float* head, int num_levels, const int* hw_host, int N, int A, int ch, int64_t* keys, hipStream_t stream) {
  RpnLevels L;
  if (!head || !keys || N < 1 || A < 1 || ch < 5 * A) return UTV2_EARG;
  if (int e = fill_rpn_levels(L, num_levels, N, hw_host, nullptr, A)) return e;
  long long rows = 0;
  for (int l = 0; l < num_levels; ++l) rows += (long long)N * hw_host[l];
  const long long total = rows * A;
  hipLaunchKernelGGL(rpn_rank_keys_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, L, head, N, A, ch, total, (long long*)keys);
  return utv2_launch_status();
}

int utv2_rpn_decode(const int64_t* top, int maxk, const float* head, const float* anchors, const float* image_hw, int num_levels,
                    const int* hw_host, const int* k_host, int N, int A, int ch, const float* weights_host, float scale_clamp,
                    float min_size, float* boxes, float* scores, int* lvls, unsigned char* keep, hipStream_t stream) {
  RpnLevels L;
  if (!top || !head || !anchors || !image_hw || !k_host || !weights_host || !boxes || !scores || !lvls || !keep || N < 1 || A < 1 ||
      ch < 5 * A)
    return UTV2_EARG;
  if (int e = fill_rpn_levels(L, num_levels, N, hw_host, k_host, A)) return e;
  for (int l = 0; l < num_levels; ++l)
    if (k_host[l] < 0 || k_host[l] > maxk) return UTV2_EARG;
  const int K = L.out0[num_levels];
  if (K == 0) return UTV2_OK;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3(cdiv(K, 256), N), dim3(256), 0, stream, L, (const long long*)top, maxk, head, anchors, image_hw,
                     N, A, ch, weights_host[0], weights_host[1], weights_host[2], weights_host[3], scale_clamp, min_size, boxes, scores,
                     lvls, keep);
  return utv2_launch_status();
}

static int fill_rpn_loss_args(RpnLossArgs& a, const float* obj, const float* deltas, int head, int num_levels, const int* hw_host, int N,
                              int batch, int img0, int A, int ch, int R, const float* anchors, const int64_t* pos_idx, const unsigned char* pos_valid, int npos,
                              const int64_t* neg_idx, const unsigned char* neg_valid, int nneg, const int* matched,
                              const unsigned char* has_gt, const float* gt_boxes, const float* gt_scores, int G, const float* weights_host) {
  if (!obj || !deltas || !pos_idx || !pos_valid || !neg_idx || !neg_valid || N < 1 || R < 1 || npos < 0 || nneg < 0) return UTV2_EARG;
  a.L.n = 0;
  if (head) {
    if (A < 1 || ch < 5 * A) return UTV2_EARG;
    // the head output holds `batch` images per level; this call's N images are [img0, img0 + N) of them
    if (batch < N || img0 < 0 || img0 + N > batch) return UTV2_EARG;
    if (int e = fill_rpn_levels(a.L, num_levels, batch, hw_host, nullptr, A)) return e;
    for (int l = 0; l < num_levels; ++l) a.L.row0[l] += img0 * hw_host[l];
  } else if (batch != N || img0 != 0) {
    return UTV2_EARG;
  }
  a.head = head; a.N = N; a.A = A; a.ch = ch; a.R = R; a.npos = npos; a.nneg = nneg; a.G = G;
  a.obj = obj; a.deltas = deltas; a.anchors = anchors;
  a.pos_idx = (const long long*)pos_idx; a.pos_valid = pos_valid; a.neg_idx = (const long long*)neg_idx; a.neg_valid = neg_valid;
  a.matched = matched; a.has_gt = has_gt; a.gt_boxes = gt_boxes; a.gt_scores = gt_scores;
  a.wx = weights_host ? weights_host[0] : 1.f; a.wy = weights_host ? weights_host[1] : 1.f;
  a.ww = weights_host ? weights_host[2] : 1.f; a.wh = weights_host ? weights_host[3] : 1.f;
  return UTV2_OK;
}

int utv2_rpn_loss_fwd_range(const float* obj, const float* deltas, int head, int num_levels, const int* hw_host, int N, int batch, int img0,
                            int A, int ch, int R, const float* anchors, const int64_t* pos_idx, const unsigned char* pos_valid, int npos,
                            const int64_t* neg_idx, const unsigned char* neg_valid, int nneg, const int* matched, const unsigned char* has_gt,
                            const float* gt_boxes, const float* gt_scores, int G, const float* weights_host, float* sums, float* gobj,
                            float* gdl, hipStream_t stream) {
  RpnLossArgs a;
  if (!anchors || !matched || !has_gt || !gt_boxes || !weights_host || !sums || !gobj || !gdl || G < 1) return UTV2_EARG;
  if (int e = fill_rpn_loss_args(a, obj, deltas, head, num_levels, hw_host, N, batch, img0, A, ch, R, anchors, pos_idx, pos_valid, npos, neg_idx,
                                 neg_valid, nneg, matched, has_gt, gt_boxes, gt_scores, G, weights_host))
    return e;
  hipLaunchKernelGGL(rpn_loss_fwd_kernel, dim3(1), dim3(256), 0, stream, a, sums, gobj, gdl);
  return utv2_launch_status();
}

int utv2_rpn_loss_fwd(const float* obj, const float* deltas, int head, int num_levels, const int* hw_host, int N, int A, int ch, int R,
                      const float* anchors, const int64_t* pos_idx, const unsigned char* pos_valid, int npos, const int64_t* neg_idx,
                      const unsigned char* neg_valid, int nneg, const int* matched, const unsigned char* has_gt, const float* gt_boxes,
                      const float* gt_scores, int G, const float* weights_host, float* sums, float* gobj, float* gdl, hipStream_t stream) {
  return utv2_rpn_loss_fwd_range(obj, deltas, head, num_levels, hw_host, N, N, 0, A, ch, R, anchors, pos_idx, pos_valid, npos, neg_idx, neg_valid,
                                 nneg, matched, has_gt, gt_boxes, gt_scores, G, weights_host, sums, gobj, gdl, stream);
}

int utv2_rpn_loss_bwd_range(const float* gobj, const float* gdl, const float* gout_cls, const float* gout_loc, int head, int num_levels,
                            const int* hw_host, int N, int batch, int img0, int A, int ch, int R, const int64_t* pos_idx,
                            const unsigned char* pos_valid, int npos, const int64_t* neg_idx, const unsigned char* neg_valid, int nneg,
                            float* grad_obj, float* grad_deltas, hipStream_t stream) {
  RpnLossArgs a;
  if (!gobj || !gdl || !gout_cls || !gout_loc) return UTV2_EARG;
  if (int e = fill_rpn_loss_args(a, grad_obj, grad_deltas, head, num_levels, hw_host, N, batch, img0, A, ch, R, nullptr, pos_idx, pos_valid, npos, neg_idx,
                                 neg_valid, nneg, nullptr, nullptr, nullptr, nullptr, 1, nullptr))
    return e;
  const int total = N * (npos + nneg);
  if (total == 0) return UTV2_OK;
  hipLaunchKernelGGL(rpn_loss_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, stream, a, gobj, gdl, gout_cls, gout_loc, grad_obj, grad_deltas);
  return utv2_launch_status();
}

int utv2_rpn_loss_bwd(const float* gobj, const float* gdl, const float* gout_cls, const float* gout_loc, int head, int num_levels,
                      const int* hw_host, int N, int A, int ch, int R, const int64_t* pos_idx, const unsigned char* pos_valid, int npos,
                      const int64_t* neg_idx, const unsigned char* neg_valid, int nneg, float* grad_obj, float* grad_deltas,
                      hipStream_t stream) {
  return utv2_rpn_loss_bwd_range(gobj, gdl, gout_cls, gout_loc, head, num_levels, hw_host, N, N, 0, A, ch, R, pos_idx, pos_valid, npos, neg_idx,
                                 neg_valid, nneg, grad_obj, grad_deltas, stream);
}

int utv2_rpn_sample_keys(const float* max_iou, const unsigned char* lowq, const unsigned char* gt_valid, int G, const float* keys, int N,
                         int R, float lo, float hi, int64_t* out, hipStream_t stream) {
  if (!max_iou || !lowq || !gt_valid || !keys || !out || N < 1 || R < 1 || G < 1) return UTV2_EARG;
  hipLaunchKernelGGL(rpn_sample_keys_kernel, dim3(cdiv(R, 256), N), dim3(256), 0, stream, max_iou, lowq, gt_valid, G, keys, N, R, lo, hi,
                     (long long*)out);
  return utv2_launch_status();
}

int utv2_rpn_sample_unpack(const int64_t* top, int k, int N, int npos_max, int nneg_max, const unsigned char* gt_valid, int G,
                           int64_t* pos_idx, unsigned char* pos_valid, int64_t* neg_idx, unsigned char* neg_valid, unsigned char* has_gt,
                           hipStream_t stream) {
  if (!top || !gt_valid || !pos_idx || !pos_valid || !neg_idx || !neg_valid || !has_gt || N < 1 || k < 1 || npos_max < 0 || nneg_max < 0 ||
      G < 1)
    return UTV2_EARG;
  hipLaunchKernelGGL(rpn_sample_unpack_kernel, dim3(N), dim3(256), 0, stream, (const long long*)top, k, N, npos_max, nneg_max, gt_valid, G,
                     (long long*)pos_idx, pos_valid, (long long*)neg_idx, neg_valid, has_gt);
  return utv2_launch_status();
}

int utv2_roi_sample(const float* boxes, const unsigned char* valid, const float* max_iou, const int* argmax, const float* keys, int N, int P,
                    const float* gt_boxes, const int* gt_classes, const unsigned char* gt_valid, const float* gt_scores, const float* gt_std,
                    int G, float iou_thr, int num_classes, int batch, int nfg_max, float* out_boxes, int64_t* out_classes,
                    float* out_gt_boxes, unsigned char* out_valid, int64_t* out_idx, float* out_conf, float* out_std, hipStream_t stream) {
  if (!boxes || !valid || !max_iou || !argmax || !keys || !gt_boxes || !gt_classes || !gt_valid || !out_boxes || !out_classes ||
      !out_gt_boxes || !out_valid || !out_idx || N < 1 || P < 1 || P > ROI_SAMPLE_MAX || G < 1 || batch < 1 || nfg_max < 0 || nfg_max > batch)
    return UTV2_EARG;
  int npow2 = 2;
  while (npow2 < P) npow2 <<= 1;
  hipLaunchKernelGGL(roi_sample_kernel, dim3(N), dim3(512), 0, stream, boxes, valid, max_iou, argmax, keys, P, gt_boxes, gt_classes, gt_valid,
                     gt_scores, gt_std, G, iou_thr, num_classes, batch, nfg_max, npow2, out_boxes, (long long*)out_classes, out_gt_boxes,
                     out_valid, (long long*)out_idx, out_conf, out_std);
  return utv2_launch_status();
}

int utv2_roi_box_loss(const float* deltas, const float* stdl, int64_t ld, const int64_t* cls, const float* prop, const float* gtb,
                      const float* gstd, int R, int num_classes, int mode, float wx, float wy, float scale_clamp, float ts_better,
                      float t_cert, float* sum, float* gdeltas, float* gstd_out, hipStream_t stream) {
  if (!deltas || !stdl || !cls || !prop || !gtb || !sum || !gdeltas || !gstd_out || R < 0 || ld < 4 || mode < 0 || mode > 3) return UTV2_EARG;
  hipLaunchKernelGGL(roi_box_loss_kernel, dim3(1), dim3(256), 0, stream, deltas, stdl, (long long)ld, (const long long*)cls, prop, gtb, gstd, R,
                     num_classes, mode, wx, wy, scale_clamp, ts_better, t_cert, sum, gdeltas, gstd_out);
  return utv2_launch_status();
}

}  // extern "C"
