// Batched (class-aware) NMS and pairwise box IoU for gfx950.
//
// Replaces torchvision.ops.nms / detectron2 batched_nms reached from
// ubteacher/layers/ml_nms.py:27 (FCOS, thr 0.6), D2 find_top_rpn_proposals (thr 0.7) and D2
// fast_rcnn_inference (fast_rcnn.py:1112-1119, thr 0.5), and D2 pairwise_iou (rpn.py:117-120,
// roi_heads.py:226-228).
//
// Semantics pinned by oracle/ (tie rule declared there because upstream leaves it open):
//   * candidates are ordered by (score desc, slot index asc);
//   * class-aware suppression uses torchvision's coordinate trick exactly as written:
//     box' = box + class * (max_coord + 1) in fp32, then plain NMS on box';
//   * suppress j (later in the order) when inter / (area_i + area_j - inter) > thr (strict);
//   * kept slots are returned in descending-score order; optional post-top-k keeps every kept
//     slot whose score >= the k-th kept score (kthvalue rule, fcos_outputs.py:1309-1318).
// Stage 1 (one block per image): max-coordinate reduce + LDS bitonic sort of 64-bit keys.
// Stage 2: 64x64 IoU bit-mask tiles, boxes staged in LDS, upper triangle only.
// Stage 3 (one wave per image): chunked scan - 64 rows resolved with register bit ops and
//          v_readlane broadcasts, then the kept rows' mask words OR-ed in coalesced.
#include "common.h"

__device__ __forceinline__ unsigned order_bits_desc(float s) {
  unsigned u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-order transform
  return ~u;                                       // descending
}

__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                      const int* __restrict__ cls, const unsigned char* __restrict__ valid, int M,
                                                      int Mpad, int class_aware, float* __restrict__ sboxes,
                                                      int* __restrict__ sidx, int* __restrict__ scls, int* __restrict__ nvalid) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // [Mpad]
  __shared__ float redmax[16];
  __shared__ int redcnt[16];
  const int n = blockIdx.x;
  const float* b = boxes + (size_t)n * M * 4;
  float mx = -INFINITY;
  int cnt = 0;
  for (int i = threadIdx.x; i < Mpad; i += blockDim.x) {
    unsigned long long k = ~0ull;
    if (i < M && valid[(size_t)n * M + i]) {
      k = ((unsigned long long)order_bits_desc(scores[(size_t)n * M + i]) << 32) | (unsigned)i;
      mx = fmaxf(mx, fmaxf(fmaxf(b[i * 4], b[i * 4 + 1]), fmaxf(b[i * 4 + 2], b[i * 4 + 3])));
      ++cnt;
    }
    keys[i] = k;
  }
  mx = wave_reduce_max(mx);
  float cf = wave_reduce_sum((float)cnt);
  if ((threadIdx.x & 63) == 0) { redmax[threadIdx.x >> 6] = mx; redcnt[threadIdx.x >> 6] = (int)cf; }
  __syncthreads();
  float maxc = -INFINITY;
  int total = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { maxc = fmaxf(maxc, redmax[w]); total += redcnt[w]; }
  // bitonic sort ascending
  for (int k = 2; k <= Mpad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < Mpad; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], c = keys[ixj];
          const bool up = ((i & k) == 0);
          if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) nvalid[n] = total;
  const float off1 = maxc + 1.f;
  for (int i = threadIdx.x; i < Mpad; i += blockDim.x) {
    const unsigned long long k = keys[i];
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    int id = -1, c = 0;
    if (k != ~0ull) {
      id = (int)(k & 0xFFFFFFFFull);
      c = class_aware ? cls[(size_t)n * M + id] : 0;
      const float offs = class_aware ? (float)c * off1 : 0.f;
      o = make_float4(b[id * 4] + offs, b[id * 4 + 1] + offs, b[id * 4 + 2] + offs, b[id * 4 + 3] + offs);
    }
    ((float4*)sboxes)[(size_t)n * Mpad + i] = o;
    sidx[(size_t)n * Mpad + i] = id;
    scls[(size_t)n * Mpad + i] = c;
  }
}

__device__ __forceinline__ bool iou_gt(const float4& a, const float4& b, float thr) {
  const float aa = (a.z - a.x) * (a.w - a.y), ab = (b.z - b.x) * (b.w - b.y);
  const float w = fmaxf(0.f, fminf(a.z, b.z) - fmaxf(a.x, b.x));
  const float h = fmaxf(0.f, fminf(a.w, b.w) - fmaxf(a.y, b.y));
  const float inter = w * h;
  return inter / (aa + ab - inter) > thr;
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ sboxes, const int* __restrict__ nvalid, int Mpad,
                                                    float thr, unsigned long long* __restrict__ mask) {
  const int n = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  const int nv = nvalid[n];
  if (rb * 64 >= nv || cb * 64 >= nv) return;
  __shared__ float4 cbox[64];
  const int t = threadIdx.x;
  cbox[t] = ((const float4*)sboxes)[(size_t)n * Mpad + cb * 64 + t];
  __syncthreads();
  const int i = rb * 64 + t;
  unsigned long long bits = 0;
  if (i < nv) {
    const float4 a = ((const float4*)sboxes)[(size_t)n * Mpad + i];
    const int jn = min(64, nv - cb * 64);
    for (int j = 0; j < jn; ++j) {
      const int gj = cb * 64 + j;
      if (gj > i && iou_gt(a, cbox[j], thr)) bits |= 1ull << j;
    }
  }
  const int W = Mpad >> 6;
  mask[((size_t)n * Mpad + i) * W + cb] = bits;
}

__device__ __forceinline__ unsigned long long bcast64(unsigned long long v, int lane) {
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)(v & 0xFFFFFFFFull), lane);
  const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), lane);
  return ((unsigned long long)hi << 32) | lo;
}

#define NMS_MAXW 4  // words per lane: Mpad <= 64*64*4 = 16384
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ sidx,
                                                    const int* __restrict__ nvalid, const float* __restrict__ scores, int M,
                                                    int Mpad, int post_topk, int max_out, int* __restrict__ keep,
                                                    int* __restrict__ keep_count) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const int W = Mpad >> 6;
  const int nv = nvalid[n];
  const int nchunks = (nv + 63) >> 6;
  unsigned long long removed[NMS_MAXW];
#pragma unroll
  for (int s = 0; s < NMS_MAXW; ++s) removed[s] = 0ull;
  int count = 0;
  const unsigned long long* mrow = mask + (size_t)n * Mpad * W;
  for (int c = 0; c < nchunks; ++c) {
    // removed word c lives in lane c%64, slot c/64
    unsigned long long rw = 0ull;
#pragma unroll
    for (int s = 0; s < NMS_MAXW; ++s)
      if ((c >> 6) == s) rw = bcast64(removed[s], c & 63);
    const int row = c * 64 + lane;
    const unsigned long long diag = (row < nv) ? mrow[(size_t)row * W + c] : 0ull;
    const int rows_here = min(64, nv - c * 64);
    unsigned long long keepbits = 0ull;
    for (int b = 0; b < rows_here; ++b) {
      const unsigned long long db = bcast64(diag, b);
      if (!((rw >> b) & 1ull)) { keepbits |= 1ull << b; rw |= db; }
    }
    // emit kept slots of this chunk in order
    if (row < nv && ((keepbits >> lane) & 1ull)) {
      const int pos = count + __popcll(keepbits & ((1ull << lane) - 1ull));
      if (pos < max_out) keep[(size_t)n * max_out + pos] = sidx[(size_t)n * Mpad + row];
    }
    count += __popcll(keepbits);
    // OR the kept rows' mask words into removed (words > c only matter)
    unsigned long long kb = keepbits;
    while (kb) {
      const int b = __ffsll((long long)kb) - 1;
      kb &= kb - 1ull;
      const unsigned long long* r = mrow + (size_t)(c * 64 + b) * W;
#pragma unroll
      for (int s = 0; s < NMS_MAXW; ++s) {
        const int w = s * 64 + lane;
        if (w < nchunks && w > c) removed[s] |= r[w];
      }
    }
    // kthvalue rule below keeps only scores >= the post_topk-th kept score; candidates are in descending score order,
    // so once post_topk are kept and the next chunk starts below that score nothing later can survive: stop scanning
    if (post_topk > 0 && post_topk <= max_out && count >= post_topk && (c + 1) * 64 < nv) {
      __syncthreads();  // single wave: keep[] stores of this wave are visible to its loads
      const float kth = scores[(size_t)n * M + keep[(size_t)n * max_out + post_topk - 1]];
      const float nxt = scores[(size_t)n * M + sidx[(size_t)n * Mpad + (c + 1) * 64]];
      if (nxt < kth) break;
    }
  }
  if (count > max_out) count = max_out;
  __syncthreads();  // single wave: orders the keep[] stores before the reads below
  // kthvalue rule: keep all with score >= score of the post_topk-th kept
  if (post_topk > 0 && count > post_topk) {
    const float thr = scores[(size_t)n * M + keep[(size_t)n * max_out + post_topk - 1]];
    int c2 = 0;
    for (int base = 0; base < count; base += 64) {
      const int i = base + lane;
      const bool ok = i < count && scores[(size_t)n * M + keep[(size_t)n * max_out + i]] >= thr;
      c2 += __popcll(__ballot(ok));
    }
    count = c2;
  }
  if (lane == 0) keep_count[n] = count;
  for (int i = count + lane; i < max_out; i += 64) keep[(size_t)n * max_out + i] = -1;
}

// Class-parallel scan: boxes of different classes never suppress each other (the coordinate trick makes their IoU 0), so the
// greedy chain splits into independent per-class chains.  Wave w of the block owns the candidates with class % NMS_PW == w
// and walks only those (in global score order): the serial part - one v_readlane round per candidate - shrinks by the number
// of busy waves (5 FPN levels for the RPN, 80 categories for FCOS / the ROI head).  The per-wave early exit is the same rule
// as above applied to the wave's own kept list (a sub-list of the global one, so its k-th score bounds the global k-th).
#define NMS_PW 8
#define NMS_SYNC 8   // chunks between the global kept-count checks (power of two)
__global__ __launch_bounds__(64 * NMS_PW) void nms_scan_par_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ sidx,
                                                                   const int* __restrict__ scls, const int* __restrict__ nvalid,
                                                                   const float* __restrict__ scores, int M, int Mpad, int post_topk,
                                                                   int max_out, int* __restrict__ keep, int* __restrict__ keep_count) {
  __shared__ unsigned long long keepw[64 * NMS_MAXW];  // kept bits per 64-candidate chunk, all classes
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int W = Mpad >> 6;
  const int nv = nvalid[n];
  const int nchunks = (nv + 63) >> 6;
  for (int i = threadIdx.x; i < 64 * NMS_MAXW; i += blockDim.x) keepw[i] = 0ull;
  __syncthreads();
  const unsigned long long* mrow = mask + (size_t)n * Mpad * W;
  const int* cl = scls + (size_t)n * Mpad;
  const int* si = sidx + (size_t)n * Mpad;
  const float* sc = scores + (size_t)n * M;
  unsigned long long removed[NMS_MAXW];
#pragma unroll
  for (int s = 0; s < NMS_MAXW; ++s) removed[s] = 0ull;
  int count = 0;
  bool have_kth = false, done = false;
  float kth = 0.f;
  __shared__ int wkept[NMS_PW];
  for (int c = 0; c < nchunks; ++c) {
    // Without a post_topk tie rule only the first max_out kept candidates (in global score order) are emitted: every NMS_SYNC chunks
    // the waves add up what they have kept in the chunks so far (all of them final: every wave is past them) and the whole block
    // stops once that reaches max_out - the RPN keeps 1000 of 10 000 candidates and would otherwise walk all of them.
    if (post_topk <= 0 && c > 0 && (c & (NMS_SYNC - 1)) == 0) {
      if (lane == 0) wkept[wv] = count;
      __syncthreads();
      int total = 0;
#pragma unroll
      for (int w = 0; w < NMS_PW; ++w) total += wkept[w];
      __syncthreads();
      if (total >= max_out) break;   // block-uniform
    }
    if (done) continue;
    const int row = c * 64 + lane;
    const bool mine = row < nv && (int)((unsigned)cl[row] % NMS_PW) == wv;
    const unsigned long long cm = __ballot(mine);
    if (cm == 0ull) continue;  // wave-uniform
    if (have_kth) {            // post_topk already kept by this wave: stop once its candidates fall below that score
      const int first = c * 64 + (__ffsll((long long)cm) - 1);
      if (sc[si[first]] < kth) { done = true; continue; }
    }
    unsigned long long rw = 0ull;
#pragma unroll
    for (int s = 0; s < NMS_MAXW; ++s)
      if ((c >> 6) == s) rw = bcast64(removed[s], c & 63);
    const unsigned long long diag = mine ? mrow[(size_t)row * W + c] : 0ull;
    unsigned long long keepbits = 0ull, todo = cm;
    while (todo) {
      const int b = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const unsigned long long db = bcast64(diag, b);
      if (!((rw >> b) & 1ull)) { keepbits |= 1ull << b; rw |= db; }
    }
    if (keepbits) {
      if (lane == 0) atomicOr(&keepw[c], keepbits);
      if (post_topk > 0 && !have_kth && count + __popcll(keepbits) >= post_topk) {
        // the post_topk-th kept candidate of this wave sits in this chunk: its score bounds the global k-th score
        int need = post_topk - count;
        unsigned long long kb = keepbits;
        int b = 0;
        while (need > 0) { b = __ffsll((long long)kb) - 1; kb &= kb - 1ull; --need; }
        kth = sc[si[c * 64 + b]];
        have_kth = true;
      }
      count += __popcll(keepbits);
      unsigned long long kb = keepbits;
      while (kb) {
        const int b = __ffsll((long long)kb) - 1;
        kb &= kb - 1ull;
        const unsigned long long* r = mrow + (size_t)(c * 64 + b) * W;
#pragma unroll
        for (int s = 0; s < NMS_MAXW; ++s) {
          const int w = s * 64 + lane;
          if (w < nchunks && w > c) removed[s] |= r[w];
        }
      }
    }
  }
  __syncthreads();
  // wave 0: emit the kept slots of all classes in global (descending score) order
  int total = 0;
  if (wv == 0) {
    for (int c = 0; c < nchunks; ++c) {
      const unsigned long long kb = keepw[c];
      if (kb == 0ull) continue;
      const int row = c * 64 + lane;
      if ((kb >> lane) & 1ull) {
        const int pos = total + __popcll(kb & ((1ull << lane) - 1ull));
        if (pos < max_out) keep[(size_t)n * max_out + pos] = si[row];
      }
      total += __popcll(kb);
    }
  }
  __syncthreads();  // orders the keep[] stores before the reads below
  if (wv != 0) return;
  int cnt = total > max_out ? max_out : total;
  // kthvalue rule: keep all with score >= score of the post_topk-th kept
  if (post_topk > 0 && cnt > post_topk) {
    const float thr = sc[keep[(size_t)n * max_out + post_topk - 1]];
    int c2 = 0;
    for (int base = 0; base < cnt; base += 64) {
      const int i = base + lane;
      const bool ok = i < cnt && sc[keep[(size_t)n * max_out + i]] >= thr;
      c2 += __popcll(__ballot(ok));
    }
    cnt = c2;
  }
  if (lane == 0) keep_count[n] = cnt;
  for (int i = cnt + lane; i < max_out; i += 64) keep[(size_t)n * max_out + i] = -1;
}

// pairwise IoU  (D2 pairwise_iou [D2-recall]): iou = inter > 0 ? inter / (a1 + a2 - inter) : 0
__global__ __launch_bounds__(256) void box_iou_kernel(const float* __restrict__ a, const float* __restrict__ b, int A, int B,
                                                    float* __restrict__ out) {
  // out[A][B]; block covers 256 columns of b for one row tile of 8 a-boxes
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i0 = blockIdx.y * 8;
  float4 bb = make_float4(0, 0, 0, 0);
  if (j < B) bb = ((const float4*)b)[j];
  const float ab = (bb.z - bb.x) * (bb.w - bb.y);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = i0 + k;
    if (i >= A || j >= B) continue;
    const float4 aa = ((const float4*)a)[i];
    const float area = (aa.z - aa.x) * (aa.w - aa.y);
    const float w = fmaxf(fminf(aa.z, bb.z) - fmaxf(aa.x, bb.x), 0.f);
    const float h = fmaxf(fminf(aa.w, bb.w) - fmaxf(aa.y, bb.y), 0.f);
    const float inter = w * h;
    out[(size_t)i * B + j] = inter > 0.f ? inter / (area + ab - inter) : 0.f;
  }
}

static inline int next_pow2(int v) {
  int p = 64;
  while (p < v) p <<= 1;
  return p;
}

extern "C" {

int utv2_nms_mpad(int M) { return next_pow2(M); }

// workspace bytes: sorted boxes + sorted idx + sorted class + nvalid + mask
int64_t utv2_nms_workspace_bytes(int N, int M) {
  const int64_t Mpad = next_pow2(M);
  return N * (Mpad * 16 + Mpad * 4 + Mpad * 4 + 64 + Mpad * (Mpad / 64) * 8);
}

// boxes [N][M][4] xyxy, scores [N][M], cls [N][M] (int32), valid [N][M] (u8)
// keep [N][max_out] (slot indices, -1 padded, descending score), keep_count [N]
int utv2_nms_batched(const float* boxes, const float* scores, const int* cls, const unsigned char* valid, int N, int M,
                     float iou_thr, int class_aware, int post_topk, int max_out, int* keep, int* keep_count, void* ws,
                     hipStream_t stream) {
  if (!boxes || !scores || !valid || !keep || !keep_count || !ws || M < 1 || (class_aware && !cls)) return UTV2_EARG;
  const int Mpad = next_pow2(M);
  if (Mpad > 64 * 64 * NMS_MAXW) return UTV2_EARG;
  char* p = (char*)ws;
  float* sboxes = (float*)p; p += (size_t)N * Mpad * 16;
  int* sidx = (int*)p; p += (size_t)N * Mpad * 4;
  int* scls = (int*)p; p += (size_t)N * Mpad * 4;
  int* nvalid = (int*)p; p += 64 * (size_t)N;
  unsigned long long* mask = (unsigned long long*)p;
  const size_t lds = (size_t)Mpad * 8;
  (void)hipFuncSetAttribute((const void*)nms_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(nms_sort_kernel, dim3(N), dim3(1024), lds, stream, boxes, scores, cls, valid, M, Mpad, class_aware, sboxes,
                     sidx, scls, nvalid);
  const int nb = Mpad / 64;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb, N), dim3(64), 0, stream, (const float*)sboxes, (const int*)nvalid, Mpad,
                     iou_thr, mask);
  if (class_aware)  // independent per-class chains: NMS_PW waves per image
    hipLaunchKernelGGL(nms_scan_par_kernel, dim3(N), dim3(64 * NMS_PW), 0, stream, (const unsigned long long*)mask, (const int*)sidx,
                       (const int*)scls, (const int*)nvalid, scores, M, Mpad, post_topk, max_out, keep, keep_count);
  else
    hipLaunchKernelGGL(nms_scan_kernel, dim3(N), dim3(64), 0, stream, (const unsigned long long*)mask, (const int*)sidx,
                       (const int*)nvalid, scores, M, Mpad, post_topk, max_out, keep, keep_count);
  return utv2_launch_status();
}

int utv2_box_iou(const float* a, const float* b, int A, int B, float* out, hipStream_t stream) {
  if (!a || !b || !out) return UTV2_EARG;
  if (A == 0 || B == 0) return UTV2_OK;
  hipLaunchKernelGGL(box_iou_kernel, dim3(cdiv(B, 256), cdiv(A, 8)), dim3(256), 0, stream, a, b, A, B, out);
  return utv2_launch_status();
}

}  // extern "C"
