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
// Class-aware calls run stages 2-3 per class bucket (see "Bucketed pipeline" below).
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
      for (int t = threadIdx.x; t < (Mpad >> 1); t += blockDim.x) {  // pair t: (i, i + j) with bit j of i clear
        const int i = 2 * t - (t & (j - 1));
        const int ixj = i + j;
        const unsigned long long a = keys[i], c = keys[ixj];
        const bool up = ((i & k) == 0);
        if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
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

// ---------------------------------------------------------------------------------------------
// Bucketed pipeline.  Boxes of different classes never suppress each other (the coordinate trick makes their IoU exactly 0), so the
// greedy chain splits into independent per-class chains.  After the global sort the candidates are stably partitioned into NMS_NB
// buckets by class % NMS_NB (any class ids: two classes that share a bucket still have IoU 0 through the trick); every bucket gets its
// own small IoU bit matrix (rows of W_b = chunks of the bucket words, instead of one M x M matrix: 13x fewer tiles and 8x shorter rows
// for the RPN's 5 levels x 2000) and its own scanning wave (grid NMS_NB x N instead of one block per image walking every chunk of
// the image); the kept candidates are marked in a global-order bit array from which one wave per image emits the result, exactly as
// the single-chain scan does.  Non-class-aware calls are the one-bucket case.
#define NMS_NB 32
#define NMS_BTAB (2 * NMS_NB + 2)  // per image: boff[NMS_NB + 1] (64-aligned first position of each bucket in bucket order), bcnt[NMS_NB], tiles

// one block per image: stable partition of the global order by bucket.  gpos[p] = global position of bucket-order position p (-1 =
// padding), cbox[p] its (offset) box; toff[b] = first IoU tile of bucket b (upper triangle incl. diagonal, row-major), moff[b] = first
// mask word.  The 16 waves own contiguous runs of chunks: histogram, prefix, then the same walk with running counters.
__global__ __launch_bounds__(1024) void nms_bucket_kernel(const float* __restrict__ sboxes, const int* __restrict__ scls,
                                                        const int* __restrict__ nvalid, int Mpad, int Ptot, float* __restrict__ cbox,
                                                        int* __restrict__ gpos, int* __restrict__ btab, long long* __restrict__ moff,
                                                        int* __restrict__ toff, unsigned long long* __restrict__ keepw) {
  __shared__ int wcount[16][NMS_NB];
  __shared__ int boff_s[NMS_NB + 1];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nv = nvalid[n];
  const int nchunks = (nv + 63) >> 6;
  for (int i = tid; i < (Mpad >> 6); i += blockDim.x) keepw[(size_t)n * (Mpad >> 6) + i] = 0ull;
  for (int i = tid; i < 16 * NMS_NB; i += blockDim.x) (&wcount[0][0])[i] = 0;
  __syncthreads();
  const int per = (nchunks + 15) >> 4;
  const int c0 = wv * per, c1 = min(nchunks, c0 + per);
  const int* cl = scls + (size_t)n * Mpad;
  volatile int* mine = wcount[wv];
  for (int c = c0; c < c1; ++c) {
    const int row = c * 64 + lane;
    const int bk = row < nv ? (int)((unsigned)cl[row] % NMS_NB) : -1;
    unsigned long long todo = __ballot(bk >= 0);
    while (todo) {
      const int cur = __builtin_amdgcn_readlane(bk, __ffsll((long long)todo) - 1);
      const unsigned long long m = __ballot(bk == cur);
      if (lane == 0) mine[cur] += __popcll(m);
      todo &= ~m;
    }
  }
  __syncthreads();
  if (tid < NMS_NB) {  // per bucket: the waves' counts become their start offsets inside the bucket
    int run = 0;
    for (int w = 0; w < 16; ++w) { const int v = wcount[w][tid]; wcount[w][tid] = run; run += v; }
    btab[n * NMS_BTAB + NMS_NB + 1 + tid] = run;
    boff_s[tid] = run;  // count, for the prefix below
  }
  __syncthreads();
  if (tid == 0) {
    int pos = 0, tiles = 0;
    long long words = 0;
    for (int b = 0; b < NMS_NB; ++b) {
      const int cnt = boff_s[b], wb = (cnt + 63) >> 6;
      boff_s[b] = pos;
      btab[n * NMS_BTAB + b] = pos;
      toff[n * (NMS_NB + 1) + b] = tiles;
      moff[n * NMS_NB + b] = words;
      pos += wb * 64;
      tiles += wb * (wb + 1) / 2;
      words += (long long)wb * 64 * wb;
    }
    boff_s[NMS_NB] = pos;
    btab[n * NMS_BTAB + NMS_NB] = pos;
    toff[n * (NMS_NB + 1) + NMS_NB] = tiles;
    btab[n * NMS_BTAB + 2 * NMS_NB + 1] = tiles;
  }
  __syncthreads();
  for (int i = tid; i < boff_s[NMS_NB]; i += blockDim.x) gpos[(size_t)n * Ptot + i] = -1;   // padding slots stay -1
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    const int row = c * 64 + lane;
    const int bk = row < nv ? (int)((unsigned)cl[row] % NMS_NB) : -1;
    unsigned long long todo = __ballot(bk >= 0);
    int p = -1;
    while (todo) {
      const int cur = __builtin_amdgcn_readlane(bk, __ffsll((long long)todo) - 1);
      const unsigned long long m = __ballot(bk == cur);
      const int base = mine[cur];
      if (bk == cur) p = boff_s[cur] + base + __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) mine[cur] = base + __popcll(m);
      todo &= ~m;
    }
    if (p >= 0) {
      gpos[(size_t)n * Ptot + p] = row;
      ((float4*)cbox)[(size_t)n * Ptot + p] = ((const float4*)sboxes)[(size_t)n * Mpad + row];
    }
  }
}

// tile t of image n: bucket by the tile prefix, (row chunk, column chunk) of the bucket's upper triangle from the triangular index
__global__ __launch_bounds__(64) void nms_mask_bucket_kernel(const float* __restrict__ cbox, const int* __restrict__ btab,
                                                           const long long* __restrict__ moff, const int* __restrict__ toff, int Ptot,
                                                           float thr, unsigned long long* __restrict__ mask, long long maskwords) {
  const int n = blockIdx.y, t = blockIdx.x;
  const int* tf = toff + n * (NMS_NB + 1);
  if (t >= tf[NMS_NB]) return;
  int b = 0;
#pragma unroll
  for (int i = 1; i < NMS_NB; ++i)
    if (t >= tf[i]) b = i;
  const int* bt = btab + n * NMS_BTAB;
  const int cnt = bt[NMS_NB + 1 + b], wb = (cnt + 63) >> 6, p0 = bt[b];
  const int tt = t - tf[b];
  // row r of the triangle starts at tile r * wb - r (r - 1) / 2
  int r = (int)(((2.f * wb + 1.f) - sqrtf((2.f * wb + 1.f) * (2.f * wb + 1.f) - 8.f * (float)tt)) * 0.5f);
  r = max(0, min(r, wb - 1));
  while (r + 1 < wb && (r + 1) * wb - (r + 1) * r / 2 <= tt) ++r;
  while (r > 0 && r * wb - r * (r - 1) / 2 > tt) --r;
  const int cb = r + (tt - (r * wb - r * (r - 1) / 2));
  __shared__ float4 colbox[64];
  const int lane = threadIdx.x;
  colbox[lane] = ((const float4*)cbox)[(size_t)n * Ptot + p0 + cb * 64 + lane];
  __syncthreads();
  const int i = r * 64 + lane;  // bucket-local row
  unsigned long long bits = 0;
  if (i < cnt) {
    const float4 a = ((const float4*)cbox)[(size_t)n * Ptot + p0 + i];
    const int jn = min(64, cnt - cb * 64);
    for (int j = 0; j < jn; ++j)
      if (cb * 64 + j > i && iou_gt(a, colbox[j], thr)) bits |= 1ull << j;
  }
  mask[(size_t)n * maskwords + moff[n * NMS_NB + b] + (size_t)i * wb + cb] = bits;
}

// one wave per (bucket, image): the chunked greedy scan over the bucket's own matrix; kept candidates are marked at their GLOBAL position
__global__ __launch_bounds__(64) void nms_scan_bucket_kernel(const unsigned long long* __restrict__ mask, long long maskwords,
                                                           const int* __restrict__ btab, const long long* __restrict__ moff,
                                                           const int* __restrict__ gpos, int Ptot, const int* __restrict__ sidx,
                                                           const float* __restrict__ scores, int M, int Mpad, int post_topk, int max_out,
                                                           unsigned long long* __restrict__ keepw) {
  const int b = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
  const int* bt = btab + n * NMS_BTAB;
  const int cnt = bt[NMS_NB + 1 + b];
  if (cnt == 0) return;
  const int wb = (cnt + 63) >> 6, p0 = bt[b];
  const unsigned long long* mrow = mask + (size_t)n * maskwords + moff[n * NMS_NB + b];
  const int* gp = gpos + (size_t)n * Ptot + p0;
  const int* si = sidx + (size_t)n * Mpad;
  const float* sc = scores + (size_t)n * M;
  unsigned long long* kw = keepw + (size_t)n * (Mpad >> 6);
  unsigned long long removed[NMS_MAXW];
#pragma unroll
  for (int s = 0; s < NMS_MAXW; ++s) removed[s] = 0ull;
  int count = 0;
  bool have_kth = false;
  float kth = 0.f;
  for (int c = 0; c < wb; ++c) {
    // without a post_topk tie rule only the first max_out kept candidates of the IMAGE are emitted: a bucket that has kept that many
    // on its own can stop; with one, the bucket's own post_topk-th kept score bounds the image's k-th score from below
    if (post_topk <= 0 && count >= max_out) break;
    const int row = c * 64 + lane;
    const int g = row < cnt ? gp[row] : -1;
    if (have_kth && sc[si[__builtin_amdgcn_readfirstlane(g)]] < kth) break;
    unsigned long long rw = 0ull;
#pragma unroll
    for (int s = 0; s < NMS_MAXW; ++s)
      if ((c >> 6) == s) rw = bcast64(removed[s], c & 63);
    const unsigned long long diag = row < cnt ? mrow[(size_t)row * wb + c] : 0ull;
    unsigned long long keepbits = 0ull;
    const int jn = min(64, cnt - c * 64);
    for (int j = 0; j < jn; ++j) {
      const unsigned long long db = bcast64(diag, j);
      if (!((rw >> j) & 1ull)) { keepbits |= 1ull << j; rw |= db; }
    }
    if (keepbits == 0ull) continue;
    if ((keepbits >> lane) & 1ull) atomicOr(&kw[g >> 6], 1ull << (g & 63));
    if (post_topk > 0 && !have_kth && count + __popcll(keepbits) >= post_topk) {
      int need = post_topk - count;
      unsigned long long kb = keepbits;
      int j = 0;
      while (need > 0) { j = __ffsll((long long)kb) - 1; kb &= kb - 1ull; --need; }
      kth = sc[si[__builtin_amdgcn_readlane(g, j)]];
      have_kth = true;
    }
    count += __popcll(keepbits);
    // OR the kept rows into the removed set, 8 rows' loads in flight at a time (one row per round trip left the wave waiting ~0.5 us
    // per kept candidate: 1000 kept = the whole kernel)
    unsigned long long kb = keepbits;
    while (kb) {
      unsigned long long v[8][NMS_MAXW];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool on = kb != 0ull;
        const int j = on ? __ffsll((long long)kb) - 1 : 0;
        kb &= kb - 1ull;   // 0 stays 0
        const unsigned long long* r = mrow + (size_t)(c * 64 + j) * wb;
#pragma unroll
        for (int s = 0; s < NMS_MAXW; ++s) {
          const int w = s * 64 + lane;
          v[u][s] = (on && w < wb && w > c) ? r[w] : 0ull;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int s = 0; s < NMS_MAXW; ++s) removed[s] |= v[u][s];
    }
  }
}

// one wave per image: the kept slots of all buckets in global (descending score) order, the kthvalue rule, -1 padding
__global__ __launch_bounds__(64) void nms_emit_kernel(const unsigned long long* __restrict__ keepw, const int* __restrict__ sidx,
                                                    const int* __restrict__ nvalid, const float* __restrict__ scores, int M, int Mpad,
                                                    int post_topk, int max_out, int* __restrict__ keep, int* __restrict__ keep_count) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const int nchunks = (nvalid[n] + 63) >> 6;
  const int* si = sidx + (size_t)n * Mpad;
  const float* sc = scores + (size_t)n * M;
  int total = 0;
  for (int c = 0; c < nchunks && total < max_out; ++c) {
    const unsigned long long kb = keepw[(size_t)n * (Mpad >> 6) + c];
    if (kb == 0ull) continue;
    if ((kb >> lane) & 1ull) {
      const int pos = total + __popcll(kb & ((1ull << lane) - 1ull));
      if (pos < max_out) keep[(size_t)n * max_out + pos] = si[c * 64 + lane];
    }
    total += __popcll(kb);
  }
  __syncthreads();  // single wave: orders the keep[] stores before the reads below
  int cnt = total > max_out ? max_out : total;
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

static inline int64_t align16(int64_t v) { return (v + 15) & ~(int64_t)15; }

// workspace layout: sorted boxes, sorted idx, sorted class, nvalid, then either the single M x M mask (non-class-aware) or the bucket
// tables + per-bucket masks (class-aware)
struct NmsLayout {
  int Mpad, Wt, Ptot;
  int64_t maskwords;  // per image, bucketed
  int64_t sboxes, sidx, scls, nvalid, tail;          // byte offsets
  int64_t cbox, gpos, btab, toff, moff, keepw, bmask, total;
};
static NmsLayout nms_layout(int N, int M) {
  NmsLayout L;
  L.Mpad = next_pow2(M);
  L.Wt = cdiv(M, 64) + NMS_NB;
  L.Ptot = L.Wt * 64;
  L.maskwords = (int64_t)64 * L.Wt * L.Wt;
  int64_t o = 0;
  L.sboxes = o; o += (int64_t)N * L.Mpad * 16;
  L.sidx = o; o += (int64_t)N * L.Mpad * 4;
  L.scls = o; o += (int64_t)N * L.Mpad * 4;
  L.nvalid = o; o += 64 * (int64_t)N;
  L.tail = o;
  const int64_t single = o + (int64_t)N * L.Mpad * (L.Mpad / 64) * 8;
  L.cbox = o; o = align16(o + (int64_t)N * L.Ptot * 16);
  L.gpos = o; o = align16(o + (int64_t)N * L.Ptot * 4);
  L.btab = o; o = align16(o + (int64_t)N * NMS_BTAB * 4);
  L.toff = o; o = align16(o + (int64_t)N * (NMS_NB + 1) * 4);
  L.moff = o; o = align16(o + (int64_t)N * NMS_NB * 8);
  L.keepw = o; o = align16(o + (int64_t)N * (L.Mpad / 64) * 8);
  L.bmask = o; o += (int64_t)N * L.maskwords * 8;
  L.total = o > single ? o : single;
  return L;
}

int64_t utv2_nms_workspace_bytes(int N, int M) { return nms_layout(N, M).total; }

// boxes [N][M][4] xyxy, scores [N][M], cls [N][M] (int32), valid [N][M] (u8)
// keep [N][max_out] (slot indices, -1 padded, descending score), keep_count [N]
int utv2_nms_batched(const float* boxes, const float* scores, const int* cls, const unsigned char* valid, int N, int M,
                     float iou_thr, int class_aware, int post_topk, int max_out, int* keep, int* keep_count, void* ws,
                     hipStream_t stream) {
  if (!boxes || !scores || !valid || !keep || !keep_count || !ws || M < 1 || (class_aware && !cls)) return UTV2_EARG;
  const NmsLayout L = nms_layout(N, M);
  const int Mpad = L.Mpad;
  if (Mpad > 64 * 64 * NMS_MAXW) return UTV2_EARG;
  char* p = (char*)ws;
  float* sboxes = (float*)(p + L.sboxes);
  int* sidx = (int*)(p + L.sidx);
  int* scls = (int*)(p + L.scls);
  int* nvalid = (int*)(p + L.nvalid);
  const size_t lds = (size_t)Mpad * 8;
  (void)hipFuncSetAttribute((const void*)nms_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(nms_sort_kernel, dim3(N), dim3(1024), lds, stream, boxes, scores, cls, valid, M, Mpad, class_aware, sboxes,
                     sidx, scls, nvalid);
  if (!class_aware) {
    unsigned long long* mask = (unsigned long long*)(p + L.tail);
    const int nb = Mpad / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nb, nb, N), dim3(64), 0, stream, (const float*)sboxes, (const int*)nvalid, Mpad,
                       iou_thr, mask);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(N), dim3(64), 0, stream, (const unsigned long long*)mask, (const int*)sidx,
                       (const int*)nvalid, scores, M, Mpad, post_topk, max_out, keep, keep_count);
    return utv2_launch_status();
  }
  float* cbox = (float*)(p + L.cbox);
  int* gpos = (int*)(p + L.gpos);
  int* btab = (int*)(p + L.btab);
  int* toff = (int*)(p + L.toff);
  long long* moff = (long long*)(p + L.moff);
  unsigned long long* keepw = (unsigned long long*)(p + L.keepw);
  unsigned long long* bmask = (unsigned long long*)(p + L.bmask);
  hipLaunchKernelGGL(nms_bucket_kernel, dim3(N), dim3(1024), 0, stream, (const float*)sboxes, (const int*)scls, (const int*)nvalid, Mpad,
                     L.Ptot, cbox, gpos, btab, moff, toff, keepw);
  // tiles of an image <= Wr (Wr + 1) / 2 with Wr = chunks actually occupied <= ceil(M / 64) + NMS_NB: the surplus workgroups return at once
  const int Wr = cdiv(M, 64) + (M < NMS_NB ? M : NMS_NB);
  hipLaunchKernelGGL(nms_mask_bucket_kernel, dim3(Wr * (Wr + 1) / 2, N), dim3(64), 0, stream, (const float*)cbox, (const int*)btab,
                     (const long long*)moff, (const int*)toff, L.Ptot, iou_thr, bmask, (long long)L.maskwords);
  hipLaunchKernelGGL(nms_scan_bucket_kernel, dim3(NMS_NB, N), dim3(64), 0, stream, (const unsigned long long*)bmask, (long long)L.maskwords,
                     (const int*)btab, (const long long*)moff, (const int*)gpos, L.Ptot, (const int*)sidx, scores, M, Mpad, post_topk,
                     max_out, keepw);
  hipLaunchKernelGGL(nms_emit_kernel, dim3(N), dim3(64), 0, stream, (const unsigned long long*)keepw, (const int*)sidx, (const int*)nvalid,
                     scores, M, Mpad, post_topk, max_out, keep, keep_count);
  return utv2_launch_status();
}

int utv2_box_iou(const float* a, const float* b, int A, int B, float* out, hipStream_t stream) {
  if (!a || !b || !out) return UTV2_EARG;
  if (A == 0 || B == 0) return UTV2_OK;
  hipLaunchKernelGGL(box_iou_kernel, dim3(cdiv(B, 256), cdiv(A, 8)), dim3(256), 0, stream, a, b, A, B, out);
  return utv2_launch_status();
}

}  // extern "C"
