// Exact per-row top-k of 64-bit ranking keys (the pre-NMS top-k of FCOS' predict_proposals,
// fcos_outputs.py:1238-1241: `per_candidate_scores.topk(pre_nms_top_n, sorted=False)` for every (image, FPN level)).
//
// keys are the sortable int64 of utv2_fcos_rank_keys: (score bits << 32) | (~flat index) for candidates, -1 otherwise -
// all candidates of a row are distinct, so "the k largest, descending" is unique.  torch.topk's multi-block radix select
// costs ~17 tiny launches per call over a DENSE matrix; this is an MSD radix select with 6 digits over ragged rows
// (every (image, level) row has its own width, nothing is padded):
//   per digit:  hist   - one read of the keys, atomics only for candidates that still match the decided prefix (sparse)
//               pick   - per row: scan the 2048-bin histogram from the top, fix the digit, update the remaining count; since round 3
//                        every block of the NEXT kernel of the chain (hist of the next digit / collect) redoes the pick of its row in
//                        its prologue (2048 bins, one block-wide scan) instead of a launch of its own: 9 launches per call, not 15
//   collect   - keys >= the k-th key go to the output (arbitrary order), then one workgroup per row bitonic-sorts them.
// Deterministic (the selected SET is exact, the sort fixes the order); no host synchronisation.
#include "common.h"

#define TK_BINS 2048
#define TK_MAXK 8192   // the bitonic sort of the selected keys runs in LDS: 8192 x 8 B = 64 KB (opt-in above 48 KB, see the launcher)

struct TkState {
  unsigned long long prefix;  // decided high bits of the k-th largest key
  int krem;                   // how many keys of the current prefix group are still to be taken
  int count;                  // collect cursor
};

__constant__ const int tk_shift[6] = {52, 41, 30, 20, 10, 0};
__constant__ const int tk_bits[6] = {11, 11, 11, 10, 10, 10};

// state slot d = the state BEFORE digit d is decided (slot 0: nothing decided); histogram d = the counts of digit d
__global__ __launch_bounds__(256) void topk_init_kernel(TkState* __restrict__ st, unsigned* __restrict__ hist, int* __restrict__ cursors,
                                                       int ncursors, int rows, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) { st[i].prefix = 0ull; st[i].krem = k; st[i].count = 0; }
  if (i < ncursors) cursors[i] = 0;
  for (size_t j = i; j < (size_t)6 * rows * TK_BINS; j += (size_t)gridDim.x * blockDim.x) hist[j] = 0u;
}

// The pick of digit `digit` for one row, by a whole 256-thread block: the digit of the k-th largest key among the keys matching the
// prefix (scan of the 2048-bin histogram from the top), the new prefix / remaining count.  Every block of a row computes the same.
__device__ __forceinline__ TkState tk_pick(const unsigned* __restrict__ h, int digit, TkState in) {
  __shared__ unsigned wsum[4];
  __shared__ int chosen;
  __shared__ unsigned above;
  const int t = threadIdx.x;
  const int shift = tk_shift[digit], nb = 1 << tk_bits[digit];
  constexpr int PER = TK_BINS / 256;
  unsigned loc[PER], sum = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {          // thread t owns bins [t*8, t*8+8) counted from the TOP (descending digit order)
    const int d = nb - 1 - (t * PER + j);
    loc[j] = d >= 0 ? h[d] : 0u;
    sum += loc[j];
  }
  unsigned inc = sum;                        // inclusive scan over the block: wave scan + wave totals
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(inc, o, 64);
    if ((t & 63) >= o) inc += v;
  }
  if ((t & 63) == 63) wsum[t >> 6] = inc;
  if (t == 0) { chosen = -1; above = 0u; }
  __syncthreads();
  unsigned run = inc - sum;
  for (int w = 0; w < (t >> 6); ++w) run += wsum[w];
  if (in.krem > 0) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int d = nb - 1 - (t * PER + j);
      if (d >= 0 && run < (unsigned)in.krem && run + loc[j] >= (unsigned)in.krem) { chosen = d; above = run; }
      run += loc[j];
    }
  }
  __syncthreads();
  TkState out = in;
  if (chosen >= 0) {
    out.prefix = in.prefix | ((unsigned long long)chosen << shift);
    out.krem = in.krem - (int)above;
  } else {
    out.krem = 0;  // fewer than k candidates match: take them all (remaining digits stay 0 => threshold = prefix)
  }
  __syncthreads();   // chosen / above / wsum may be reused by the caller's next pick
  return out;
}

// digit > 0: the block first decides digit - 1 of its row from that digit's histogram and state slot digit - 1 (block x == 0 of the row
// stores the result as slot `digit` for the next kernel of the chain), then counts digit `digit`
__global__ __launch_bounds__(256) void topk_hist_kernel(const long long* __restrict__ keys, const long long* __restrict__ row_off,
                                                       TkState* __restrict__ st, unsigned* __restrict__ hist, int digit, int rows) {
  // block-private LDS histogram, flushed with one global atomic per non-empty bin: the candidates of a row share their
  // high (exponent) bits, so global atomics straight from the lanes would serialise on a handful of addresses
  __shared__ unsigned lh[TK_BINS];
  const int row = blockIdx.y;
  const long long base = row_off[row], width = row_off[row + 1] - base;
  if (blockIdx.x > 0 && (long long)blockIdx.x * blockDim.x >= width) return;  // block-uniform; block 0 carries the state along
  TkState cur = st[(size_t)(digit > 0 ? digit - 1 : 0) * rows + row];
  if (digit > 0) {
    cur = tk_pick(hist + ((size_t)(digit - 1) * rows + row) * TK_BINS, digit - 1, cur);
    if (blockIdx.x == 0 && threadIdx.x == 0) st[(size_t)digit * rows + row] = cur;
  }
  const int shift = tk_shift[digit], bits = tk_bits[digit];
  const unsigned long long prefix = cur.prefix;
  const int hi = shift + bits;  // bits >= hi are decided
  for (int i = threadIdx.x; i < TK_BINS; i += blockDim.x) lh[i] = 0u;
  __syncthreads();
  for (long long i0 = (long long)blockIdx.x * blockDim.x; i0 < width; i0 += (long long)gridDim.x * blockDim.x) {
    const long long i = i0 + threadIdx.x;
    int bin = -1;
    if (i < width) {
      const long long key = keys[base + i];
      const unsigned long long u = (unsigned long long)key;
      if (key >= 0 && !(hi < 64 && (u >> hi) != (prefix >> hi))) bin = (int)((u >> shift) & ((1u << bits) - 1u));
    }
    // DENSE rows (every anchor of an RPN level is a candidate) put a whole wave into one or two bins of the leading digits: the lanes
    // that share the first active lane's bin are counted by ONE atomic (64 same-address LDS atomics serialise), the rest one by one
    const unsigned long long act = __ballot(bin >= 0);
    if (act) {
      const int lead = __builtin_amdgcn_readfirstlane(__shfl(bin, __ffsll((long long)act) - 1, 64));
      const unsigned long long same = __ballot(bin == lead);
      if ((threadIdx.x & 63) == __ffsll((long long)same) - 1) atomicAdd(&lh[lead], (unsigned)__popcll(same));
      if (bin >= 0 && bin != lead) atomicAdd(&lh[bin], 1u);
    }
  }
  __syncthreads();
  unsigned* h = hist + ((size_t)digit * rows + row) * TK_BINS;
  for (int i = threadIdx.x; i < TK_BINS; i += blockDim.x) {
    const unsigned v = lh[i];
    if (v) atomicAdd(&h[i], v);
  }
}

// keys >= the k-th key go to one of TK_SPREAD staging lists of the row (same-address atomics serialise at ~50 ns each:
// one cursor for a whole row would cost k of them back to back).  The block first decides the last digit of its row.
#define TK_SPREAD 16
__global__ __launch_bounds__(256) void topk_collect_kernel(const long long* __restrict__ keys, const long long* __restrict__ row_off,
                                                          const TkState* __restrict__ st, const unsigned* __restrict__ hist,
                                                          int* __restrict__ cursors, long long* __restrict__ stage, int k, int rows) {
  const int row = blockIdx.y;
  const long long base = row_off[row], width = row_off[row + 1] - base;
  if ((long long)blockIdx.x * blockDim.x >= width) return;
  const TkState fin = tk_pick(hist + ((size_t)5 * rows + row) * TK_BINS, 5, st[(size_t)5 * rows + row]);
  const long long thr = (long long)fin.prefix;  // the k-th largest key (or 0 when the row has fewer than k candidates)
  const int lane_list = (blockIdx.x + (threadIdx.x >> 6)) % TK_SPREAD;
  int* cur = cursors + row * TK_SPREAD + lane_list;
  long long* dst = stage + ((size_t)row * TK_SPREAD + lane_list) * k;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < width; i += (long long)gridDim.x * blockDim.x) {
    const long long key = keys[base + i];
    if (key < 0 || key < thr) continue;
    const int pos = atomicAdd(cur, 1);
    if (pos < k) dst[pos] = key;
  }
}

// one block per row: gather the staging lists, descending bitonic sort (padded with -1 to a power of two) in LDS
__global__ __launch_bounds__(256) void topk_sort_kernel(const int* __restrict__ cursors, const long long* __restrict__ stage,
                                                       long long* __restrict__ out, int k, int kpad) {
  extern __shared__ long long sk[];
  __shared__ int start[TK_SPREAD + 1];
  const int row = blockIdx.x;
  if (threadIdx.x == 0) {
    int run = 0;
    for (int j = 0; j < TK_SPREAD; ++j) {
      start[j] = run;
      int c = cursors[row * TK_SPREAD + j];
      if (c > k) c = k;
      run += c;
    }
    start[TK_SPREAD] = run < kpad ? run : kpad;  // == min(k, #candidates): exactly the selected set
  }
  for (int i = threadIdx.x; i < kpad; i += blockDim.x) sk[i] = -1ll;
  __syncthreads();
  for (int j = 0; j < TK_SPREAD; ++j) {
    const int s0 = start[j], n = (j + 1 < TK_SPREAD ? start[j + 1] : start[TK_SPREAD]) - s0;
    const long long* src = stage + ((size_t)row * TK_SPREAD + j) * k;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
      if (s0 + i < kpad) sk[s0 + i] = src[i];
  }
  __syncthreads();
  for (int size = 2; size <= kpad; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < kpad / 2; i += blockDim.x) {
        const int lo = 2 * i - (i & (stride - 1));
        const int hi = lo + stride;
        const bool desc = (lo & size) == 0;
        const long long a = sk[lo], b = sk[hi];
        if (desc ? a < b : a > b) { sk[lo] = b; sk[hi] = a; }
      }
      __syncthreads();
    }
  }
  long long* o = out + (size_t)row * k;
  for (int i = threadIdx.x; i < k; i += blockDim.x) o[i] = sk[i];
}

extern "C" {

static inline size_t tk_align(size_t n) { return (n + 255) & ~(size_t)255; }

int64_t utv2_topk_rows_workspace_bytes(int rows, int k) {
  return (int64_t)(tk_align((size_t)6 * rows * sizeof(TkState)) + tk_align((size_t)6 * rows * TK_BINS * sizeof(unsigned)) +
                   tk_align((size_t)rows * TK_SPREAD * sizeof(int)) + (size_t)rows * TK_SPREAD * k * sizeof(long long));
}

// keys: ragged rows, row r = keys[row_off[r] .. row_off[r+1]) (row_off: device int64[rows+1]); max_width = widest row (host).
// out[rows][k] = the k largest keys of every row in descending order, -1 padded.  1 <= k <= 8192.
int utv2_topk_rows_i64(const long long* keys, const long long* row_off, int rows, int64_t max_width, int k, long long* out, void* ws,
                       hipStream_t stream) {
  if (!keys || !row_off || !out || !ws || rows <= 0 || k < 1 || k > TK_MAXK || max_width < 1) return UTV2_EARG;
  char* w = (char*)ws;
  TkState* st = (TkState*)w;            // [6][rows]: slot d = before digit d is decided
  w += tk_align((size_t)6 * rows * sizeof(TkState));
  unsigned* hist = (unsigned*)w;        // [6][rows][TK_BINS]
  w += tk_align((size_t)6 * rows * TK_BINS * sizeof(unsigned));
  int* cursors = (int*)w;
  w += tk_align((size_t)rows * TK_SPREAD * sizeof(int));
  long long* stage = (long long*)w;
  int gx = cdiv(max_width, 256 * 16);
  if (gx > 256) gx = 256;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(topk_init_kernel, dim3(cdiv((size_t)6 * rows * TK_BINS, 256 * 8)), dim3(256), 0, stream, st, hist, cursors, rows * TK_SPREAD,
                     rows, k);
  for (int d = 0; d < 6; ++d)
    hipLaunchKernelGGL(topk_hist_kernel, dim3(gx, rows), dim3(256), 0, stream, keys, row_off, st, hist, d, rows);
  hipLaunchKernelGGL(topk_collect_kernel, dim3(gx, rows), dim3(256), 0, stream, keys, row_off, (const TkState*)st, (const unsigned*)hist, cursors,
                     stage, k, rows);
  int kpad = 2;
  while (kpad < k) kpad <<= 1;
  static LdsOptIn sort_opt_in;
  sort_opt_in({(const void*)topk_sort_kernel}, TK_MAXK * (int)sizeof(long long));
  hipLaunchKernelGGL(topk_sort_kernel, dim3(rows), dim3(256), kpad * sizeof(long long), stream, (const int*)cursors,
                     (const long long*)stage, out, k, kpad);
  return utv2_launch_status();
}

}  // extern "C"
