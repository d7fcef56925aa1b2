// The LOCK-STEP schedules of the 256 x 256 tile (round 2), superseded by the ping-pong kernels that ship in csrc/conv_bf16.hip
// (conv_igemm_bf16_pp / _rs: rounds 2 / 5; conv_wgrad_bf16_pp: round 5) and moved here in round 6: they were the A/B partners of those
// kernels (UTV2_PP=0, UTV2_WGRAD_PP=0 - bit-identical outputs / slabs, 6-12 % slower) and no longer appear in any trace of the step.
// Kept as source for probes: include AFTER csrc/conv_bf16.hip (ConvArgs16, Wgrad16Args, epilogue_rows, the zero page, mfma wrappers,
// lds_read_tr16 and the gptr_t / lptr_t typedefs come from there).  Nothing in the product or its tests builds this file.
#pragma once
// 256 x 256 tile, 8 waves (2 x 4), 128 x 64 per wave, BK = 64, operands staged by LDS-DMA (global_load_lds_dwordx4) into a
// double-buffered 128 KB LDS image, one workgroup per CU.  Against the 128 x 128 / 4-wave kernel above: 6 fragment reads feed
// 8 MFMAs per k16 step (was 4 : 4), a barrier every 32 MFMAs per wave (was 16), the loads of the next chunk have 2048+ matrix-pipe
// cycles to land, no staging VGPRs and no ds_write pass.  Same hoisted im2col addressing, zero page, source-side swizzle and
// epilogue.  For deep MFMA-bound layers with K >= 256 on plain NHWC bf16 inputs; launched on whole rounds of 256 tiles, the
// remaining output rows go to the 128 x 128 kernel (ConvArgs16::m_begin).
#define W8_SP 2  // k16 steps over which the 8 LDS-DMA pieces of the next chunk are issued (1: 951 TF, 2: 968 TF, 4: 933 TF on the tower convs)
template <bool ML, typename TO>
__global__ __launch_bounds__(512) void conv_igemm_bf16_w8(ConvArgs16 p) {
  constexpr int BM = 256, BN = 256, BK = 64, ROWB = BK * 2;
  constexpr int SLOTS = 8, RPP = 512 / SLOTS;                      // 64 rows staged per pass of the 512 threads
  constexpr int TM = 4, TN = 2, AP = BM / RPP, BP = BN / RPP, KS = BK / 16;
  constexpr int ABUF = BM * ROWB, BBUF = BN * ROWB;
  constexpr int STAGE = 2 * (ABUF + BBUF), PATCH = 8 * 32 * (TN * 32 + 4) * 4;
  static_assert(STAGE >= PATCH, "epilogue patches must fit the staging LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* As = smem;
  unsigned char* Bs = smem + 2 * ABUF;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keeps the LDS-DMA bases (M0) in SGPRs
  const int wm = wid >> 2, wn = wid & 3;
  const int tilesN = (p.K + BN - 1) / BN;
  const int nwg = gridDim.x;
  int tile;
  {
    const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mt = tile / tilesN, nt = tile - mt * tilesN;
  const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
  const int lrow = tid / SLOTS, slot = tid % SLOTS;
  const int kslot = slot ^ ((lrow >> 1) & 7);  // source-side swizzle: the lane fetches the k-slot that belongs in its physical slot
  const int ntaps = p.KH * p.KW;
  const int goff = p.groups > 1 ? (n0 / (p.K / p.groups)) * p.C : 0;

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
  int boff[BP];
  bool bvalid[BP];
#pragma unroll
  for (int j = 0; j < BP; ++j) {
    const int co = n0 + lrow + RPP * j;
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
  constexpr int NP = AP + BP;
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
  const int wrow0 = wid * (64 / SLOTS);  // one wave instruction fills 1 KB = 8 consecutive rows of the stage
  auto issue_piece = [&](int buf, int q) {
    if (q < AP) {
      const h16_t* src = ((amask[q] >> utap) & 1u) ? xb + (unsigned)(aoff[q] + ukh * awc[q] + ua) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + buf * ABUF + (wrow0 + RPP * q) * ROWB), 16, 0, 0);
    } else {
      const h16_t* src = bvalid[q - AP] ? p.w + (unsigned)(boff[q - AP] + ub) : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + buf * BBUF + (wrow0 + RPP * (q - AP)) * ROWB), 16, 0, 0);
    }
  };

  const int frow = lane & 31, fh = lane >> 5;
  const int swz = (frow >> 1) & 7;
  int koff[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) koff[s] = ((2 * s + fh) ^ swz) * 16;
  const int arow = (wm * 128 + frow) * ROWB, brow = (wn * 64 + frow) * ROWB;

  cursor_next();
#pragma unroll
  for (int q = 0; q < NP; ++q) issue_piece(0, q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  auto iteration = [&](int buf, auto do_load) {
    constexpr bool LOAD = decltype(do_load)::value;
    if constexpr (LOAD) cursor_next();
    const unsigned char* ab = As + buf * ABUF + arow;
    const unsigned char* bb = Bs + buf * BBUF + brow;
    bf16x8_t a[2][TM], b[2][TN];  // fragments of k16 step s+1 are read BEFORE the MFMAs of step s are issued
    auto read_frags = [&](int set, int s) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[set][i] = *(const bf16x8_t*)(ab + i * 32 * ROWB + koff[s]);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[set][j] = *(const bf16x8_t*)(bb + j * 32 * ROWB + koff[s]);
    };
    read_frags(0, 0);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) read_frags((s + 1) & 1, s + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma_32x32x16(a[s & 1][i], b[s & 1][j], acc[i][j]);
      if constexpr (LOAD) {  // all 8 pieces go out behind the MFMAs of the first two k16 steps: at least half a chunk to land
        if (s < W8_SP) {
#pragma unroll
          for (int q = s * NP / W8_SP; q < (s + 1) * NP / W8_SP; ++q) issue_piece(buf ^ 1, q);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;
  int kc = 0;
  for (; kc + 1 < nchunks; ++kc) iteration(kc & 1, yes{});
  iteration(kc & 1, no{});

  // K % 4 == 0 is guaranteed by the launcher
  float* patch = (float*)smem + wid * (32 * (TN * 32 + 4));
  // two explicit calls (a loop the optimizer declines to unroll would index the accumulator registers dynamically: scratch)
  epilogue_rows<TN, TO>(*(const f32x16(*)[2][TN]) & acc[0], patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual,
                        p.relu, p.accumulate, m0 + wm * 128, n0 + wn * 64, p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits);
  epilogue_rows<TN, TO>(*(const f32x16(*)[2][TN]) & acc[2], patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual,
                        p.relu, p.accumulate, m0 + wm * 128 + 64, n0 + wn * 64, p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits);
}


// wgrad for the deep 3x3 layers (K % 256 == 0, C % 256 == 0, bf16 x and dY): the counterpart of conv_igemm_bf16_w8.
// 256 (co) x 256 (k = one tap x 256 input channels) output tile, 8 waves (2 x 4) of 128 x 64, chunks of 64 pixels, both operand
// tiles ([64 pixels][256 channels] bf16 = 32 KB each) staged by LDS-DMA into a double-buffered 128 KB image, one workgroup per CU.
// Against the 128 x 128 / 4-wave kernel above: 12 transposing reads feed 8 MFMAs per k16 step (was 8 : 4), one barrier per 32 MFMAs
// per wave (was 8), no ds_write pass and no staging VGPRs - per chunk the LDS array serves 24 read cycles per wave and k16 step where
// the small tile's reads + writes kept it as busy as the matrix pipe itself.  Tower shape (M = 268 800): 0.52 -> 0.39 ms.
// LDS rows are unpadded (a 1 KB DMA instruction fills two pixel rows); the 64-byte block b of pixel row r lives at block b ^ (r & 3),
// applied on the SOURCE address of the DMA, so the 4 rows x 64 B a 32-lane half of ds_read_b64_tr_b16 touches fall on 64 distinct
// banks.  Work items (pixel split x tile) are laid out so that a split's tiles - which share the dY chunk and the X neighbourhood -
// run on one XCD (block b runs on XCD b % 8: speed only).  Slabs and the fixed-order reduction are the small kernel's.
//
// Memory instructions of the K loop are inline asm with hand-placed waits, because the compiler
//  * orders every LDS load it knows about behind ALL pending LDS-DMA (a vmcnt(0) in front of each fragment read that follows a DMA
//    issue: the loads of the next chunk would have to land before the current chunk is consumed), and
//  * waits for a loop-carried global load right where its result is first used - in the middle of the DMA issue sequence.
// The pixel geometry (rowinfo) is fetched by plain VMEM loads two chunks ahead, issued BEHIND the DMA pieces of the iteration and left
// in flight by a counted vmcnt(4) in front of a bare s_barrier (loads return in order; __syncthreads() would drain vmcnt).  As scalar
// loads they sat behind every lgkmcnt(0) of the fragment reads (SMEM shares that counter and returns out of order): 0.398 -> 0.387 ms.
// Measured and NOT kept (tools/bench_wgrad_pf.py, same shape): one discarded dword load per 128-byte line 1-3 chunks ahead of the DMA
// as an L2 prefetch (0.51 ms: the in-order vmcnt makes every piece wait for the older HBM-miss load); a three-stage ring of 48-pixel
// chunks with the DMA two chunks ahead and a counted vmcnt(9) in front of a bare s_barrier (0.396 ms: no gain, the DMA stream is not
// latency-bound).  With the fragment reads / MFMAs / DMA switched off in turn (UTV2_WGRAD_DEBUG_KNOBS): MFMAs alone 0.25 ms, MFMAs +
// reads 0.32, DMA alone 0.30 (64 KB per CU every 1.8 us = 8.8 TB/s out of the L2s, each dY chunk fetched by nine tiles), all 0.39.
#ifdef UTV2_WGRAD_DEBUG_KNOBS
#define WG8_DMA (!(p.debug & 16))
#define WG8_MFMA (!(p.debug & 32))
#define WG8_READ (!(p.debug & 64))
#else
#define WG8_DMA true
#define WG8_MFMA true
#define WG8_READ true
#endif
#define WGRAD_W8_BP 64
__global__ __launch_bounds__(512) void conv_wgrad_bf16_w8(Wgrad16Args p) {
  constexpr int BP = WGRAD_W8_BP, ROWB = 512, OPB = BP * ROWB, STAGE = 2 * OPB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int tilesN = p.Kred >> 8, tiles = (p.K >> 8) * tilesN;
  const int per = gridDim.x >> 3;
  const int wi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);   // (spreading a split's tiles over the XCDs instead: 0.39 -> 0.42 ms)
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

  const int total_chunks = (p.M + BP - 1) / BP;
  const int chunk_begin = split * p.chunks_per_split;
  int chunk_end = chunk_begin + p.chunks_per_split;
  if (chunk_end > total_chunks) chunk_end = total_chunks;

  // DMA role of the lane: piece q of an operand = pixel rows 8*wid + 2q + (lane >> 5) of the chunk, 16 bytes at physical slot lane & 31
  const int hr = lane >> 5, slot = lane & 31;
  int choff[2];  // source channel of the lane's 16 bytes for pieces with (2q + hr) & 3 == hr (q even) / 2 + hr (q odd)
#pragma unroll
  for (int o = 0; o < 2; ++o) choff[o] = (((slot >> 2) ^ ((2 * o + hr) & 3)) << 5) + ((slot & 3) << 3);
  const h16_t* __restrict__ xb = (const h16_t*)p.x;
  const h16_t* __restrict__ dyb = (const h16_t*)p.dy;
  const h16_t* zero = (const h16_t*)g_zero64;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x2 ri[2][4];           // rowinfo of the lane's four im2col rows, two chunks in flight (set = parity of the chunk it belongs to)
  const int rowl = 8 * wid + hr;  // + 2q
  // (two explicit copies per helper: `set` must be a compile-time constant - an asm result has to land in its final registers -
  // and inline asm inside a generic lambda cannot name the captured array)
#define WG8_RLOAD(SET)                                                                             \
  [&](int chunk) {                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                \
      int m = chunk * BP + rowl + 2 * q;                                                           \
      m = m < p.M ? m : p.M - 1;                                                                   \
      const int2* src = p.rowinfo + m;                                                             \
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(ri[SET][q]) : "v"(src));               \
    }                                                                                              \
  }
  auto rload0 = WG8_RLOAD(0);   // asm: invisible to the compiler's waitcnt insertion; covered by the counted vmcnt waits
  auto rload1 = WG8_RLOAD(1);
#undef WG8_RLOAD
  const h16_t* bsrc[4];    // source of the lane's 16 bytes of the four im2col pieces of the chunk staged in this iteration
#define WG8_BSRC(SET)                                                                                              \
  [&](int chunk) {                                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                \
      const int m = chunk * BP + rowl + 2 * q;                                                                     \
      const int W = ri[SET][q].y >> 16;                                                                            \
      const h16_t* s0 = xb + (unsigned)((ri[SET][q].x + dh * W + dw) * p.xs + ci0 + choff[q & 1]);                \
      const bool ok = (m < p.M) & (chunk < chunk_end) & ((ri[SET][q].y >> tap) & 1);                               \
      bsrc[q] = ok ? s0 : zero;                                                                                    \
    }                                                                                                              \
  }
  auto bsrc0 = WG8_BSRC(0);
  auto bsrc1 = WG8_BSRC(1);
#undef WG8_BSRC
  auto issue_piece = [&](int buf, int chunk, int q8) {  // q8 0..3: dY pieces, 4..7: im2col pieces
    const int q = q8 & 3;
    unsigned char* dst = smem + buf * STAGE + (q8 < 4 ? 0 : OPB) + (8 * wid + 2 * q) * ROWB;
    const h16_t* src;
    if (q8 < 4) {
      const int m = chunk * BP + rowl + 2 * q;
      const h16_t* s0 = dyb + (unsigned)(m * p.K + i0 + choff[q & 1]);
      src = ((m < p.M) & (chunk < chunk_end)) ? s0 : zero;   // past the split's end: zeros (keeps the vmcnt arithmetic uniform)
    } else {
      src = bsrc[q];
    }
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
  };

  // transposing fragment reads: lane (G = lane >> 4, t = lane & 15) addresses pixel row 8*(G>>1) + (t>>2) (+4 for the second half of
  // the 8-deep operand, + 16 per k16 step), the 8 bytes at 32*(G&1) + 8*(t&3) of logical 64-byte block L, stored at block L ^ (t>>2)
  const int G = lane >> 4, t = lane & 15, r3 = t >> 2;
  const unsigned lrow = (unsigned)(size_t)(lptr_t)smem + (8 * (G >> 1) + r3) * ROWB + 32 * (G & 1) + 8 * (t & 3);
  unsigned aoff[4], boff[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = lrow + (((wm * 4 + i) ^ r3) << 6);
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = OPB + lrow + (((wn * 2 + j) ^ r3) << 6);
  typedef h16_t frag_t __attribute__((ext_vector_type(8)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define READ_FRAGS(set, S)                                                                         \
  TR_READ(al[set][0], ab[0], (S) * 16 * ROWB); TR_READ(ah[set][0], ab[0], ((S) * 16 + 4) * ROWB);  \
  TR_READ(al[set][1], ab[1], (S) * 16 * ROWB); TR_READ(ah[set][1], ab[1], ((S) * 16 + 4) * ROWB);  \
  TR_READ(al[set][2], ab[2], (S) * 16 * ROWB); TR_READ(ah[set][2], ab[2], ((S) * 16 + 4) * ROWB);  \
  TR_READ(al[set][3], ab[3], (S) * 16 * ROWB); TR_READ(ah[set][3], ab[3], ((S) * 16 + 4) * ROWB);  \
  TR_READ(bl[set][0], bb[0], (S) * 16 * ROWB); TR_READ(bh[set][0], bb[0], ((S) * 16 + 4) * ROWB);  \
  TR_READ(bl[set][1], bb[1], (S) * 16 * ROWB); TR_READ(bh[set][1], bb[1], ((S) * 16 + 4) * ROWB)
  // all fragment reads issued so far have landed; ties the registers so that no MFMA moves above the wait
#define WAIT_FRAGS(set)                                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                          \
               : "+v"(al[set][0]), "+v"(al[set][1]), "+v"(al[set][2]), "+v"(al[set][3]), "+v"(ah[set][0]), "+v"(ah[set][1]), \
                 "+v"(ah[set][2]), "+v"(ah[set][3]), "+v"(bl[set][0]), "+v"(bl[set][1]), "+v"(bh[set][0]), "+v"(bh[set][1]))
  // all but the newest N VMEM operations of the wave (rowinfo loads, LDS-DMA pieces) have completed; ties the rowinfo registers
#define WAIT_VMEM(N)                                                                                                      \
  asm volatile("s_waitcnt vmcnt(" #N ")"                                                                                  \
               : "+v"(ri[0][0]), "+v"(ri[0][1]), "+v"(ri[0][2]), "+v"(ri[0][3]), "+v"(ri[1][0]), "+v"(ri[1][1]), "+v"(ri[1][2]), \
                 "+v"(ri[1][3]) : : "memory")

  // Iteration ch consumes chunk ch from stage ch & 1, issues the 8 DMA pieces of chunk ch+1 (their geometry arrived an iteration ago:
  // set (ch+1) & 1) and, BEHIND them, the 4 geometry loads of chunk ch+3 into the set the pieces just released.  The wait in front of
  // the barrier leaves those 4 loads in flight (vmcnt(4): loads return in order, so the pieces and the older geometry are in) - they
  // get a whole further chunk to come back from HBM; the barrier is a bare s_barrier (__syncthreads() would drain vmcnt).
  auto iteration = [&](int ch, int buf, auto setc) {
    constexpr bool LOAD = true;
    constexpr int SET = decltype(setc)::value;
    unsigned ab[4], bb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ab[i] = aoff[i] + buf * STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) bb[j] = boff[j] + buf * STAGE;
    s16x4 al[2][4], ah[2][4], bl[2][2], bh[2][2];  // low / high pixel quads of the 8-deep operands, two sets (k16 step parity)
#ifdef UTV2_WGRAD_DEBUG_KNOBS
    for (int u = 0; u < 2; ++u) {
      for (int i = 0; i < 4; ++i) al[u][i] = ah[u][i] = s16x4{0, 0, 0, 0};
      for (int j = 0; j < 2; ++j) bl[u][j] = bh[u][j] = s16x4{0, 0, 0, 0};
    }
#endif
    auto mfmas = [&](int set) {
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
      MFMA_BURST_BEGIN;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_32x32x16(a[i], b[j], acc[i][j]);
      MFMA_BURST_END;
    };
    if constexpr (SET == 0) bsrc0(ch + 1); else bsrc1(ch + 1);
    // k16 step s: [fragments of step s have landed] -> issue the reads of step s+1 -> 8 MFMAs -> memory work of the next chunks
    if (WG8_READ) { READ_FRAGS(0, 0); }
    WAIT_FRAGS(0);
    if (WG8_READ) { READ_FRAGS(1, 1); }
    if (WG8_MFMA) mfmas(0);
    if constexpr (LOAD) {
      if (WG8_DMA) {
#pragma unroll
        for (int q8 = 4; q8 < 8; ++q8) issue_piece(buf ^ 1, ch + 1, q8);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    WAIT_FRAGS(1);
    if (WG8_READ) { READ_FRAGS(0, 2); }
    if (WG8_MFMA) mfmas(1);
    if constexpr (LOAD) {
      if (WG8_DMA) {
#pragma unroll
        for (int q8 = 0; q8 < 4; ++q8) issue_piece(buf ^ 1, ch + 1, q8);
      }
      if constexpr (SET == 0) rload0(ch + 3); else rload1(ch + 3);
    }
    __builtin_amdgcn_sched_barrier(0);
    WAIT_FRAGS(0);
    if (WG8_READ) { READ_FRAGS(1, 3); }
    if (WG8_MFMA) mfmas(0);
    __builtin_amdgcn_sched_barrier(0);
    WAIT_FRAGS(1);
    if (WG8_MFMA) mfmas(1);
    __builtin_amdgcn_sched_barrier(0);
#ifdef UTV2_WGRAD_DEBUG_KNOBS
    if (WG8_DMA) { WAIT_VMEM(4); } else { WAIT_VMEM(0); }
#else
    WAIT_VMEM(4);
#endif
    __builtin_amdgcn_s_barrier();
  };
  if (chunk_begin < chunk_end) {
    using set0 = std::integral_constant<int, 0>;
    using set1 = std::integral_constant<int, 1>;
    // chunk k's geometry lives in set (k - chunk_begin) & 1
    rload0(chunk_begin);
    rload1(chunk_begin + 1);
    WAIT_VMEM(0);
    bsrc0(chunk_begin);
#pragma unroll
    for (int q8 = 0; q8 < 8; ++q8) issue_piece(0, chunk_begin, q8);
    rload0(chunk_begin + 2);
    WAIT_VMEM(0);
    __builtin_amdgcn_s_barrier();
    for (int ch = chunk_begin; ch < chunk_end;) {
      iteration(ch, 0, set1{});          // stages chunk ch+1 (odd offset): its geometry is in set 1; refills set 1 with chunk ch+3
      if (++ch >= chunk_end) break;
      iteration(ch, 1, set0{});
      ++ch;
    }
    WAIT_VMEM(0);   // the trailing (zero-page) pieces must not land in LDS after the workgroup has gone
  }
#undef TR_READ
#undef READ_FRAGS
#undef WAIT_FRAGS
#undef WAIT_VMEM

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

