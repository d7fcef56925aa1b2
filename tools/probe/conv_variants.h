// Two experimental schedules of the 256 x 256 forward / dgrad tile of csrc/conv_bf16.hip, built and measured in round 5 and NOT shipped
// (DESIGN section 9): both produce bit-identical outputs to conv_igemm_bf16_pp and neither beats it.  Included by q4_trace.hip AFTER
// conv_bf16.hip (they use its ConvArgs16, prologue geometry, zero page and epilogue_rows).
#pragma once
#include <utility>
// ---------------------------------------------------------------------------------------------
// The same 256 x 256 tile on FOUR waves - one per SIMD, each a 128 x 128 sub-tile (16 accumulator blocks = 256 of the 512 registers a
// lone wave on its SIMD may hold).  Against the 8-wave ping-pong above: 8 fragment reads feed 16 MFMAs per k16 step (12 : 16 there, half
// the LDS read traffic per FLOP), ONE s_barrier per 32-channel segment (32 MFMAs per wave) instead of two hard slot hand-offs, and nothing
// two waves of a SIMD have to take turns for - the wave software-pipelines itself: the fragments of k16 step t+1 are read (asm, register-
// tied waits) and the LDS-DMA pieces of a later segment are issued in the issue gaps BEHIND the MFMAs of step t (one filler per gap; the
// matrix pipe takes a 32x32x16 every 32 cycles, a wave can issue ~5 other instructions meanwhile).
// LDS image: a ring of four 32 KB segment slots ([slot][operand][256 rows][64 B], 16-byte k-slot XOR-swizzled with (row >> 2) & 3 on the
// source side as above).  Wave w stages rows [64 w, 64 w + 64) of both operands: 8 pieces (16 rows x 64 B) per segment.  Barrier B_s sits
// at the head of the SECOND k16 step of segment s: every wave has then read all of segment s (its second step's fragments were fetched
// during the first), so behind B_s the slot of segment s is free for segment s + 4 ... issued as: pieces 0-3 (im2col rows) of segment
// s + 4 behind B_s in that step, pieces 4-7 (weight rows) in the first step of segment s + 1; each wave waits (counted vmcnt) for ITS
// pieces of segment s + 1 before B_s, so behind B_s segment s + 1 is complete and its first fragments are read in the same step.  A
// piece is first read >= 4 k16 steps (2048+ matrix-pipe cycles) after its issue.  Same accumulation order as the other kernels (k16
// steps in order, fp32): bit-identical outputs; same epilogue (four 64 x 64 calls per wave, transposed accumulator blocks).
__device__ uint4 g_zero128[8] = {};   // zero page of this kernel's pieces (never written)
#ifdef UTV2_Q4_TRACE
// tools/probe/q4_trace.hip: s_memtime at the phase boundaries (tile entry, main loop start, main loop end, tile end) of the first 16 tiles
// of workgroup 0 and of a mid-grid workgroup, wave 0; [..][16][4] = {s_memtime of the kernel entry, s_memtime at exit, s_memrealtime delta, 0}
__device__ unsigned long long g_q4_phase[2][17][4];
#define Q4_STAMP(k)                                                                                         \
  if (tr_on && tr_tile < 16) g_q4_phase[tr_wg][tr_tile][k] = __builtin_amdgcn_s_memtime();
#else
#define Q4_STAMP(k)
#endif
#ifndef Q4_DBG
#define Q4_DBG 0   // timing experiments only (results are then WRONG): 1 no DMA in the loop, 2 no fragment reads, 4 no barriers, 8 no epilogue, 16 no source arithmetic
#endif
template <typename F, int... I>
__device__ __forceinline__ void static_for_seq(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <typename F>
__device__ __forceinline__ void static_for16(F&& f) {
  static_for_seq(f, std::make_integer_sequence<int, 16>{});
}
template <bool ML, typename TO>
__global__ __launch_bounds__(256) void conv_igemm_bf16_q4(ConvArgs16 p) {
  constexpr int BM = 256, BN = 256, BK = 64, SEGB = 64;
  constexpr int OPSEG = 256 * SEGB, SLOT = 2 * OPSEG;  // 16 KB per operand, 32 KB per ring slot
  constexpr int PATCH = 4 * 32 * (2 * 32 + 4) * 4;
  static_assert(4 * SLOT >= PATCH, "epilogue patches must fit the staging LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int tilesN = (p.K + BN - 1) / BN;
  const int nwg = p.ntiles > 0 ? p.ntiles : (int)gridDim.x;   // persistent grid as conv_igemm_bf16_pp
#ifdef UTV2_Q4_TRACE
  const bool tr_on = (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 + 3) && tid == 0;
  const int tr_wg = blockIdx.x == 0 ? 0 : 1;
  int tr_tile = 0;
  const unsigned long long tr_t0 = __builtin_amdgcn_s_memtime(), tr_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
  Q4_STAMP(0);
  int tile;
  {
    const int bid = vb, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int mt = tile / tilesN, nt = tile - mt * tilesN;
  const int m0 = p.m_begin + mt * BM, n0 = nt * BN;
  // DMA role of the lane: rows 64 * wid + 16 * j + (lane >> 2) of either operand, the k-slot that belongs in physical slot lane & 3
  const int prow = lane >> 2;
  const int kslot = (lane & 3) ^ ((lane >> 4) & 3);
  const int ntaps = p.KH * p.KW;
  const int goff = p.groups > 1 ? (n0 / (p.K / p.groups)) * p.C : 0;

  int aoff[4], awc[4];
  unsigned amask[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wid * 64 + j * 16 + prow;
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
  int boff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) boff[j] = (n0 + wid * 64 + j * 16 + prow) * p.Kred + kslot * 8;   // K % 256 == 0 (launcher): every row exists

  f32x16 acc[2][4][2];   // [column half][row block][column block of the half]: a (row pair, column half) is one epilogue call
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[h][i][j][e] = 0.f;

  const h16_t* __restrict__ xb = (const h16_t*)p.x;
  const int nchunks = ntaps * (p.C / BK), nseg = 2 * nchunks;
  int kh = 0, kw = 0, c0 = 0, tap = 0;
  const h16_t* zero = (const h16_t*)g_zero128;
  bool in_loop = false;   // (Q4_DBG)
  // sources of the wave's 8 pieces of one 64-channel chunk (segment 0; segment 1 = + 32 elements): 0-3 im2col rows, 4-7 weight rows;
  // prep_piece(ps, j) fills ps[j] and ps[4 + j] for the chunk under the cursor (one call per issue gap), cursor_next() moves on
  auto prep_piece = [&](const h16_t* (&ps)[8], int j) {
    if ((Q4_DBG & 16) && in_loop) return;
    ps[j] = ((amask[j] >> tap) & 1u) ? xb + (unsigned)(aoff[j] + kh * awc[j] + kw * p.xs + c0) : zero;
    ps[4 + j] = p.w + (unsigned)(boff[j] + tap * p.C + c0);
  };
  auto cursor_next = [&]() {
    ++tap;
    if (++kw == p.KW) {
      kw = 0;
      if (++kh == p.KH) { kh = 0; tap = 0; c0 += BK; }
    }
  };
  auto prep_chunk = [&](const h16_t* (&ps)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) prep_piece(ps, j);
    cursor_next();
  };
  unsigned char* const dma_row = smem + (wid * 64) * SEGB;  // wave-uniform (M0)
  auto issue_piece = [&](const h16_t* src, int slot, int q) {   // q: 0-3 operand A, 4-7 operand B
    if ((Q4_DBG & 1) && in_loop) return;
    unsigned char* d = dma_row + slot * SLOT + (q >> 2) * OPSEG + (q & 3) * 16 * SEGB;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)d, 16, 0, 0);
  };

  const int frow = lane & 31, fh = lane >> 5, fx = (frow >> 2) & 3;
  const unsigned lbase = (unsigned)(size_t)(lptr_t)smem;
  unsigned a_addr[2], b_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_addr[ks] = lbase + (wm * 128 + frow) * SEGB + (((ks * 2 + fh) ^ fx) << 4);
    b_addr[ks] = lbase + OPSEG + (wn * 128 + frow) * SEGB + (((ks * 2 + fh) ^ fx) << 4);
  }
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 fa[2][4], fb[2][4];
#define Q4_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define Q4_WAIT_FRAGS(S)                                                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                                     \
               : "+v"(fa[S][0]), "+v"(fa[S][1]), "+v"(fa[S][2]), "+v"(fa[S][3]), "+v"(fb[S][0]), "+v"(fb[S][1]), "+v"(fb[S][2]), \
                 "+v"(fb[S][3]))
#define Q4_PIN __builtin_amdgcn_sched_barrier(0)
  // one k16 step: 16 MFMAs on fragment set S; filler n goes out right behind MFMA n
  auto step = [&](auto set_tag, auto&& filler) {
    constexpr int S = decltype(set_tag)::value;
    static_for16([&](auto n_) {
      constexpr int n = decltype(n_)::value, i = n >> 2, j = n & 3;
      // operands swapped: acc is the TRANSPOSED block (rows = channels, columns = pixels), see epilogue_rows<.., TR>
      acc[j >> 1][i][j & 1] = mfma_32x32x16(__builtin_bit_cast(bf16x8_t, fb[S][j]), __builtin_bit_cast(bf16x8_t, fa[S][i]), acc[j >> 1][i][j & 1]);
      Q4_PIN;
      filler(n_);
      Q4_PIN;
    });
  };
  using set0 = std::integral_constant<int, 0>;
  using set1 = std::integral_constant<int, 1>;
  // the 8 fragment reads of k16 step ks of the segment in ring slot `slot` into set S, read n behind MFMA n (n = 0..7)
#define Q4_READ_N(S, n, aa, bb)                                               \
  if constexpr ((Q4_DBG & 2) != 0) {}                                         \
  else if constexpr ((n) < 4) { Q4_READ(fb[S][(n) & 3], bb, ((n) & 3) * 2048); }   \
  else if constexpr ((n) < 8) { Q4_READ(fa[S][(n) & 3], aa, ((n) & 3) * 2048); }

  const h16_t* pa[8];   // chunk whose pieces are being issued
  const h16_t* pn[8];   // the chunk after it
  // prologue: segments 0, 1 (chunk 0), 2 and the im2col pieces of 3 (chunk 1)
  prep_chunk(pa);
#pragma unroll
  for (int q = 0; q < 8; ++q) issue_piece(pa[q], 0, q);
#pragma unroll
  for (int q = 0; q < 8; ++q) issue_piece(pa[q] + 32, 1, q);
  if (nchunks > 1) {
    prep_chunk(pa);
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_piece(pa[q], 2, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece(pa[q] + 32, 3, q);
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  Q4_PIN;
  __builtin_amdgcn_s_barrier();
  Q4_PIN;
  {
    const unsigned aa = a_addr[0], bb = b_addr[0];
    Q4_READ(fb[0][0], bb, 0); Q4_READ(fb[0][1], bb, 2048); Q4_READ(fb[0][2], bb, 4096); Q4_READ(fb[0][3], bb, 6144);
    Q4_READ(fa[0][0], aa, 0); Q4_READ(fa[0][1], aa, 2048); Q4_READ(fa[0][2], aa, 4096); Q4_READ(fa[0][3], aa, 6144);
    Q4_WAIT_FRAGS(0);
  }
  // Chunk c = segments 2c, 2c + 1 (ring slots 2 (c & 1), 2 (c & 1) + 1).  On entry: set 0 holds the fragments of (2c, step 0); `pa` = chunk
  // c + 1, whose segment-1 weight pieces are still to go out; issued so far: everything up to the im2col pieces of segment 2c + 3.
  in_loop = true;
  Q4_STAMP(1);
  for (int c = 0; c < nchunks; ++c) {
    const int sl = (c & 1) * 2;                       // ring slot of segment 2c (2c + 1: sl + 1; 2c + 2: sl ^ 2; 2c + 3: (sl ^ 2) + 1)
    const bool more1 = c + 1 < nchunks, more2 = c + 2 < nchunks;
    const unsigned s0 = sl * SLOT, s1 = s0 + SLOT, s2 = (sl ^ 2) * SLOT;
    // ---- segment 2c, step 0: read (2c, 1) -> set 1; weight pieces of segment 2c + 3; sources of chunk c + 2
    {
      const unsigned aa = a_addr[1] + s0, bb = b_addr[1] + s0;
      step(set0{}, [&](auto n_) {
        constexpr int n = decltype(n_)::value;
        Q4_READ_N(1, n, aa, bb)
        if constexpr (n >= 8 && !(n & 1)) { if (more2) prep_piece(pn, (n - 8) / 2); }
        if constexpr (n >= 9 && (n & 1)) { if (more1) issue_piece(pa[4 + (n - 9) / 2] + 32, (sl ^ 2) + 1, 4 + (n - 9) / 2); }
        if constexpr (n == 15) { if (more2) cursor_next(); }
      });
      Q4_WAIT_FRAGS(1);
    }
    // ---- segment 2c, step 1: B_2c; read (2c + 1, 0) -> set 0; im2col pieces of segment 2c + 4 (chunk c + 2) into the slot of segment 2c
    {
      if (more1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // own pieces of segment 2c + 1 have landed (2c + 2, 2c + 3 may be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      Q4_PIN;
      if (!(Q4_DBG & 4)) __builtin_amdgcn_s_barrier();
      Q4_PIN;
      const unsigned aa = a_addr[0] + s1, bb = b_addr[0] + s1;
      step(set1{}, [&](auto n_) {
        constexpr int n = decltype(n_)::value;
        Q4_READ_N(0, n, aa, bb)
        if constexpr (n >= 9 && (n & 1)) { if (more2) issue_piece(pn[(n - 9) / 2], sl, (n - 9) / 2); }
      });
      Q4_WAIT_FRAGS(0);
    }
    // ---- segment 2c + 1, step 0: read (2c + 1, 1) -> set 1; weight pieces of segment 2c + 4
    {
      const unsigned aa = a_addr[1] + s1, bb = b_addr[1] + s1;
      step(set0{}, [&](auto n_) {
        constexpr int n = decltype(n_)::value;
        Q4_READ_N(1, n, aa, bb)
        if constexpr (n >= 9 && (n & 1)) { if (more2) issue_piece(pn[4 + (n - 9) / 2], sl, 4 + (n - 9) / 2); }
      });
      Q4_WAIT_FRAGS(1);
    }
    // ---- segment 2c + 1, step 1: B_(2c+1); read (2c + 2, 0) -> set 0; im2col pieces of segment 2c + 5 into the slot of segment 2c + 1
    {
      if (more2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (more1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      Q4_PIN;
      if (!(Q4_DBG & 4)) __builtin_amdgcn_s_barrier();
      Q4_PIN;
      const unsigned aa = a_addr[0] + s2, bb = b_addr[0] + s2;
      step(set1{}, [&](auto n_) {
        constexpr int n = decltype(n_)::value;
        if (more1) { Q4_READ_N(0, n, aa, bb) }
        if constexpr (n >= 9 && (n & 1)) { if (more2) issue_piece(pn[(n - 9) / 2] + 32, sl + 1, (n - 9) / 2); }
      });
      Q4_WAIT_FRAGS(0);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) pa[q] = pn[q];
  }
#undef Q4_READ
#undef Q4_READ_N
#undef Q4_WAIT_FRAGS
#undef Q4_PIN
  Q4_STAMP(2);
  __syncthreads();   // every wave is done with the ring: the epilogue patches overlay it

  float* patch = (float*)smem + wid * (32 * (2 * 32 + 4));
  // four explicit calls (a loop the optimizer declines to unroll would index the accumulator registers dynamically: scratch)
#define Q4_EPI(H, R)                                                                                                                   \
  epilogue_rows<2, TO, true, true>(*(const f32x16(*)[2][2]) & acc[H][2 * (R)], patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual, \
                                   p.relu, p.accumulate, m0 + wm * 128 + 64 * (R), n0 + wn * 128 + 64 * (H), p.M, p.K, (const TO*)p.mask,       \
                                   (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits)
  if (!(Q4_DBG & 8) || acc[0][0][0][0] == 12345.678f) {
  Q4_EPI(0, 0);
  Q4_EPI(0, 1);
  Q4_EPI(1, 0);
  Q4_EPI(1, 1);
  }
#undef Q4_EPI
  __syncthreads();   // persistent grid: every wave is done with its epilogue patch before the next tile's first DMA pieces land there
#ifdef UTV2_Q4_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  Q4_STAMP(3);
  ++tr_tile;
#endif
  }
#ifdef UTV2_Q4_TRACE
  if (tr_on) {
    g_q4_phase[tr_wg][16][0] = tr_t0;
    g_q4_phase[tr_wg][16][1] = __builtin_amdgcn_s_memtime();
    g_q4_phase[tr_wg][16][2] = __builtin_amdgcn_s_memrealtime() - tr_r0;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// The self-pipelined stream of conv_igemm_bf16_q4 on EIGHT waves (2 x 4, 128 x 64 per wave: the wave tile, DMA roles and epilogue of the
// ping-pong kernel), two FREE-RUNNING waves per SIMD: no LOAD / COMPUTE slots and no hand-off barriers - each wave reads the fragments of
// k16 step t + 1 and issues its LDS-DMA pieces in the issue gaps behind its 8 MFMAs of step t, and the matrix pipe takes whichever of the
// two waves has an MFMA ready: the ~50-cycle issue stall of one wave's LDS-DMA piece (measured on the four-wave kernel, where nothing
// covers it: 750 cycles per 64-deep chunk) is covered by the partner's MFMAs.  One s_barrier per 32-channel segment (B_s at the head of
// the segment's second step, as in q4); ring of four 32 KB slots; per wave and segment 4 pieces (2 im2col, 2 weight), 12 fragment reads,
// 16 MFMAs.  Same accumulation order: bit-identical outputs.
#ifdef UTV2_F8_TRACE
__device__ unsigned long long g_f8_phase[2][17][4];
#define F8_STAMP(k)                                                                                         \
  if (tr_on && tr_tile < 16) g_f8_phase[tr_wg][tr_tile][k] = __builtin_amdgcn_s_memtime();
#else
#define F8_STAMP(k)
#endif
#ifndef F8_BUF
#define F8_BUF 1   // LDS-DMA pieces as buffer loads (32-bit offsets) instead of global loads (64-bit pointers)
#endif
#ifndef F8_STAGGER
#define F8_STAGGER 1   // the two waves of a SIMD issue their LDS-DMA pieces in different gaps
#endif
#ifndef F8_DBG
#define F8_DBG 0   // timing experiments only (results are then WRONG): 1 no DMA in the loop, 2 no fragment reads, 4 no barriers, 8 no epilogue, 16 L2-resident im2col sources, 32 no vmcnt waits
#endif
template <typename F>
__device__ __forceinline__ void static_for8(F&& f) {
  static_for_seq(f, std::make_integer_sequence<int, 8>{});
}
template <bool ML, typename TO>
__global__ __launch_bounds__(512) void conv_igemm_bf16_f8(ConvArgs16 p) {
  constexpr int BM = 256, BN = 256, BK = 64, SEGB = 64;
  constexpr int OPSEG = 256 * SEGB, SLOT = 2 * OPSEG;  // 16 KB per operand, 32 KB per ring slot
  constexpr int PATCH = 8 * 32 * (2 * 32 + 4) * 4;
  static_assert(4 * SLOT >= PATCH, "epilogue patches must fit the staging LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 2, wn = wid & 3;
  const int tilesN = (p.K + BN - 1) / BN;
  const int nwg = p.ntiles > 0 ? p.ntiles : (int)gridDim.x;   // persistent grid as conv_igemm_bf16_pp
#ifdef UTV2_F8_TRACE
  const bool tr_on = (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2 + 3) && tid == 0;
  const int tr_wg = blockIdx.x == 0 ? 0 : 1;
  int tr_tile = 0;
  const unsigned long long tr_t0 = __builtin_amdgcn_s_memtime(), tr_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
  F8_STAMP(0);
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
      if (F8_DBG & 16) aoff[j] = (aoff[j] & 0x3ffff) + 0x40000;   // every im2col piece from one 512 KB window (L2-resident)
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
#pragma unroll
  for (int j = 0; j < 2; ++j) boff[j] = (n0 + wid * 32 + j * 16 + prow) * p.Kred + kslot * 8;   // K % 256 == 0 (launcher): every row exists

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const h16_t* __restrict__ xb = (const h16_t*)p.x;
  const int nchunks = ntaps * (p.C / BK);
  int kh = 0, kw = 0, c0 = 0, tap = 0;
  const h16_t* zero = (const h16_t*)g_zero128;
  bool in_loop = false;   // (F8_DBG)
  // sources of the wave's 4 pieces of one 64-channel chunk (segment 0; segment 1 = + 64 bytes): 0-1 im2col rows, 2-3 weight rows.
  // F8_BUF: as 32-bit byte offsets into two buffer descriptors (x, w) - buffer_load_dwordx4 ... offen lds takes ONE address register per
  // lane instead of two, an out-of-image tap is an offset beyond num_records (the load then writes zeros: no zero page, no select of a
  // 64-bit pointer) and segment 1 is the scalar offset 64 of the same registers
#if F8_BUF
  typedef unsigned src_t;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x80000000u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x80000000u, 0x00020000);
  auto prep_piece = [&](src_t (&ps)[4], int j) {
    ps[j] = ((amask[j] >> tap) & 1u) ? (unsigned)(aoff[j] + kh * awc[j] + kw * p.xs + c0) * 2u : 0x80000000u;
    ps[2 + j] = (unsigned)(boff[j] + tap * p.C + c0) * 2u;
  };
#else
  typedef const h16_t* src_t;
  auto prep_piece = [&](src_t (&ps)[4], int j) {
    ps[j] = ((amask[j] >> tap) & 1u) ? xb + (unsigned)(aoff[j] + kh * awc[j] + kw * p.xs + c0) : zero;
    ps[2 + j] = p.w + (unsigned)(boff[j] + tap * p.C + c0);
  };
#endif
  auto cursor_next = [&]() {
    ++tap;
    if (++kw == p.KW) {
      kw = 0;
      if (++kh == p.KH) { kh = 0; tap = 0; c0 += BK; }
    }
  };
  auto prep_chunk = [&](src_t (&ps)[4]) {
    prep_piece(ps, 0);
    prep_piece(ps, 1);
    cursor_next();
  };
  unsigned char* const dma_row = smem + (wid * 32) * SEGB;  // wave-uniform (M0)
  auto issue_piece = [&](src_t src, auto seg1_tag, int slot, int q) {   // q: 0-1 operand A, 2-3 operand B; seg1: the chunk's second segment
    constexpr bool SEG1 = decltype(seg1_tag)::value;
    if ((F8_DBG & 1) && in_loop) return;
    if ((F8_DBG & 128) && in_loop && (q & 1)) return;                 // half the pieces
    if ((F8_DBG & 64) && in_loop) src = (src_t)(lane * 16);           // every piece = the same contiguous KB (L1-resident)
    unsigned char* d = dma_row + slot * SLOT + (q >> 1) * OPSEG + (q & 1) * 16 * SEGB;
#if F8_BUF
    if (q < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)d, 16, (int)src, SEG1 ? 64 : 0, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)d, 16, (int)src, SEG1 ? 64 : 0, 0, 0);
#else
    __builtin_amdgcn_global_load_lds((gptr_t)(src + (SEG1 ? 32 : 0)), (lptr_t)d, 16, 0, 0);
#endif
  };
  using seg0 = std::integral_constant<bool, false>;
  using seg1 = std::integral_constant<bool, true>;

  const int frow = lane & 31, fh = lane >> 5, fx = (frow >> 2) & 3;
  const unsigned lbase = (unsigned)(size_t)(lptr_t)smem;
  unsigned a_addr[2], b_addr[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_addr[ks] = lbase + (wm * 128 + frow) * SEGB + (((ks * 2 + fh) ^ fx) << 4);
    b_addr[ks] = lbase + OPSEG + (wn * 64 + frow) * SEGB + (((ks * 2 + fh) ^ fx) << 4);
  }
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 fa[2][4], fb[2][2];
#define F8_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define F8_WAIT_FRAGS(S) \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[S][0]), "+v"(fa[S][1]), "+v"(fa[S][2]), "+v"(fa[S][3]), "+v"(fb[S][0]), "+v"(fb[S][1]))
#define F8_PIN __builtin_amdgcn_sched_barrier(0)
  // one k16 step: 8 MFMAs on fragment set S; filler n goes out right behind MFMA n
  auto step = [&](auto set_tag, auto&& filler) {
    constexpr int S = decltype(set_tag)::value;
#ifdef F8_PRIO   // round 5: the wave keeps the matrix pipe for its step of 8 (two waves of a SIMD alternating 32x32x16 MFMAs lose a third of the pipe: mfma_peak.hip)
    __builtin_amdgcn_s_setprio(F8_PRIO);
#endif
    static_for8([&](auto n_) {
      constexpr int n = decltype(n_)::value, i = n >> 1, j = n & 1;
      // operands swapped: acc is the TRANSPOSED block (rows = channels, columns = pixels), see epilogue_rows<.., TR>
      acc[i][j] = mfma_32x32x16(__builtin_bit_cast(bf16x8_t, fb[S][j]), __builtin_bit_cast(bf16x8_t, fa[S][i]), acc[i][j]);
      F8_PIN;
      filler(n_);
      F8_PIN;
    });
#ifdef F8_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  using set0 = std::integral_constant<int, 0>;
  using set1 = std::integral_constant<int, 1>;
  // the 6 fragment reads of a k16 step into set S, read n behind MFMA n (n = 0..5)
#define F8_READ_N(S, n, aa, bb)                                               \
  if constexpr ((F8_DBG & 2) != 0) {}                                         \
  else if constexpr ((n) < 2) { F8_READ(fb[S][(n) & 1], bb, ((n) & 1) * 2048); }   \
  else if constexpr ((n) < 6) { F8_READ(fa[S][(n) - 2], aa, ((n) - 2) * 2048); }

  src_t pa[4];   // chunk whose pieces are being issued
  src_t pn[4];   // the chunk after it
  // prologue: segments 0, 1 (chunk 0), 2 and the im2col pieces of 3 (chunk 1)
  prep_chunk(pa);
#pragma unroll
  for (int q = 0; q < 4; ++q) issue_piece(pa[q], seg0{}, 0, q);
#pragma unroll
  for (int q = 0; q < 4; ++q) issue_piece(pa[q], seg1{}, 1, q);
  if (nchunks > 1) {
    prep_chunk(pa);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue_piece(pa[q], seg0{}, 2, q);
#pragma unroll
    for (int q = 0; q < 2; ++q) issue_piece(pa[q], seg1{}, 3, q);
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  F8_PIN;
  __builtin_amdgcn_s_barrier();
  F8_PIN;
  {
    const unsigned aa = a_addr[0], bb = b_addr[0];
    F8_READ(fb[0][0], bb, 0); F8_READ(fb[0][1], bb, 2048);
    F8_READ(fa[0][0], aa, 0); F8_READ(fa[0][1], aa, 2048); F8_READ(fa[0][2], aa, 4096); F8_READ(fa[0][3], aa, 6144);
    F8_WAIT_FRAGS(0);
  }
  // Chunk c = segments 2c, 2c + 1 (ring slots 2 (c & 1), 2 (c & 1) + 1).  On entry: set 0 holds the fragments of (2c, step 0); `pa` = chunk
  // c + 1, whose segment-1 weight pieces are still to go out; issued so far: everything up to the im2col pieces of segment 2c + 3.
  in_loop = true;
  F8_STAMP(1);
  // The two waves of a SIMD (w, w + 4) run the same stream one MFMA apart; an LDS-DMA piece holds its wave ~60 cycles, so the halves
  // issue theirs in DIFFERENT gaps (half 0 behind MFMAs 1 and 5, half 1 behind 3 and 7): while one wave sits in a piece the partner's
  // MFMAs keep the pipe busy (with both in gaps 6 / 7 the pipe idled ~400 cycles per chunk).  The 6 fragment reads take the other gaps.
  auto run_loop = [&](auto half_tag) {
    constexpr int HF = decltype(half_tag)::value;
    constexpr int D0 = (F8_STAGGER && HF) ? 3 : (F8_STAGGER ? 1 : 6), D1 = (F8_STAGGER && HF) ? 7 : (F8_STAGGER ? 5 : 7);
    // read index (0..5) issued in gap n, -1 = none; F8_STAGGER half 0: gaps 0 2 2 3 4 6, half 1: gaps 0 1 2 4 5 6
#define F8_GAP_READS(S, n, aa, bb)                                                                              \
    if constexpr (!F8_STAGGER) { F8_READ_N(S, n, aa, bb) }                                                       \
    else if constexpr (HF == 0) {                                                                                \
      if constexpr (n == 0) { F8_READ_N(S, 0, aa, bb) }                                                          \
      if constexpr (n == 2) { F8_READ_N(S, 1, aa, bb) F8_READ_N(S, 2, aa, bb) }                                  \
      if constexpr (n == 3) { F8_READ_N(S, 3, aa, bb) }                                                          \
      if constexpr (n == 4) { F8_READ_N(S, 4, aa, bb) }                                                          \
      if constexpr (n == 6) { F8_READ_N(S, 5, aa, bb) }                                                          \
    } else {                                                                                                     \
      if constexpr (n < 3) { F8_READ_N(S, n, aa, bb) }                                                           \
      if constexpr (n >= 4 && n < 7) { F8_READ_N(S, n - 1, aa, bb) }                                             \
    }
  for (int c = 0; c < nchunks; ++c) {
    const int sl = (c & 1) * 2;                       // ring slot of segment 2c (2c + 1: sl + 1; 2c + 2: sl ^ 2; 2c + 3: (sl ^ 2) + 1)
    const bool more1 = c + 1 < nchunks, more2 = c + 2 < nchunks;
    const unsigned s0 = sl * SLOT, s1 = s0 + SLOT, s2 = (sl ^ 2) * SLOT;
    // ---- segment 2c, step 0: read (2c, 1) -> set 1; weight pieces of segment 2c + 3; sources of chunk c + 2
    {
      const unsigned aa = a_addr[1] + s0, bb = b_addr[1] + s0;
      step(set0{}, [&](auto n_) {
        constexpr int n = decltype(n_)::value;
        F8_GAP_READS(1, n, aa, bb)
        if constexpr (n == (HF ? 1 : 3) || n == (HF ? 5 : 7)) { if (more2) prep_piece(pn, n == (HF ? 1 : 3) ? 0 : 1); }
        if constexpr (n == D0 || n == D1) { if (more1) issue_piece(pa[2 + (n == D1)], seg1{}, (sl ^ 2) + 1, 2 + (n == D1)); }
        if constexpr (n == 7) { if (more2) cursor_next(); }
      });
      F8_WAIT_FRAGS(1);
    }
    // ---- segment 2c, step 1: B_2c; read (2c + 1, 0) -> set 0; im2col pieces of segment 2c + 4 (chunk c + 2) into the slot of segment 2c
    {
      if (F8_DBG & 32) {}
      else if (more1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // own pieces of segment 2c + 1 have landed (2c + 2, 2c + 3 may be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      F8_PIN;
      if (!(F8_DBG & 4)) __builtin_amdgcn_s_barrier();
      F8_PIN;
      const unsigned aa = a_addr[0] + s1, bb = b_addr[0] + s1;
      step(set1{}, [&](auto n_) {
        constexpr int n = decltype(n_)::value;
        F8_GAP_READS(0, n, aa, bb)
        if constexpr (n == D0 || n == D1) { if (more2) issue_piece(pn[n == D1], seg0{}, sl, n == D1); }
      });
      F8_WAIT_FRAGS(0);
    }
    // ---- segment 2c + 1, step 0: read (2c + 1, 1) -> set 1; weight pieces of segment 2c + 4
    {
      const unsigned aa = a_addr[1] + s1, bb = b_addr[1] + s1;
      step(set0{}, [&](auto n_) {
        constexpr int n = decltype(n_)::value;
        F8_GAP_READS(1, n, aa, bb)
        if constexpr (n == D0 || n == D1) { if (more2) issue_piece(pn[2 + (n == D1)], seg0{}, sl, 2 + (n == D1)); }
      });
      F8_WAIT_FRAGS(1);
    }
    // ---- segment 2c + 1, step 1: B_(2c+1); read (2c + 2, 0) -> set 0; im2col pieces of segment 2c + 5 into the slot of segment 2c + 1
    {
      if (F8_DBG & 32) {}
      else if (more2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (more1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      F8_PIN;
      if (!(F8_DBG & 4)) __builtin_amdgcn_s_barrier();
      F8_PIN;
      const unsigned aa = a_addr[0] + s2, bb = b_addr[0] + s2;
      step(set1{}, [&](auto n_) {
        constexpr int n = decltype(n_)::value;
        if (more1) { F8_GAP_READS(0, n, aa, bb) }
        if constexpr (n == D0 || n == D1) { if (more2) issue_piece(pn[n == D1], seg1{}, sl + 1, n == D1); }
      });
      F8_WAIT_FRAGS(0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) pa[q] = pn[q];
  }
#undef F8_GAP_READS
  };
  if (wm == 0) run_loop(std::integral_constant<int, 0>{});
  else run_loop(std::integral_constant<int, 1>{});
#undef F8_READ
#undef F8_READ_N
#undef F8_WAIT_FRAGS
#undef F8_PIN
  F8_STAMP(2);
  __syncthreads();   // every wave is done with the ring: the epilogue patches overlay it

  float* patch = (float*)smem + wid * (32 * (2 * 32 + 4));
  if (!(F8_DBG & 8) || acc[0][0][0] == 12345.678f) {
  // two explicit calls (a loop the optimizer declines to unroll would index the accumulator registers dynamically: scratch)
  epilogue_rows<2, TO, true, true>(*(const f32x16(*)[2][2]) & acc[0], patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual,
                        p.relu, p.accumulate, m0 + wm * 128, n0 + wn * 64, p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits);
  epilogue_rows<2, TO, true, true>(*(const f32x16(*)[2][2]) & acc[2], patch, lane, (TO*)p.y, p.scale, p.bias, (const TO*)p.residual,
                        p.relu, p.accumulate, m0 + wm * 128 + 64, n0 + wn * 64, p.M, p.K, (const TO*)p.mask, (const TO*)p.post_mask, p.ldy, p.gn_part, p.bits);
  }
  __syncthreads();   // persistent grid: every wave is done with its epilogue patch before the next tile's first DMA pieces land there
#ifdef UTV2_F8_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  F8_STAMP(3);
  ++tr_tile;
#endif
  }
#ifdef UTV2_F8_TRACE
  if (tr_on) {
    g_f8_phase[tr_wg][16][0] = tr_t0;
    g_f8_phase[tr_wg][16][1] = __builtin_amdgcn_s_memtime();
    g_f8_phase[tr_wg][16][2] = __builtin_amdgcn_s_memrealtime() - tr_r0;
  }
#endif
}

