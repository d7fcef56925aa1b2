// Slot timeline of conv_igemm_bf16_pp on the FCOS tower shape: s_memtime at both ends of every LOAD / COMPUTE slot of chunks 8..15 of
// workgroup 0, per wave, and the shader clock (s_memtime ticks per s_memrealtime tick).  -DPP_NO_DMA / -DPP_NO_READ switch the DMA
// issue / the fragment reads off (results are then garbage: timing only).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ../../include -o pp_trace pp_trace.hip && ./pp_trace
#define UTV2_PP_TRACE 1
#include "../../unbiased-teacher-v2_amd/csrc/conv_bf16.hip"
#include <stdio.h>
#include <vector>
int main() {
  const int N = 8, H = 100, W = 168, C = 256, K = 256;
  const size_t P = (size_t)N * H * W;
  std::vector<unsigned short> hx(P * C), hw((size_t)K * 9 * C);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 16) & 0x3ff) - ((s >> 31) << 15)); }
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3a00 + ((s >> 16) & 0x1ff)); }
  void *x, *w, *y;
  hipMalloc(&x, P * C * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&y, P * K * 2);
  hipMemcpy(x, hx.data(), P * C * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  ConvArgs16 a{};
  a.x = x; a.w = (const __bf16*)w; a.y = y;
  a.N = N; a.H = H; a.W = W; a.C = C; a.OH = H; a.OW = W; a.K = K; a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1; a.in_dil = 1;
  a.Kred = 9 * C; a.M = (int)P; a.xs = C; a.groups = 1; a.ldy = K; a.gn_part = nullptr; a.m_begin = 0;
  const int smem = 2 * (256 + 256) * 128 + 8192;
  hipFuncSetAttribute((const void*)conv_igemm_bf16_pp<false, __bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int tiles = (int)(P / 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((conv_igemm_bf16_pp<false, __bf16>), dim3(tiles), dim3(512), smem, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("launch %d: %.3f ms  %.1f TF (%d tiles)\n", it, ms, 2.0 * tiles * 256 * K * 9 * C / ms / 1e9, tiles);
  }
  unsigned tr[8 * 128];
  hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_pp_trace), sizeof(tr));
  printf("workgroup 0: %u shader cycles in %u x 10 ns -> %.3f GHz, %.0f cycles per chunk\n", tr[126], tr[127], tr[126] / (10.0 * tr[127]), tr[126] / 36.0);
  // 8 stamps per chunk: (start, end) of LOAD0, COMPUTE0, LOAD1, COMPUTE1 of the wave's own stream; in between: barrier waits
  for (int wv = 0; wv < 8; wv += 4) {
    const unsigned* t = tr + wv * 128;
    printf("wave %d (L b C b L b C b):", wv);
    for (int c = 2; c < 6; ++c) {
      printf("  |");
      for (int k = 0; k < 8; ++k) printf(" %4u", t[c * 8 + k + 1] - t[c * 8 + k]);
    }
    printf("   chunk period %u\n", (t[48] - t[16]) / 4);
  }
  unsigned long long ph[2][5];
  hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_pp_phase), sizeof(ph));
  for (int b = 0; b < 2; ++b)
    printf("workgroup %s: geometry %llu  prologue DMA + wait %llu  main loop %llu  epilogue incl. store drain %llu  cycles (total %llu)\n",
           b ? "mid-grid" : "0", ph[b][1] - ph[b][0], ph[b][2] - ph[b][1], ph[b][3] - ph[b][2], ph[b][4] - ph[b][3], ph[b][4] - ph[b][0]);
  return 0;
}
