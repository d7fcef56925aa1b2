// Phase timeline of the two experimental schedules of conv_variants.h (conv_igemm_bf16_q4, _f8) and the launch time of the shipped conv_igemm_bf16_pp beside them, on the paired FCOS tower launch of the
// student batch: [268800, 512] -> [268800, 512], groups = 2, 3x3, GroupNorm partials, 2048 tiles on a persistent grid of 256 workgroups.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ../../include [-DQ4_DBG=n] -o q4_trace q4_trace.hip && ./q4_trace
#define UTV2_Q4_TRACE 1
#define UTV2_F8_TRACE 1
#include "../../unbiased-teacher-v2_amd/csrc/conv_bf16.hip"
#include "conv_variants.h"
#include <stdio.h>
#include <vector>
int main() {
  const int N = 12, C = 256, K = 512, G = 2;
  const int LH[5] = {100, 50, 25, 13, 7}, LW[5] = {168, 84, 42, 21, 11};
  ConvArgs16 a{};
  a.M = fill_levels16(a.lt, 5, N, LH, LW);
  const size_t P = (size_t)a.M;
  std::vector<unsigned short> hx(P * C * G), hw((size_t)K * 9 * C);
  unsigned s = 12345;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (s >> 30) ? (unsigned short)(0x3c00 + ((s >> 16) & 0x3ff)) : 0; }   // ReLU-like: a quarter zeros
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3a00 + ((s >> 16) & 0x1ff) - ((s >> 31) << 15)); }
  void *x, *w, *y; float* gp;
  hipMalloc(&x, hx.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&y, P * K * 2); hipMalloc(&gp, (P / 32 + 1) * (K / 8) * 8);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  a.x = x; a.w = (const h16_t*)w; a.y = y;
  a.N = N; a.C = C; a.K = K; a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1; a.in_dil = 1;
  a.Kred = 9 * C; a.xs = C * G; a.groups = G; a.ldy = K; a.gn_part = gp; a.m_begin = 0; a.relu = 0;
  const int smem = 2 * (256 + 256) * 128;
  hipFuncSetAttribute((const void*)conv_igemm_bf16_pp<true, h16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipFuncSetAttribute((const void*)conv_igemm_bf16_q4<true, h16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipFuncSetAttribute((const void*)conv_igemm_bf16_f8<true, h16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int tilesN = K / 256, rounds = (int)(P / 256) * tilesN / 256, tiles = rounds * 256;
  ConvArgs16 m = a;
  m.M = tiles / tilesN * 256;
  m.ntiles = tiles;
  unsigned long long sums[3] = {0, 1, 2};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 3; ++which)
    for (int it = 0; it < (getenv("REPS") ? atoi(getenv("REPS")) : 4); ++it) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL((conv_igemm_bf16_pp<true, h16_t>), dim3(256), dim3(512), smem, 0, m);
      else if (which == 1) hipLaunchKernelGGL((conv_igemm_bf16_q4<true, h16_t>), dim3(256), dim3(256), smem, 0, m);
      else hipLaunchKernelGGL((conv_igemm_bf16_f8<true, h16_t>), dim3(getenv("F8_GRID") ? atoi(getenv("F8_GRID")) : 256), dim3(512), smem, 0, m);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (it == 3) {   // checksum of y and the GroupNorm partials: the three schedules must agree bit for bit
        std::vector<unsigned> hy(P * K / 2), hg((P / 32) * (K / 8) * 2);
        hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hg.data(), gp, hg.size() * 4, hipMemcpyDeviceToHost);
        unsigned long long h = 1469598103934665603ull;
        for (size_t i = 0; i < (size_t)m.M * K / 2; ++i) h = (h ^ hy[i]) * 1099511628211ull;
        for (size_t i = 0; i < (size_t)(m.M / 32) * (K / 8) * 2; ++i) h = (h ^ hg[i]) * 1099511628211ull;
        sums[which] = h;
        hipMemset(y, 0xff, P * K * 2);
      }
      if (it && which == 0) {
        unsigned long long ck[2];
        hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_pp_clock), sizeof(ck));
        printf("   (pp workgroup 0: %llu cycles at %.3f GHz)\n", ck[0], ck[0] / (10.0 * ck[1]));
      }
      if (it) printf("%s launch %d: %.3f ms  %.1f TF (%d tiles, %d rounds)\n", which == 2 ? "f8" : which ? "q4" : "pp", it, ms, 2.0 * m.M * K * 9 * C / ms / 1e9, tiles, rounds);
    }
  printf("outputs (y rows of the %d tiles + GroupNorm partials) bit-identical across pp / q4 / f8: %s\n", tiles, sums[0] == sums[1] && sums[1] == sums[2] ? "yes" : "NO");
  for (int k = 0; k < 2; ++k) {
  unsigned long long ph[2][17][4];
  if (k == 0) hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_q4_phase), sizeof(ph));
  else hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_f8_phase), sizeof(ph));
  for (int b = 0; b < 2; ++b) {
    const double ghz = (double)(ph[b][16][1] - ph[b][16][0]) / (10.0 * ph[b][16][2]);
    printf("%s workgroup %s: %llu cycles, %.3f GHz\n  tile:  prologue  mainloop  (per chunk)  epilogue   total   gap to next\n", k ? "f8" : "q4", b ? "mid" : "0", ph[b][16][1] - ph[b][16][0], ghz);
    for (int t = 0; t < rounds && t < 16; t += (b ? 3 : 1))
      printf("  %2d  %9llu %9llu   %8.0f  %9llu %9llu %9lld\n", t, ph[b][t][1] - ph[b][t][0], ph[b][t][2] - ph[b][t][1], (ph[b][t][2] - ph[b][t][1]) / 36.0,
             ph[b][t][3] - ph[b][t][2], ph[b][t][3] - ph[b][t][0], t + 1 < rounds ? (long long)(ph[b][t + 1][0] - ph[b][t][3]) : 0ll);
  }
  }
  return 0;
}
