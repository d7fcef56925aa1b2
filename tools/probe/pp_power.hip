// Where does the board's power go under the shipped 256-tile ping-pong conv kernel?  The paired FCOS tower launch of the student batch
// ([268800, 512] -> [268800, 512], groups 2, 3x3, GroupNorm partials; 2048 tiles on the persistent grid of 256 workgroups) launched
// back to back for a few seconds, with parts of the kernel switched off at compile time (timing / power builds - results are garbage):
//   (none)        the shipped kernel            -DPP_NO_MFMA   no matrix instructions (DMA, fragment reads, barriers, epilogue stay)
//   -DPP_NO_DMA   no global -> LDS DMA in the main loop          -DPP_NO_READ   no ds_read of the fragments
// and combinations.  Reports launch time, tile rate, shader clock (workgroup 0's s_memtime against the 100 MHz counter) and the board
// power of the busiest card (hwmon power1_average of every card, 50 Hz, settled part).
// build: tools/probe/build_pp_power.sh      usage: pp_power_<variant> [seconds]
#define UTV2_PP_TRACE 1
#include "../../unbiased-teacher-v2_amd/csrc/conv_bf16.hip"
#include <stdio.h>
#include <dirent.h>
#include <atomic>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

static std::vector<std::string> power_files() {
  std::vector<std::string> out;
  DIR* d = opendir("/sys/class/drm");
  if (!d) return out;
  while (dirent* e = readdir(d)) {
    std::string n = e->d_name;
    if (n.rfind("card", 0) != 0 || n.find('-') != std::string::npos) continue;
    std::string hw = "/sys/class/drm/" + n + "/device/hwmon";
    DIR* h = opendir(hw.c_str());
    if (!h) continue;
    while (dirent* f = readdir(h)) {
      std::string fn = f->d_name;
      if (fn.rfind("hwmon", 0) != 0) continue;
      for (const char* leaf : {"/power1_average", "/power1_input"}) {
        std::string p = hw + "/" + fn + leaf;
        if (FILE* t = fopen(p.c_str(), "r")) { fclose(t); out.push_back(p); break; }
      }
    }
    closedir(h);
  }
  closedir(d);
  return out;
}
static double read_num(const std::string& p) {
  double v = 0;
  if (FILE* f = fopen(p.c_str(), "r")) { if (fscanf(f, "%lf", &v) != 1) v = 0; fclose(f); }
  return v;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  const int N = 12, C = 256, K = 512, G = 2;
  const int LH[5] = {100, 50, 25, 13, 7}, LW[5] = {168, 84, 42, 21, 11};
  ConvArgs16 a{};
  a.M = fill_levels16(a.lt, 5, N, LH, LW);
  const size_t P = (size_t)a.M;
  std::vector<unsigned short> hx(P * C * G), hw((size_t)K * 9 * C);
  unsigned s = 12345;
  // fp16 bit patterns: activations ReLU-like (half zeros, the rest in [1, 2)), weights +-[0.75, 1)
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (s >> 31) ? (unsigned short)(0x3c00 + ((s >> 16) & 0x3ff)) : 0; }
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3a00 + ((s >> 16) & 0x1ff) - ((s >> 31) << 15)); }
  void *x, *w, *y; float* gp;
  hipMalloc(&x, hx.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&y, P * K * 2); hipMalloc(&gp, (P / 32 + 1) * (K / 8) * 8);
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  a.x = x; a.w = (const h16_t*)w; a.y = y;
  a.N = N; a.C = C; a.K = K; a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1; a.in_dil = 1;
  a.Kred = 9 * C; a.xs = C * G; a.groups = G; a.ldy = K; a.gn_part = gp; a.m_begin = 0; a.relu = 0;
  const int smem = 2 * (256 + 256) * 128 + 8192;
  hipFuncSetAttribute((const void*)conv_igemm_bf16_pp<true, h16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const int tilesN = K / 256, rounds = (int)(P / 256) * tilesN / 256, tiles = rounds * 256;
  ConvArgs16 m = a;
  m.M = tiles / tilesN * 256;
  m.ntiles = tiles;
  auto launch = [&]() { hipLaunchKernelGGL((conv_igemm_bf16_pp<true, h16_t>), dim3(256), dim3(512), smem, 0, m); };
  launch(); hipDeviceSynchronize();
  const std::vector<std::string> pf = power_files();
  std::atomic<bool> stop{false};
  std::vector<std::vector<double>> rows;
  std::vector<double> stamps;
  const auto t0 = std::chrono::steady_clock::now();
  auto now = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  std::thread sampler([&]() {
    while (!stop) {
      std::vector<double> r;
      for (auto& p : pf) r.push_back(read_num(p) / 1e6);
      rows.push_back(r); stamps.push_back(now());
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  });
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int n = 0, n_settled = 0;
  bool mark = false;
  while (now() < secs) {
    if (!mark && now() > secs * 0.4) { hipEventRecord(e0); mark = true; }
    for (int i = 0; i < 10; ++i) launch();
    n += 10; if (mark) n_settled += 10;
    hipDeviceSynchronize();
  }
  hipEventRecord(e1); hipEventSynchronize(e1);
  stop = true; sampler.join();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long ck[2] = {0, 1};
  hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_pp_clock), sizeof(ck));
  // busiest card over the settled part
  double best = 0; int bi = -1;
  for (size_t c = 0; c < pf.size(); ++c) {
    double sum = 0; int k = 0;
    for (size_t r = 0; r < rows.size(); ++r) if (stamps[r] > secs * 0.4) { sum += rows[r][c]; ++k; }
    if (k && sum / k > best) { best = sum / k; bi = (int)c; }
  }
  const char* variant =
#if defined(PP_NO_MFMA) && defined(PP_NO_DMA) && defined(PP_NO_READ)
      "barriers + epilogue only";
#elif defined(PP_NO_MFMA) && defined(PP_NO_DMA)
      "no MFMA, no DMA (fragment reads + epilogue)";
#elif defined(PP_NO_MFMA) && defined(PP_NO_READ)
      "no MFMA, no reads (DMA + epilogue)";
#elif defined(PP_NO_DMA) && defined(PP_NO_READ)
      "MFMA only (no DMA, no reads)";
#elif defined(PP_A_EVERY)
      PP_A_EVERY == 3 ? "im2col DMA for 1 tap in 3 (row-span emulation)" : "im2col DMA for 1 tap in 9 (halo-patch emulation)";
#elif defined(PP_NO_MFMA)
      "no MFMA";
#elif defined(PP_NO_DMA)
      "no DMA";
#elif defined(PP_NO_READ)
      "no fragment reads";
#else
      "shipped kernel";
#endif
  const double per = ms / n_settled;
  printf("%-44s %7.3f ms / launch  %7.1f TF-equivalent  %6.2f Mtiles/s  clock %.3f GHz  power %6.1f W (%s)\n", variant, per,
         2.0 * m.M * K * 9 * C / per / 1e9, tiles / per / 1e3, ck[0] / (10.0 * ck[1]), best, bi >= 0 ? pf[bi].c_str() + 15 : "no hwmon");
  return 0;
}
