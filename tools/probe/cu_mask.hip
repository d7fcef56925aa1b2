// Which CUs does a stream created with hipExtStreamCreateWithCUMask use on a multi-XCD part?  Launches a grid of busy workgroups on a
// masked stream and counts the distinct (XCC, SE, CU) ids they ran on.  usage: cu_mask <bits set per 32-bit word pattern, hex> [words]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <map>
#include <vector>

__global__ void where(unsigned* out, int spin) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2] = hw;
    out[blockIdx.x * 2 + 1] = (xcc & 0xf) | (a == 0.f ? 16 : 0);
  }
}

int main(int argc, char** argv) {
  const unsigned pattern = argc > 1 ? strtoul(argv[1], nullptr, 16) : 0xffffffffu;
  const int words = argc > 2 ? atoi(argv[2]) : 8;
  std::vector<uint32_t> mask(words, pattern);
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask.data());
  printf("create: %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 1;
  const int G = 8192;
  unsigned* d;
  hipMalloc(&d, G * 8);
  hipLaunchKernelGGL(where, dim3(G), dim3(256), 0, s, d, 20000);
  e = hipStreamSynchronize(s);
  printf("sync: %s\n", hipGetErrorString(e));
  std::vector<unsigned> h(G * 2);
  hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::set<unsigned>> per;
  for (int i = 0; i < G; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
    per[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  int total = 0;
  for (auto& kv : per) {
    printf("xcc %u: %zu CUs\n", kv.first, kv.second.size());
    total += kv.second.size();
  }
  printf("pattern %08x x %d words -> %d distinct CUs\n", pattern, words, total);
  return 0;
}
