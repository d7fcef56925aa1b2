// What does the matrix pipe of THIS board sustain with nothing else in the way?  A register-only MFMA loop (no LDS, no memory in the
// loop) with the register pattern of the tower kernel's compute slot (4 x 2 accumulator blocks per wave, 6 operand fragments), launched
// on the whole chip for seconds: TFLOP/s, the shader clock it held (s_memtime against the 100 MHz s_memrealtime), for
//   shape 0: v_mfma_f32_32x32x16_f16 (the shipped kernels' instruction)      shape 1: v_mfma_f32_16x16x32_f16
//   shape 2: v_mfma_f32_32x32x16_bf16
// and operand data: 0 = N(0,1) values, 1 = the same with half the A values zeroed (post-ReLU activations), 2 = all zeros.
// The board's power limit, not the instruction issue rate, sets the figure: it is the practical MFMA ceiling the conv kernels'
// roofline fractions are to be read against (bench.py roofline.sustained_clock_ghz).
// usage: mfma_peak [seconds per case] [workgroups] [waves per SIMD 1|2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPE, bool PRIO>
__global__ __launch_bounds__(512) void mfma_loop(const i32x4* __restrict__ in, float* __restrict__ out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  unsigned long long c0 = 0, r0 = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  i32x4 fa[2][4], fb[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[ks][i] = in[(ks * 4 + i) * 64 + lane];
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[ks][j] = in[(8 + ks * 2 + j) * 64 + lane];
  }
  if constexpr (SHAPE == 1) {
    // 16x16x32: the same 128 x 64 wave tile is 8 x 4 blocks; 8 x 4 x 4 = 128 accumulator registers as well
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fb[j & 1][j >> 1]), __builtin_bit_cast(f16x8, fa[i & 1][i >> 1]),
                                                            acc[i][j], 0, 0, 0);
      asm volatile("" : "+v"(fa[0][0]), "+v"(fb[0][0]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
  } else {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(3);   // the wave keeps the matrix pipe for its burst of 16
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if constexpr (SHAPE == 0)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fb[ks][j]), __builtin_bit_cast(f16x8, fa[ks][i]), acc[i][j], 0, 0, 0);
            else
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[ks][j]), __builtin_bit_cast(bf16x8, fa[ks][i]), acc[i][j], 0, 0, 0);
          }
      if constexpr (PRIO) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_sleep(1); }
      asm volatile("" : "+v"(fa[0][0]), "+v"(fb[0][0]));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][7];
    if (s == 12345.678f) out[threadIdx.x] = s;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = __builtin_amdgcn_s_memtime() - c0;
    clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static unsigned short f2b(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 3.0;
  const int wgs = argc > 2 ? atoi(argv[2]) : 256;
  const int wps = argc > 3 ? atoi(argv[3]) : 2;
  const int threads = wps > 2 ? 256 : 256 * wps;   // wps 3: 256-thread workgroups, two per CU when workgroups = 512
  const int ndata = argc > 4 ? atoi(argv[4]) : 3;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs, clockRate %d kHz; %d workgroups x %d threads\n", prop.name, prop.multiProcessorCount, prop.clockRate, wgs, threads);
  i32x4* din; float* dout; unsigned long long* dclk;
  CK(hipMalloc(&din, 12 * 64 * 16)); CK(hipMalloc(&dout, 4096)); CK(hipMalloc(&dclk, 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* sname[4] = {"32x32x16_f16", "16x16x32_f16", "32x32x16_bf16", "32x32x16_f16+prio"};
  const char* dname[3] = {"N(0,1)", "N(0,1), half of A zero", "zeros"};
  const int iters = 4096;   // 16 x 32x32x16 per iteration per wave
  for (int shape = 0; shape < 4; ++shape)
    for (int data = 0; data < ndata; ++data) {
      std::mt19937 rng(7); std::normal_distribution<float> nd(0.f, 1.f); std::bernoulli_distribution half(0.5);
      std::vector<unsigned short> h(12 * 64 * 8);
      for (size_t i = 0; i < h.size(); ++i) {
        float v = data == 2 ? 0.f : nd(rng);
        const bool isA = i < 8 * 64 * 8;
        if (data == 1 && isA && half(rng)) v = 0.f;
        h[i] = shape == 2 ? f2b(v) : f2h(v);   // (shape 3 = shape 0's data)
      }
      CK(hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice));
      auto launch = [&]() {
        if (shape == 0) hipLaunchKernelGGL((mfma_loop<0, false>), dim3(wgs), dim3(threads), 0, 0, din, dout, iters, dclk);
        else if (shape == 1) hipLaunchKernelGGL((mfma_loop<1, false>), dim3(wgs), dim3(threads), 0, 0, din, dout, iters, dclk);
        else if (shape == 2) hipLaunchKernelGGL((mfma_loop<2, false>), dim3(wgs), dim3(threads), 0, 0, din, dout, iters, dclk);
        else hipLaunchKernelGGL((mfma_loop<0, true>), dim3(wgs), dim3(threads), 0, 0, din, dout, iters, dclk);
      };
      launch(); CK(hipDeviceSynchronize());
      // run for `secs`, time the last third
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms1 = 0; CK(hipEventElapsedTime(&ms1, e0, e1));
      const int n = (int)(secs * 1000.0 / ms1) + 3;
      for (int i = 0; i < 2 * n / 3; ++i) launch();
      CK(hipEventRecord(e0));
      const int m = n - 2 * n / 3;
      for (int i = 0; i < m; ++i) launch();
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long clk[2]; CK(hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost));
      const double flop = (double)wgs * (threads / 64) * iters * 16.0 * 32768.0;
      const double tf = flop * m / (ms * 1e-3) / 1e12;
      const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
      // issue-rate ceiling at that clock: 4 SIMDs x 1 MFMA pipe, 32768 FLOP per 32 cycles (8 passes x 4 cycles)
      const double ceil_at_clock = (wgs < prop.multiProcessorCount ? wgs : prop.multiProcessorCount) * 4.0 * 1024.0 * ghz * 1e9 / 1e12;
      printf("%-14s %-24s %8.1f TFLOP/s  clock %.3f GHz  pipe busy %.3f  (%.3f ms / launch, settled over %d launches)\n", sname[shape], dname[data], tf, ghz,
             tf / ceil_at_clock, ms / m, m);
      fflush(stdout);
    }
  return 0;
}
