// Micro-benchmark: per-CU operand delivery rate into LDS on gfx950, L2-resident data (one 512-thread workgroup per CU re-reads its own
// buffer): LDS-DMA (global_load_lds_dwordx4) vs global_load_dwordx4 -> VGPR (-> ds_write_b128), drained per batch or one batch in flight.
//   hipcc --offload-arch=gfx950 -O3 -o dma_bw dma_bw.hip && ./dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: DMA, drain every batch (8 pieces / wave = 64 KB / WG); mode 1: DMA, previous batch stays in flight (vmcnt(8));
// mode 2: loads to VGPR, drain; mode 3: loads to VGPR + ds_write_b128; mode 4: half DMA (4 pieces) + half VGPR+ds_write
template <int MODE>
__global__ __launch_bounds__(512) void k(const unsigned char* src, int iters, int bytes_per_wg, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned char* base = src + (size_t)blockIdx.x * bytes_per_wg;
  f32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (MODE >= 5) {
      // strided pieces as the conv kernels fetch them: MODE 5: 8 rows x 128 B, MODE 6: 16 rows x 64 B, MODE 7: 32 rows x 32 B; row pitch 512 B
      constexpr int RB = MODE == 5 ? 128 : MODE == 6 ? 64 : 32, LPR = RB / 16, RPP = 64 / LPR;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int piece = q * 8 + wid;                       // 64 pieces of 1 KB = 64 KB, laid out as 512 rows x 128 B in a 512-B-pitch image
        const int row = (piece * RPP + lane / LPR) % 128, col = (piece * RPP + lane / LPR) / 128;
        const unsigned char* s = base + (size_t)row * 512 + col * RB + (lane % LPR) * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(smem + buf * 65536 + piece * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else if (MODE == 0 || MODE == 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned char* s = base + (size_t)((q * 8 + wid) * 1024 + lane * 16);
        __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(smem + buf * 65536 + (q * 8 + wid) * 1024), 16, 0, 0);
      }
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else if (MODE == 2 || MODE == 3) {
      f32x4 v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = *(const f32x4*)(base + (size_t)((q * 8 + wid) * 1024 + lane * 16));
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (MODE == 3) *(f32x4*)(smem + buf * 65536 + (q * 8 + wid) * 1024 + lane * 16) = v[q];
        else acc += v[q];
      }
      __builtin_amdgcn_s_barrier();
    } else {
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *(const f32x4*)(base + (size_t)((q * 8 + wid) * 1024 + lane * 16));
#pragma unroll
      for (int q = 4; q < 8; ++q) {
        const unsigned char* s = base + (size_t)((q * 8 + wid) * 1024 + lane * 16);
        __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)(smem + buf * 65536 + (q * 8 + wid) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) *(f32x4*)(smem + buf * 65536 + (q * 8 + wid) * 1024 + lane * 16) = v[q];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  acc += *(f32x4*)(smem + tid * 16);
  if (acc[0] == 12345.678f) sink[0] = acc[1];
}

template <int MODE>
void run(const char* name, const unsigned char* d, float* sink, int nwg, int iters) {
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(512), 131072, 0, d, 10, 65536, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(512), 131072, 0, d, iters, 65536, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nwg * iters * 65536;
  printf("%-44s %7.3f ms  %7.1f GB/s per CU  %6.2f TB/s chip  %.2f us / 64 KB\n", name, ms, bytes / nwg / ms / 1e6, bytes / ms / 1e9, 1e3 * ms / iters);
}

int main() {
  const int nwg = 256, iters = 2000;
  unsigned char* d; float* sink;
  hipMalloc(&d, (size_t)nwg * 65536 + 4096);
  hipMemset(d, 1, (size_t)nwg * 65536 + 4096);
  hipMalloc(&sink, 64);
  run<0>("LDS-DMA, drained per 64 KB", d, sink, nwg, iters);
  run<1>("LDS-DMA, one batch in flight", d, sink, nwg, iters);
  run<2>("global_load x4 -> VGPR, drained", d, sink, nwg, iters);
  run<3>("global_load x4 -> VGPR -> ds_write_b128", d, sink, nwg, iters);
  run<4>("half LDS-DMA + half VGPR/ds_write", d, sink, nwg, iters);
  run<5>("LDS-DMA, pieces of 8 rows x 128 B, pitch 512", d, sink, nwg, iters);
  run<6>("LDS-DMA, pieces of 16 rows x 64 B, pitch 512", d, sink, nwg, iters);
  run<7>("LDS-DMA, pieces of 32 rows x 32 B, pitch 512", d, sink, nwg, iters);
  return 0;
}
