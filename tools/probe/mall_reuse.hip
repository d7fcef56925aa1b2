// Does a re-read of data that was just streamed through come back faster than HBM (the 256 MB memory-side Infinity Cache, per-XCD L2)?
// For sizes 8 MB .. 1 GB: (a) read pass over a buffer that was just READ, (b) read pass over a buffer that was just WRITTEN, against
// (c) a cold read (a 2 GB buffer streamed in between).  Build: hipcc --offload-arch=gfx950 -O3 -o mall_reuse mall_reuse.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ p, size_t n, float* out) {
  f32x4 s[4] = {};
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t st = (size_t)gridDim.x * 256;
  for (; i + 3 * st < n; i += 4 * st) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] += p[i + u * st];
  }
  for (; i < n; i += st) s[0] += p[i];
  f32x4 t = (s[0] + s[1]) + (s[2] + s[3]);
  if (t[0] + t[1] + t[2] + t[3] == 12345.678f) *out = 1.f;
}
__global__ __launch_bounds__(256) void wr(f32x4* __restrict__ p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t st = (size_t)gridDim.x * 256;
  for (; i < n; i += st) p[i] = f32x4{v, v, v, v};
}
int main() {
  const size_t big = 2ull << 30;
  float *a, *flush, *out;
  hipMalloc(&a, 1ull << 30); hipMalloc(&flush, big); hipMalloc(&out, 4);
  hipMemset(a, 0, 1ull << 30); hipMemset(flush, 0, big);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 8;
  auto timed = [&](auto f) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
  printf("%8s %14s %14s %14s %14s\n", "MB", "cold read", "re-read", "read-after-wr", "write");
  for (size_t mb : {8, 16, 32, 64, 128, 192, 256, 384, 512, 1024}) {
    const size_t bytes = mb << 20, n = bytes / 16;
    float best[4] = {1e9f, 1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 5; ++rep) {
      rd<<<grid, 256>>>((const f32x4*)flush, big / 16, out);                       // evict
      float c = timed([&] { rd<<<grid, 256>>>((const f32x4*)a, n, out); });      // cold
      float r = timed([&] { rd<<<grid, 256>>>((const f32x4*)a, n, out); });      // just read
      rd<<<grid, 256>>>((const f32x4*)flush, big / 16, out);
      float w = timed([&] { wr<<<grid, 256>>>((f32x4*)a, n, 1.f); });
      float rw = timed([&] { rd<<<grid, 256>>>((const f32x4*)a, n, out); });     // just written
      best[0] = c < best[0] ? c : best[0]; best[1] = r < best[1] ? r : best[1]; best[2] = rw < best[2] ? rw : best[2]; best[3] = w < best[3] ? w : best[3];
    }
    printf("%8zu %9.0f GB/s %9.0f GB/s %9.0f GB/s %9.0f GB/s\n", mb, bytes / best[0] / 1e6, bytes / best[1] / 1e6, bytes / best[2] / 1e6, bytes / best[3] / 1e6);
  }
  return 0;
}
