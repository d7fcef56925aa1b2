// Shared helpers for the utv2 HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define UTV2_OK 0
#define UTV2_EARG (-1000)

// All entry points return 0 on success, -(hipError_t) on a launch error,
// UTV2_EARG on a bad argument.  They never synchronise the stream.
static inline int utv2_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? UTV2_OK : -(int)e;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define WAVE 64

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}

__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, WAVE));
  return v;
}

// Block-wide sum; result valid in thread 0.  `red` must hold >= blockDim.x/64 floats.
__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = v;
  __syncthreads();
  const int nw = (blockDim.x + 63) >> 6;
  float r = (threadIdx.x < nw) ? red[threadIdx.x] : 0.f;
  if (wid == 0) r = wave_reduce_sum(r);
  __syncthreads();
  return r;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
