#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* addr_in, short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
  __syncthreads();
  int a = addr_in[threadIdx.x];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  int h_addr[64]; short h_out[256];
  int *d_addr; short* d_out;
  hipMalloc(&d_addr, 256); hipMalloc(&d_out, 512);
  for (int mode = 0; mode < 3; ++mode) {
    for (int l = 0; l < 64; ++l) {
      if (mode == 0) h_addr[l] = l * 4;               // lane l -> 4 contiguous shorts at l*4 (row l of a [64][4] matrix)
      if (mode == 1) h_addr[l] = (l & 15) * 64 + (l >> 4) * 4;  // lane-in-group i -> row i (stride 64 shorts), group g -> col block g*4
      if (mode == 2) h_addr[l] = 0;
    }
    hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("lane %2d (addr %4d): %4d %4d %4d %4d\n", l, h_addr[l], h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]); }
  }
  return 0;
}
