// Two-crop data path on the GPU: the weak view (Detectron2 ResizeShortestEdge + RandomFlip,
// ubteacher/data/dataset_mapper.py:40,97-99) and the strong view (ubteacher/data/detection_utils.py:8-46: torchvision
// ColorJitter / RandomGrayscale on PIL images, the reference's GaussianBlur = PIL.ImageFilter.GaussianBlur
// (data/transforms/augmentation_impl.py:7-22), ToTensor -> 3 x RandomErasing(value="random") -> ToPILImage).
// The pixel arithmetic of all of it is Pillow's integer / float32 arithmetic (Resample.c, Blend.c, Convert.c, BoxBlur.c);
// these kernels reproduce it BIT-EXACTLY on uint8 [H][W][3] images resident in HBM (oracle/aug_oracle.py, pinned against Pillow).
// All of it is byte-wise HBM-bound work: one thread per pixel, consecutive threads on consecutive pixels, no LDS needed.
// Builds with -ffp-contract=off: the float expressions below must NOT be fused into FMAs.
#include "common.h"
#include "utv2.h"

#define AUG_PRECISION_BITS 22  // Resample.c: 32 - 8 - 2
#define AUG_MAX_KSIZE 256

__device__ __forceinline__ unsigned char clip8(int v) { return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ---- Resample.c precompute_coeffs + normalize_coeffs_8bpc (bilinear / triangle filter), one thread per output index ---------------
// bounds[2*xx] = first input index, bounds[2*xx+1] = tap count; kk[xx*ksize + k] = 8.22 fixed-point weight
__global__ __launch_bounds__(256) void aug_coeffs_kernel(int in_size, int out_size, int ksize, int* __restrict__ bounds,
                                                        int* __restrict__ kk) {
  const int xx = blockIdx.x * blockDim.x + threadIdx.x;
  if (xx >= out_size) return;
  const double scale = (double)in_size / (double)out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double ss = 1.0 / filterscale;
  const double center = (xx + 0.5) * scale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in_size) xmax = in_size;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double a = (x + xmin - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    ww += a < 1.0 ? 1.0 - a : 0.0;
  }
  int* k = kk + (size_t)xx * ksize;
  for (int x = 0; x < ksize; ++x) {
    double w = 0.0;
    if (x < xmax) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      w = a < 1.0 ? 1.0 - a : 0.0;
      if (ww != 0.0) w /= ww;
    }
    k[x] = w < 0.0 ? (int)(-0.5 + w * (double)(1 << AUG_PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << AUG_PRECISION_BITS));
  }
  bounds[2 * xx] = xmin;
  bounds[2 * xx + 1] = xmax;
}

// horizontal pass: src [H][W][3] -> dst [H][OW][3]
__global__ __launch_bounds__(256) void aug_resample_h_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int H,
                                                            int W, int OW, int ksize, const int* __restrict__ bounds,
                                                            const int* __restrict__ kk) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * OW) return;
  const int y = (int)(idx / OW), xx = (int)(idx - (size_t)y * OW);
  const int x0 = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = kk + (size_t)xx * ksize;
  const unsigned char* row = src + ((size_t)y * W + x0) * 3;
  int a0 = 1 << (AUG_PRECISION_BITS - 1), a1 = a0, a2 = a0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    a0 += row[3 * t] * w;
    a1 += row[3 * t + 1] * w;
    a2 += row[3 * t + 2] * w;
  }
  unsigned char* o = dst + idx * 3;
  o[0] = clip8(a0 >> AUG_PRECISION_BITS);
  o[1] = clip8(a1 >> AUG_PRECISION_BITS);
  o[2] = clip8(a2 >> AUG_PRECISION_BITS);
}

// vertical pass (+ optional horizontal flip of the result): src [H][W][3] -> dst [OH][W][3]
__global__ __launch_bounds__(256) void aug_resample_v_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int H,
                                                            int W, int OH, int ksize, const int* __restrict__ bounds,
                                                            const int* __restrict__ kk, int flip) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)OH * W) return;
  const int yy = (int)(idx / W), x = (int)(idx - (size_t)yy * W);
  const int y0 = bounds[2 * yy], n = bounds[2 * yy + 1];
  const int* k = kk + (size_t)yy * ksize;
  const unsigned char* col = src + ((size_t)y0 * W + x) * 3;
  int a0 = 1 << (AUG_PRECISION_BITS - 1), a1 = a0, a2 = a0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    const unsigned char* p = col + (size_t)t * W * 3;
    a0 += p[0] * w;
    a1 += p[1] * w;
    a2 += p[2] * w;
  }
  unsigned char* o = dst + ((size_t)yy * W + (flip ? W - 1 - x : x)) * 3;
  o[0] = clip8(a0 >> AUG_PRECISION_BITS);
  o[1] = clip8(a1 >> AUG_PRECISION_BITS);
  o[2] = clip8(a2 >> AUG_PRECISION_BITS);
}

__global__ __launch_bounds__(256) void aug_copy_flip_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int H, int W,
                                                           int flip) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * W) return;
  const int y = (int)(idx / W), x = (int)(idx - (size_t)y * W);
  const unsigned char* p = src + idx * 3;
  unsigned char* o = dst + ((size_t)y * W + (flip ? W - 1 - x : x)) * 3;
  o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
}

// ---- Convert.c rgb2l ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int luma(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// sum of L over the image (for ImageStat mean): block partials -> one 64-bit atomic per block
__global__ __launch_bounds__(256) void aug_luma_sum_kernel(const unsigned char* __restrict__ img, size_t npix, unsigned long long* __restrict__ sum) {
  __shared__ unsigned long long red[4];
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x)
    acc += (unsigned)luma(img[3 * i], img[3 * i + 1], img[3 * i + 2]);
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sum, red[0] + red[1] + red[2] + red[3]);
}

// mean = int(sum / count + 0.5) in double (ImageStat.Stat(...).mean[0] is a Python float division)
__global__ void aug_mean_finish_kernel(const unsigned long long* __restrict__ sum, size_t npix, int* __restrict__ mean) {
  if (threadIdx.x == 0 && blockIdx.x == 0) mean[0] = (int)((double)sum[0] / (double)npix + 0.5);
}

// ---- Blend.c ImagingBlend(degenerate, image, alpha) in place; mode 0: degenerate 0 (Brightness), 1: *mean (Contrast), 2: L (Color)
__global__ __launch_bounds__(256) void aug_blend_kernel(unsigned char* __restrict__ img, size_t npix, int mode, float alpha,
                                                       const int* __restrict__ mean) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  unsigned char* p = img + 3 * i;
  const int v0 = p[0], v1 = p[1], v2 = p[2];
  int deg = 0;
  if (mode == 1) deg = mean[0];
  else if (mode == 2) deg = luma(v0, v1, v2);
  const bool inside = alpha >= 0.f && alpha <= 1.f;
  const int v[3] = {v0, v1, v2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float prod = alpha * (float)(v[c] - deg);
    const float t = (float)deg + prod;
    int o;
    if (inside) o = (int)t;
    else o = t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
    p[c] = (unsigned char)o;
  }
}

__global__ __launch_bounds__(256) void aug_grayscale_kernel(unsigned char* __restrict__ img, size_t npix) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  unsigned char* p = img + 3 * i;
  const unsigned char l = (unsigned char)luma(p[0], p[1], p[2]);
  p[0] = l; p[1] = l; p[2] = l;
}

// ---- Convert.c rgb2hsv -> uint8 wrap-add on H -> hsv2rgb (torchvision F_pil.adjust_hue) ------------------------------------------------
__global__ __launch_bounds__(256) void aug_hue_kernel(unsigned char* __restrict__ img, size_t npix, int shift) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= npix) return;
  unsigned char* px = img + 3 * idx;
  const int r = px[0], g = px[1], b = px[2];
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = cr / (float)maxc;
    const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
    float h;
    if (r == maxc) h = (float)((double)bc - (double)gc);
    else if (g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    int t = (int)((double)h * 255.0);
    uh = t < 0 ? 0 : (t > 255 ? 255 : t);
    t = (int)((double)s * 255.0);
    us = t < 0 ? 0 : (t > 255 ? 255 : t);
  }
  const int hh = (uh + shift) & 255;
  if (us == 0) {
    px[0] = (unsigned char)uv; px[1] = (unsigned char)uv; px[2] = (unsigned char)uv;
    return;
  }
  const float x = (float)hh * 6.0f / 255.0f;
  const float fi = floorf(x);
  const float f = x - fi;
  const float fs = (float)us / 255.0f;
  const float vf = (float)uv;
  const int p = clip8((int)floorf(vf * (1.0f - fs) + 0.5f));
  const int q = clip8((int)floorf(vf * (1.0f - fs * f) + 0.5f));
  const int t = clip8((int)floorf(vf * (1.0f - fs * (1.0f - f)) + 0.5f));
  int ro, go, bo;
  switch (((int)fi) % 6) {
    case 0: ro = uv; go = t; bo = p; break;
    case 1: ro = q; go = uv; bo = p; break;
    case 2: ro = p; go = uv; bo = t; break;
    case 3: ro = p; go = q; bo = uv; break;
    case 4: ro = t; go = p; bo = uv; break;
    default: ro = uv; go = p; bo = q; break;
  }
  px[0] = (unsigned char)ro; px[1] = (unsigned char)go; px[2] = (unsigned char)bo;
}

// ---- BoxBlur.c ImagingLineBoxBlur: one box pass along x (vertical == 0) or y, edge pixels replicated; src != dst -------------------
__global__ __launch_bounds__(256) void aug_box_blur_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int H, int W,
                                                          int vertical, int radius, unsigned ww, unsigned fw) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * W) return;
  const int y = (int)(idx / W), x = (int)(idx - (size_t)y * W);
  const int pos = vertical ? y : x, len = vertical ? H : W;
  const size_t step = vertical ? (size_t)W * 3 : 3;
  const unsigned char* line = src + (vertical ? (size_t)x * 3 : (size_t)y * W * 3);
  unsigned a0 = 0, a1 = 0, a2 = 0;
  for (int k = -radius; k <= radius; ++k) {
    int q = pos + k;
    q = q < 0 ? 0 : (q > len - 1 ? len - 1 : q);
    const unsigned char* p = line + (size_t)q * step;
    a0 += p[0]; a1 += p[1]; a2 += p[2];
  }
  int ql = pos - radius - 1, qr = pos + radius + 1;
  ql = ql < 0 ? 0 : ql;
  qr = qr > len - 1 ? len - 1 : qr;
  const unsigned char* pl = line + (size_t)ql * step;
  const unsigned char* pr = line + (size_t)qr * step;
  unsigned char* o = dst + idx * 3;
  o[0] = (unsigned char)((a0 * ww + (unsigned)(pl[0] + pr[0]) * fw + (1u << 23)) >> 24);
  o[1] = (unsigned char)((a1 * ww + (unsigned)(pl[1] + pr[1]) * fw + (1u << 23)) >> 24);
  o[2] = (unsigned char)((a2 * ww + (unsigned)(pl[2] + pr[2]) * fw + (1u << 23)) >> 24);
}

// ---- ToTensor -> RandomErasing(value = normal noise) -> ToPILImage on the rectangle: (noise * 255).byte() ----------------------------------
__global__ __launch_bounds__(256) void aug_erase_kernel(unsigned char* __restrict__ img, int W, int i0, int j0, int h, int w,
                                                       const float* __restrict__ noise) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)h * w) return;
  const int y = (int)(idx / w), x = (int)(idx - (size_t)y * w);
  unsigned char* o = img + ((size_t)(i0 + y) * W + j0 + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = noise[(size_t)c * h * w + idx] * 255.0f;
    o[c] = (unsigned char)((long long)truncf(v) & 255);
  }
}

__global__ __launch_bounds__(256) void aug_hwc_to_chw_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, size_t npix) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  dst[i] = src[3 * i];
  dst[npix + i] = src[3 * i + 1];
  dst[2 * npix + i] = src[3 * i + 2];
}

static int aug_ksize(int in_size, int out_size) {
  double scale = (double)in_size / (double)out_size;
  if (scale < 1.0) scale = 1.0;
  return (int)ceil(scale) * 2 + 1;
}

extern "C" {

int64_t utv2_aug_resize_workspace_bytes(int H, int W, int OH, int OW) {
  if (H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return UTV2_EARG;
  const int64_t kx = aug_ksize(W, OW), ky = aug_ksize(H, OH);
  return (int64_t)H * OW * 3 + 64 + ((int64_t)OW * (kx + 2) + (int64_t)OH * (ky + 2)) * 4 + 64;
}

int utv2_aug_resize_bilinear_u8(const unsigned char* src, int H, int W, unsigned char* dst, int OH, int OW, int flip, void* ws,
                                hipStream_t stream) {
  if (!src || !dst || !ws || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || src == dst) return UTV2_EARG;
  const int kx = aug_ksize(W, OW), ky = aug_ksize(H, OH);
  if (kx > AUG_MAX_KSIZE || ky > AUG_MAX_KSIZE) return UTV2_EARG;
  unsigned char* tmp = (unsigned char*)ws;
  const size_t tmp_bytes = (((size_t)H * OW * 3) + 63) & ~(size_t)63;
  int* bx = (int*)(tmp + tmp_bytes);
  int* kkx = bx + 2 * OW;
  int* by = kkx + (size_t)OW * kx;
  int* kky = by + 2 * OH;
  const unsigned char* vsrc = src;
  if (OW != W) {
    hipLaunchKernelGGL(aug_coeffs_kernel, dim3(cdiv(OW, 256)), dim3(256), 0, stream, W, OW, kx, bx, kkx);
    unsigned char* hdst = (OH != H) ? tmp : dst;
    if (OH == H && flip) hdst = tmp;
    hipLaunchKernelGGL(aug_resample_h_kernel, dim3(cdiv((int64_t)H * OW, 256)), dim3(256), 0, stream, src, hdst, H, W, OW, kx,
                       (const int*)bx, (const int*)kkx);
    vsrc = hdst;
  }
  if (OH != H) {
    hipLaunchKernelGGL(aug_coeffs_kernel, dim3(cdiv(OH, 256)), dim3(256), 0, stream, H, OH, ky, by, kky);
    hipLaunchKernelGGL(aug_resample_v_kernel, dim3(cdiv((int64_t)OH * OW, 256)), dim3(256), 0, stream, vsrc, dst, H, OW, OH, ky,
                       (const int*)by, (const int*)kky, flip);
  } else if (vsrc != dst) {
    hipLaunchKernelGGL(aug_copy_flip_kernel, dim3(cdiv((int64_t)OH * OW, 256)), dim3(256), 0, stream, vsrc, dst, OH, OW, flip);
  }
  return utv2_launch_status();
}

// mean_out (device int) = int(mean(L) + 0.5); sum_ws: 8 device bytes
int utv2_aug_gray_mean_u8(const unsigned char* img, int64_t npix, void* sum_ws, int* mean_out, hipStream_t stream) {
  if (!img || !sum_ws || !mean_out || npix <= 0) return UTV2_EARG;
  hipError_t e = hipMemsetAsync(sum_ws, 0, 8, stream);
  if (e != hipSuccess) return -(int)e;
  int blocks = cdiv(npix, 256 * 8);
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(aug_luma_sum_kernel, dim3(blocks), dim3(256), 0, stream, img, (size_t)npix, (unsigned long long*)sum_ws);
  hipLaunchKernelGGL(aug_mean_finish_kernel, dim3(1), dim3(64), 0, stream, (const unsigned long long*)sum_ws, (size_t)npix, mean_out);
  return utv2_launch_status();
}

int utv2_aug_blend_u8(unsigned char* img, int64_t npix, int mode, float alpha, const int* mean, hipStream_t stream) {
  if (!img || npix <= 0 || mode < 0 || mode > 2 || (mode == 1 && !mean)) return UTV2_EARG;
  hipLaunchKernelGGL(aug_blend_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, stream, img, (size_t)npix, mode, alpha, mean);
  return utv2_launch_status();
}

int utv2_aug_grayscale_u8(unsigned char* img, int64_t npix, hipStream_t stream) {
  if (!img || npix <= 0) return UTV2_EARG;
  hipLaunchKernelGGL(aug_grayscale_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, stream, img, (size_t)npix);
  return utv2_launch_status();
}

int utv2_aug_hue_u8(unsigned char* img, int64_t npix, int shift, hipStream_t stream) {
  if (!img || npix <= 0 || shift < 0 || shift > 255) return UTV2_EARG;
  hipLaunchKernelGGL(aug_hue_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, stream, img, (size_t)npix, shift);
  return utv2_launch_status();
}

int utv2_aug_box_blur_u8(const unsigned char* src, unsigned char* dst, int H, int W, int vertical, int radius, int ww, int fw,
                         hipStream_t stream) {
  if (!src || !dst || src == dst || H <= 0 || W <= 0 || radius < 0 || ww < 0 || fw < 0) return UTV2_EARG;
  hipLaunchKernelGGL(aug_box_blur_kernel, dim3(cdiv((int64_t)H * W, 256)), dim3(256), 0, stream, src, dst, H, W, vertical, radius, (unsigned)ww, (unsigned)fw);
  return utv2_launch_status();
}

int utv2_aug_erase_u8(unsigned char* img, int H, int W, int i, int j, int h, int w, const float* noise, hipStream_t stream) {
  if (!img || !noise || h <= 0 || w <= 0 || i < 0 || j < 0 || i + h > H || j + w > W) return UTV2_EARG;
  hipLaunchKernelGGL(aug_erase_kernel, dim3(cdiv((int64_t)h * w, 256)), dim3(256), 0, stream, img, W, i, j, h, w, noise);
  return utv2_launch_status();
}

int utv2_aug_hwc_to_chw_u8(const unsigned char* src, unsigned char* dst, int64_t npix, hipStream_t stream) {
  if (!src || !dst || src == dst || npix <= 0) return UTV2_EARG;
  hipLaunchKernelGGL(aug_hwc_to_chw_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, stream, src, dst, (size_t)npix);
  return utv2_launch_status();
}

}  // extern "C"
