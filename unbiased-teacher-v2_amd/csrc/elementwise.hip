// HBM-bound elementwise / reduction kernels of the UTv2 step (gfx950).
// All arithmetic fp32, one rounding per written operation (the library is built with
// -ffp-contract=off) so results are comparable op-for-op with the reference's eager
// PyTorch expressions.
#include <stdlib.h>
#include "common.h"

// ---------------------------------------------------------------------------------------------
// Teacher EMA  (ubteacher/engine/trainer.py:468-486 == :950-968)
//   new_teacher = student * (1 - keep) + teacher * keep      evaluated exactly as written:
//   two rounded products and one rounded sum; a = (float)(1 - keep), b = (float)keep.
// One launch over the whole flat state arena: 12 B/element algorithmic traffic.
// mirror16 (optional): the 16-bit copy of the arena the mixed-precision convs read (RNE of the value just written) - saves the
// conversion pass that would re-read the arena at the start of the next forward.
__global__ __launch_bounds__(256) void ema_axpby_f32(float* __restrict__ teacher, const float* __restrict__ student,
                                                   size_t n4, size_t n, float a, float b, h16_t* __restrict__ mirror16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f32x4* t4 = (f32x4*)teacher;
  const f32x4* s4 = (const f32x4*)student;
  for (size_t j = i; j < n4; j += stride) {
    f32x4 t = t4[j], s = s4[j], r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = __fadd_rn(__fmul_rn(s[e], a), __fmul_rn(t[e], b));
    t4[j] = r;
    if (mirror16) {
      bf16x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (h16_t)r[e];
      ((bf16x4_t*)mirror16)[j] = o;
    }
  }
  // tail
  for (size_t j = n4 * 4 + i; j < n; j += stride) {
    const float r = __fadd_rn(__fmul_rn(student[j], a), __fmul_rn(teacher[j], b));
    teacher[j] = r;
    if (mirror16) mirror16[j] = (h16_t)r;
  }
}

// ---------------------------------------------------------------------------------------------
// SGD with momentum + weight decay on a flat arena (D2 build_optimizer: torch.optim.SGD,
// momentum 0.9, nesterov off).  g = grad*gscale + wd*p ; buf = mom*buf + g ; p -= lr*buf.
// `gscale_ptr` (optional, device) multiplies the gradient (1/world for DDP mean, loss-scale
// inverse); grad is zeroed afterwards if zero_grad != 0 (saves the separate memset pass).
__global__ __launch_bounds__(256) void sgd_momentum_f32(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                      size_t n, float lr, float mom, float wd, float gscale,
                                                      int zero_grad, h16_t* __restrict__ mirror16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float pv = p[i];
    float gv = __fmul_rn(g[i], gscale);
    gv = __fadd_rn(gv, __fmul_rn(wd, pv));
    const float mv = __fadd_rn(__fmul_rn(mom, m[i]), gv);
    m[i] = mv;
    const float pn = __fsub_rn(pv, __fmul_rn(lr, mv));
    p[i] = pn;
    if (mirror16) mirror16[i] = (h16_t)pn;
    if (zero_grad) g[i] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Dynamic loss scaling for the fp16 AMP mode, with torch.cuda.amp.GradScaler's semantics (the reference: engine/trainer.py:207,
// 424-426 `scaler.scale(losses).backward(); scaler.step(optimizer); scaler.update()`), entirely on the device - no host read of
// the inf flag.  state = fp32 {scale, found_inf, growth_tracker}:
//   backward runs on scale * loss;  amp_found_inf marks non-finite gradients;  sgd_momentum_amp unscales (g / scale) and applies
//   the update unless found_inf (GradScaler.step skips optimizer.step());  amp_update_scale: found_inf ? scale *= backoff, tracker = 0
//   : (++tracker == interval ? scale *= growth, tracker = 0), then clears found_inf.
__global__ __launch_bounds__(256) void amp_found_inf_f32(const float* __restrict__ g, size_t n4, size_t n, float* __restrict__ state) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < n4; i += stride) {
    const f32x4 v = ((const f32x4*)g)[i];
    // x - x is 0 for finite x, NaN for +-inf and NaN
    const float t = (v[0] - v[0]) + (v[1] - v[1]) + (v[2] - v[2]) + (v[3] - v[3]);
    bad |= !(t == 0.f);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (size_t k = n4 * 4; k < n; ++k) bad |= !((g[k] - g[k]) == 0.f);
  if (__any(bad) && (threadIdx.x & 63) == 0) state[1] = 1.0f;   // every writer stores the same value: order-free
}

__global__ __launch_bounds__(256) void sgd_momentum_amp_f32(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          size_t n, float lr, float mom, float wd, float gscale,
                                                          const float* __restrict__ state, h16_t* __restrict__ mirror16) {
  if (state[1] != 0.f) return;                       // non-finite gradients: the step is skipped (a fresh mirror16 stays fresh)
  const float inv = 1.0f / state[0];                 // the scale is a power of two: exact
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float pv = p[i];
    float gv = __fmul_rn(__fmul_rn(g[i], inv), gscale);
    gv = __fadd_rn(gv, __fmul_rn(wd, pv));
    const float mv = __fadd_rn(__fmul_rn(mom, m[i]), gv);
    m[i] = mv;
    const float pn = __fsub_rn(pv, __fmul_rn(lr, mv));
    p[i] = pn;
    if (mirror16) mirror16[i] = (h16_t)pn;
  }
}

__global__ void amp_update_scale_f32(float* __restrict__ state, float growth, float backoff, int interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float scale = state[0], tracker = state[2];
  if (state[1] != 0.f) {
    scale *= backoff;
    tracker = 0.f;
  } else {
    tracker += 1.f;
    if (tracker >= (float)interval) {
      const float grown = scale * growth;
      if (grown - grown == 0.f) scale = grown;     // only while the grown scale is finite
      tracker = 0.f;
    }
  }
  state[0] = scale;
  state[1] = 0.f;
  state[2] = tracker;
}

// ---------------------------------------------------------------------------------------------
// out = (mask_y ? (y > 0 ? dy : 0) : dy) * (scale ? scale[c] : 1)     on [M][C]
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_scale_k(const T* __restrict__ dy, const T* __restrict__ y,
                                                      const float* __restrict__ scale, T* __restrict__ out,
                                                      size_t n4, int C4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    f32x4 g = ld4(dy, i);
    if (y) {
      const f32x4 yy = ld4(y, i);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
    }
    if (scale) {
      const f32x4 s = ((const f32x4*)scale)[i % C4];
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] *= s[e];
    }
    st4(out, i, g);
  }
}

// y = a + b (used for gradient fan-in)
__global__ __launch_bounds__(256) void add_f32(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                                             size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) o[i] = a[i] + b[i];
}

// ---------------------------------------------------------------------------------------------
// 3x3 stride-2 pad-1 max pool, NHWC (ResNet stem; frozen => forward only)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_k(const TI* __restrict__ x, TO* __restrict__ y, int N, int H,
                                                         int W, int C4, int OH, int OW) {
  const size_t total = (size_t)N * OH * OW * C4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    size_t t = i;
    const int c = (int)(t % C4); t /= C4;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH); t /= OH;
    const int n = (int)t;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int ih = oh * 2 - 1 + dh;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int iw = ow * 2 - 1 + dw;
        if ((unsigned)iw >= (unsigned)W) continue;
        const f32x4 v = ld4(x, ((size_t)(n * H + ih) * W + iw) * C4 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    st4(y, i, m);
  }
}

// Backward of the 3x3 stride-2 pad-1 max pool (a trainable stem, MODEL.BACKBONE.FREEZE_AT < 1) as a GATHER: input pixel (h, w) collects
// dpool of every window whose arg-max it is - ATen's rule: the FIRST maximum in (kh, kw) scan order (`val > maxval`), so ties (frequent
// after a ReLU) go to one pixel, deterministically; relu != 0 also applies the mask x > 0 of the ReLU in front of the pool.
template <typename TX, typename TG>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_nhwc_k(const TX* __restrict__ x, const TG* __restrict__ dpool, TG* __restrict__ dx, int N,
                                                             int H, int W, int C4, int OH, int OW, int relu) {
  const size_t total = (size_t)N * H * W * C4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    size_t t = i;
    const int c = (int)(t % C4); t /= C4;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int n = (int)t;
    const f32x4 mine = ld4(x, i);
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    // windows that contain (h, w): 2 oh - 1 <= h <= 2 oh + 1, i.e. oh = h / 2 (and h / 2 + 1 for odd h)
    for (int oh = h / 2; oh <= (h + 1) / 2; ++oh) {
      if (oh >= OH) continue;
      for (int ow = w / 2; ow <= (w + 1) / 2; ++ow) {
        if (ow >= OW) continue;
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
          const int ih = oh * 2 - 1 + dh;
          if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
          for (int dw = 0; dw < 3; ++dw) {
            const int iw = ow * 2 - 1 + dw;
            if ((unsigned)iw >= (unsigned)W) continue;
            const f32x4 v = ld4(x, ((size_t)(n * H + ih) * W + iw) * C4 + c);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; arg[e] = dh * 3 + dw; }
          }
        }
        const int me = (h - (oh * 2 - 1)) * 3 + (w - (ow * 2 - 1));
        const f32x4 d = ld4(dpool, ((size_t)(n * OH + oh) * OW + ow) * C4 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (arg[e] == me) g[e] += d[e];
      }
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = mine[e] > 0.f ? g[e] : 0.f;
    }
    st4(dx, i, g);
  }
}

// bf16 -> bf16 form with 8 channels (16 bytes) per thread and 32-bit index arithmetic (max is exact: same result as the generic kernel)
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_bf16x8_k(const h16_t* __restrict__ x, h16_t* __restrict__ y, int N, int H, int W,
                                                                int C8, int OH, int OW) {
  const unsigned total = (unsigned)N * OH * OW * C8;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    unsigned t = i;
    const int c = (int)(t % C8); t /= C8;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH); t /= OH;
    const int n = (int)t;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int ih = oh * 2 - 1 + dh;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int iw = ow * 2 - 1 + dw;
        if ((unsigned)iw >= (unsigned)W) continue;
        const bf16x8_t v = ((const bf16x8_t*)x)[((size_t)(n * H + ih) * W + iw) * C8 + c];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (float)v[e]);
      }
    }
    bf16x8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16_t)m[e];
    ((bf16x8_t*)y)[i] = o;
  }
}

// FPN top-down: out[n,h,w,:] = lateral[n,h,w,:] + top[n,h/2,w/2,:]   (nearest x2)
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_add_nhwc_k(const T* __restrict__ lat, const T* __restrict__ top,
                                                           T* __restrict__ out, int N, int H, int W, int C4) {
  const size_t total = (size_t)N * H * W * C4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int TH = H >> 1, TW = W >> 1;
  for (; i < total; i += stride) {
    size_t t = i;
    const int c = (int)(t % C4); t /= C4;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int n = (int)t;
    const f32x4 a = ld4(lat, i);
    const f32x4 b = ld4(top, ((size_t)(n * TH + (h >> 1)) * TW + (w >> 1)) * C4 + c);
    st4(out, i, a + b);
  }
}

// backward of the nearest x2 upsample: dtop[n,h,w,:] (+)= sum of the 2x2 block of g
template <typename T>
__global__ __launch_bounds__(256) void downsample2x_sum_nhwc_k(const T* __restrict__ g, T* __restrict__ dtop, int N,
                                                             int TH, int TW, int C4, int accumulate) {
  const size_t total = (size_t)N * TH * TW * C4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int H = TH * 2, W = TW * 2;
  for (; i < total; i += stride) {
    size_t t = i;
    const int c = (int)(t % C4); t /= C4;
    const int w = (int)(t % TW); t /= TW;
    const int h = (int)(t % TH); t /= TH;
    const int n = (int)t;
    const size_t b = ((size_t)(n * H + 2 * h) * W + 2 * w) * C4 + c;
    f32x4 s = ld4(g, b) + ld4(g, b + C4) + ld4(g, b + (size_t)W * C4) + ld4(g, b + (size_t)W * C4 + C4);
    if (accumulate) s += ld4(dtop, i);
    st4(dtop, i, s);
  }
}

// dgrad of a stride-2 1x1 conv: the compact gradient [N][TH][TW][C] lands on the even pixels of [N][H][W][C], zeros elsewhere
template <typename T>
__global__ __launch_bounds__(256) void zero_interleave2x_nhwc_k(const T* __restrict__ src, const T* __restrict__ mask, T* __restrict__ dst,
                                                              int N, int H, int W, int TH, int TW, int C4) {
  const size_t total = (size_t)N * H * W * C4;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    size_t t = i;
    const int c = (int)(t % C4); t /= C4;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int n = (int)t;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!((h | w) & 1)) {
      v = ld4(src, ((size_t)(n * TH + (h >> 1)) * TW + (w >> 1)) * C4 + c);
      if (mask) {  // optional [N][H][W][C]: the gradient flows into a ReLU output, masked where it is made
        const f32x4 mk = ld4(mask, i);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = mk[q] > 0.f ? v[q] : 0.f;
      }
    }
    st4(dst, i, v);
  }
}

// The same on 16-bit tensors, 8 channels per thread, with the two fusions the stride-2 bottlenecks' backward wants: `add` (optional,
// [N][H][W][C]) is summed in (the gradient another consumer of the same activation produced: no elementwise add pass in autograd) and
// the ReLU mask of the even pixels can come as a bit plane (mask_bits, see epilogue_rows).  `add` is NOT masked here: its producer did.
__global__ __launch_bounds__(256) void zero_interleave2x_add_h16_k(const h16_t* __restrict__ src, const h16_t* __restrict__ mask,
                                                                  const unsigned char* __restrict__ mask_bits,
                                                                  const h16_t* __restrict__ add, h16_t* __restrict__ dst, int N, int H,
                                                                  int W, int TH, int TW, int C8) {
  const size_t total = (size_t)N * H * W * C8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    size_t t = i;
    const int c = (int)(t % C8); t /= C8;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int n = (int)t;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = 0.f;
    if (!((h | w) & 1)) {
      const bf16x8_t sv = *(const bf16x8_t*)(src + (((size_t)(n * TH + (h >> 1)) * TW + (w >> 1)) * C8 + c) * 8);
      unsigned mb = 0xffu;
      if (mask_bits) mb = mask_bits[i];
      else if (mask) {
        const bf16x8_t mk = *(const bf16x8_t*)(mask + i * 8);
        mb = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) mb |= ((float)mk[q] > 0.f ? 1u : 0u) << q;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = ((mb >> q) & 1u) ? (float)sv[q] : 0.f;
    }
    bf16x8_t o;
    if (add) {
      const bf16x8_t a = *(const bf16x8_t*)(add + i * 8);
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (h16_t)(v[q] + (float)a[q]);
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (h16_t)v[q];
    }
    *(bf16x8_t*)(dst + i * 8) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// Image normalisation + CHW -> padded NHWC4 (one_stage_detector.py:88-90 / D2 preprocess_image +
// ImageList.from_tensors): dst[n,h,w,c] = (src[c,h,w] - mean[c]) / std[c] for h<H,w<W, else 0.
template <typename T>
__global__ __launch_bounds__(256) void preprocess_chw_to_nhwc4(const T* __restrict__ src, float* __restrict__ dst, int H, int W,
                                                             int Hp, int Wp, float m0, float m1, float m2, float s0,
                                                             float s1, float s2) {
  const size_t total = (size_t)Hp * Wp;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int h = (int)(i / Wp), w = (int)(i % Wp);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (h < H && w < W) {
      const size_t o = (size_t)h * W + w, hw = (size_t)H * W;
      v[0] = ((float)src[o] - m0) / s0;
      v[1] = ((float)src[hw + o] - m1) / s1;
      v[2] = ((float)src[2 * hw + o] - m2) / s2;
    }
    ((f32x4*)dst)[i] = v;
  }
}

// bf16 form for the MFMA stem: the image lands at pixel offset (3, 3) of a [Hp+6][Wp+8][4] bf16 buffer whose border the
// caller zeroed once (rows pitch Wp+8); the padded canvas beyond the image is written as zeros here.
template <typename T>
__global__ __launch_bounds__(256) void preprocess_chw_to_nhwc4_bf16pad(const T* __restrict__ src, h16_t* __restrict__ dst, int H, int W,
                                                                     int Hp, int Wp, float m0, float m1, float m2, float s0,
                                                                     float s1, float s2) {
  const size_t total = (size_t)Hp * Wp;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int h = (int)(i / Wp), w = (int)(i % Wp);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (h < H && w < W) {
      const size_t o = (size_t)h * W + w, hw = (size_t)H * W;
      v[0] = ((float)src[o] - m0) / s0;
      v[1] = ((float)src[hw + o] - m1) / s1;
      v[2] = ((float)src[2 * hw + o] - m2) / s2;
    }
    st4(dst, (size_t)(h + 3) * (Wp + 8) + (w + 3), v);
  }
}

// the same for a whole batch in ONE launch (blockIdx.y = image): a training step preprocesses 12 + 4 images and each launch, with its
// gap, sat at the very start of the step's critical path
#define PRE_MAX_IMGS 32
struct PreBatch {
  const void* src[PRE_MAX_IMGS];
  int H[PRE_MAX_IMGS], W[PRE_MAX_IMGS];
};
template <typename T>
__global__ __launch_bounds__(256) void preprocess_batch_bf16pad(PreBatch b, h16_t* __restrict__ dst, int Hp, int Wp, float m0, float m1,
                                                              float m2, float s0, float s1, float s2) {
  const int n = blockIdx.y, H = b.H[n], W = b.W[n];
  const T* __restrict__ src = (const T*)b.src[n];
  h16_t* out = dst + (size_t)n * (Hp + 6) * (Wp + 8) * 4;
  const size_t total = (size_t)Hp * Wp;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int h = (int)(i / Wp), w = (int)(i % Wp);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (h < H && w < W) {
      const size_t o = (size_t)h * W + w, hw = (size_t)H * W;
      v[0] = ((float)src[o] - m0) / s0;
      v[1] = ((float)src[hw + o] - m1) / s1;
      v[2] = ((float)src[2 * hw + o] - m2) / s2;
    }
    st4(out, (size_t)(h + 3) * (Wp + 8) + (w + 3), v);
  }
}

// ---------------------------------------------------------------------------------------------
// FrozenBatchNorm fold for every BN layer at once (D2 FrozenBatchNorm2d.forward [D2-recall]):
//   scale = w * rsqrt(var + eps);  shift = b - mean * scale
__global__ void frozenbn_fold_f32(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ mean,
                                  const float* __restrict__ var, float* __restrict__ scale, float* __restrict__ shift,
                                  int n, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float s = w[i] * (1.0f / sqrtf(var[i] + eps));
    scale[i] = s;
    shift[i] = b[i] - mean[i] * s;
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm(G groups) + ReLU (fcos/fcos.py:263-264, 283) over SEGMENTS of a [rows][C] matrix: a segment is
// one (image, FPN level) = a run of consecutive rows that shares statistics, so ONE launch normalises every
// level of a shared tower.  Work is cut into chunks of GN_ROWS rows that never straddle a segment.
//   stage 1  per chunk: partial sum / sumsq per group            -> part[chunk][G][2]
//   stage 2  per (segment, group): mean / rstd (double combine)   -> mean, rstd [seg][G]
//   stage 3  per chunk: y = relu((x - mean) * rstd * gamma + beta)
#define GN_ROWS 256
#define GN_MAX_SEG 160
struct GnSegs {
  int nseg;
  int rev;                     // apply passes walk the chunks LAST to FIRST: the pass before them (statistics / partial sums, or the conv
                               // that wrote x) ended on the tensor's tail, which is what the 256 MB Infinity Cache still holds
  int row0[GN_MAX_SEG + 1];    // first row of each segment (prefix sums)
  int chunk0[GN_MAX_SEG + 1];  // first chunk of each segment
};

__device__ __forceinline__ void gn_locate(const GnSegs& sg, int chunk, int& seg, int& r0, int& r1) {
  int s = 0;
  for (int i = 1; i < sg.nseg; ++i)
    if (chunk >= sg.chunk0[i]) s = i;
  seg = s;
  r0 = sg.row0[s] + (chunk - sg.chunk0[s]) * GN_ROWS;
  r1 = r0 + GN_ROWS;
  if (r1 > sg.row0[s + 1]) r1 = sg.row0[s + 1];
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_partial(GnSegs sg, const T* __restrict__ x, float* __restrict__ part, int C, int G) {
  // thread t owns channel quad c4 = t % C4 and row lane t / C4 (deterministic reduction order)
  __shared__ float red[2][256];
  int seg, r0, r1;
  gn_locate(sg, blockIdx.x, seg, r0, r1);
  const int C4 = C >> 2;
  const int cpg4 = (C / G) >> 2;  // channel quads per group
  const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4, RL = blockDim.x / C4;
  float s = 0.f, q = 0.f;
  int row = r0 + rl;
  // four independent loads in flight per thread, accumulated in row order (same arithmetic as the rolled loop: the reduction was
  // latency-bound at 2.5 TB/s with one dependent 8-byte load per iteration)
  for (; row + 3 * RL < r1; row += 4 * RL) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = ld4(x, (size_t)(row + u * RL) * C4 + c4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
      q += (v[u][0] * v[u][0] + v[u][1] * v[u][1]) + (v[u][2] * v[u][2] + v[u][3] * v[u][3]);
    }
  }
  for (; row < r1; row += RL) {
    const f32x4 v = ld4(x, (size_t)row * C4 + c4);
    s += (v[0] + v[1]) + (v[2] + v[3]);
    q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  float* out = part + (size_t)blockIdx.x * G * 2;
  for (int g = threadIdx.x; g < G; g += blockDim.x) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < RL; ++k)
      for (int j = 0; j < cpg4; ++j) {
        a += red[0][k * C4 + g * cpg4 + j];
        b += red[1][k * C4 + g * cpg4 + j];
      }
    out[g * 2] = a;
    out[g * 2 + 1] = b;
  }
}

// one block per segment: thread (g = t % G, lane = t / G) sums every (256/G)-th chunk, LDS-combined in order
__global__ __launch_bounds__(256) void gn_stats_final(GnSegs sg, const float* __restrict__ part, float* __restrict__ mean,
                                                    float* __restrict__ rstd, int G, int cpg, float eps) {
  __shared__ double red[2][256];
  const int seg = blockIdx.x;
  const int g = threadIdx.x % G, ln = threadIdx.x / G, L = blockDim.x / G;
  double s = 0.0, q = 0.0;
  for (int c = sg.chunk0[seg] + ln; c < sg.chunk0[seg + 1]; c += L) {
    const float* p = part + ((size_t)c * G + g) * 2;
    s += (double)p[0];
    q += (double)p[1];
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (ln == 0) {
    for (int k = 1; k < L; ++k) { s += red[0][k * G + g]; q += red[1][k * G + g]; }
    const double cnt = (double)(sg.row0[seg + 1] - sg.row0[seg]) * cpg;
    const double m = s / cnt;
    double var = q / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean[seg * G + g] = (float)m;
    rstd[seg * G + g] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// The same from the conv epilogue's partials (epilogue_rows gn_part: fp32 [ceil(rows / 32)][G][2] per 32-row block and group; 8 channels
// per group): one block per segment; the 32-row blocks that lie inside the segment come from `part32`, the (< 32) rows in front of
// the first and behind the last whole block - segments start at arbitrary rows - are summed here from x itself.  Thread (g = t % G,
// lane = t / G); double accumulation, fixed order.
__global__ __launch_bounds__(256) void gn_stats_final_p32(GnSegs sg, const float* __restrict__ part32, const h16_t* __restrict__ x,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int G, int C, float eps) {
  // grid (segments, G / 8): a workgroup owns 8 groups of one segment, thread (group, lane of 32) walks every 32nd block
  __shared__ double red[2][256];
  const int seg = blockIdx.x;
  const int gl = threadIdx.x & 7, ln = threadIdx.x >> 3, L = 32;
  const int g = blockIdx.y * 8 + gl;
  const int r0 = sg.row0[seg], r1 = sg.row0[seg + 1];
  int b0 = (r0 + 31) >> 5, b1 = r1 >> 5;      // whole blocks [b0, b1)
  if (b1 < b0) b1 = b0;                        // the segment lies inside one block
  double s = 0.0, q = 0.0;
  if (g < G) {
    for (int b = b0 + ln; b < b1; b += L) {
      const float2 p = *(const float2*)(part32 + ((size_t)b * G + g) * 2);
      s += (double)p.x;
      q += (double)p.y;
    }
    // edge rows: [r0, min(b0 * 32, r1)) and [max(b1 * 32, head end), r1)
    const int h1 = (b0 << 5) < r1 ? (b0 << 5) : r1;
    const int t0 = (b1 << 5) > h1 ? (b1 << 5) : h1;
    for (int pass = 0; pass < 2; ++pass) {
      const int e0 = pass == 0 ? r0 : t0, e1 = pass == 0 ? h1 : r1;
      for (int r = e0 + ln; r < e1; r += L) {
        const bf16x8_t v = *(const bf16x8_t*)(x + (size_t)r * C + g * 8);
        float fs = 0.f, fq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; fs += f; fq += f * f; }
        s += (double)fs;
        q += (double)fq;
      }
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (ln == 0 && g < G) {
    for (int k = 1; k < L; ++k) { s += red[0][k * 8 + gl]; q += red[1][k * 8 + gl]; }
    const double cnt = (double)(r1 - r0) * 8.0;
    const double m = s / cnt;
    double var = q / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean[seg * G + g] = (float)m;
    rstd[seg * G + g] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

__device__ __forceinline__ int gn_chunk_of_block(const GnSegs& sg) {
  const int nch = sg.chunk0[sg.nseg], b = (int)blockIdx.x;
  return (sg.rev && b < nch) ? nch - 1 - b : b;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_relu(GnSegs sg, const T* __restrict__ x, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, T* __restrict__ y, int C, int G, int relu,
                                                   unsigned* __restrict__ bits = nullptr) {
  // bits (optional; C % 32 == 0): the ReLU mask as a bit plane, bit (row * C + c) = y > 0 BEFORE the rounding to T (the reference's ReLU
  // follows an fp32 GroupNorm under autocast; a positive value below T's range still passes its gradient) - the plane the conv epilogues
  // read in place of a sign tensor (common.h EpiBits).  8 neighbouring lanes (32 channels of one row) assemble one 32-bit word.
  int seg, r0, r1;
  gn_locate(sg, gn_chunk_of_block(sg), seg, r0, r1);
  const int C4 = C >> 2, cpg = C / G;
  const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4, RL = blockDim.x / C4;
  const int g = (c4 * 4) / cpg;
  const float m = mean[seg * G + g], r = rstd[seg * G + g];
  const f32x4 ga = ((const f32x4*)gamma)[c4], be = ((const f32x4*)beta)[c4];
  auto apply = [&](const f32x4 v, size_t i) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t = (v[e] - m) * r * ga[e] + be[e];
      o[e] = relu ? fmaxf(t, 0.f) : t;
    }
    st4(y, i, o);
    if (bits) {   // (wave-uniform; the 8 lanes of a word run the same rows)
      unsigned b = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) b |= (o[e] > 0.f ? 1u : 0u) << e;   // the fp32 value's sign: the mask gn_bwd_partial / gn_bwd_apply recompute from x
      b <<= 4 * (threadIdx.x & 7);
      b |= __shfl_xor(b, 1, 64);
      b |= __shfl_xor(b, 2, 64);
      b |= __shfl_xor(b, 4, 64);
      if ((threadIdx.x & 7) == 0) bits[i >> 3] = b;
    }
  };
  int row = r0 + rl;
  for (; row + 3 * RL < r1; row += 4 * RL) {  // four independent row loads in flight per thread
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = ld4(x, (size_t)(row + u * RL) * C4 + c4);
#pragma unroll
    for (int u = 0; u < 4; ++u) apply(v[u], (size_t)(row + u * RL) * C4 + c4);
  }
  for (; row < r1; row += RL) apply(ld4(x, (size_t)row * C4 + c4), (size_t)row * C4 + c4);
}

// backward stage 1: per (chunk, c): A = sum g*xhat, B = sum g   with g = dy * (y > 0)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_partial(GnSegs sg, const T* __restrict__ dy, const T* __restrict__ y,
                                                    const T* __restrict__ x, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ part, int C, int G,
                                                    int relu) {
  __shared__ float red[2][256 * 4];
  int seg, r0, r1;
  gn_locate(sg, blockIdx.x, seg, r0, r1);
  const int C4 = C >> 2, cpg = C / G;
  const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4, RL = blockDim.x / C4;
  const int g = (c4 * 4) / cpg;
  const float m = mean[seg * G + g], r = rstd[seg * G + g];
  f32x4 A = {0.f, 0.f, 0.f, 0.f}, B = {0.f, 0.f, 0.f, 0.f};
  // ReLU mask: with beta given it is RECOMPUTED from x with the forward's exact expression (same sign as the stored y)
  // instead of re-reading y - one tensor read less in each backward pass
  const bool remask = relu && beta != nullptr;
  f32x4 ga = {0.f, 0.f, 0.f, 0.f}, be = {0.f, 0.f, 0.f, 0.f};
  if (remask) { ga = ((const f32x4*)gamma)[c4]; be = ((const f32x4*)beta)[c4]; }
  auto accumulate = [&](f32x4 gg, const f32x4 xx, size_t o) {
    if (remask) {
#pragma unroll
      for (int e = 0; e < 4; ++e) gg[e] = ((xx[e] - m) * r * ga[e] + be[e]) > 0.f ? gg[e] : 0.f;
    } else if (relu) {
      const f32x4 yy = ld4(y, o);
#pragma unroll
      for (int e = 0; e < 4; ++e) gg[e] = yy[e] > 0.f ? gg[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      A[e] += gg[e] * ((xx[e] - m) * r);
      B[e] += gg[e];
    }
  };
  int row = r0 + rl;
  for (; row + 3 * RL < r1; row += 4 * RL) {  // 8 independent loads in flight per thread, accumulated in row order
    f32x4 gv[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t o = (size_t)(row + u * RL) * C4 + c4;
      gv[u] = ld4(dy, o);
      xv[u] = ld4(x, o);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) accumulate(gv[u], xv[u], (size_t)(row + u * RL) * C4 + c4);
  }
  for (; row < r1; row += RL) {
    const size_t o = (size_t)row * C4 + c4;
    accumulate(ld4(dy, o), ld4(x, o), o);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[0][threadIdx.x * 4 + e] = A[e];
    red[1][threadIdx.x * 4 + e] = B[e];
  }
  __syncthreads();
  if (rl == 0) {
    for (int k = 1; k < RL; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        A[e] += red[0][(k * C4 + c4) * 4 + e];
        B[e] += red[1][(k * C4 + c4) * 4 + e];
      }
    float* out = part + ((size_t)blockIdx.x * C + c4 * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      out[e * 2] = A[e];
      out[e * 2 + 1] = B[e];
    }
  }
}

// backward stage 2: AB[seg][c][2] = sum over the segment's chunks; s1 = sum_c gamma*A, s2 = sum_c gamma*B per (seg, group).
// Grid (segments, C / 64): a workgroup owns 64 channels (whole groups: cpg divides 64) of one segment, thread (channel, lane) walks
// every 4th chunk with four loads in flight and the four lanes are combined in a fixed order - one block per segment with one walk per
// channel was pure load latency (78 us per launch on the backward's critical path for 0.3 MB of partials).
__global__ __launch_bounds__(256) void gn_bwd_reduce(GnSegs sg, const float* __restrict__ part, const float* __restrict__ gamma,
                                                   float* __restrict__ AB, float* __restrict__ s12, int C, int G) {
  __shared__ float ra[256], rb[256], sh[64][2];
  const int seg = blockIdx.x, cl = threadIdx.x & 63, ln = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl, cpg = C / G;
  const int k0 = sg.chunk0[seg], k1 = sg.chunk0[seg + 1];
  float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    for (int k = k0 + ln; k < k1; k += 16) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (k + 4 * q < k1) {
          const float2 v = *(const float2*)(part + ((size_t)(k + 4 * q) * C + c) * 2);
          a[q] += v.x;
          b[q] += v.y;
        }
    }
  }
  ra[threadIdx.x] = (a[0] + a[1]) + (a[2] + a[3]);
  rb[threadIdx.x] = (b[0] + b[1]) + (b[2] + b[3]);
  __syncthreads();
  if (ln == 0 && c < C) {
    const float sa = (ra[cl] + ra[cl + 64]) + (ra[cl + 128] + ra[cl + 192]);
    const float sb = (rb[cl] + rb[cl + 64]) + (rb[cl + 128] + rb[cl + 192]);
    AB[((size_t)seg * C + c) * 2] = sa;
    AB[((size_t)seg * C + c) * 2 + 1] = sb;
    sh[cl][0] = sa * gamma[c];
    sh[cl][1] = sb * gamma[c];
  }
  __syncthreads();
  const int gpb = 64 / cpg;   // groups per block
  if ((int)threadIdx.x < gpb && blockIdx.y * 64 + threadIdx.x * cpg < C) {
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < cpg; ++k) {
      s1 += sh[threadIdx.x * cpg + k][0];
      s2 += sh[threadIdx.x * cpg + k][1];
    }
    const int g = blockIdx.y * gpb + threadIdx.x;
    s12[(seg * G + g) * 2] = s1;
    s12[(seg * G + g) * 2 + 1] = s2;
  }
}

// The same from the dgrad epilogue's partials (common.h EpiBits::gnb_part: fp32 [ceil(rows / 64)][C][2] = {sum g, sum g * x} per 64-row
// block and channel, g = dy * mask as stored in `g`): A = sum g * xhat = rstd * (sum g x - mean * sum g), B = sum g.  The 64-row blocks
// that lie inside the segment come from part64; the rows in front of the first and behind the last whole block - segments start at
// arbitrary rows - are summed here from g and x.  Same grid and outputs as gn_bwd_reduce; double accumulation, fixed order.
__global__ __launch_bounds__(256) void gn_bwd_reduce_p64(GnSegs sg, const float* __restrict__ part64, const h16_t* __restrict__ gy,
                                                       const h16_t* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                       float* __restrict__ AB, float* __restrict__ s12, int C, int G) {
  __shared__ double r0s[256], r1s[256];
  __shared__ float sh[64][2];
  const int seg = blockIdx.x, cl = threadIdx.x & 63, ln = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl, cpg = C / G;
  const int r0 = sg.row0[seg], r1 = sg.row0[seg + 1];
  int b0 = (r0 + 63) >> 6, b1 = r1 >> 6;      // whole blocks [b0, b1)
  if (b1 < b0) b1 = b0;                        // the segment lies inside one block
  double t0 = 0.0, t1 = 0.0;
  if (c < C) {
    int k = b0 + ln;
    for (; k + 12 < b1; k += 16) {   // four loads in flight per thread
      float2 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *(const float2*)(part64 + ((size_t)(k + 4 * q) * C + c) * 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) { t0 += (double)v[q].x; t1 += (double)v[q].y; }
    }
    for (; k < b1; k += 4) {
      const float2 v = *(const float2*)(part64 + ((size_t)k * C + c) * 2);
      t0 += (double)v.x;
      t1 += (double)v.y;
    }
    const int h1 = (b0 << 6) < r1 ? (b0 << 6) : r1;
    const int e2 = (b1 << 6) > h1 ? (b1 << 6) : h1;
    for (int pass = 0; pass < 2; ++pass) {
      const int e0 = pass == 0 ? r0 : e2, e1 = pass == 0 ? h1 : r1;
      for (int r = e0 + ln; r < e1; r += 4) {
        const float gv = (float)gy[(size_t)r * C + c], xv = (float)x[(size_t)r * C + c];
        t0 += (double)gv;
        t1 += (double)(gv * xv);
      }
    }
  }
  r0s[threadIdx.x] = t0;
  r1s[threadIdx.x] = t1;
  __syncthreads();
  if (ln == 0 && c < C) {
    const double S0 = (r0s[cl] + r0s[cl + 64]) + (r0s[cl + 128] + r0s[cl + 192]);
    const double S1 = (r1s[cl] + r1s[cl + 64]) + (r1s[cl + 128] + r1s[cl + 192]);
    const int g = c / cpg;
    const float sa = (float)((double)rstd[seg * G + g] * (S1 - (double)mean[seg * G + g] * S0));
    const float sb = (float)S0;
    AB[((size_t)seg * C + c) * 2] = sa;
    AB[((size_t)seg * C + c) * 2 + 1] = sb;
    sh[cl][0] = sa * gamma[c];
    sh[cl][1] = sb * gamma[c];
  }
  __syncthreads();
  const int gpb = 64 / cpg;   // groups per block
  if ((int)threadIdx.x < gpb && blockIdx.y * 64 + threadIdx.x * cpg < C) {
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < cpg; ++k) {
      s1 += sh[threadIdx.x * cpg + k][0];
      s2 += sh[threadIdx.x * cpg + k][1];
    }
    const int g = blockIdx.y * gpb + threadIdx.x;
    s12[(seg * G + g) * 2] = s1;
    s12[(seg * G + g) * 2 + 1] = s2;
  }
}

// dgamma / dbeta += sums of AB over the segments.  Runs as the prologue of the first cdiv(C, 64) workgroups of gn_bwd_apply: a launch of
// its own sat, with its gap, between gn_bwd_reduce and gn_bwd_apply on the backward's critical path although only the optimizer reads it.
__device__ __forceinline__ void gn_bwd_param(const float* __restrict__ AB, float* __restrict__ dgamma, float* __restrict__ dbeta, int S, int C,
                                             int block) {
  // 64 channels x 4 segment parts per block (one thread per channel walked the S segments alone: 16 us of pure load latency);
  // the parts are combined in a fixed order
  __shared__ float ra[256], rb[256];
  const int c = block * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float a = 0.f, b = 0.f;
  if (c < C)
    for (int n = part; n < S; n += 4) {
      const float2 v = *(const float2*)(AB + ((size_t)n * C + c) * 2);
      a += v.x;
      b += v.y;
    }
  ra[threadIdx.x] = a;
  rb[threadIdx.x] = b;
  __syncthreads();
  if (part == 0 && c < C) {
    dgamma[c] += (ra[threadIdx.x] + ra[threadIdx.x + 64]) + (ra[threadIdx.x + 128] + ra[threadIdx.x + 192]);
    dbeta[c] += (rb[threadIdx.x] + rb[threadIdx.x + 64]) + (rb[threadIdx.x + 128] + rb[threadIdx.x + 192]);
  }
}

// backward stage 3: dx = rstd * (g*gamma - (s2 + xhat*s1)/cnt)
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply(GnSegs sg, const T* __restrict__ dy, const T* __restrict__ y,
                                                  const T* __restrict__ x, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, const float* __restrict__ s12,
                                                  T* __restrict__ dx, int C, int G, int relu, const float* __restrict__ AB,
                                                  float* __restrict__ dgamma, float* __restrict__ dbeta, int param_blocks,
                                                  float* __restrict__ colpart) {
  if ((int)blockIdx.x < param_blocks) gn_bwd_param(AB, dgamma, dbeta, sg.nseg, C, blockIdx.x);
  int seg, r0, r1;
  const int chunk = gn_chunk_of_block(sg);
  gn_locate(sg, chunk, seg, r0, r1);
  const int C4 = C >> 2, cpg = C / G;
  const int c4 = threadIdx.x % C4, rl = threadIdx.x / C4, RL = blockDim.x / C4;
  const int g = (c4 * 4) / cpg;
  const float m = mean[seg * G + g], r = rstd[seg * G + g];
  const float s1 = s12[(seg * G + g) * 2], s2 = s12[(seg * G + g) * 2 + 1];
  const float inv_cnt = 1.0f / ((float)(sg.row0[seg + 1] - sg.row0[seg]) * cpg);
  const f32x4 ga = ((const f32x4*)gamma)[c4];
  const bool remask = relu && beta != nullptr;
  f32x4 be = {0.f, 0.f, 0.f, 0.f};
  if (remask) be = ((const f32x4*)beta)[c4];
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};   // colpart: column sums of the rows this thread writes
  auto apply = [&](f32x4 gg, const f32x4 xx, size_t i) {
    if (remask) {
#pragma unroll
      for (int e = 0; e < 4; ++e) gg[e] = ((xx[e] - m) * r * ga[e] + be[e]) > 0.f ? gg[e] : 0.f;
    } else if (relu) {
      const f32x4 yy = ld4(y, i);
#pragma unroll
      for (int e = 0; e < 4; ++e) gg[e] = yy[e] > 0.f ? gg[e] : 0.f;
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xx[e] - m) * r;
      o[e] = r * (gg[e] * ga[e] - (s2 + xh * s1) * inv_cnt);
    }
    st4(dx, i, o);
    if (colpart) {
#pragma unroll
      for (int e = 0; e < 4; ++e) csum[e] += (float)(T)o[e];   // the value as stored (rounded to T)
    }
  };
  // 8 independent loads in flight per thread: inside the step this kernel shares the chip with the weight-gradient stream and gets a
  // fraction of the CUs, so the bandwidth one CU sustains decides its time (324 us per launch in the step against ~100 alone)
  int row = r0 + rl;
  for (; row + 3 * RL < r1; row += 4 * RL) {
    f32x4 gv[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t i = (size_t)(row + u * RL) * C4 + c4;
      gv[u] = ld4(dy, i);
      xv[u] = ld4(x, i);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) apply(gv[u], xv[u], (size_t)(row + u * RL) * C4 + c4);
  }
  for (; row < r1; row += RL) {
    const size_t i = (size_t)row * C4 + c4;
    apply(ld4(dy, i), ld4(x, i), i);
  }
  if (colpart) {   // colpart[chunk][c] = column sums of this chunk of dx, row lanes combined in a fixed order
    __shared__ float cred[256 * 4];
#pragma unroll
    for (int e = 0; e < 4; ++e) cred[threadIdx.x * 4 + e] = csum[e];
    __syncthreads();
    if (rl == 0 && chunk < sg.chunk0[sg.nseg]) {
      for (int k = 1; k < RL; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[e] += cred[(k * C4 + c4) * 4 + e];
      *(f32x4*)(colpart + (size_t)chunk * C + c4 * 4) = csum;
    }
  }
}

static const bool g_gn_reverse = [] { const char* v = getenv("UTV2_GN_REVERSE"); return !(v && v[0] == '0'); }();   // A/B switch, read once

static int gn_fill(GnSegs& sg, int nseg, const int* seg_rows) {
  sg.nseg = nseg;
  sg.rev = g_gn_reverse ? 1 : 0;
  int r = 0, c = 0;
  for (int s = 0; s < nseg; ++s) {
    sg.row0[s] = r;
    sg.chunk0[s] = c;
    r += seg_rows[s];
    c += (seg_rows[s] + GN_ROWS - 1) / GN_ROWS;
  }
  for (int s = nseg; s <= GN_MAX_SEG; ++s) { sg.row0[s] = r; sg.chunk0[s] = c; }
  return c;
}

static inline int grid_for(size_t n, int block = 256, int cap = 256 * 16) {
  size_t b = (n + block - 1) / block;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename T>
static void gn_fwd_launch(const GnSegs& sg, int chunks, int nseg, const void* x, const float* gamma, const float* beta, void* y,
                          float* mean, float* rstd, float* ws, int C, int G, float eps, int relu, hipStream_t stream) {
  hipLaunchKernelGGL(gn_stats_partial<T>, dim3(chunks), dim3(256), 0, stream, sg, (const T*)x, ws, C, G);
  hipLaunchKernelGGL(gn_stats_final, dim3(nseg), dim3(256), 0, stream, sg, (const float*)ws, mean, rstd, G, C / G, eps);
  hipLaunchKernelGGL(gn_apply_relu<T>, dim3(chunks), dim3(256), 0, stream, sg, (const T*)x, (const float*)mean, (const float*)rstd,
                     gamma, beta, (T*)y, C, G, relu);
}

template <typename T>
static void gn_bwd_launch(const GnSegs& sg, int chunks, int nseg, const void* dy, const void* y, const void* x, const float* mean,
                          const float* rstd, const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta,
                          float* ws, int C, int G, int relu, float* colpart, hipStream_t stream) {
  float* part = ws;
  float* AB = ws + (size_t)chunks * C * 2;
  float* s12 = AB + (size_t)nseg * C * 2;
  hipLaunchKernelGGL(gn_bwd_partial<T>, dim3(chunks), dim3(256), 0, stream, sg, (const T*)dy, (const T*)y, (const T*)x, mean, rstd,
                     gamma, beta, part, C, G, relu);
  hipLaunchKernelGGL(gn_bwd_reduce, dim3(nseg, cdiv(C, 64)), dim3(256), 0, stream, sg, (const float*)part, gamma, AB, s12, C, G);
  const int pb = cdiv(C, 64);   // workgroups that also run gn_bwd_param (a workgroup past the last chunk finds no rows)
  hipLaunchKernelGGL(gn_bwd_apply<T>, dim3(chunks > pb ? chunks : pb), dim3(256), 0, stream, sg, (const T*)dy, (const T*)y, (const T*)x, mean,
                     rstd, gamma, beta, (const float*)s12, (T*)dx, C, G, relu, (const float*)AB, dgamma, dbeta, pb, colpart);
}

extern "C" {

int utv2_ema_axpby(float* teacher, const float* student, int64_t n, double keep_rate, hipStream_t stream) {
  if (!teacher || !student || n < 0) return UTV2_EARG;
  if (n == 0) return UTV2_OK;
  if (((uintptr_t)teacher | (uintptr_t)student) & 15) return UTV2_EARG;
  const float a = (float)(1.0 - keep_rate), b = (float)keep_rate;
  hipLaunchKernelGGL(ema_axpby_f32, dim3(grid_for((size_t)n / 4)), dim3(256), 0, stream, teacher, student, (size_t)n / 4,
                     (size_t)n, a, b, (h16_t*)nullptr);
  return utv2_launch_status();
}

// the same + mirror16[i] = the library's 16-bit rounding of the new teacher[i] (8-byte aligned)
int utv2_ema_axpby_m16(float* teacher, const float* student, void* mirror16, int64_t n, double keep_rate, hipStream_t stream) {
  if (!teacher || !student || !mirror16 || n < 0) return UTV2_EARG;
  if (n == 0) return UTV2_OK;
  if ((((uintptr_t)teacher | (uintptr_t)student) & 15) || ((uintptr_t)mirror16 & 7)) return UTV2_EARG;
  const float a = (float)(1.0 - keep_rate), b = (float)keep_rate;
  hipLaunchKernelGGL(ema_axpby_f32, dim3(grid_for((size_t)n / 4)), dim3(256), 0, stream, teacher, student, (size_t)n / 4,
                     (size_t)n, a, b, (h16_t*)mirror16);
  return utv2_launch_status();
}

int utv2_sgd_momentum(float* param, float* grad, float* mom_buf, int64_t n, float lr, float momentum, float weight_decay,
                      float grad_scale, int zero_grad, hipStream_t stream) {
  if (!param || !grad || !mom_buf || n < 0) return UTV2_EARG;
  if (n == 0) return UTV2_OK;
  hipLaunchKernelGGL(sgd_momentum_f32, dim3(grid_for((size_t)n)), dim3(256), 0, stream, param, grad, mom_buf, (size_t)n, lr,
                     momentum, weight_decay, grad_scale, zero_grad, (h16_t*)nullptr);
  return utv2_launch_status();
}

// utv2_sgd_momentum / utv2_sgd_momentum_amp that also write mirror16[i] = the library's 16-bit rounding of the new param[i]
int utv2_sgd_momentum_m16(float* param, float* grad, float* mom_buf, void* mirror16, int64_t n, float lr, float momentum, float weight_decay,
                          float grad_scale, int zero_grad, hipStream_t stream) {
  if (!param || !grad || !mom_buf || !mirror16 || n < 0) return UTV2_EARG;
  if (n == 0) return UTV2_OK;
  hipLaunchKernelGGL(sgd_momentum_f32, dim3(grid_for((size_t)n)), dim3(256), 0, stream, param, grad, mom_buf, (size_t)n, lr,
                     momentum, weight_decay, grad_scale, zero_grad, (h16_t*)mirror16);
  return utv2_launch_status();
}

int utv2_sgd_momentum_amp_m16(float* param, const float* grad, float* mom_buf, void* mirror16, int64_t n, float lr, float momentum,
                              float weight_decay, float grad_scale, const float* state, hipStream_t stream) {
  if (!param || !grad || !mom_buf || !mirror16 || !state || n < 0) return UTV2_EARG;
  if (n == 0) return UTV2_OK;
  hipLaunchKernelGGL(sgd_momentum_amp_f32, dim3(grid_for((size_t)n)), dim3(256), 0, stream, param, grad, mom_buf, (size_t)n, lr, momentum,
                     weight_decay, grad_scale, state, (h16_t*)mirror16);
  return utv2_launch_status();
}

// state: device fp32[3] = {loss scale, found_inf flag, growth tracker} (see amp_found_inf_f32)
int utv2_amp_found_inf(const float* grad, int64_t n, float* state, hipStream_t stream) {
  if (!grad || !state || n < 0 || ((uintptr_t)grad & 15)) return UTV2_EARG;
  if (n == 0) return UTV2_OK;
  hipLaunchKernelGGL(amp_found_inf_f32, dim3(grid_for((size_t)n / 4 + 1)), dim3(256), 0, stream, grad, (size_t)n / 4, (size_t)n, state);
  return utv2_launch_status();
}

int utv2_sgd_momentum_amp(float* param, const float* grad, float* mom_buf, int64_t n, float lr, float momentum, float weight_decay,
                          float grad_scale, const float* state, hipStream_t stream) {
  if (!param || !grad || !mom_buf || !state || n < 0) return UTV2_EARG;
  if (n == 0) return UTV2_OK;
  hipLaunchKernelGGL(sgd_momentum_amp_f32, dim3(grid_for((size_t)n)), dim3(256), 0, stream, param, grad, mom_buf, (size_t)n, lr, momentum,
                     weight_decay, grad_scale, state, (h16_t*)nullptr);
  return utv2_launch_status();
}

int utv2_amp_update_scale(float* state, float growth_factor, float backoff_factor, int growth_interval, hipStream_t stream) {
  if (!state || growth_interval < 1) return UTV2_EARG;
  hipLaunchKernelGGL(amp_update_scale_f32, dim3(1), dim3(64), 0, stream, state, growth_factor, backoff_factor, growth_interval);
  return utv2_launch_status();
}

// out[M][C] = (y? relu-mask by y : 1) * dy * (scale? scale[c] : 1).  C % 4 == 0.  dy / y / out are `dtype`.
int utv2_relu_bwd_scale(const void* dy, const void* y, const float* scale, void* out, int64_t M, int C, int dtype,
                        hipStream_t stream) {
  if (!dy || !out || (C & 3) || (dtype != UTV2_F32 && dtype != UTV2_BF16)) return UTV2_EARG;
  const size_t n4 = (size_t)M * C / 4;
  if (n4 == 0) return UTV2_OK;
  if (dtype == UTV2_BF16)
    hipLaunchKernelGGL(relu_bwd_scale_k<h16_t>, dim3(grid_for(n4)), dim3(256), 0, stream, (const h16_t*)dy, (const h16_t*)y,
                       scale, (h16_t*)out, n4, C / 4);
  else
    hipLaunchKernelGGL(relu_bwd_scale_k<float>, dim3(grid_for(n4)), dim3(256), 0, stream, (const float*)dy, (const float*)y,
                       scale, (float*)out, n4, C / 4);
  return utv2_launch_status();
}

int utv2_add(const float* a, const float* b, float* out, int64_t n, hipStream_t stream) {
  if (!a || !b || !out) return UTV2_EARG;
  if (n == 0) return UTV2_OK;
  hipLaunchKernelGGL(add_f32, dim3(grid_for((size_t)n)), dim3(256), 0, stream, a, b, out, (size_t)n);
  return utv2_launch_status();
}

// x is `x_dtype`, y is `y_dtype` (the stem conv writes fp32, the bf16 activation pipeline starts here)
// dx[n][h][w][c] = sum of dpool over the windows whose (first) arg-max pixel (h, w) is; relu: times (x > 0).  x: the pool's INPUT
// (x_dtype), dpool / dx: g_dtype.
int utv2_maxpool3x3s2_bwd_nhwc(const void* x, int x_dtype, const void* dpool, void* dx, int g_dtype, int N, int H, int W, int C, int OH,
                               int OW, int relu, hipStream_t stream) {
  if (!x || !dpool || !dx || (C & 3)) return UTV2_EARG;
  const dim3 g(grid_for((size_t)N * H * W * C / 4, 256, 1 << 16)), b(256);
  if (x_dtype == UTV2_F32 && g_dtype == UTV2_F32)
    hipLaunchKernelGGL((maxpool3x3s2_bwd_nhwc_k<float, float>), g, b, 0, stream, (const float*)x, (const float*)dpool, (float*)dx, N, H, W, C / 4, OH, OW, relu);
  else if (x_dtype == UTV2_BF16 && g_dtype == UTV2_BF16)
    hipLaunchKernelGGL((maxpool3x3s2_bwd_nhwc_k<h16_t, h16_t>), g, b, 0, stream, (const h16_t*)x, (const h16_t*)dpool, (h16_t*)dx, N, H, W, C / 4, OH, OW, relu);
  else if (x_dtype == UTV2_BF16 && g_dtype == UTV2_F32)
    hipLaunchKernelGGL((maxpool3x3s2_bwd_nhwc_k<h16_t, float>), g, b, 0, stream, (const h16_t*)x, (const float*)dpool, (float*)dx, N, H, W, C / 4, OH, OW, relu);
  else
    return UTV2_EARG;
  return utv2_launch_status();
}

int utv2_maxpool3x3s2_nhwc(const void* x, int x_dtype, void* y, int y_dtype, int N, int H, int W, int C, int OH, int OW,
                           hipStream_t stream) {
  if (!x || !y || (C & 3)) return UTV2_EARG;
  const dim3 g(grid_for((size_t)N * OH * OW * C / 4, 256, 1 << 16)), b(256);
  if (x_dtype == UTV2_F32 && y_dtype == UTV2_F32)
    hipLaunchKernelGGL((maxpool3x3s2_nhwc_k<float, float>), g, b, 0, stream, (const float*)x, (float*)y, N, H, W, C / 4, OH, OW);
  else if (x_dtype == UTV2_F32 && y_dtype == UTV2_BF16)
    hipLaunchKernelGGL((maxpool3x3s2_nhwc_k<float, h16_t>), g, b, 0, stream, (const float*)x, (h16_t*)y, N, H, W, C / 4, OH, OW);
  else if (x_dtype == UTV2_BF16 && y_dtype == UTV2_BF16 && (C & 7) == 0 && (size_t)N * OH * OW * C / 8 < (1ull << 31) &&
           (size_t)N * H * W < (1ull << 31))
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_bf16x8_k, dim3(grid_for((size_t)N * OH * OW * C / 8, 256, 1 << 16)), b, 0, stream,
                       (const h16_t*)x, (h16_t*)y, N, H, W, C / 8, OH, OW);
  else if (x_dtype == UTV2_BF16 && y_dtype == UTV2_BF16)
    hipLaunchKernelGGL((maxpool3x3s2_nhwc_k<h16_t, h16_t>), g, b, 0, stream, (const h16_t*)x, (h16_t*)y, N, H, W, C / 4, OH, OW);
  else
    return UTV2_EARG;
  return utv2_launch_status();
}

int utv2_upsample2x_add_nhwc(const void* lateral, const void* top, void* out, int N, int H, int W, int C, int dtype,
                             hipStream_t stream) {
  if (!lateral || !top || !out || (C & 3) || (H & 1) || (W & 1)) return UTV2_EARG;
  const dim3 g(grid_for((size_t)N * H * W * C / 4, 256, 1 << 16)), b(256);
  if (dtype == UTV2_BF16)
    hipLaunchKernelGGL(upsample2x_add_nhwc_k<h16_t>, g, b, 0, stream, (const h16_t*)lateral, (const h16_t*)top, (h16_t*)out, N,
                       H, W, C / 4);
  else if (dtype == UTV2_F32)
    hipLaunchKernelGGL(upsample2x_add_nhwc_k<float>, g, b, 0, stream, (const float*)lateral, (const float*)top, (float*)out, N, H,
                       W, C / 4);
  else
    return UTV2_EARG;
  return utv2_launch_status();
}

int utv2_downsample2x_sum_nhwc(const void* g, void* dtop, int N, int TH, int TW, int C, int accumulate, int dtype,
                               hipStream_t stream) {
  if (!g || !dtop || (C & 3)) return UTV2_EARG;
  const dim3 gr(grid_for((size_t)N * TH * TW * C / 4, 256, 1 << 16)), b(256);
  if (dtype == UTV2_BF16)
    hipLaunchKernelGGL(downsample2x_sum_nhwc_k<h16_t>, gr, b, 0, stream, (const h16_t*)g, (h16_t*)dtop, N, TH, TW, C / 4,
                       accumulate);
  else if (dtype == UTV2_F32)
    hipLaunchKernelGGL(downsample2x_sum_nhwc_k<float>, gr, b, 0, stream, (const float*)g, (float*)dtop, N, TH, TW, C / 4, accumulate);
  else
    return UTV2_EARG;
  return utv2_launch_status();
}

// dst[n,2i,2j,:] = src[n,i,j,:], zero elsewhere.  TH = (H+1)/2, TW = (W+1)/2.
int utv2_zero_interleave2x_nhwc(const void* src, const void* mask, void* dst, int N, int H, int W, int C, int dtype, hipStream_t stream) {
  if (!src || !dst || (C & 3)) return UTV2_EARG;
  const int TH = (H + 1) / 2, TW = (W + 1) / 2;
  const dim3 g(grid_for((size_t)N * H * W * C / 4, 256, 1 << 16)), b(256);
  if (dtype == UTV2_BF16)
    hipLaunchKernelGGL(zero_interleave2x_nhwc_k<h16_t>, g, b, 0, stream, (const h16_t*)src, (const h16_t*)mask, (h16_t*)dst, N, H, W, TH, TW, C / 4);
  else if (dtype == UTV2_F32)
    hipLaunchKernelGGL(zero_interleave2x_nhwc_k<float>, g, b, 0, stream, (const float*)src, (const float*)mask, (float*)dst, N, H, W, TH, TW, C / 4);
  else
    return UTV2_EARG;
  return utv2_launch_status();
}

// 16-bit tensors, C % 8 == 0: dst = add + (even pixel ? (mask ? src : 0) : 0); mask as a 16-bit tensor OR as a bit plane (uint8 [N][H][W][C / 8]);
// mask, mask_bits and add are optional
int utv2_zero_interleave2x_add_nhwc(const void* src, const void* mask, const void* mask_bits, const void* add, void* dst, int N, int H, int W,
                                    int C, hipStream_t stream) {
  if (!src || !dst || (C & 7) || (mask && mask_bits)) return UTV2_EARG;
  const int TH = (H + 1) / 2, TW = (W + 1) / 2;
  const dim3 g(grid_for((size_t)N * H * W * C / 8, 256, 1 << 16)), b(256);
  hipLaunchKernelGGL(zero_interleave2x_add_h16_k, g, b, 0, stream, (const h16_t*)src, (const h16_t*)mask, (const unsigned char*)mask_bits,
                     (const h16_t*)add, (h16_t*)dst, N, H, W, TH, TW, C / 8);
  return utv2_launch_status();
}

// src: one image [3][H][W], uint8 (is_u8) or fp32; dst: one padded NHWC4 image [Hp][Wp][4]
int utv2_preprocess_image(const void* src, int is_u8, float* dst, int H, int W, int Hp, int Wp, const float* mean3_host,
                          const float* std3_host, hipStream_t stream) {
  if (!src || !dst || H > Hp || W > Wp) return UTV2_EARG;
  const float m0 = mean3_host[0], m1 = mean3_host[1], m2 = mean3_host[2];
  const float s0 = std3_host[0], s1 = std3_host[1], s2 = std3_host[2];
  const int g = grid_for((size_t)Hp * Wp, 256, 1 << 16);
  if (is_u8)
    hipLaunchKernelGGL((preprocess_chw_to_nhwc4<unsigned char>), dim3(g), dim3(256), 0, stream, (const unsigned char*)src,
                       dst, H, W, Hp, Wp, m0, m1, m2, s0, s1, s2);
  else
    hipLaunchKernelGGL((preprocess_chw_to_nhwc4<float>), dim3(g), dim3(256), 0, stream, (const float*)src, dst, H, W, Hp,
                       Wp, m0, m1, m2, s0, s1, s2);
  return utv2_launch_status();
}

// dst16: one image slot [Hp+6][Wp+8][4] bf16 of the zero-bordered stem input (see utv2_conv2d_stem_fwd_bf16)
int utv2_preprocess_image_bf16pad(const void* src, int is_u8, void* dst16, int H, int W, int Hp, int Wp, const float* mean3_host,
                                  const float* std3_host, hipStream_t stream) {
  if (!src || !dst16 || H > Hp || W > Wp) return UTV2_EARG;
  const float m0 = mean3_host[0], m1 = mean3_host[1], m2 = mean3_host[2];
  const float s0 = std3_host[0], s1 = std3_host[1], s2 = std3_host[2];
  const int g = grid_for((size_t)Hp * Wp, 256, 1 << 16);
  if (is_u8)
    hipLaunchKernelGGL((preprocess_chw_to_nhwc4_bf16pad<unsigned char>), dim3(g), dim3(256), 0, stream, (const unsigned char*)src,
                       (h16_t*)dst16, H, W, Hp, Wp, m0, m1, m2, s0, s1, s2);
  else
    hipLaunchKernelGGL((preprocess_chw_to_nhwc4_bf16pad<float>), dim3(g), dim3(256), 0, stream, (const float*)src, (h16_t*)dst16,
                       H, W, Hp, Wp, m0, m1, m2, s0, s1, s2);
  return utv2_launch_status();
}

// N <= 32 images (device pointers in src_host, sizes in H_host / W_host) into the N slots of dst16 = bf16 [N][Hp+6][Wp+8][4]
int utv2_preprocess_images_bf16pad(const void* const* src_host, int is_u8, void* dst16, const int* H_host, const int* W_host, int N,
                                   int Hp, int Wp, const float* mean3_host, const float* std3_host, hipStream_t stream) {
  if (!src_host || !dst16 || !H_host || !W_host || !mean3_host || !std3_host || N < 1 || N > PRE_MAX_IMGS) return UTV2_EARG;
  PreBatch b;
  for (int i = 0; i < PRE_MAX_IMGS; ++i) {
    b.src[i] = i < N ? src_host[i] : nullptr;
    b.H[i] = i < N ? H_host[i] : 0;
    b.W[i] = i < N ? W_host[i] : 0;
    if (i < N && (!src_host[i] || H_host[i] > Hp || W_host[i] > Wp || H_host[i] < 0 || W_host[i] < 0)) return UTV2_EARG;
  }
  const float m0 = mean3_host[0], m1 = mean3_host[1], m2 = mean3_host[2];
  const float s0 = std3_host[0], s1 = std3_host[1], s2 = std3_host[2];
  const int g = grid_for((size_t)Hp * Wp, 256, 1 << 12);
  if (is_u8)
    hipLaunchKernelGGL((preprocess_batch_bf16pad<unsigned char>), dim3(g, N), dim3(256), 0, stream, b, (h16_t*)dst16, Hp, Wp, m0, m1, m2, s0,
                       s1, s2);
  else
    hipLaunchKernelGGL((preprocess_batch_bf16pad<float>), dim3(g, N), dim3(256), 0, stream, b, (h16_t*)dst16, Hp, Wp, m0, m1, m2, s0, s1, s2);
  return utv2_launch_status();
}

int utv2_frozenbn_fold(const float* w, const float* b, const float* mean, const float* var, float* scale, float* shift,
                       int n, float eps, hipStream_t stream) {
  if (!w || !b || !mean || !var || !scale || !shift) return UTV2_EARG;
  if (n == 0) return UTV2_OK;
  hipLaunchKernelGGL(frozenbn_fold_f32, dim3(cdiv(n, 256)), dim3(256), 0, stream, w, b, mean, var, scale, shift, n, eps);
  return utv2_launch_status();
}

// Segmented API: seg_rows_host = host int[nseg] (rows of each (image, level) segment, consecutive in memory).
int64_t utv2_groupnorm_seg_workspace_floats(int nseg, const int* seg_rows_host, int C) {
  int64_t chunks = 0;
  for (int s = 0; s < nseg; ++s) chunks += (seg_rows_host[s] + GN_ROWS - 1) / GN_ROWS;
  return chunks * C * 2 + (int64_t)nseg * C * 2 + (int64_t)nseg * C;
}

static int gn_check(int nseg, int C, int G) {
  return !(nseg < 1 || nseg > GN_MAX_SEG || (C & 3) || ((C / G) & 3) || C / 4 > 256 || (256 % (C / 4)) || (256 % G) || (C % G) || (64 % (C / G)));
}

// x,y: [rows][C] of `dtype`; mean,rstd: fp32 [nseg][G] (saved for backward).  Statistics and arithmetic are fp32.
int utv2_groupnorm_relu_seg_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                float* ws, int nseg, const int* seg_rows_host, int C, int G, float eps, int relu, int dtype,
                                hipStream_t stream) {
  if (!x || !y || !mean || !rstd || !ws || !gn_check(nseg, C, G) || (dtype != UTV2_F32 && dtype != UTV2_BF16)) return UTV2_EARG;
  GnSegs sg;
  const int chunks = gn_fill(sg, nseg, seg_rows_host);
  if (dtype == UTV2_BF16) gn_fwd_launch<h16_t>(sg, chunks, nseg, x, gamma, beta, y, mean, rstd, ws, C, G, eps, relu, stream);
  else gn_fwd_launch<float>(sg, chunks, nseg, x, gamma, beta, y, mean, rstd, ws, C, G, eps, relu, stream);
  return utv2_launch_status();
}

// bf16 x with 8 channels per group whose statistics partials the producing conv left in part32 (utv2_conv2d_ml_fwd_bf16_g gn_part:
// fp32 [ceil(rows / 32)][G][2]): statistics from the partials (+ the segment-edge rows read from x), then the apply pass - one tensor
// pass less than utv2_groupnorm_relu_seg_fwd.
// relu_bits (optional; C % 32 == 0): [rows * C / 8] bytes, bit (row * C + c) = y > 0 in fp32 (before the 16-bit rounding; the mask
// utv2_groupnorm_relu_seg_bwd recomputes from x when it is given beta) - the plane utv2_conv2d_ml_fwd_bf16_gnb reads
int utv2_groupnorm_relu_seg_fwd_p32b(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                     const float* part32, int nseg, const int* seg_rows_host, int C, int G, float eps, int relu,
                                     void* relu_bits, hipStream_t stream) {
  if (!x || !y || !mean || !rstd || !part32 || !gn_check(nseg, C, G) || C != 8 * G || (relu_bits && (C & 31))) return UTV2_EARG;
  GnSegs sg;
  const int chunks = gn_fill(sg, nseg, seg_rows_host);
  hipLaunchKernelGGL(gn_stats_final_p32, dim3(nseg, cdiv(G, 8)), dim3(256), 0, stream, sg, part32, (const h16_t*)x, mean, rstd, G, C, eps);
  hipLaunchKernelGGL(gn_apply_relu<h16_t>, dim3(chunks), dim3(256), 0, stream, sg, (const h16_t*)x, (const float*)mean, (const float*)rstd,
                     gamma, beta, (h16_t*)y, C, G, relu, (unsigned*)relu_bits);
  return utv2_launch_status();
}

int utv2_groupnorm_relu_seg_fwd_p32(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                    const float* part32, int nseg, const int* seg_rows_host, int C, int G, float eps, int relu,
                                    hipStream_t stream) {
  return utv2_groupnorm_relu_seg_fwd_p32b(x, gamma, beta, y, mean, rstd, part32, nseg, seg_rows_host, C, G, eps, relu, nullptr, stream);
}

// GroupNorm (+ ReLU) backward whose first reduction the producing dgrad's epilogue already made (utv2_conv2d_ml_fwd_bf16_gnb): g = the
// gradient with the ReLU mask applied (16-bit [rows][C]), part64 = that conv's gnb_part.  Two launches (segment sums; apply) instead of
// three, and one pass over g and x instead of two.  Outputs as utv2_groupnorm_relu_seg_bwd_colsum; ws: utv2_groupnorm_seg_workspace_floats.
int utv2_groupnorm_seg_bwd_p64(const void* g, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx,
                               float* dgamma, float* dbeta, float* ws, int nseg, const int* seg_rows_host, int C, int G,
                               const float* part64, float* colsum_part, hipStream_t stream) {
  if (!g || !x || !dx || !ws || !part64 || !mean || !rstd || !gamma || !gn_check(nseg, C, G)) return UTV2_EARG;
  GnSegs sg;
  const int chunks = gn_fill(sg, nseg, seg_rows_host);
  float* AB = ws;
  float* s12 = AB + (size_t)nseg * C * 2;
  hipLaunchKernelGGL(gn_bwd_reduce_p64, dim3(nseg, cdiv(C, 64)), dim3(256), 0, stream, sg, part64, (const h16_t*)g, (const h16_t*)x, mean,
                     rstd, gamma, AB, s12, C, G);
  const int pb = cdiv(C, 64);
  hipLaunchKernelGGL(gn_bwd_apply<h16_t>, dim3(chunks > pb ? chunks : pb), dim3(256), 0, stream, sg, (const h16_t*)g, (const h16_t*)nullptr,
                     (const h16_t*)x, mean, rstd, gamma, (const float*)nullptr, (const float*)s12, (h16_t*)dx, C, G, 0, (const float*)AB, dgamma,
                     dbeta, pb, colsum_part);
  return utv2_launch_status();
}

// dx written (`dtype`, like dy / y / x); dgamma/dbeta (fp32) accumulated (+=).  The ReLU mask comes from y, or - when
// beta is given - is recomputed from x (y may then be null).
int64_t utv2_groupnorm_seg_chunks(int nseg, const int* seg_rows_host) {
  int64_t chunks = 0;
  for (int s = 0; s < nseg; ++s) chunks += (seg_rows_host[s] + GN_ROWS - 1) / GN_ROWS;
  return chunks;
}

// colsum_part (optional, fp32 [utv2_groupnorm_seg_chunks()][C]): per-chunk column sums of dx as stored - the partial sums of the bias
// gradient of the convolution that produced x (its dY is this dx), for free while dx is in registers.
int utv2_groupnorm_relu_seg_bwd_colsum(const void* dy, const void* y, const void* x, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, float* ws,
                                       int nseg, const int* seg_rows_host, int C, int G, int relu, int dtype, float* colsum_part,
                                       hipStream_t stream) {
  if (!dy || !x || !dx || !ws || !gn_check(nseg, C, G) || (dtype != UTV2_F32 && dtype != UTV2_BF16) || (relu && !y && !beta))
    return UTV2_EARG;
  GnSegs sg;
  const int chunks = gn_fill(sg, nseg, seg_rows_host);
  if (dtype == UTV2_BF16)
    gn_bwd_launch<h16_t>(sg, chunks, nseg, dy, y, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, ws, C, G, relu, colsum_part, stream);
  else gn_bwd_launch<float>(sg, chunks, nseg, dy, y, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, ws, C, G, relu, colsum_part, stream);
  return utv2_launch_status();
}

int utv2_groupnorm_relu_seg_bwd(const void* dy, const void* y, const void* x, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, float* ws,
                                int nseg, const int* seg_rows_host, int C, int G, int relu, int dtype, hipStream_t stream) {
  return utv2_groupnorm_relu_seg_bwd_colsum(dy, y, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, ws, nseg, seg_rows_host, C, G, relu, dtype,
                                            nullptr, stream);
}

// Dense [N][HW][C] convenience forms (one segment per image).
int64_t utv2_groupnorm_workspace_floats(int N, int HW, int C) {
  int rows[GN_MAX_SEG];
  if (N > GN_MAX_SEG) return -1;
  for (int i = 0; i < N; ++i) rows[i] = HW;
  return utv2_groupnorm_seg_workspace_floats(N, rows, C);
}

int utv2_groupnorm_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            float* ws, int N, int HW, int C, int G, float eps, int relu, hipStream_t stream) {
  int rows[GN_MAX_SEG];
  if (N > GN_MAX_SEG) return UTV2_EARG;
  for (int i = 0; i < N; ++i) rows[i] = HW;
  return utv2_groupnorm_relu_seg_fwd(x, gamma, beta, y, mean, rstd, ws, N, rows, C, G, eps, relu, UTV2_F32, stream);
}

int utv2_groupnorm_relu_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                            const float* gamma, float* dx, float* dgamma, float* dbeta, float* ws, int N, int HW, int C,
                            int G, int relu, hipStream_t stream) {
  int rows[GN_MAX_SEG];
  if (N > GN_MAX_SEG) return UTV2_EARG;
  for (int i = 0; i < N; ++i) rows[i] = HW;
  return utv2_groupnorm_relu_seg_bwd(dy, y, x, mean, rstd, gamma, nullptr, dx, dgamma, dbeta, ws, N, rows, C, G, relu, UTV2_F32, stream);
}

}  // extern "C"
