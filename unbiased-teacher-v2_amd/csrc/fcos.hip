// FCOS-specific kernels of the UTv2 step (gfx950): target assignment, fused sigmoid-focal,
// fused positive-location losses (Integral + centerness BCE + GIoU + NLL + pseudo "teacher better"
// L1), ranking keys and box decode for the teacher's pseudo-label NMS.
//
// Dense layout used by every kernel here ("level-first", the ordering the reference builds with
// _transpose / cat at ubteacher/modeling/fcos/fcos_outputs.py:634-647,227-290): row index
//   p = N * level_off[l] + n * HW_l + hw,   level_off[l] = sum_{l'<l} HW_l'
// so each level's NHWC head output [N][H_l][W_l][C] is one contiguous row range and no
// permute/cat is ever materialised.
#include "common.h"

#define MAX_LEVELS 8

struct LevelTable {
  int num_levels;
  int H[MAX_LEVELS], W[MAX_LEVELS], stride[MAX_LEVELS];
  int off[MAX_LEVELS + 1];  // prefix sums of H*W
  float soi_lo[MAX_LEVELS], soi_hi[MAX_LEVELS];
};

__device__ __forceinline__ int level_of(const LevelTable& t, int loc) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < MAX_LEVELS; ++i)
    if (i < t.num_levels && loc >= t.off[i]) l = i;
  return l;
}

// ---------------------------------------------------------------------------------------------
// Target assignment  (fcos_outputs.py:649-698 _get_ground_truth, :772-906
// compute_targets_for_locations with CENTER_SAMPLE False, ignore_near False).
// gt arrays are padded to MAXG per image with a validity byte (the thresholded pseudo-label
// mask, pseudo_generator.py:85, stays on the device; compaction order == index order).
// One thread per (image, location); gts of the image staged in LDS.
//   labels[p]      class id, num_classes for background, -1 for "location dropped" (the keep_locations filter of :310-311: the
//                  image has no gt and drop_empty & 1 (:804-815), or drop_empty & 2 = ignore_near (:841-848) and the location
//                  lies inside a box but in no box's centre-sampling region)
//   reg_targets[p] ltrb / stride of the min-area matching gt (gt 0 when none, as the reference)
//   bvars[p]       teacher reg_pred_std of the matched gt (99999 for background, 0 if no gt)
//   gt_inds[p]     matched gt slot (or -1 when the image has no gt)
#define TG_MAXG 256
__global__ __launch_bounds__(256) void fcos_targets_kernel(LevelTable lt, int N, int MAXG, const float* __restrict__ gt_boxes,
                                                         const int* __restrict__ gt_classes,
                                                         const unsigned char* __restrict__ gt_valid,
                                                         const float* __restrict__ gt_std, int num_classes, int drop_empty,
                                                         float center_radius, const unsigned char* __restrict__ img_active,
                                                         int gt_img0, int gt_imgs, int* __restrict__ labels,
                                                         float* __restrict__ reg_targets, float* __restrict__ bvars,
                                                         int* __restrict__ gt_inds) {
  __shared__ float sb[TG_MAXG][4];
  __shared__ float sarea[TG_MAXG];
  __shared__ int sidx[TG_MAXG];
  __shared__ int scount;
  const int n = blockIdx.y;
  const int L = lt.off[lt.num_levels];
  // the gt arrays describe images [gt_img0, gt_img0 + gt_imgs) of the batch; the others belong to another loss branch
  const int gi = n - gt_img0;
  const bool in_range = gi >= 0 && gi < gt_imgs;
  if (threadIdx.x == 0) {
    // ordered compaction of the valid gts (serial: MAXG <= 256)
    int c = 0;
    if (in_range)
      for (int g = 0; g < MAXG; ++g)
        if (gt_valid[gi * MAXG + g]) sidx[c++] = g;
    scount = c;
  }
  __syncthreads();
  const int G = scount;
  for (int k = threadIdx.x; k < G; k += blockDim.x) {
    const float* b = gt_boxes + ((size_t)gi * MAXG + sidx[k]) * 4;
    sb[k][0] = b[0]; sb[k][1] = b[1]; sb[k][2] = b[2]; sb[k][3] = b[3];
    sarea[k] = (b[2] - b[0]) * (b[3] - b[1]);
  }
  __syncthreads();
  const int loc = blockIdx.x * blockDim.x + threadIdx.x;
  if (loc >= L) return;
  const int l = level_of(lt, loc);
  const int hw = loc - lt.off[l];
  const int HWl = lt.H[l] * lt.W[l];
  const int y = hw / lt.W[l], x = hw - y * lt.W[l];
  const float s = (float)lt.stride[l];
  const float xs = (float)(x * lt.stride[l]) + (float)(lt.stride[l] / 2);
  const float ys = (float)(y * lt.stride[l]) + (float)(lt.stride[l] / 2);
  const size_t p = (size_t)N * lt.off[l] + (size_t)n * HWl + hw;
  // an inactive image (it belongs to the other loss branch of a fused student pass) is ignored like a dropped one
  const bool inactive = !in_range || (img_active != nullptr && !img_active[n]);
  if (G == 0 || inactive) {
    labels[p] = ((drop_empty & 1) || inactive) ? -1 : num_classes;
    gt_inds[p] = -1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { reg_targets[p * 4 + e] = 0.f; bvars[p * 4 + e] = 0.f; }
    return;
  }
  const float INF = 100000000.f;
  float best = INF;
  int bi = 0;
  // CENTER_SAMPLE (get_sample_region, fcos_outputs.py:700-770): positives only inside the radius*stride square around the box
  // centre, clipped to the box; the reference returns an all-False mask when the first box's centre x is 0
  const bool cs = center_radius > 0.f;
  const float sr = s * center_radius;
  const bool cs_none = cs && (sb[0][0] + sb[0][2]) * 0.5f * (float)L == 0.f;
  bool inside_any = false, sampled_any = false;
  for (int k = 0; k < G; ++k) {
    const float lft = xs - sb[k][0], top = ys - sb[k][1], rgt = sb[k][2] - xs, bot = sb[k][3] - ys;
    float mn = fminf(fminf(lft, top), fminf(rgt, bot));
    inside_any |= mn > 0.f;
    const float mx = fmaxf(fmaxf(lft, top), fmaxf(rgt, bot));
    if (cs) {
      const float cx = (sb[k][0] + sb[k][2]) * 0.5f, cy = (sb[k][1] + sb[k][3]) * 0.5f;
      const float xmin = cx - sr, ymin = cy - sr, xmax = cx + sr, ymax = cy + sr;
      const float x0 = xmin > sb[k][0] ? xmin : sb[k][0], y0 = ymin > sb[k][1] ? ymin : sb[k][1];
      const float x1 = xmax > sb[k][2] ? sb[k][2] : xmax, y1 = ymax > sb[k][3] ? sb[k][3] : ymax;
      mn = cs_none ? -1.f : fminf(fminf(xs - x0, ys - y0), fminf(x1 - xs, y1 - ys));
    }
    sampled_any |= mn > 0.f;
    float a = sarea[k];
    if (!(mn > 0.f)) a = INF;
    if (!(mx >= lt.soi_lo[l] && mx <= lt.soi_hi[l])) a = INF;
    if (a < best) { best = a; bi = k; }
  }
  const bool bg = (best == INF);
  const int g = sidx[bi];
  const bool ignored = (drop_empty & 2) && inside_any && !sampled_any;
  labels[p] = ignored ? -1 : bg ? num_classes : gt_classes[gi * MAXG + g];
  gt_inds[p] = g;
  reg_targets[p * 4 + 0] = (xs - sb[bi][0]) / s;
  reg_targets[p * 4 + 1] = (ys - sb[bi][1]) / s;
  reg_targets[p * 4 + 2] = (sb[bi][2] - xs) / s;
  reg_targets[p * 4 + 3] = (sb[bi][3] - ys) / s;
#pragma unroll
  for (int e = 0; e < 4; ++e) bvars[p * 4 + e] = bg ? 99999.0f : (gt_std ? gt_std[((size_t)gi * MAXG + g) * 4 + e] : 0.f);
}

// ---------------------------------------------------------------------------------------------
// Sigmoid focal loss (fvcore sigmoid_focal_loss_jit [fvcore-recall], called at
// fcos_outputs.py:329-335,619-625) with the one-hot target built on the fly from labels.
//   p = sigmoid(x); ce = max(x,0) - x*t + log1p(exp(-|x|)); p_t = p*t + (1-p)(1-t)
//   loss = ce * (1-p_t)^gamma * (alpha*t + (1-alpha)(1-t))
// fwd: deterministic two-stage sum -> partial[gridDim.x];  rows with label < 0 are skipped.
// One exponential serves both the sigmoid and the softplus: e = exp(-|x|) in (0, 1], p = 1 / (1 + e) or e / (1 + e), log1p(e) = log(1 + e)
// (its series below 1e-2, where 1 + e loses digits).  Hardware exp2 / log2 / reciprocal (1 ulp each): ~25 instructions per element
// instead of ~100 through the libm calls - the two focal kernels of a step were 3x off the HBM roofline on the serial loss tail.
__device__ __forceinline__ float focal_term(float x, float t, float alpha, float gamma, float* dldx) {
  const float e = __expf(-fabsf(x));
  const float r = __frcp_rn(1.f + e);
  const float p = x >= 0.f ? r : e * r;
  const float l1p = e < 1e-2f ? e * (1.f - e * (0.5f - e * (1.f / 3.f))) : __logf(1.f + e);   // below 1e-2 the rounding of 1 + e would cost log() digits
  const float ce = fmaxf(x, 0.f) - x * t + l1p;
  const float pt = p * t + (1.f - p) * (1.f - t);
  const float om = 1.f - pt;
  const float mod = (gamma == 2.f) ? om * om : powf(om, gamma);
  const float at = alpha * t + (1.f - alpha) * (1.f - t);
  if (dldx) {
    // dL/dx = a_t (2t-1) (1-pt)^g [ g*pt*log(pt) - (1-pt) ],  log(pt) = -ce
    *dldx = at * (2.f * t - 1.f) * mod * (gamma * pt * (-ce) - om);
  }
  return ce * mod * at;
}

__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ logits, const int* __restrict__ labels, size_t P,
                                                      int C, float alpha, float gamma, float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  if ((C & 3) == 0 && P * (size_t)C < (1ull << 32)) {  // a quad of logits per thread: one 16-byte load, one 32-bit division per quad
    const unsigned C4 = (unsigned)C >> 2, total4 = (unsigned)(P * (size_t)C4);
    for (unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += gridDim.x * blockDim.x) {
      const unsigned row = i4 / C4;
      const int c = (int)(i4 - row * C4) * 4;
      const int lab = labels[row];
      if (lab < 0) continue;
      const f32x4 v = ((const f32x4*)logits)[i4];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc += focal_term(v[e], lab == c + e ? 1.f : 0.f, alpha, gamma, nullptr);
    }
  } else {
    const size_t total = P * (size_t)C;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const size_t row = i / C;
      const int c = (int)(i - row * C);
      const int lab = labels[row];
      if (lab < 0) continue;
      acc += focal_term(logits[i], lab == c ? 1.f : 0.f, alpha, gamma, nullptr);
    }
  }
  const float s = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// dlogits = coef[0] * dL/dx   (coef = upstream grad / num_pos_avg, a device scalar: no host sync)
// gscale (optional device scalar): a further factor of the gradient (the upstream gradient of a node that owns several of these launches).
// accumulate: rows with a label < 0 (ignored locations / images of another loss branch) are left untouched and the others are ADDED to
// dlogits - the second, third, ... loss branch of a fused pass writes into the first one's gradient instead of into a tensor of its own
// that an elementwise add pass would then have to fold in.
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ logits, const int* __restrict__ labels, size_t P,
                                                      int C, float alpha, float gamma, const float* __restrict__ coef,
                                                      float* __restrict__ dlogits, const float* __restrict__ gscale, int accumulate) {
  const float k = coef[0] * (gscale ? gscale[0] : 1.f);
  if ((C & 3) == 0 && P * (size_t)C < (1ull << 32)) {
    const unsigned C4 = (unsigned)C >> 2, total4 = (unsigned)(P * (size_t)C4);
    for (unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += gridDim.x * blockDim.x) {
      const unsigned row = i4 / C4;
      const int c = (int)(i4 - row * C4) * 4;
      const int lab = labels[row];
      if (accumulate && lab < 0) continue;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (accumulate) g = ((const f32x4*)dlogits)[i4];
      if (lab >= 0) {
        const f32x4 v = ((const f32x4*)logits)[i4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float ge;
          focal_term(v[e], lab == c + e ? 1.f : 0.f, alpha, gamma, &ge);
          g[e] += ge * k;
        }
      }
      ((f32x4*)dlogits)[i4] = g;
    }
    return;
  }
  const size_t total = P * (size_t)C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const size_t row = i / C;
    const int c = (int)(i - row * C);
    const int lab = labels[row];
    if (accumulate && lab < 0) continue;
    float g = 0.f;
    if (lab >= 0) {
      focal_term(logits[i], lab == c ? 1.f : 0.f, alpha, gamma, &g);
      g *= k;
    }
    dlogits[i] = accumulate ? dlogits[i] + g : g;
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int n, int ncols,
                                                           float* __restrict__ out) {
  // out[c] = sum_r partial[r*ncols + c]: block c, thread t adds rows t, t+256, ... in double, then a fixed-order
  // LDS tree - deterministic, and ~n/256 dependent loads deep instead of n.
  __shared__ double red[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int r = threadIdx.x; r < n; r += 256) s += (double)partial[(size_t)r * ncols + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = (float)red[0];
}

// ---------------------------------------------------------------------------------------------
// Fused positive-location terms (fcos_outputs.py:340-416 supervised; :514-590 pseudo;
// Integral :44-77; compute_ctrness_targets :80-88; compute_iou_targets :91-129;
// IOULoss giou layers/iou_loss.py:26-76; NLLoss layers/kl_loss.py:69-105).
// box rows: [reg logits 4*(R+1) | std 4 | ctr 1 | pad], row stride BS floats.
// Sums produced (LT_NSUM columns):
//   0 n_pos  1 sum ctr_t  2 sum bce(ctr)  3 sum (1-giou)*ctr_t  4 sum nll_i*iou_i
//   5 n_sel  6 sum_sel |d - t|
#define LT_NSUM 8
struct LocTerms {
  float d[4];      // Integral ltrb
  float ctr_t, iou, giou_l, nll;
};

template <int R1>
__device__ __forceinline__ void integral4(const float* __restrict__ z, float* d, float (*prob)[R1]) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < R1; ++j) m = fmaxf(m, z[b * R1 + j]);
    float s = 0.f, e[R1];
#pragma unroll
    for (int j = 0; j < R1; ++j) { e[j] = expf(z[b * R1 + j] - m); s += e[j]; }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < R1; ++j) {
      const float pj = e[j] / s;
      if (prob) prob[b][j] = pj;
      acc += pj * (float)j;
    }
    d[b] = acc;
  }
}

__device__ __forceinline__ float min_grad(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }  // d min(a,b)/da
__device__ __forceinline__ float max_grad(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }

// giou loss (1 - giou) on ltrb with +1 smoothing on the IoU only; optionally its gradient wrt d
// loc_type (MODEL.FCOS.LOC_LOSS_TYPE, iou_loss.py:64-69): 0 giou (1 - giou), 1 iou (-log iou), 2 linear_iou (1 - iou)
__device__ __forceinline__ float giou_ltrb(const float* d, const float* t, float* iou_out, float* grad, int loc_type = 0) {
  const float ta = (t[0] + t[2]) * (t[1] + t[3]);
  const float pw = d[0] + d[2], ph = d[1] + d[3];
  const float pa = pw * ph;
  const float wi = fminf(d[0], t[0]) + fminf(d[2], t[2]);
  const float hi = fminf(d[3], t[3]) + fminf(d[1], t[1]);
  const float gw = fmaxf(d[0], t[0]) + fmaxf(d[2], t[2]);
  const float gh = fmaxf(d[3], t[3]) + fmaxf(d[1], t[1]);
  const float ac = gw * gh;
  const float I = wi * hi;
  const float U = ta + pa - I;
  const float iou = (I + 1.f) / (U + 1.f);
  const float giou = iou - (ac - U) / ac;
  if (iou_out) *iou_out = iou;
  if (grad) {
    // index 0: left, 1: top, 2: right, 3: bottom
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const bool horiz = (b == 0 || b == 2);
      const float dpa = horiz ? ph : pw;
      const float dI = horiz ? min_grad(d[b], t[b]) * hi : min_grad(d[b], t[b]) * wi;
      const float dac = horiz ? max_grad(d[b], t[b]) * gh : max_grad(d[b], t[b]) * gw;
      const float dU = dpa - dI;
      const float diou = (dI * (U + 1.f) - (I + 1.f) * dU) / ((U + 1.f) * (U + 1.f));
      // giou = iou - 1 + U/ac
      const float dgiou = diou + dU / ac - U * dac / (ac * ac);
      // three plain assignments, not one nested ?: - hipcc (ROCm 7.2) mis-folded `t == 1 ? -diou / iou : -diou` into the un-divided
      // numerator for t == 2 (caught by tests/test_fcos_kernels_gpu.py::test_supervised_loss_variants_vs_reference_golden[loclinear])
      float g = -diou;
      if (loc_type == 1) g = g / iou;
      if (loc_type == 0) g = -dgiou;
      grad[b] = g;
    }
  }
  return loc_type == 0 ? 1.f - giou : (loc_type == 1 ? -logf(iou) : 1.f - iou);
}

// flags of the positive-location kernels (config-reachable variants; every shipped YAML uses 0)
#define LT_QUALITY_IOU 1  // MODEL.FCOS.QUALITY_EST "iou": the centerness target is IoU(pred.detach, target) (fcos_outputs.py:355-359)
#define LT_KLLOSS 2       // MODEL.FCOS.KL_LOSS_TYPE "klloss": column 4 = sum_b exp(-std_b)*smoothL1_1(d_b - t_b) + std_b/2 (kl_loss.py:11-66)
#define LT_LOC_SHIFT 2    // bits 2-3: loc_type
#define LT_KL_WCTR 16     // with LT_KLLOSS: each positive's KL term is weighted by its centerness / quality target (MODEL.FCOS.LOC_FUN_ALL
                          // "weight_ctr_sum" / "weight_ctr_mean", kl_loss.py:52-58) instead of 1 ("sum" / "mean")

template <int R1>
__global__ __launch_bounds__(128) void fcos_loc_fwd_kernel(const int* __restrict__ labels, const float* __restrict__ box, int BS,
                                                         const float* __restrict__ reg_targets, const float* __restrict__ bvars,
                                                         size_t P, int num_classes, float ts_better, float ts_cert, int flags,
                                                         float* __restrict__ partial) {
  __shared__ float red[2];
  float acc[LT_NSUM];
#pragma unroll
  for (int k = 0; k < LT_NSUM; ++k) acc[k] = 0.f;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < P; i += stride) {
    const int lab = labels[i];
    if (lab < 0 || lab == num_classes) continue;
    const float* row = box + i * BS;
    float d[4], t[4];
    integral4<R1>(row, d, (float(*)[R1]) nullptr);
#pragma unroll
    for (int b = 0; b < 4; ++b) t[b] = reg_targets[i * 4 + b];
    float ctr_t = sqrtf((fminf(t[0], t[2]) / fmaxf(t[0], t[2])) * (fminf(t[1], t[3]) / fmaxf(t[1], t[3])));
    float iou;
    const float gl = giou_ltrb(d, t, &iou, nullptr, (flags >> LT_LOC_SHIFT) & 3);
    if (flags & LT_QUALITY_IOU) ctr_t = iou;
    float nll = 0.f;
    const float* sp = row + 4 * R1;
    if (flags & LT_KLLOSS) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float n = fabsf(d[b] - t[b]);
        nll += expf(-sp[b]) * (n < 1.f ? 0.5f * n * n : n - 0.5f) + 0.5f * sp[b];
      }
      iou = (flags & LT_KL_WCTR) ? ctr_t : 1.f;  // KLLoss ignores the IoU weight; LOC_FUN_ALL weight_ctr_*: the centerness target
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float sg = 1.f / (1.f + expf(-sp[b]));
        const float sq = sg * sg;
        const float df = t[b] - d[b];
        nll += (df * df) / (2.f * sq) + 0.5f * logf(sq);
      }
      nll += 2.f * logf(2.f * 3.14159265358979323846f);
    }
    const float c = row[4 * R1 + 4];
    const float bce = fmaxf(c, 0.f) - c * ctr_t + log1pf(expf(-fabsf(c)));
    acc[0] += 1.f;
    acc[1] += ctr_t;
    acc[2] += bce;
    acc[3] += gl * ctr_t;
    acc[4] += nll * iou;
    if (bvars) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float cs = 1.f - 1.f / (1.f + expf(-sp[b]));
        const float ct = 1.f - 1.f / (1.f + expf(-bvars[i * 4 + b]));
        if (ct > ts_cert && ct > cs + ts_better) {
          acc[5] += 1.f;
          acc[6] += fabsf(d[b] - t[b]);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < LT_NSUM; ++k) {
    const float s = block_reduce_sum(acc[k], red);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * LT_NSUM + k] = s;
  }
}

// coef: device floats [c_bce, c_giou, c_nll, c_l1]: the fully normalised upstream factors
//   d(total)/d(sum bce), d/d(sum giou*ctr), d/d(sum nll*iou), d/d(sum_sel |d-t|)
// writes the whole gradient row (zeros for background / pad channels).
template <int R1>
__global__ __launch_bounds__(128) void fcos_loc_bwd_kernel(const int* __restrict__ labels, const float* __restrict__ box, int BS,
                                                         const float* __restrict__ reg_targets, const float* __restrict__ bvars,
                                                         size_t P, int num_classes, float ts_better, float ts_cert, int flags,
                                                         const float* __restrict__ coef, float* __restrict__ dbox,
                                                         const float* __restrict__ gscale, int coef8, int accumulate) {
  // coef8: coef = d total / d sums[8] of the forward (the factors sit at [2], [3], [4], [6]); gscale: optional further device factor;
  // accumulate: only the positive rows are touched, their gradient ADDED to dbox (see focal_bwd_kernel)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float gs = gscale ? gscale[0] : 1.f;
  const float c_bce = coef[coef8 ? 2 : 0] * gs, c_giou = coef[coef8 ? 3 : 1] * gs, c_nll = coef[coef8 ? 4 : 2] * gs,
              c_l1 = coef[coef8 ? 6 : 3] * gs;
  for (; i < P; i += stride) {
    const int lab = labels[i];
    float* grow = dbox + i * BS;
    if (lab < 0 || lab == num_classes) {
      if (!accumulate)
        for (int k = 0; k < BS; k += 4) *(f32x4*)(grow + k) = f32x4{0.f, 0.f, 0.f, 0.f};
      continue;
    }
    const float* row = box + i * BS;
    float d[4], t[4], prob[4][R1];
    integral4<R1>(row, d, prob);
#pragma unroll
    for (int b = 0; b < 4; ++b) t[b] = reg_targets[i * 4 + b];
    float ctr_t = sqrtf((fminf(t[0], t[2]) / fmaxf(t[0], t[2])) * (fminf(t[1], t[3]) / fmaxf(t[1], t[3])));
    float iou, gg[4];
    giou_ltrb(d, t, &iou, gg, (flags >> LT_LOC_SHIFT) & 3);
    if (flags & LT_QUALITY_IOU) ctr_t = iou;  // detached target
    const float* sp = row + 4 * R1;
    float dd[4], ds[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float sg = 1.f / (1.f + expf(-sp[b]));
      const float df = t[b] - d[b];
      if (flags & LT_KLLOSS) {
        // kl_b = exp(-s) * sl1(n) + s/2, n = |d - t|: d/dd = exp(-s) * min(n, 1) * sign(d - t); d/ds = 1/2 - exp(-s) * sl1(n)
        const float n = fabsf(df), es = expf(-sp[b]);
        const float sgn = df < 0.f ? 1.f : (df > 0.f ? -1.f : 0.f);
        const float wk = (flags & LT_KL_WCTR) ? ctr_t : 1.f;   // a detached target
        dd[b] = c_giou * ctr_t * gg[b] + c_nll * wk * es * fminf(n, 1.f) * sgn;
        ds[b] = c_nll * wk * (0.5f - es * (n < 1.f ? 0.5f * n * n : n - 0.5f));
      } else {
        // nll_b = df^2/(2 sg^2) + log(sg);  d/dd = -df/sg^2 ; d/ds = (1-sg) * (1 - df^2/sg^2)
        dd[b] = c_giou * ctr_t * gg[b] + c_nll * iou * (-df / (sg * sg));
        ds[b] = c_nll * iou * (1.f - sg) * (1.f - (df * df) / (sg * sg));
      }
      if (bvars) {
        const float cs = 1.f - sg;
        const float ct = 1.f - 1.f / (1.f + expf(-bvars[i * 4 + b]));
        if (ct > ts_cert && ct > cs + ts_better) {
          const float e = d[b] - t[b];
          dd[b] += c_l1 * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
        }
      }
    }
    const float c = row[4 * R1 + 4];
    const float dctr = c_bce * (1.f / (1.f + expf(-c)) - ctr_t);
    if (accumulate) {
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int j = 0; j < R1; ++j) grow[b * R1 + j] += dd[b] * prob[b][j] * ((float)j - d[b]);
#pragma unroll
      for (int b = 0; b < 4; ++b) grow[4 * R1 + b] += ds[b];
      grow[4 * R1 + 4] += dctr;
      continue;
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int j = 0; j < R1; ++j) grow[b * R1 + j] = dd[b] * prob[b][j] * ((float)j - d[b]);
#pragma unroll
    for (int b = 0; b < 4; ++b) grow[4 * R1 + b] = ds[b];
    grow[4 * R1 + 4] = dctr;
    for (int k = 4 * R1 + 5; k < BS; ++k) grow[k] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Ranking keys for the pre-NMS top-k (fcos_outputs.py:1146-1195,1238-1241).
// key = (float bits of ranking score) << 32 | (0xFFFFFFFF - flat index) for candidates
// (sigmoid(logit) > thr), else -1: descending int64 order == (score desc, index asc).
// method: 0 cls, 1 cls_n_ctr, 2 ctr, 3 cls_n_loc
__global__ __launch_bounds__(256) void fcos_rank_keys_kernel(const float* __restrict__ logits, const float* __restrict__ box, int BS,
                                                           int R4, int HW, int C, float thr, int method,
                                                           long long* __restrict__ keys, size_t key_stride) {
  // grid.y = image; rows of this (level, image): [HW][C]
  const int n = blockIdx.y;
  const size_t total = (size_t)HW * C;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const size_t hw = i / C;
    const size_t row = (size_t)n * HW + hw;
    const float p = 1.f / (1.f + expf(-logits[row * C + (i - hw * C)]));
    long long key = -1;
    if (p > thr) {
      float r = p;
      const float* br = box + row * BS;
      if (method == 1) r = p * (1.f / (1.f + expf(-br[R4 + 4])));
      else if (method == 2) r = 1.f / (1.f + expf(-br[R4 + 4]));
      else if (method == 3) {
        float m = 0.f;
#pragma unroll
        for (int b = 0; b < 4; ++b) m += 1.f - 1.f / (1.f + expf(-br[R4 + b]));
        r = p * (m / 4.f);
      }
      key = ((long long)__float_as_uint(r) << 32) | (long long)(0xFFFFFFFFu - (unsigned)i);
    }
    keys[(size_t)n * key_stride + i] = key;
  }
}

// Decode the selected candidates of one level (fcos_outputs.py:1093-1104,1258-1296):
// in : topkeys [N][K] (from the keys above, sorted or not; -1 = empty slot)
// out (slot s = level_slot0 + k of image n, MAXC slots per image):
//   boxes[4] scores cls(int) loc[2] ctr cls_conf std[4] level valid
template <int R1>
__global__ __launch_bounds__(128) void fcos_decode_kernel(const long long* __restrict__ topkeys, int K, const float* __restrict__ logits,
                                                        const float* __restrict__ box, int BS, int HW, int Wl, int C, int stride,
                                                        int level, int method, int MAXC, int slot0, float* __restrict__ oboxes,
                                                        float* __restrict__ oscores, int* __restrict__ ocls, float* __restrict__ oloc,
                                                        float* __restrict__ octr, float* __restrict__ oconf, float* __restrict__ ostd,
                                                        int* __restrict__ olevel, unsigned char* __restrict__ ovalid) {
  const int n = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const size_t s = (size_t)n * MAXC + slot0 + k;
  const long long key = topkeys[(size_t)n * K + k];
  if (key < 0) {
    ovalid[s] = 0;
    oscores[s] = -1.f;
    ocls[s] = 0;
    olevel[s] = level;
#pragma unroll
    for (int e = 0; e < 4; ++e) { oboxes[s * 4 + e] = 0.f; ostd[s * 4 + e] = 0.f; }
    oloc[s * 2] = 0.f; oloc[s * 2 + 1] = 0.f; octr[s] = 0.f; oconf[s] = 0.f;
    return;
  }
  const unsigned flat = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFll);
  const float rank = __uint_as_float((unsigned)(key >> 32));
  const int hw = flat / C, c = flat - hw * C;
  const size_t row = (size_t)n * HW + hw;
  const float* br = box + row * BS;
  float d[4];
  integral4<R1>(br, d, (float(*)[R1]) nullptr);
  const int y = hw / Wl, x = hw - y * Wl;
  const float lx = (float)(x * stride) + (float)(stride / 2), ly = (float)(y * stride) + (float)(stride / 2);
  const float fs = (float)stride;
  oboxes[s * 4 + 0] = lx - d[0] * fs;
  oboxes[s * 4 + 1] = ly - d[1] * fs;
  oboxes[s * 4 + 2] = lx + d[2] * fs;
  oboxes[s * 4 + 3] = ly + d[3] * fs;
  oscores[s] = (method == 1 || method == 3) ? sqrtf(rank) : rank;
  ocls[s] = c;
  oloc[s * 2] = lx; oloc[s * 2 + 1] = ly;
  octr[s] = 1.f / (1.f + expf(-br[4 * R1 + 4]));
  oconf[s] = 1.f / (1.f + expf(-logits[row * C + c]));
#pragma unroll
  for (int e = 0; e < 4; ++e) ostd[s * 4 + e] = br[4 * R1 + e];
  olevel[s] = level;
  ovalid[s] = 1;
}

// in-place y[:, 0:ncols] *= s[0] on rows of stride BS (Scale layer, fcos/fcos.py:22-28,356-357)
__global__ __launch_bounds__(256) void scale_cols_kernel(float* __restrict__ y, size_t rows, int BS, int ncols, const float* __restrict__ s) {
  const size_t total = rows * (size_t)ncols;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float k = s[0];
  for (; i < total; i += stride) {
    const size_t r = i / ncols;
    const int c = (int)(i - r * ncols);
    y[r * BS + c] *= k;
  }
}

// backward of the above: g[:, 0:ncols] *= s ; partial[b] = sum g_in * y_post   (ds = sum / s)
__global__ __launch_bounds__(256) void scale_cols_bwd_kernel(float* __restrict__ g, const float* __restrict__ ypost, size_t rows, int BS,
                                                           int ncols, const float* __restrict__ s, float* __restrict__ partial) {
  __shared__ float red[4];
  const size_t total = rows * (size_t)ncols;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float k = s[0];
  float acc = 0.f;
  for (; i < total; i += stride) {
    const size_t r = i / ncols;
    const int c = (int)(i - r * ncols);
    const float gv = g[r * BS + c];
    acc += gv * ypost[r * BS + c];
    g[r * BS + c] = gv * k;
  }
  const float t = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// The Scale layers of ALL FPN levels of a level-first matrix in one launch (fcos/fcos.py:22-28,306-340: one learnable scalar per level):
// rows [row0[l], row0[l + 1]) belong to level l; blockIdx.y = level.
#define SC_MAXL 8
struct ScaleLevels {
  long long row0[SC_MAXL + 1];
  const float* s[SC_MAXL];
  float* sgrad[SC_MAXL];
  int nlev;
};

__global__ __launch_bounds__(256) void scale_cols_ml_kernel(float* __restrict__ y, ScaleLevels L, int BS, int ncols) {
  const int l = blockIdx.y;
  float* yl = y + (size_t)L.row0[l] * BS;
  const size_t total = (size_t)(L.row0[l + 1] - L.row0[l]) * (size_t)ncols;
  const float k = L.s[l][0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / ncols;
    const int c = (int)(i - r * ncols);
    yl[r * BS + c] *= k;
  }
}

// backward: g[:, 0:ncols] *= s_l ; partial[l][b] = sum g_in * y_post over the block's share of level l.
// VEC4 (row stride and ncols multiples of 4, 16-byte aligned base): one float4 per thread and pass - contiguous 16-byte accesses over the
// whole row (the groups at and beyond ncols are skipped), 32-bit index arithmetic; the scalar form walked ncols-element rows with a
// 64-bit division per element on 256 blocks per level (136 us per launch on the student batch: 86 MB at 1.3 TB/s).
template <bool VEC4>
__global__ __launch_bounds__(256) void scale_cols_bwd_ml_kernel(float* __restrict__ g, const float* __restrict__ ypost, ScaleLevels L, int BS,
                                                              int ncols, float* __restrict__ partial) {
  __shared__ float red[4];
  const int l = blockIdx.y;
  float* gl = g + (size_t)L.row0[l] * BS;
  const float* yl = ypost + (size_t)L.row0[l] * BS;
  const float k = L.s[l][0];
  float acc = 0.f;
  if constexpr (VEC4) {
    const unsigned q = (unsigned)BS >> 2, qn = (unsigned)ncols >> 2;                 // float4 groups per row / of them scaled
    const unsigned total = (unsigned)(L.row0[l + 1] - L.row0[l]) * q;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
      if (i % q >= qn) continue;
      f32x4 gv = *(const f32x4*)(gl + (size_t)i * 4);
      const f32x4 yv = *(const f32x4*)(yl + (size_t)i * 4);
      acc += ((gv[0] * yv[0] + gv[1] * yv[1]) + (gv[2] * yv[2] + gv[3] * yv[3]));
      gv[0] *= k; gv[1] *= k; gv[2] *= k; gv[3] *= k;
      *(f32x4*)(gl + (size_t)i * 4) = gv;
    }
  } else {
    const size_t total = (size_t)(L.row0[l + 1] - L.row0[l]) * (size_t)ncols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
      const size_t r = i / ncols;
      const int c = (int)(i - r * ncols);
      const float gv = gl[r * BS + c];
      acc += gv * yl[r * BS + c];
      gl[r * BS + c] = gv * k;
    }
  }
  const float t = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) partial[(size_t)l * gridDim.x + blockIdx.x] = t;
}

// The same backward with the result written as the zero-padded 16-bit matrix the prediction conv's dgrad / weight gradient read
// (out16 [rows][cpad], cpad >= BS): out[:, :ncols] = g * s_l, out[:, ncols:BS] = g, out[:, BS:] = 0; g itself is NOT modified.  One pass
// instead of three (clone of the incoming gradient + in-place scale + pad / convert): 224 MB instead of 568 MB on the student batch.
__global__ __launch_bounds__(256) void scale_cols_bwd_ml_pad16_kernel(const float* __restrict__ g, const float* __restrict__ ypost, ScaleLevels L,
                                                                    int BS, int ncols, float* __restrict__ partial, h16_t* __restrict__ out,
                                                                    int cpad) {
  __shared__ float red[4];
  const int l = blockIdx.y;
  const float* gl = g + (size_t)L.row0[l] * BS;
  const float* yl = ypost + (size_t)L.row0[l] * BS;
  h16_t* ol = out + (size_t)L.row0[l] * cpad;
  const float k = L.s[l][0];
  float acc = 0.f;
  const unsigned q = (unsigned)BS >> 2, qn = (unsigned)ncols >> 2, qp = (unsigned)cpad >> 2;
  const unsigned total = (unsigned)(L.row0[l + 1] - L.row0[l]) * qp;
  typedef h16_t h16x4 __attribute__((ext_vector_type(4)));
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned r = i / qp, c4 = i - r * qp;
    f32x4 gv = {0.f, 0.f, 0.f, 0.f};
    if (c4 < q) {
      gv = *(const f32x4*)(gl + ((size_t)r * q + c4) * 4);
      if (c4 < qn) {
        const f32x4 yv = *(const f32x4*)(yl + ((size_t)r * q + c4) * 4);
        acc += ((gv[0] * yv[0] + gv[1] * yv[1]) + (gv[2] * yv[2] + gv[3] * yv[3]));
        gv[0] *= k; gv[1] *= k; gv[2] *= k; gv[3] *= k;
      }
    }
    h16x4 o = {(h16_t)gv[0], (h16_t)gv[1], (h16_t)gv[2], (h16_t)gv[3]};
    *(h16x4*)(ol + (size_t)i * 4) = o;
  }
  const float t = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) partial[(size_t)l * gridDim.x + blockIdx.x] = t;
}

// one block per level: d s_l += (sum of the level's partials, fixed order) / s_l
__global__ __launch_bounds__(256) void scale_cols_bwd_ml_final(ScaleLevels L, const float* __restrict__ partial, int nb) {
  __shared__ float red[4];
  const int l = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) acc += partial[(size_t)l * nb + i];
  const float t = block_reduce_sum(acc, red);
  if (threadIdx.x == 0) L.sgrad[l][0] += t / L.s[l][0];
}

static LevelTable make_table(int num_levels, const int* H, const int* W, const int* strides, const float* soi) {
  LevelTable t;
  t.num_levels = num_levels;
  int off = 0;
  for (int l = 0; l < MAX_LEVELS; ++l) {
    if (l < num_levels) {
      t.H[l] = H[l]; t.W[l] = W[l]; t.stride[l] = strides[l];
      t.soi_lo[l] = soi ? soi[2 * l] : 0.f; t.soi_hi[l] = soi ? soi[2 * l + 1] : 0.f;
      t.off[l] = off;
      off += H[l] * W[l];
    } else {
      t.H[l] = t.W[l] = t.stride[l] = 0; t.soi_lo[l] = t.soi_hi[l] = 0.f; t.off[l] = off;
    }
  }
  t.off[num_levels] = off;
  for (int l = num_levels + 1; l <= MAX_LEVELS; ++l) t.off[l] = off;
  return t;
}

// ---------------------------------------------------------------------------------------------
// Scalar tail of the FCOS losses of one fused student pass (fcos_outputs.py:317-321,340-416 supervised; :447-631 pseudo; the loss
// weighting of engine/trainer.py:396-417): from the raw sums of the loss kernels to the six normalised losses, their weighted total and
// d(total)/d(every raw sum) in ONE single-thread launch (as ATen scalar ops this was ~60 forward + ~70 backward launches per step, with
// the GPU idle between them).  Normalisers carry no gradient (they are counts / target sums).
//   flags: 1 = KL_LOSS on the supervised branch, 2 = KL type "klloss" (mean over positives x 4), 4 = UNIFY_CTRCLS, 8 = tsbetter pseudo loc
//   wmul / wdiv [6]: the weighted loss k enters the total as value * wmul[k] / wdiv[k]; k = cls, loc, ctr, cls_pseudo, ctr_pseudo, loc_pseudo
//   rec [8] = {cls, loc, ctr, cls_pseudo, ctr_pseudo, loc_pseudo, teacher_better_student, total}
//   coef [26] = d total / d {focal_sup[1], sums_sup[8], focal_cls[1], sums_cls[8], sums_reg[8]}
struct FcosCombineArgs {
  const float* focal_sup;
  const float* sums_sup;
  const float* focal_cls;
  const float* sums_cls;
  const float* sums_reg;
  const float* norm;   // optional [6]: all-reduced (n_pos, sum ctr) of the sup / cls / reg branches
  float world, kl_weight;
  int flags;
  float wmul[6], wdiv[6];
};

__global__ void fcos_loss_combine_kernel(FcosCombineArgs a, float* __restrict__ rec, float* __restrict__ coef) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i < 26; ++i) coef[i] = 0.f;
  const bool kl = a.flags & 1, kl4 = a.flags & 2, unify = a.flags & 4, tsb = a.flags & 8;
  float g[6];
  for (int k = 0; k < 6; ++k) g[k] = a.wmul[k] / a.wdiv[k];
  const float* S = a.sums_sup;
  const float npa_s = fmaxf((a.norm ? a.norm[0] : S[0]) / a.world, 1.f), den_s = fmaxf((a.norm ? a.norm[1] : S[1]) / a.world, 1e-6f);
  const float cls = a.focal_sup[0] / npa_s, ctr = S[2] / npa_s;
  float loc = S[3] / den_s;
  coef[0] = g[0] / npa_s;
  coef[1 + 2] = g[2] / npa_s;
  coef[1 + 3] = g[1] / den_s;
  if (kl) {
    const float n = fmaxf(S[0], 1.f), dn = kl4 ? 4.f * n : n, w = a.kl_weight;
    loc = w * (w * (S[4] / dn)) + loc;
    coef[1 + 4] = g[1] * (w * w) / dn;
  }
  const float* C = a.sums_cls;
  const float npa_c = fmaxf((a.norm ? a.norm[2] : C[0]) / a.world, 1.f);
  const float cls_p = a.focal_cls[0] / npa_c;
  float ctr_p = C[2] / npa_c;
  coef[9] = g[3] / npa_c;
  if (unify) ctr_p = ctr_p * 0.f;
  else coef[10 + 2] = g[4] / npa_c;
  const float* R = a.sums_reg;
  float loc_p, tbs = 0.f;
  if (tsb) {
    const float d = fmaxf(R[5], 1.f);
    loc_p = R[6] / d;
    tbs = R[5];
    coef[18 + 6] = g[5] / d;
  } else {
    const float n = fmaxf(R[0], 1.f), dn = kl4 ? 4.f * n : n;
    loc_p = a.kl_weight * (R[4] / dn);
    coef[18 + 4] = g[5] * a.kl_weight / dn;
  }
  rec[0] = cls; rec[1] = loc; rec[2] = ctr; rec[3] = cls_p; rec[4] = ctr_p; rec[5] = loc_p; rec[6] = tbs;
  const float v[6] = {cls, loc, ctr, cls_p, ctr_p, loc_p};
  float total = 0.f;
  for (int k = 0; k < 6; ++k) total += v[k] * a.wmul[k] / a.wdiv[k];   // the trainer's `record * lambda / (lambda + 1)`, summed in key order
  rec[7] = total;
}

extern "C" {

// H,W,strides: host int[num_levels]; soi: host float[2*num_levels] (lo,hi per level).
int utv2_fcos_targets_range(int num_levels, const int* H, const int* W, const int* strides, const float* soi, int N, int MAXG,
                            const float* gt_boxes, const int* gt_classes, const unsigned char* gt_valid, const float* gt_std,
                            int gt_img0, int gt_imgs, int num_classes, int drop_empty, float center_radius,
                            const unsigned char* img_active, int* labels, float* reg_targets, float* bvars, int* gt_inds,
                            hipStream_t stream) {
  if (num_levels < 1 || num_levels > MAX_LEVELS || MAXG > TG_MAXG || MAXG < 1 || !gt_boxes || !gt_classes || !gt_valid ||
      !labels || !reg_targets || !bvars || !gt_inds || gt_img0 < 0 || gt_imgs < 0 || gt_img0 + gt_imgs > N)
    return UTV2_EARG;
  LevelTable t = make_table(num_levels, H, W, strides, soi);
  const int L = t.off[num_levels];
  hipLaunchKernelGGL(fcos_targets_kernel, dim3(cdiv(L, 256), N), dim3(256), 0, stream, t, N, MAXG, gt_boxes, gt_classes,
                     gt_valid, gt_std, num_classes, drop_empty, center_radius, img_active, gt_img0, gt_imgs, labels, reg_targets, bvars,
                     gt_inds);
  return utv2_launch_status();
}

int utv2_fcos_targets(int num_levels, const int* H, const int* W, const int* strides, const float* soi, int N, int MAXG,
                      const float* gt_boxes, const int* gt_classes, const unsigned char* gt_valid, const float* gt_std,
                      int num_classes, int drop_empty, float center_radius, const unsigned char* img_active, int* labels,
                      float* reg_targets, float* bvars, int* gt_inds, hipStream_t stream) {
  return utv2_fcos_targets_range(num_levels, H, W, strides, soi, N, MAXG, gt_boxes, gt_classes, gt_valid, gt_std, 0, N, num_classes,
                                 drop_empty, center_radius, img_active, labels, reg_targets, bvars, gt_inds, stream);
}

#define FOCAL_BLOCKS 1024
// loss_sum[0] = sum over rows with label >= 0 and all C classes.  ws: >= FOCAL_BLOCKS floats.
int utv2_sigmoid_focal_fwd(const float* logits, const int* labels, int64_t P, int C, float alpha, float gamma,
                           float* loss_sum, float* ws, hipStream_t stream) {
  if (!logits || !labels || !loss_sum || !ws) return UTV2_EARG;
  hipLaunchKernelGGL(focal_fwd_kernel, dim3(FOCAL_BLOCKS), dim3(256), 0, stream, logits, labels, (size_t)P, C, alpha, gamma, ws);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, FOCAL_BLOCKS, 1, loss_sum);
  return utv2_launch_status();
}

int utv2_sigmoid_focal_bwd(const float* logits, const int* labels, int64_t P, int C, float alpha, float gamma,
                           const float* coef, float* dlogits, hipStream_t stream) {
  if (!logits || !labels || !coef || !dlogits) return UTV2_EARG;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(2048), dim3(256), 0, stream, logits, labels, (size_t)P, C, alpha, gamma, coef, dlogits,
                     (const float*)nullptr, 0);
  return utv2_launch_status();
}

int utv2_sigmoid_focal_bwd_acc(const float* logits, const int* labels, int64_t P, int C, float alpha, float gamma, const float* coef,
                               const float* gscale, float* dlogits, int accumulate, hipStream_t stream) {
  if (!logits || !labels || !coef || !dlogits) return UTV2_EARG;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(2048), dim3(256), 0, stream, logits, labels, (size_t)P, C, alpha, gamma, coef, dlogits, gscale,
                     accumulate ? 1 : 0);
  return utv2_launch_status();
}

#define LOC_BLOCKS 512
// sums[8]: see LT_NSUM comment.  ws >= LOC_BLOCKS*8 floats.  reg_max+1 must be 17.
int utv2_fcos_loc_terms_fwd(const int* labels, const float* box, int box_stride, const float* reg_targets,
                            const float* bvars, int64_t P, int num_classes, int reg_max, float ts_better, float ts_cert,
                            int flags, float* sums, float* ws, hipStream_t stream) {
  if (!labels || !box || !reg_targets || !sums || !ws || reg_max != 16 || flags < 0 || ((flags >> LT_LOC_SHIFT) & 3) > 2 || flags >= 32 || box_stride < 4 * 17 + 5 || (box_stride & 3)) return UTV2_EARG;
  hipLaunchKernelGGL((fcos_loc_fwd_kernel<17>), dim3(LOC_BLOCKS), dim3(128), 0, stream, labels, box, box_stride, reg_targets,
                     bvars, (size_t)P, num_classes, ts_better, ts_cert, flags, ws);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(LT_NSUM), dim3(256), 0, stream, (const float*)ws, LOC_BLOCKS, LT_NSUM, sums);
  return utv2_launch_status();
}

int utv2_fcos_loc_terms_bwd(const int* labels, const float* box, int box_stride, const float* reg_targets,
                            const float* bvars, int64_t P, int num_classes, int reg_max, float ts_better, float ts_cert,
                            int flags, const float* coef, float* dbox, hipStream_t stream) {
  if (!labels || !box || !reg_targets || !coef || !dbox || reg_max != 16 || flags < 0 || ((flags >> LT_LOC_SHIFT) & 3) > 2 || flags >= 32 || box_stride < 4 * 17 + 5 || (box_stride & 3)) return UTV2_EARG;
  hipLaunchKernelGGL((fcos_loc_bwd_kernel<17>), dim3(cdiv(P, 128)), dim3(128), 0, stream, labels, box, box_stride, reg_targets,
                     bvars, (size_t)P, num_classes, ts_better, ts_cert, flags, coef, dbox, (const float*)nullptr, 0, 0);
  return utv2_launch_status();
}

int utv2_fcos_loc_terms_bwd_acc(const int* labels, const float* box, int box_stride, const float* reg_targets, const float* bvars,
                                int64_t P, int num_classes, int reg_max, float ts_better, float ts_cert, int flags, const float* coef8,
                                const float* gscale, float* dbox, int accumulate, hipStream_t stream) {
  if (!labels || !box || !reg_targets || !coef8 || !dbox || reg_max != 16 || flags < 0 || ((flags >> LT_LOC_SHIFT) & 3) > 2 || flags >= 32 || box_stride < 4 * 17 + 5 || (box_stride & 3)) return UTV2_EARG;
  hipLaunchKernelGGL((fcos_loc_bwd_kernel<17>), dim3(cdiv(P, 128)), dim3(128), 0, stream, labels, box, box_stride, reg_targets,
                     bvars, (size_t)P, num_classes, ts_better, ts_cert, flags, coef8, dbox, gscale, 1, accumulate ? 1 : 0);
  return utv2_launch_status();
}

// one level: logits [N][HW][C], box [N][HW][BS] -> keys [N][HW*C]
int utv2_fcos_rank_keys(const float* logits, const float* box, int box_stride, int reg_max, int N, int HW, int C, float thr,
                        int method, long long* keys, int64_t key_row_stride, hipStream_t stream) {
  if (!logits || !box || !keys || method < 0 || method > 3 || key_row_stride < (int64_t)HW * C) return UTV2_EARG;
  int gx = cdiv((int64_t)HW * C, 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(fcos_rank_keys_kernel, dim3(gx, N), dim3(256), 0, stream, logits, box, box_stride, 4 * (reg_max + 1), HW, C,
                     thr, method, keys, (size_t)key_row_stride);
  return utv2_launch_status();
}

int utv2_fcos_decode(const long long* topkeys, int K, const float* logits, const float* box, int box_stride, int reg_max,
                     int N, int HW, int Wl, int C, int stride, int level, int method, int MAXC, int slot0, float* oboxes,
                     float* oscores, int* ocls, float* oloc, float* octr, float* oconf, float* ostd, int* olevel,
                     unsigned char* ovalid, hipStream_t stream) {
  if (!topkeys || !logits || !box || reg_max != 16 || slot0 + K > MAXC) return UTV2_EARG;
  hipLaunchKernelGGL((fcos_decode_kernel<17>), dim3(cdiv(K, 128), N), dim3(128), 0, stream, topkeys, K, logits, box, box_stride,
                     HW, Wl, C, stride, level, method, MAXC, slot0, oboxes, oscores, ocls, oloc, octr, oconf, ostd, olevel,
                     ovalid);
  return utv2_launch_status();
}

int utv2_scale_cols(float* y, int64_t rows, int row_stride, int ncols, const float* s, hipStream_t stream) {
  if (!y || !s) return UTV2_EARG;
  int g = cdiv(rows * ncols, 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(scale_cols_kernel, dim3(g), dim3(256), 0, stream, y, (size_t)rows, row_stride, ncols, s);
  return utv2_launch_status();
}

// g[:, :ncols] *= s in place; dsum[0] = sum(g_in * ypost) (caller divides by s).  ws >= 1024 floats.
int utv2_scale_cols_bwd(float* g, const float* ypost, int64_t rows, int row_stride, int ncols, const float* s, float* dsum,
                        float* ws, hipStream_t stream) {
  if (!g || !ypost || !s || !dsum || !ws) return UTV2_EARG;
  int nb = cdiv(rows * ncols, 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(scale_cols_bwd_kernel, dim3(nb), dim3(256), 0, stream, g, ypost, (size_t)rows, row_stride, ncols, s, ws);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, nb, 1, dsum);
  return utv2_launch_status();
}

static int fill_scale_levels(ScaleLevels& L, int nlev, const int64_t* row0_host, const float* const* s_host, float* const* sgrad_host) {
  if (nlev < 1 || nlev > SC_MAXL || !row0_host || !s_host) return UTV2_EARG;
  L.nlev = nlev;
  for (int l = 0; l <= SC_MAXL; ++l) L.row0[l] = row0_host[l <= nlev ? l : nlev];
  for (int l = 0; l < SC_MAXL; ++l) {
    L.s[l] = l < nlev ? s_host[l] : nullptr;
    L.sgrad[l] = (l < nlev && sgrad_host) ? sgrad_host[l] : nullptr;
    if (l < nlev && (!L.s[l] || L.row0[l + 1] < L.row0[l])) return UTV2_EARG;
  }
  return UTV2_OK;
}

// y[rows of level l, 0:ncols] *= s_l[0] for every level in ONE launch.  row0_host: host int64[nlev + 1] (first row of each level, then
// the end); s_host: host array of nlev DEVICE pointers (one scalar each).
int utv2_scale_cols_ml(float* y, int nlev, const int64_t* row0_host, int row_stride, int ncols, const float* const* s_host,
                       hipStream_t stream) {
  ScaleLevels L;
  if (!y || fill_scale_levels(L, nlev, row0_host, s_host, nullptr) != UTV2_OK) return UTV2_EARG;
  int64_t most = 0;
  for (int l = 0; l < nlev; ++l) most = L.row0[l + 1] - L.row0[l] > most ? L.row0[l + 1] - L.row0[l] : most;
  int gx = cdiv(most * ncols, 256 * 4);
  if (gx > 1024) gx = 1024;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(scale_cols_ml_kernel, dim3(gx, nlev), dim3(256), 0, stream, y, L, row_stride, ncols);
  return utv2_launch_status();
}

// backward of all levels in two launches: g[rows of level l, 0:ncols] *= s_l in place and sgrad_l[0] += sum(g_in * ypost) / s_l
// (sgrad_host: host array of nlev device pointers into the gradient arena).  ws >= nlev * 256 floats.
int utv2_scale_cols_bwd_ml(float* g, const float* ypost, int nlev, const int64_t* row0_host, int row_stride, int ncols,
                           const float* const* s_host, float* const* sgrad_host, float* ws, hipStream_t stream) {
  ScaleLevels L;
  if (!g || !ypost || !ws || !sgrad_host || fill_scale_levels(L, nlev, row0_host, s_host, sgrad_host) != UTV2_OK) return UTV2_EARG;
  for (int l = 0; l < nlev; ++l)
    if (!L.sgrad[l]) return UTV2_EARG;
  const int nb = 1024;
  int64_t rows_max = 0;
  for (int l = 0; l < nlev; ++l) rows_max = rows_max > row0_host[l + 1] - row0_host[l] ? rows_max : row0_host[l + 1] - row0_host[l];
  const bool vec4 = (row_stride & 3) == 0 && (ncols & 3) == 0 && (((uintptr_t)g | (uintptr_t)ypost) & 15) == 0 &&
                    rows_max * (row_stride >> 2) < (int64_t)1 << 31;
  if (vec4) hipLaunchKernelGGL(scale_cols_bwd_ml_kernel<true>, dim3(nb, nlev), dim3(256), 0, stream, g, ypost, L, row_stride, ncols, ws);
  else hipLaunchKernelGGL(scale_cols_bwd_ml_kernel<false>, dim3(nb, nlev), dim3(256), 0, stream, g, ypost, L, row_stride, ncols, ws);
  hipLaunchKernelGGL(scale_cols_bwd_ml_final, dim3(nlev), dim3(256), 0, stream, L, (const float*)ws, nb);
  return utv2_launch_status();
}

int utv2_scale_cols_bwd_ml_pad16(const float* g, const float* ypost, int nlev, const int64_t* row0_host, int row_stride, int ncols,
                                 const float* const* s_host, float* const* sgrad_host, float* ws, void* out16, int cpad, hipStream_t stream) {
  ScaleLevels L;
  if (!g || !ypost || !ws || !sgrad_host || !out16 || fill_scale_levels(L, nlev, row0_host, s_host, sgrad_host) != UTV2_OK) return UTV2_EARG;
  for (int l = 0; l < nlev; ++l)
    if (!L.sgrad[l]) return UTV2_EARG;
  int64_t rows_max = 0;
  for (int l = 0; l < nlev; ++l) rows_max = rows_max > row0_host[l + 1] - row0_host[l] ? rows_max : row0_host[l + 1] - row0_host[l];
  if ((row_stride & 3) || (ncols & 3) || (cpad & 3) || cpad < row_stride || ncols > row_stride || ((((uintptr_t)g | (uintptr_t)ypost) & 15) != 0) ||
      (((uintptr_t)out16) & 7) != 0 || rows_max * (cpad >> 2) >= (int64_t)1 << 31)
    return UTV2_EARG;
  const int nb = 1024;
  hipLaunchKernelGGL(scale_cols_bwd_ml_pad16_kernel, dim3(nb, nlev), dim3(256), 0, stream, g, ypost, L, row_stride, ncols, ws, (h16_t*)out16, cpad);
  hipLaunchKernelGGL(scale_cols_bwd_ml_final, dim3(nlev), dim3(256), 0, stream, L, (const float*)ws, nb);
  return utv2_launch_status();
}

int utv2_fcos_loss_combine(const float* focal_sup, const float* sums_sup, const float* focal_cls, const float* sums_cls,
                           const float* sums_reg, const float* norm, float world, int flags, float kl_weight, const float* wmul_host,
                           const float* wdiv_host, float* rec, float* coef, hipStream_t stream) {
  if (!focal_sup || !sums_sup || !focal_cls || !sums_cls || !sums_reg || !wmul_host || !wdiv_host || !rec || !coef || !(world >= 1.f))
    return UTV2_EARG;
  FcosCombineArgs a;
  a.focal_sup = focal_sup; a.sums_sup = sums_sup; a.focal_cls = focal_cls; a.sums_cls = sums_cls; a.sums_reg = sums_reg; a.norm = norm;
  a.world = world; a.kl_weight = kl_weight; a.flags = flags;
  for (int k = 0; k < 6; ++k) { a.wmul[k] = wmul_host[k]; a.wdiv[k] = wdiv_host[k]; }
  hipLaunchKernelGGL(fcos_loss_combine_kernel, dim3(1), dim3(64), 0, stream, a, rec, coef);
  return utv2_launch_status();
}

}  // extern "C"
