// Fused segmentation head: bilinear upsample (align_corners=True) of the low-resolution class
// scores + CrossEntropyLoss(ignore_index) + argmax, and its backward.  Reference:
// model/pspnet.py:95,100-103 (F.interpolate -> criterion -> x.max(1)[1]) with the criterion built at
// tool/train.py:121.  The [N,classes,H,W] logits tensor (2.15 GB fp32 at bs16/473^2/150 classes) is
// never materialised in training: forward keeps only the per-pixel log-sum-exp (4 B/pixel),
// backward re-interpolates the scores from the L1-resident low-res map.
//
// Scores layout: NHWC [N,h,w,ld] with C valid channels (ld >= roundup4(C), pad = 0).
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

__device__ __forceinline__ void src_index(int o, int in, float scale, int& i0, int& i1, float& l) {
  const float s = scale * (float)o;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l = s - (float)i0;
}

// acc[0] += sum of per-pixel losses (fp64), acc[1] += number of non-ignored pixels (fp64)
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ z, int ld,
                                                     const long long* __restrict__ label,
                                                     float* __restrict__ lse_out,
                                                     long long* __restrict__ pred,
                                                     double* __restrict__ acc, int N, int h, int w,
                                                     int H, int W, int C, int ignore_index, float sh,
                                                     float sw) {
  const int total = N * H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  double loss = 0.0, cnt = 0.0;
  if (pix < total) {
    const int n = pix / (H * W);
    const int rem = pix - n * H * W;
    const int oh = rem / W, ow = rem - oh * W;
    int h0, h1, w0, w1;
    float lh, lw;
    src_index(oh, h, sh, h0, h1, lh);
    src_index(ow, w, sw, w0, w1, lw);
    const float a00 = (1.f - lh) * (1.f - lw), a01 = (1.f - lh) * lw, a10 = lh * (1.f - lw), a11 = lh * lw;
    const float* b = z + (size_t)n * h * w * ld;
    const float* p00 = b + ((size_t)h0 * w + w0) * ld;
    const float* p01 = b + ((size_t)h0 * w + w1) * ld;
    const float* p10 = b + ((size_t)h1 * w + w0) * ld;
    const float* p11 = b + ((size_t)h1 * w + w1) * ld;
    float mx = -INFINITY, sum = 0.f;
    int arg = 0;
    for (int c = 0; c < C; c += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p00 + c) * a00 +
                      *reinterpret_cast<const f32x4*>(p01 + c) * a01 +
                      *reinterpret_cast<const f32x4*>(p10 + c) * a10 +
                      *reinterpret_cast<const f32x4*>(p11 + c) * a11;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (c + k < C) {
          const float zz = v[k];
          if (zz > mx) {
            sum = sum * __expf(mx - zz) + 1.f;
            mx = zz;
            arg = c + k;
          } else {
            sum += __expf(zz - mx);
          }
        }
      }
    }
    const float lse = mx + __logf(sum);
    lse_out[pix] = lse;
    if (pred) pred[pix] = arg;
    const long long y = label[pix];
    if (y != (long long)ignore_index && y >= 0 && y < C) {
      const float zy = p00[y] * a00 + p01[y] * a01 + p10[y] * a10 + p11[y] * a11;
      loss = (double)(lse - zy);
      cnt = 1.0;
    }
  }
  // block reduce
  __shared__ double sl[4], sc[4];
  for (int o = 32; o > 0; o >>= 1) {
    loss += shfl_xor_f64(loss, o);
    cnt += shfl_xor_f64(cnt, o);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { sl[wv] = loss; sc[wv] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomic_add_f64(&acc[0], sl[0] + sl[1] + sl[2] + sl[3]);
    atomic_add_f64(&acc[1], sc[0] + sc[1] + sc[2] + sc[3]);
  }
}

// loss[0] = acc[0] / acc[1]  (NaN when every pixel is ignored, like torch)
__global__ void ce_finalize_kernel(const double* acc, float* loss) { loss[0] = (float)(acc[0] / acc[1]); }

// Gather-form backward.  Block = one low-res pixel (n,i,j); thread = (float4 class group, footprint
// partition).  dz[n,i,j,c] = gscale/count * sum_px wgt(px->ij) * (softmax_c(px) - [label(px)==c]).
template <int CG>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ z, int ld,
                                                     const long long* __restrict__ label,
                                                     const float* __restrict__ lse,
                                                     const double* __restrict__ acc,
                                                     const float* __restrict__ gloss, float gmul,
                                                     float* __restrict__ dz, int lddz, int accumulate,
                                                     int N, int h, int w, int H, int W, int C,
                                                     int ignore_index, float sh, float sw) {
  constexpr int PARTS = 256 / CG;
  __shared__ f32x4 red[256];
  const int pixl = blockIdx.x;
  const int n = pixl / (h * w);
  const int rem = pixl - n * h * w;
  const int i = rem / w, j = rem - i * w;
  const int cg = threadIdx.x % CG, part = threadIdx.x / CG;
  const int c = cg * 4;
  int oh_lo = 0, oh_hi = H - 1, ow_lo = 0, ow_hi = W - 1;
  if (sh > 0.f) { oh_lo = (int)floorf((float)(i - 1) / sh); oh_hi = (int)ceilf((float)(i + 1) / sh); }
  if (sw > 0.f) { ow_lo = (int)floorf((float)(j - 1) / sw); ow_hi = (int)ceilf((float)(j + 1) / sw); }
  oh_lo = max(oh_lo, 0); ow_lo = max(ow_lo, 0);
  oh_hi = min(oh_hi, H - 1); ow_hi = min(ow_hi, W - 1);
  const int nw = ow_hi - ow_lo + 1;
  const int cnt = (oh_hi - oh_lo + 1) * nw;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const float* b = z + (size_t)n * h * w * ld + c;
    for (int q = part; q < cnt; q += PARTS) {
      const int oh = oh_lo + q / nw, ow = ow_lo + q % nw;
      int h0, h1, w0, w1;
      float lh, lw;
      src_index(oh, h, sh, h0, h1, lh);
      src_index(ow, w, sw, w0, w1, lw);
      const float a00 = (1.f - lh) * (1.f - lw), a01 = (1.f - lh) * lw, a10 = lh * (1.f - lw), a11 = lh * lw;
      float wgt = 0.f;
      if (h0 == i && w0 == j) wgt += a00;
      if (h0 == i && w1 == j) wgt += a01;
      if (h1 == i && w0 == j) wgt += a10;
      if (h1 == i && w1 == j) wgt += a11;
      if (wgt == 0.f) continue;
      const size_t px = ((size_t)n * H + oh) * W + ow;
      const long long y = label[px];
      if (y == (long long)ignore_index || y < 0 || y >= C) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(b + ((size_t)h0 * w + w0) * ld) * a00 +
                      *reinterpret_cast<const f32x4*>(b + ((size_t)h0 * w + w1) * ld) * a01 +
                      *reinterpret_cast<const f32x4*>(b + ((size_t)h1 * w + w0) * ld) * a10 +
                      *reinterpret_cast<const f32x4*>(b + ((size_t)h1 * w + w1) * ld) * a11;
      const float l = lse[px];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (c + k < C) {
          float pk = __expf(v[k] - l);
          if ((long long)(c + k) == y) pk -= 1.f;
          a[k] += wgt * pk;
        }
      }
    }
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if (part == 0 && c < lddz) {
    for (int pp = 1; pp < PARTS; ++pp) a += red[pp * CG + cg];
    float scale = gmul / (float)acc[1];
    if (gloss) scale *= gloss[0];
    a *= scale;
    float* o = dz + (size_t)pixl * lddz + c;
    if (accumulate) a += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = a;
  }
}

}  // namespace

extern "C" {

int semseg_ce_head_fwd(const float* scores, int ld, const long long* label, float* lse,
                       long long* pred, double* acc2, float* loss, int N, int h, int w, int H, int W,
                       int C, int ignore_index, hipStream_t stream) {
  if (!scores || !label || !lse || !acc2 || !loss || (ld & 3) || ld < ((C + 3) & ~3)) return SEMSEG_EINVAL;
  if (hipMemsetAsync(acc2, 0, 2 * sizeof(double), stream) != hipSuccess) return SEMSEG_ELAUNCH;
  const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const int total = N * H * W;
  ce_fwd_kernel<<<(total + 255) / 256, 256, 0, stream>>>(scores, ld, label, lse, pred, acc2, N, h, w,
                                                       H, W, C, ignore_index, sh, sw);
  ce_finalize_kernel<<<1, 1, 0, stream>>>(acc2, loss);
  return semseg_launch_status();
}

int semseg_ce_head_bwd(const float* scores, int ld, const long long* label, const float* lse,
                       const double* acc2, const float* grad_loss, float grad_mul, float* dscores,
                       int lddz, int accumulate, int N, int h, int w, int H, int W, int C,
                       int ignore_index, hipStream_t stream) {
  if (!scores || !label || !lse || !acc2 || !dscores || (ld & 3) || (lddz & 3) ||
      ld < ((C + 3) & ~3) || lddz < ((C + 3) & ~3))
    return SEMSEG_EINVAL;
  const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const int cv = (C + 3) / 4;
  const int grid = N * h * w;
#define CE_BWD(CG)                                                                                  \
  ce_bwd_kernel<CG><<<grid, 256, 0, stream>>>(scores, ld, label, lse, acc2, grad_loss, grad_mul,   \
                                              dscores, lddz, accumulate, N, h, w, H, W, C,         \
                                              ignore_index, sh, sw)
  if (cv <= 8) CE_BWD(8);
  else if (cv <= 16) CE_BWD(16);
  else if (cv <= 32) CE_BWD(32);
  else if (cv <= 64) CE_BWD(64);
  else if (cv <= 128) CE_BWD(128);
  else return SEMSEG_EINVAL;
#undef CE_BWD
  return semseg_launch_status();
}

}  // extern "C"
