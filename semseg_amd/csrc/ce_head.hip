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

// acc[0] += sum of per-pixel losses (fp64), acc[1] += number of non-ignored pixels (fp64), acc[2] += number of labels
// that are neither ignore_index nor a class id (torch's CrossEntropyLoss raises on those; the caller checks the count)
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
  bool bad = false;
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
    } else if (y != (long long)ignore_index) {
      bad = true;
    }
  }
  {
    const unsigned long long m = __ballot(bad);      // rare path: one atomic per wave that saw an out-of-range label
    if (m && (threadIdx.x & 63) == 0) atomic_add_f64(&acc[2], (double)__popcll(m));
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
    // no valid pixel: torch's nll_loss backward leaves grad_input at 0 (only the loss is NaN)
    float scale = acc[1] > 0.0 ? gmul / (float)acc[1] : 0.f;
    if (gloss && acc[1] > 0.0) scale *= gloss[0];
    a *= scale;
    float* o = dz + (size_t)pixl * lddz + c;
    if (accumulate) a += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = a;
  }
}

// ---- cell-based backward (4x less work than the gather form: every hi-res pixel is visited once) ----
// A "cell" is the square between four adjacent low-res points.  Kernel A: one workgroup per cell walks
// the hi-res pixels whose interpolation base lies in the cell and accumulates their contribution to the
// cell's four corners; kernel B sums, for every low-res point, the matching corners of its <= 4 cells.
// Deterministic (no atomics).  cellbuf: [N][ch][cw][4][CP] floats, CP = roundup4(C).
template <int CG>
__global__ __launch_bounds__(256) void ce_bwd_cell_kernel(const float* __restrict__ z, int ld,
                                                          const long long* __restrict__ label,
                                                          const float* __restrict__ lse,
                                                          float* __restrict__ cellbuf, int N, int h,
                                                          int w, int H, int W, int C, int CP,
                                                          int ignore_index, float sh, float sw) {
  constexpr int PARTS = 256 / CG;
  __shared__ f32x4 red[4][256];
  const int ch = max(h - 1, 1), cw = max(w - 1, 1);
  int b = blockIdx.x;
  const int cc = b % cw; b /= cw;
  const int cr = b % ch;
  const int n = b / ch;
  const int cg = threadIdx.x % CG, part = threadIdx.x / CG;
  const int c = cg * 4;
  // conservative hi-res range of the cell; exact membership is re-checked per pixel
  int oh_lo = 0, oh_hi = H - 1, ow_lo = 0, ow_hi = W - 1;
  if (sh > 0.f) { oh_lo = (int)floorf((float)cr / sh) - 1; oh_hi = (int)ceilf((float)(cr + 1) / sh) + 1; }
  if (sw > 0.f) { ow_lo = (int)floorf((float)cc / sw) - 1; ow_hi = (int)ceilf((float)(cc + 1) / sw) + 1; }
  if (cr == ch - 1) oh_hi = H - 1;
  if (cc == cw - 1) ow_hi = W - 1;
  oh_lo = max(oh_lo, 0); ow_lo = max(ow_lo, 0);
  oh_hi = min(oh_hi, H - 1); ow_hi = min(ow_hi, W - 1);
  const int nw = ow_hi - ow_lo + 1;
  const int cnt = (oh_hi - oh_lo + 1) * nw;
  f32x4 a[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c < C && part < PARTS) {   // CG = 40 leaves 16 idle threads (part == PARTS)
    const float* bz = z + (size_t)n * h * w * ld + c;
    // The four low-res corners of the cell are the same for every hi-res pixel of the loop: loaded once (round 4; they were
    // re-read per pixel — 64 B per lane and pixel through the L1, ~9 GB per launch at 150 classes and 473^2, which is what
    // the kernel's 870 us were made of).  A pixel's interpolated score is formed from them with the SAME four products in the
    // same order as before: its corner weights (rw, cw below) equal (1 - lh, lh) x (1 - lw, lw) except on the last low-res
    // row / column, where the clamped index makes one weight 0 and the other 1 — adding the zero products is exact.
    const int r1 = min(cr + 1, h - 1), c1 = min(cc + 1, w - 1);
    const f32x4 z00 = *reinterpret_cast<const f32x4*>(bz + ((size_t)cr * w + cc) * ld);
    const f32x4 z01 = *reinterpret_cast<const f32x4*>(bz + ((size_t)cr * w + c1) * ld);
    const f32x4 z10 = *reinterpret_cast<const f32x4*>(bz + ((size_t)r1 * w + cc) * ld);
    const f32x4 z11 = *reinterpret_cast<const f32x4*>(bz + ((size_t)r1 * w + c1) * ld);
    for (int q = part; q < cnt; q += PARTS) {
      const int oh = oh_lo + q / nw, ow = ow_lo + q % nw;
      int h0, h1, w0, w1;
      float lh, lw;
      src_index(oh, h, sh, h0, h1, lh);
      src_index(ow, w, sw, w0, w1, lw);
      if (min(h0, ch - 1) != cr || min(w0, cw - 1) != cc) continue;
      const size_t px = ((size_t)n * H + oh) * W + ow;
      const long long y = label[px];
      if (y == (long long)ignore_index || y < 0 || y >= C) continue;
      // weights of the cell's corner rows cr, cr+1 / columns cc, cc+1 for this pixel (h1 == h0 / w1 == w0 on the last
      // low-res row / column)
      const float rw0 = (h0 == cr ? 1.f - lh : 0.f) + (h1 == cr ? lh : 0.f);
      const float rw1 = (h0 == cr + 1 ? 1.f - lh : 0.f) + (h1 == cr + 1 ? lh : 0.f);
      const float cw0 = (w0 == cc ? 1.f - lw : 0.f) + (w1 == cc ? lw : 0.f);
      const float cw1 = (w0 == cc + 1 ? 1.f - lw : 0.f) + (w1 == cc + 1 ? lw : 0.f);
      const f32x4 v = z00 * (rw0 * cw0) + z01 * (rw0 * cw1) + z10 * (rw1 * cw0) + z11 * (rw1 * cw1);
      const float l = lse[px];
      f32x4 pk;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pk[k] = (c + k < C) ? __expf(v[k] - l) : 0.f;
        if ((long long)(c + k) == y) pk[k] -= 1.f;
      }
      a[0] += pk * (rw0 * cw0);
      a[1] += pk * (rw0 * cw1);
      a[2] += pk * (rw1 * cw0);
      a[3] += pk * (rw1 * cw1);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[q][threadIdx.x] = a[q];
  __syncthreads();
  if (part == 0 && c < CP) {
    float* o = cellbuf + ((((size_t)n * ch + cr) * cw + cc) * 4) * CP + c;   // (threads >= PARTS * CG hold zeros)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 r = red[q][cg];
      for (int pp = 1; pp < PARTS; ++pp) r += red[q][pp * CG + cg];
      *reinterpret_cast<f32x4*>(o + (size_t)q * CP) = r;
    }
  }
}

__global__ __launch_bounds__(256) void ce_bwd_gather_cells_kernel(const float* __restrict__ cellbuf,
                                                                  const double* __restrict__ acc,
                                                                  const float* __restrict__ gloss,
                                                                  float gmul, float* __restrict__ dz,
                                                                  int lddz, int accumulate, int N,
                                                                  int h, int w, int CP) {
  const int ch = max(h - 1, 1), cw = max(w - 1, 1);
  const int CV = CP >> 2;
  const size_t total = (size_t)N * h * w * CV;
  // no valid pixel: torch's nll_loss backward leaves grad_input at 0 (only the loss is NaN)
  float scale = acc[1] > 0.0 ? gmul / (float)acc[1] : 0.f;
  if (gloss && acc[1] > 0.0) scale *= gloss[0];
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int c = (int)(idx % CV) * 4;
    size_t t = idx / CV;
    const int j = (int)(t % w); t /= w;
    const int i = (int)(t % h);
    const int n = (int)(t / h);
    f32x4 r = {0.f, 0.f, 0.f, 0.f};
    // cells (cr, cc) with corner (i - cr, j - cc) in {0,1}^2
    for (int dr = 0; dr < 2; ++dr) {
      const int cr = i - dr;
      if (cr < 0 || cr >= ch) continue;
      for (int dc = 0; dc < 2; ++dc) {
        const int cc = j - dc;
        if (cc < 0 || cc >= cw) continue;
        r += *reinterpret_cast<const f32x4*>(cellbuf + ((((size_t)n * ch + cr) * cw + cc) * 4 + dr * 2 + dc) * CP + c);
      }
    }
    r *= scale;
    float* o = dz + (((size_t)n * h + i) * w + j) * lddz + c;
    if (accumulate) r += *reinterpret_cast<const f32x4*>(o);
    *reinterpret_cast<f32x4*>(o) = r;
  }
}

// counts targets that are neither ignore_index nor a class id (torch raises / device-asserts on those)
__global__ __launch_bounds__(256) void label_check_kernel(const long long* __restrict__ label, size_t n,
                                                          int C, int ignore_index,
                                                          unsigned long long* __restrict__ bad) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const long long y = label[i];
    if (y != (long long)ignore_index && (y < 0 || y >= C)) ++c;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(bad, c);
}

}  // namespace

extern "C" {

int semseg_label_check(const long long* label, size_t n, int C, int ignore_index,
                       unsigned long long* bad_count_dev, hipStream_t stream) {
  if (!label || !bad_count_dev || C <= 0) return SEMSEG_EINVAL;
  if (hipMemsetAsync(bad_count_dev, 0, sizeof(unsigned long long), stream) != hipSuccess) return SEMSEG_ELAUNCH;
  size_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  label_check_kernel<<<(int)g, 256, 0, stream>>>(label, n, C, ignore_index, bad_count_dev);
  return semseg_launch_status();
}

int semseg_ce_head_fwd(const float* scores, int ld, const long long* label, float* lse,
                       long long* pred, double* acc2, float* loss, int N, int h, int w, int H, int W,
                       int C, int ignore_index, hipStream_t stream) {
  if (!scores || !label || !lse || !acc2 || !loss || (ld & 3) || ld < ((C + 3) & ~3)) return SEMSEG_EINVAL;
  if (hipMemsetAsync(acc2, 0, 3 * sizeof(double), stream) != hipSuccess) return SEMSEG_ELAUNCH;
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
                       int ignore_index, float* scratch, size_t scratch_floats, hipStream_t stream) {
  if (!scores || !label || !lse || !acc2 || !dscores || (ld & 3) || (lddz & 3) ||
      ld < ((C + 3) & ~3) || lddz < ((C + 3) & ~3))
    return SEMSEG_EINVAL;
  const float sh = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sw = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const int cv = (C + 3) / 4;
  const int CP = cv * 4;
  const int ch = h > 1 ? h - 1 : 1, cw = w > 1 ? w - 1 : 1;
  const size_t need = (size_t)N * ch * cw * 4 * CP;
  if (scratch && need <= scratch_floats && cv <= 128) {
    const int cells = N * ch * cw;
#define CE_CELL(CG)                                                                                 \
  ce_bwd_cell_kernel<CG><<<cells, 256, 0, stream>>>(scores, ld, label, lse, scratch, N, h, w, H, W, C, \
                                                    CP, ignore_index, sh, sw)
    if (cv <= 8) CE_CELL(8);
    else if (cv <= 16) CE_CELL(16);
    else if (cv <= 32) CE_CELL(32);
    else if (cv <= 40) CE_CELL(40);   // 150 classes = 38 float4: 6 pixels x 40 lanes instead of 4 x 64
    else if (cv <= 64) CE_CELL(64);
    else CE_CELL(128);
#undef CE_CELL
    size_t tot = (size_t)N * h * w * cv;
    size_t g = (tot + 255) / 256;
    if (g > 4096) g = 4096;
    ce_bwd_gather_cells_kernel<<<(int)g, 256, 0, stream>>>(scratch, acc2, grad_loss, grad_mul, dscores, lddz,
                                                          accumulate, N, h, w, CP);
    return semseg_launch_status();
  }
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
