// Device-side multi-scale sliding-window inference — the step right after the model at test time
// (reference: tool/test.py:122-146 net_process, :149-178 scale_process, :181-204 test; the same
// logic in tool/demo.py:106-189).  The reference bounces every crop through numpy (134 MB D2H per
// crop, float64 canvases on the host); here the image pyramid, the crop/normalise/flip staging, the
// softmax + flip-average + canvas accumulation, the per-scale resize-accumulate and the final argmax
// all stay in HBM.  All kernels are HBM-bound elementwise/gather ops (lanes along the image x axis).
//
// cv2.resize(..., INTER_LINEAR) on float32 (test.py:201,177) = half-pixel bilinear with edge clamp:
//   fx = (dx + 0.5) * (src_w / dst_w) - 0.5; sx = floor(fx); fx -= sx; sx < 0 -> (0, fx = 0);
//   sx >= src_w - 1 -> (src_w - 1, fx = 0).   No antialiasing (that is INTER_AREA).
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

__device__ __forceinline__ void cv_src(int d, float scale, int n, int& i0, int& i1, float& f) {
  float s = ((float)d + 0.5f) * scale - 0.5f;
  int i = (int)floorf(s);
  f = s - (float)i;
  if (i < 0) { i = 0; f = 0.f; }
  if (i >= n - 1) { i = n - 1; f = 0.f; }
  i0 = i;
  i1 = min(i + 1, n - 1);
}

inline int grid1(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

// HWC float image -> HWC float image (image pyramid level, test.py:201)
__global__ __launch_bounds__(256) void resize_hwc_kernel(const float* __restrict__ src, int Hs, int Ws,
                                                         float* __restrict__ dst, int Hd, int Wd, int C,
                                                         float sy, float sx) {
  const size_t total = (size_t)Hd * Wd * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const size_t t = i / C;
    const int x = (int)(t % Wd), y = (int)(t / Wd);
    int y0, y1, x0, x1;
    float fy, fx;
    cv_src(y, sy, Hs, y0, y1, fy);
    cv_src(x, sx, Ws, x0, x1, fx);
    const float v00 = src[((size_t)y0 * Ws + x0) * C + c], v01 = src[((size_t)y0 * Ws + x1) * C + c];
    const float v10 = src[((size_t)y1 * Ws + x0) * C + c], v11 = src[((size_t)y1 * Ws + x1) * C + c];
    dst[i] = (v00 * (1.f - fx) + v01 * fx) * (1.f - fy) + (v10 * (1.f - fx) + v11 * fx) * fy;
  }
}

// crops of the (virtually mean-padded) HWC image -> normalised NCHW batch [2*K,3,ch,cw]:
// entry 2k = crop k, 2k+1 = its horizontal flip (test.py:123-132,156,171).  Padding = mean, i.e. 0
// after normalisation.  origins[k] = (y0, x0) of crop k in UNPADDED image coordinates.
__global__ __launch_bounds__(256) void crop_norm_flip_kernel(const float* __restrict__ img, int H, int W,
                                                             const int* __restrict__ origins, int K,
                                                             int ch, int cw, float m0, float m1,
                                                             float m2, float s0, float s1, float s2,
                                                             float* __restrict__ out) {
  const size_t total = (size_t)K * 3 * ch * cw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % cw);
    size_t t = i / cw;
    const int y = (int)(t % ch); t /= ch;
    const int c = (int)(t % 3);
    const int k = (int)(t / 3);
    const int iy = origins[2 * k] + y, ix = origins[2 * k + 1] + x;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const float istd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    float v = 0.f;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = (img[((size_t)iy * W + ix) * 3 + c] - mean) / istd;
    const size_t plane = (size_t)ch * cw;
    out[((size_t)(2 * k) * 3 + c) * plane + (size_t)y * cw + x] = v;
    out[((size_t)(2 * k + 1) * 3 + c) * plane + (size_t)y * cw + (cw - 1 - x)] = v;
  }
}

// logits [2K,C,ch,cw] -> canvas[C][Hc][Wc] += (softmax(l[2k]) + flip(softmax(l[2k+1]))) / 2 on the
// crop's window, count[Hc][Wc] += 1   (test.py:139-141,172-173).  pos[k] = (y, x) in canvas coords.
// One thread per crop pixel; crops of one launch may overlap, hence atomics on the canvas.
__global__ __launch_bounds__(256) void softmax_flip_accum_kernel(const float* __restrict__ logits,
                                                                 const int* __restrict__ pos, int K,
                                                                 int C, int ch, int cw,
                                                                 float* __restrict__ canvas,
                                                                 float* __restrict__ count, int Hc,
                                                                 int Wc) {
  const size_t total = (size_t)K * ch * cw;
  const size_t plane = (size_t)ch * cw;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % cw);
    const size_t t = i / cw;
    const int y = (int)(t % ch);
    const int k = (int)(t / ch);
    const float* a = logits + (size_t)(2 * k) * C * plane + (size_t)y * cw + x;
    const float* b = logits + (size_t)(2 * k + 1) * C * plane + (size_t)y * cw + (cw - 1 - x);
    float ma = -INFINITY, mb = -INFINITY;
    for (int c = 0; c < C; ++c) {
      ma = fmaxf(ma, a[c * plane]);
      mb = fmaxf(mb, b[c * plane]);
    }
    float sa = 0.f, sb = 0.f;
    for (int c = 0; c < C; ++c) {
      sa += expf(a[c * plane] - ma);
      sb += expf(b[c * plane] - mb);
    }
    const float ia = 0.5f / sa, ib = 0.5f / sb;
    const int cy = pos[2 * k] + y, cx = pos[2 * k + 1] + x;
    const size_t cplane = (size_t)Hc * Wc;
    float* o = canvas + (size_t)cy * Wc + cx;
    for (int c = 0; c < C; ++c)
      atomicAdd(o + c * cplane, expf(a[c * plane] - ma) * ia + expf(b[c * plane] - mb) * ib);
    atomicAdd(count + (size_t)cy * Wc + cx, 1.f);
  }
}

// dst[C][Hd][Wd] += weight * resize( canvas[C][y0:y0+Hs][x0:x0+Ws] / count )   (test.py:175-177,203)
__global__ __launch_bounds__(256) void resize_accum_chw_kernel(const float* __restrict__ canvas,
                                                               const float* __restrict__ count, int Hc,
                                                               int Wc, int y0, int x0, int Hs, int Ws,
                                                               float* __restrict__ dst, int Hd, int Wd,
                                                               int C, float weight, float sy, float sx) {
  const size_t total = (size_t)C * Hd * Wd;
  const size_t cplane = (size_t)Hc * Wc;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % Wd);
    const size_t t = i / Wd;
    const int y = (int)(t % Hd);
    const int c = (int)(t / Hd);
    int ya, yb, xa, xb;
    float fy, fx;
    cv_src(y, sy, Hs, ya, yb, fy);
    cv_src(x, sx, Ws, xa, xb, fx);
    const float* p = canvas + c * cplane;
    const size_t i00 = (size_t)(y0 + ya) * Wc + x0 + xa, i01 = (size_t)(y0 + ya) * Wc + x0 + xb;
    const size_t i10 = (size_t)(y0 + yb) * Wc + x0 + xa, i11 = (size_t)(y0 + yb) * Wc + x0 + xb;
    const float v00 = p[i00] / count[i00], v01 = p[i01] / count[i01];
    const float v10 = p[i10] / count[i10], v11 = p[i11] / count[i11];
    dst[i] += weight * ((v00 * (1.f - fx) + v01 * fx) * (1.f - fy) + (v10 * (1.f - fx) + v11 * fx) * fy);
  }
}

__global__ __launch_bounds__(256) void argmax_chw_kernel(const float* __restrict__ p, long long* out,
                                                         int C, size_t plane) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
    float best = p[i];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      const float v = p[c * plane + i];
      if (v > best) { best = v; arg = c; }   // first maximum, as np.argmax
    }
    out[i] = arg;
  }
}

}  // namespace

extern "C" {

int semseg_resize_linear_hwc(const float* src, int Hs, int Ws, float* dst, int Hd, int Wd, int C,
                             hipStream_t stream) {
  if (!src || !dst || Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1 || C < 1) return SEMSEG_EINVAL;
  resize_hwc_kernel<<<grid1((size_t)Hd * Wd * C), 256, 0, stream>>>(
      src, Hs, Ws, dst, Hd, Wd, C, (float)((double)Hs / Hd), (float)((double)Ws / Wd));
  return semseg_launch_status();
}

int semseg_crop_normalize_flip(const float* img_hwc, int H, int W, const int* origins_dev, int K,
                               int crop_h, int crop_w, const float* mean3, const float* std3,
                               float* out_nchw, hipStream_t stream) {
  if (!img_hwc || !origins_dev || !out_nchw || !mean3 || !std3 || K < 1) return SEMSEG_EINVAL;
  crop_norm_flip_kernel<<<grid1((size_t)K * 3 * crop_h * crop_w), 256, 0, stream>>>(
      img_hwc, H, W, origins_dev, K, crop_h, crop_w, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2],
      out_nchw);
  return semseg_launch_status();
}

int semseg_softmax_flip_accumulate(const float* logits_nchw, const int* pos_dev, int K, int C,
                                   int crop_h, int crop_w, float* canvas_chw, float* count, int Hc,
                                   int Wc, hipStream_t stream) {
  if (!logits_nchw || !pos_dev || !canvas_chw || !count || K < 1 || C < 1) return SEMSEG_EINVAL;
  softmax_flip_accum_kernel<<<grid1((size_t)K * crop_h * crop_w), 256, 0, stream>>>(
      logits_nchw, pos_dev, K, C, crop_h, crop_w, canvas_chw, count, Hc, Wc);
  return semseg_launch_status();
}

int semseg_resize_accumulate_chw(const float* canvas_chw, const float* count, int Hc, int Wc, int y0,
                                 int x0, int Hs, int Ws, float* dst_chw, int Hd, int Wd, int C,
                                 float weight, hipStream_t stream) {
  if (!canvas_chw || !count || !dst_chw || y0 < 0 || x0 < 0 || y0 + Hs > Hc || x0 + Ws > Wc)
    return SEMSEG_EINVAL;
  resize_accum_chw_kernel<<<grid1((size_t)C * Hd * Wd), 256, 0, stream>>>(
      canvas_chw, count, Hc, Wc, y0, x0, Hs, Ws, dst_chw, Hd, Wd, C, weight, (float)((double)Hs / Hd),
      (float)((double)Ws / Wd));
  return semseg_launch_status();
}

int semseg_argmax_chw(const float* prob_chw, long long* out, int C, int H, int W, hipStream_t stream) {
  if (!prob_chw || !out || C < 1) return SEMSEG_EINVAL;
  argmax_chw_kernel<<<grid1((size_t)H * W), 256, 0, stream>>>(prob_chw, out, C, (size_t)H * W);
  return semseg_launch_status();
}

}  // extern "C"

// ---- segmentation metrics on the device: intersectionAndUnionGPU (util/util.py:55-67), called once
// per train step at tool/train.py:286 and per validation batch at :375.  One pass, LDS histograms,
// one atomic per (block, class): hist[0..K) intersection, [K..2K) output area, [2K..3K) target area.
// torch.histc(min=0, max=K-1) drops out-of-range values, so label/pred outside [0,K) (incl. the
// ignore value written into `output`) count nowhere.
namespace {
__global__ __launch_bounds__(256) void iou_hist_kernel(const long long* __restrict__ pred,
                                                       const long long* __restrict__ target, size_t n,
                                                       int K, int ignore_index,
                                                       unsigned long long* __restrict__ hist) {
  extern __shared__ unsigned int sh[];  // [3K]
  for (int i = threadIdx.x; i < 3 * K; i += 256) sh[i] = 0;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const long long t = target[i];
    long long o = pred[i];
    if (t == (long long)ignore_index) o = ignore_index;   // util/util.py:61
    if (o >= 0 && o < K) {
      atomicAdd(&sh[K + (int)o], 1u);
      if (o == t) atomicAdd(&sh[(int)o], 1u);
    }
    if (t >= 0 && t < K) atomicAdd(&sh[2 * K + (int)t], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * K; i += 256)
    if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

__global__ void iou_finalize_kernel(const unsigned long long* hist, int K, float* inter, float* uni,
                                    float* tgt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  const float i = (float)hist[c], o = (float)hist[K + c], t = (float)hist[2 * K + c];
  inter[c] = i;
  uni[c] = o + t - i;   // util/util.py:66
  tgt[c] = t;
}
}  // namespace

extern "C" int semseg_intersection_and_union(const long long* pred, const long long* target, size_t n,
                                             int K, int ignore_index, unsigned long long* hist3K,
                                             float* area_intersection, float* area_union,
                                             float* area_target, hipStream_t stream) {
  if (!pred || !target || !hist3K || !area_intersection || !area_union || !area_target || K < 1 || K > 4096)
    return SEMSEG_EINVAL;
  if (hipMemsetAsync(hist3K, 0, sizeof(unsigned long long) * 3 * K, stream) != hipSuccess) return SEMSEG_ELAUNCH;
  size_t g = (n + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  iou_hist_kernel<<<(int)g, 256, 3 * K * sizeof(unsigned int), stream>>>(pred, target, n, K, ignore_index, hist3K);
  iou_finalize_kernel<<<(K + 255) / 256, 256, 0, stream>>>(hist3K, K, area_intersection, area_union, area_target);
  return semseg_launch_status();
}
