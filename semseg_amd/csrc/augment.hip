// Device-side training input pipeline — SURVEY.md section 8(f) row 3: the reference's per-sample transform chain
// (util/transform.py:76-241, composed at tool/train.py:194-201, 209-212; called from util/dataset.py:67-69), which
// runs cv2 on CPU DataLoader workers, float32 HWC numpy arrays in and out.
//
// Design (not a transcription of the cv2 calls): all random parameters of a sample depend only on image SIZES, so
// the host (semseg_amd/transform.py) draws them first, in the reference's order, and back-propagates the final crop
// window through flip / blur / rotate / scale.  The device then computes only the pixels that can reach the output:
// every stage writes just its needed region (ROI) of its virtual full-size image.  The decoded image stays uint8 in
// HBM (3 B/px over PCIe instead of 12); the last stage fuses flip + pad + crop + channel order + (x-mean)/std +
// HWC->CHW + label->int64 and writes straight into the batch tensor the train step reads.  One launch per stage
// ROUND for the whole batch: blockIdx.y = sample, each block reads its sample's op descriptor and branches on the op
// kind (uniform per block).  All stages are gathers of a few MB: HBM/L2-bound, lanes along x.
//
// Arithmetic follows OpenCV 4.x imgproc for the argument combinations the reference uses (restated with citations in
// oracle/cv2_restated.py); float expressions are written in the published order with FMA contraction switched off for this
// file, so that the result does not depend on the compiler:
//   resize  INTER_LINEAR float: fx=(float)((dx+.5)*scale-.5), horizontal pass then vertical pass in float;
//           INTER_NEAREST: min(floor(dx*scale), n-1)
//   warpAffine: coordinates in 1/1024 px fixed point (AB_BITS 10), 1/32 px bilinear table (INTER_BITS 5),
//           BORDER_CONSTANT; NEAREST for labels
//   GaussianBlur sigma 0: tabulated 3/5/7 kernels, symmetric row then column pass, BORDER_REFLECT_101
#include "common.h"
#include "../../include/semseg_hip.h"

#pragma clang fp contract(off)

namespace {

typedef semseg_aug_op Op;

struct View {
  const unsigned char* img8;
  const float* img32;
  const unsigned char* lab;
  int y0, x0, h, w;   // materialised ROI of the virtual image
};

__device__ __forceinline__ View src_view(const Op& o) {
  View v;
  v.img8 = o.src_u8 ? (const unsigned char*)o.src_img : nullptr;
  v.img32 = o.src_u8 ? nullptr : (const float*)o.src_img;
  v.lab = (const unsigned char*)o.src_lab;
  v.y0 = o.src_y0; v.x0 = o.src_x0; v.h = o.src_h; v.w = o.src_w;
  return v;
}

// pixel (y, x) of the virtual image, channel c.  The planner guarantees (y, x) is inside the ROI; the clamp only
// keeps a planner bug from reading outside the arena.
__device__ __forceinline__ size_t roi_index(const View& v, int y, int x) {
  const int ly = max(min(y - v.y0, v.h - 1), 0), lx = max(min(x - v.x0, v.w - 1), 0);
  return (size_t)ly * v.w + lx;
}
__device__ __forceinline__ float px(const View& v, int y, int x, int c) {
  const size_t i = roi_index(v, y, x) * 3 + c;
  return v.img8 ? (float)v.img8[i] : v.img32[i];
}
__device__ __forceinline__ unsigned char lb(const View& v, int y, int x) { return v.lab[roi_index(v, y, x)]; }

// Plain operators, NOT the __fmul_rn/__fadd_rn header wrappers: those are compiled under the default
// -ffp-contract=fast and carry the `contract` flag into this file when inlined (observed: 1-ulp differences from
// fused multiply-adds); the pragma above only governs expressions written in this file.
__device__ __forceinline__ float mulf(float a, float b) { return a * b; }
__device__ __forceinline__ float addf(float a, float b) { return a + b; }
__device__ __forceinline__ float subf(float a, float b) { return a - b; }
__device__ __forceinline__ double muld(double a, double b) { return a * b; }
__device__ __forceinline__ double addd(double a, double b) { return a + b; }

// ---- resize (RandScale / Resize) ----
__device__ __forceinline__ void linear_coeff(int d, double scale, int n, int& s0, int& s1, float& f) {
  float fv = (float)(muld((double)d + 0.5, scale) - 0.5);
  int s = (int)floorf(fv);
  fv = subf(fv, (float)s);
  if (s < 0) { s = 0; fv = 0.f; }
  if (s >= n - 1) { s = n - 1; fv = 0.f; }
  s0 = s; s1 = min(s + 1, n - 1); f = fv;
}

__device__ void op_resize(const Op& o, int pix) {
  const View v = src_view(o);
  const int ly = pix / o.dst_w, lx = pix - ly * o.dst_w;
  const int dy = o.dst_y0 + ly, dx = o.dst_x0 + lx;
  int xa, xb, ya, yb;
  float fx, fy;
  linear_coeff(dx, o.p[0], o.src_W, xa, xb, fx);
  linear_coeff(dy, o.p[1], o.src_H, ya, yb, fy);
  const float a0 = subf(1.f, fx), b0 = subf(1.f, fy);
  float* out = (float*)o.dst_img + (size_t)pix * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float r0 = addf(mulf(px(v, ya, xa, c), a0), mulf(px(v, ya, xb, c), fx));
    const float r1 = addf(mulf(px(v, yb, xa, c), a0), mulf(px(v, yb, xb, c), fx));
    out[c] = addf(mulf(r0, b0), mulf(r1, fy));
  }
  const int nx = min((int)floor(muld((double)dx, o.p[0])), o.src_W - 1);
  const int ny = min((int)floor(muld((double)dy, o.p[1])), o.src_H - 1);
  ((unsigned char*)o.dst_lab)[pix] = lb(v, ny, nx);
}

// ---- warpAffine (RandRotate) ----
__device__ __forceinline__ long long cv_round(double v) { return (long long)rint(v); }
__device__ __forceinline__ int sat_short(long long v) { return (int)min(max(v, -32768LL), 32767LL); }

__device__ void op_rotate(const Op& o, int pix) {
  const View v = src_view(o);
  const int ly = pix / o.dst_w, lx = pix - ly * o.dst_w;
  const int dy = o.dst_y0 + ly, dx = o.dst_x0 + lx;
  const int W = o.src_W, H = o.src_H;
  const long long adelta = cv_round(muld(muld(o.p[0], (double)dx), 1024.0));
  const long long bdelta = cv_round(muld(muld(o.p[3], (double)dx), 1024.0));
  const long long X0 = cv_round(muld(addd(muld(o.p[1], (double)dy), o.p[2]), 1024.0));
  const long long Y0 = cv_round(muld(addd(muld(o.p[4], (double)dy), o.p[5]), 1024.0));
  // label: INTER_NEAREST, round_delta = 512
  {
    const int sx = sat_short((X0 + 512 + adelta) >> 10), sy = sat_short((Y0 + 512 + bdelta) >> 10);
    const bool in = sx >= 0 && sx < W && sy >= 0 && sy < H;
    ((unsigned char*)o.dst_lab)[pix] = in ? lb(v, sy, sx) : (unsigned char)o.pad_lab;
  }
  // image: INTER_LINEAR, round_delta = 16, 1/32 px grid
  const long long X = (X0 + 16 + adelta) >> 5, Y = (Y0 + 16 + bdelta) >> 5;
  const int sx = sat_short(X >> 5), sy = sat_short(Y >> 5);
  const float fx = mulf((float)(int)(X & 31), 0.03125f), fy = mulf((float)(int)(Y & 31), 0.03125f);
  const float gx = subf(1.f, fx), gy = subf(1.f, fy);
  const float w00 = mulf(gy, gx), w01 = mulf(gy, fx), w10 = mulf(fy, gx), w11 = mulf(fy, fx);
  float* out = (float*)o.dst_img + (size_t)pix * 3;
  if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) {
    out[0] = o.pad[0]; out[1] = o.pad[1]; out[2] = o.pad[2];
    return;
  }
  const bool x0in = sx >= 0 && sx < W, x1in = sx + 1 >= 0 && sx + 1 < W;
  const bool y0in = sy >= 0 && sy < H, y1in = sy + 1 >= 0 && sy + 1 < H;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float cv = o.pad[c];
    const float v0 = (x0in && y0in) ? px(v, sy, sx, c) : cv;
    const float v1 = (x1in && y0in) ? px(v, sy, sx + 1, c) : cv;
    const float v2 = (x0in && y1in) ? px(v, sy + 1, sx, c) : cv;
    const float v3 = (x1in && y1in) ? px(v, sy + 1, sx + 1, c) : cv;
    out[c] = addf(addf(addf(mulf(v0, w00), mulf(v1, w01)), mulf(v2, w10)), mulf(v3, w11));
  }
}

// ---- GaussianBlur (RandomGaussianBlur) ----
__device__ __forceinline__ int reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}

__constant__ float kGauss[4][4] = {   // centre weight first, then the pairs outwards (small_gaussian_tab)
    {1.f, 0.f, 0.f, 0.f}, {0.5f, 0.25f, 0.f, 0.f}, {0.375f, 0.25f, 0.0625f, 0.f}, {0.28125f, 0.21875f, 0.109375f, 0.03125f}};

__device__ void op_blur(const Op& o, int pix) {
  const View v = src_view(o);
  const int ly = pix / o.dst_w, lx = pix - ly * o.dst_w;
  const int dy = o.dst_y0 + ly, dx = o.dst_x0 + lx;
  const int half = o.ksize >> 1;
  const float* k = kGauss[half];
  int xs[7];
  for (int j = -half; j <= half; ++j) xs[j + half] = reflect101(dx + j, o.src_W);
  float* out = (float*)o.dst_img + (size_t)pix * 3;
  for (int c = 0; c < 3; ++c) {
    float rows[7];
    for (int r = -half; r <= half; ++r) {
      const int yy = reflect101(dy + r, o.src_H);
      float acc = mulf(px(v, yy, xs[half], c), k[0]);
      for (int j = 1; j <= half; ++j)
        acc = addf(acc, mulf(addf(px(v, yy, xs[half - j], c), px(v, yy, xs[half + j], c)), k[j]));
      rows[r + half] = acc;
    }
    float acc = mulf(rows[half], k[0]);
    for (int j = 1; j <= half; ++j) acc = addf(acc, mulf(addf(rows[half - j], rows[half + j]), k[j]));
    out[c] = acc;
  }
  ((unsigned char*)o.dst_lab)[pix] = lb(v, dy, dx);
}

// ---- index maps: flip / pad / crop / channel order, optionally fused with ToTensor + Normalize ----
__device__ void op_gather(const Op& o, int pix) {
  const View v = src_view(o);
  const int ly = pix / o.dst_w, lx = pix - ly * o.dst_w;
  int y = o.dst_y0 + ly, x = o.dst_x0 + lx;
  bool swapped = false, filled = false;
  float val[3];
  unsigned char lab = 0;
  for (int i = o.n_maps - 1; i >= 0; --i) {
    const semseg_aug_map& m = o.maps[i];
    swapped ^= (m.swap_rb != 0);
    y = m.sy * y + m.oy;
    x = m.sx * x + m.ox;
    if (y < 0 || y >= m.in_h || x < 0 || x >= m.in_w) {
      val[0] = m.pad[swapped ? 2 : 0]; val[1] = m.pad[1]; val[2] = m.pad[swapped ? 0 : 2];
      lab = (unsigned char)m.pad_lab;
      filled = true;
      break;
    }
  }
  if (!filled) {
    val[0] = px(v, y, x, swapped ? 2 : 0); val[1] = px(v, y, x, 1); val[2] = px(v, y, x, swapped ? 0 : 2);
    lab = lb(v, y, x);
  }
  if (o.out_chw) {
    const size_t plane = (size_t)o.dst_h * o.dst_w;
    float* out = (float*)o.dst_img;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float t = val[c];
      if (o.normalize) {
        t = subf(t, o.mean[c]);
        if (o.normalize > 1) t = t / o.std[c];
      }
      out[c * plane + pix] = t;
    }
    ((long long*)o.dst_lab)[pix] = (long long)lab;
  } else {
    float* out = (float*)o.dst_img + (size_t)pix * 3;
    out[0] = val[0]; out[1] = val[1]; out[2] = val[2];
    ((unsigned char*)o.dst_lab)[pix] = lab;
  }
}

__global__ __launch_bounds__(256) void aug_round_kernel(const Op* __restrict__ ops) {
  const Op& o = ops[blockIdx.y];
  if (o.kind == SEMSEG_AUG_NONE) return;
  const int total = o.dst_h * o.dst_w;
  for (int pix = blockIdx.x * 256 + threadIdx.x; pix < total; pix += gridDim.x * 256) {
    switch (o.kind) {
      case SEMSEG_AUG_RESIZE: op_resize(o, pix); break;
      case SEMSEG_AUG_ROTATE: op_rotate(o, pix); break;
      case SEMSEG_AUG_BLUR: op_blur(o, pix); break;
      default: op_gather(o, pix); break;
    }
  }
}

}  // namespace

extern "C" {

int semseg_augment_round(const semseg_aug_op* ops_dev, int n_samples, int max_pixels, hipStream_t stream) {
  if (!ops_dev || n_samples < 1 || max_pixels < 0) return SEMSEG_EINVAL;
  if (max_pixels == 0) return SEMSEG_OK;
  int bx = (max_pixels + 255) / 256;
  if (bx > 4096) bx = 4096;
  aug_round_kernel<<<dim3(bx, n_samples), 256, 0, stream>>>(ops_dev);
  return semseg_launch_status();
}

int semseg_aug_op_size(void) { return (int)sizeof(semseg_aug_op); }

}  // extern "C"
