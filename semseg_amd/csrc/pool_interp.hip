// HBM-bound spatial ops of the path on NHWC fp32 (16-byte lane accesses along channels):
//  * MaxPool2d(3, stride 2, pad 1)            — model/resnet.py:115 (layer0.9 via model/pspnet.py:46)
//  * AdaptiveAvgPool2d(bin)                   — model/pspnet.py:14 (PPM)
//  * bilinear interpolate, align_corners=True — model/pspnet.py:25, model/psanet.py:61,78-79,97
// Backward kernels are gather-formulated (each thread owns an input-gradient element).  Only the
// tiny PPM tensors (a few low-res cells with map-sized windows) split their window over several
// workgroups and merge with fp32 atomics.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

inline int flat_grid(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

// ---------------- MaxPool 3x3 s2 p1 ----------------
// idx stores, per output element, the winning tap code r*3+s (first maximum in scan order, as
// torch's CPU kernel does), packed 4 channels per uint32.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x,
                                                          float* __restrict__ y,
                                                          uint32_t* __restrict__ idx, int N, int H,
                                                          int W, int C, int Ho, int Wo) {
  const int CV = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * CV;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % CV);
    size_t t = i / CV;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t code[4] = {0, 0, 0, 0};
    bool first = true;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int ih = oh * 2 + r - 1, iw = ow * 2 + s - 1;
        if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + ih) * W + iw) * C + c4 * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // NaN propagates like torch: (v > best) || isnan(v)
          if (first || v[k] > best[k] || v[k] != v[k]) {
            best[k] = v[k];
            code[k] = r * 3 + s;
          }
        }
        first = false;
      }
    *reinterpret_cast<f32x4*>(y + i * 4) = best;
    idx[i] = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                          const uint32_t* __restrict__ idx,
                                                          float* __restrict__ dx, int N, int H,
                                                          int W, int C, int Ho, int Wo) {
  const int CV = C >> 2;
  const size_t total = (size_t)N * H * W * CV;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % CV);
    size_t t = i / CV;
    const int iw = (int)(t % W); t /= W;
    const int ih = (int)(t % H);
    const int n = (int)(t / H);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // windows containing (ih, iw): oh with ih = 2*oh + r - 1, r in 0..2
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int th = ih + 1 - r;
      if (th < 0 || (th & 1)) continue;
      const int oh = th >> 1;
      if (oh >= Ho) continue;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int tw = iw + 1 - s;
        if (tw < 0 || (tw & 1)) continue;
        const int ow = tw >> 1;
        if (ow >= Wo) continue;
        const size_t o = ((size_t)(n * Ho + oh) * Wo + ow) * CV + c4;
        const uint32_t code = idx[o];
        const f32x4 g = *reinterpret_cast<const f32x4*>(dy + o * 4);
        const uint32_t me = r * 3 + s;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (((code >> (8 * k)) & 0xff) == me) acc[k] += g[k];
      }
    }
    *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
  }
}

// ---------------- AdaptiveAvgPool2d for a list of bins (PPM) ----------------
// grid.x = N * total_cells, grid.y = channel groups of 1024; output for bin b at
// y + off_b + ((n*b + i)*b + j)*C.   Window rows [floor(i*H/b), ceil((i+1)*H/b)).
struct PoolBins {
  int nb;
  int bin[4];
  int cell_start[5];      // prefix of bin^2
  long long out_off[4];   // float offset of each bin's [N,b,b,C] output
};

__global__ __launch_bounds__(256) void adaptive_pool_fwd_kernel(const float* __restrict__ x, int ldx,
                                                                float* __restrict__ y, PoolBins pb,
                                                                int N, int H, int W, int C) {
  const int cells = pb.cell_start[pb.nb];
  const int n = blockIdx.x / cells;
  int cell = blockIdx.x - n * cells;
  int b = 0;
  while (b + 1 < pb.nb && cell >= pb.cell_start[b + 1]) ++b;
  cell -= pb.cell_start[b];
  const int bin = pb.bin[b];
  const int i = cell / bin, j = cell - i * bin;
  const int h0 = (i * H) / bin, h1 = ((i + 1) * H + bin - 1) / bin;
  const int w0 = (j * W) / bin, w1 = ((j + 1) * W + bin - 1) / bin;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  // the window's rows are split over gridDim.z workgroups (big windows: bin 1 = whole map) and merged
  // with fp32 atomics into the pre-zeroed output
  const int nz = gridDim.z;
  const int rows = h1 - h0;
  const int per = (rows + nz - 1) / nz;
  const int hs = h0 + blockIdx.z * per, he = min(h1, hs + per);
  if (hs >= he) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int h = hs; h < he; ++h)
    for (int w = w0; w < w1; ++w)
      acc += *reinterpret_cast<const f32x4*>(x + ((size_t)(n * H + h) * W + w) * ldx + c);
  const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
  acc *= inv;
  float* o = y + pb.out_off[b] + ((size_t)(n * bin + i) * bin + j) * C + c;
  if (nz == 1) {
    *reinterpret_cast<f32x4*>(o) = acc;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(o + k, acc[k]);
  }
}

// dx[n,h,w,c] = base[n,h,w,c] + sum_bins sum_{cells containing (h,w)} dpool[cell]/area
__global__ __launch_bounds__(256) void adaptive_pool_bwd_kernel(const float* __restrict__ base,
                                                                int ldbase,
                                                                const float* __restrict__ dpool,
                                                                PoolBins pb, float* __restrict__ dx,
                                                                int lddx, int N, int H, int W,
                                                                int C) {
  const int pix = blockIdx.x;
  const int n = pix / (H * W);
  const int rem = pix - n * H * W;
  const int h = rem / W, w = rem - h * W;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (base) acc = *reinterpret_cast<const f32x4*>(base + (size_t)pix * ldbase + c);
  for (int b = 0; b < pb.nb; ++b) {
    const int bin = pb.bin[b];
    for (int i = 0; i < bin; ++i) {
      const int h0 = (i * H) / bin, h1 = ((i + 1) * H + bin - 1) / bin;
      if (h < h0 || h >= h1) continue;
      for (int j = 0; j < bin; ++j) {
        const int w0 = (j * W) / bin, w1 = ((j + 1) * W + bin - 1) / bin;
        if (w < w0 || w >= w1) continue;
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
        const f32x4 g = *reinterpret_cast<const f32x4*>(
            dpool + pb.out_off[b] + ((size_t)(n * bin + i) * bin + j) * C + c);
        acc += g * inv;
      }
    }
  }
  *reinterpret_cast<f32x4*>(dx + (size_t)pix * lddx + c) = acc;
}

// ---------------- bilinear, align_corners=True ----------------
__device__ __forceinline__ void src_index(int o, int in, float scale, int& i0, int& i1, float& l) {
  const float s = scale * (float)o;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l = s - (float)i0;
}

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, int ldx,
                                                           float* __restrict__ y, int ldy, int N,
                                                           int Hi, int Wi, int Ho, int Wo, int C,
                                                           float sh, float sw) {
  const int CV = C >> 2;
  const size_t total = (size_t)N * Ho * Wo * CV;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c4 = (int)(i % CV);
    size_t t = i / CV;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    int h0, h1, w0, w1;
    float lh, lw;
    src_index(oh, Hi, sh, h0, h1, lh);
    src_index(ow, Wi, sw, w0, w1, lw);
    const float* b = x + (size_t)n * Hi * Wi * ldx + c4 * 4;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((size_t)h0 * Wi + w0) * ldx);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((size_t)h0 * Wi + w1) * ldx);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((size_t)h1 * Wi + w0) * ldx);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((size_t)h1 * Wi + w1) * ldx);
    const float a00 = (1.f - lh) * (1.f - lw), a01 = (1.f - lh) * lw, a10 = lh * (1.f - lw), a11 = lh * lw;
    const f32x4 r = v00 * a00 + v01 * a01 + v10 * a10 + v11 * a11;
    *reinterpret_cast<f32x4*>(y + ((size_t)(n * Ho + oh) * Wo + ow) * ldy + c4 * 4) = r;
  }
}

// Gather-form backward: block = one low-res pixel (n,i,j) x 64 float4 channel lanes x 4 footprint
// partitions; the footprint of (i,j) is every hi-res pixel whose source coordinate lies within
// (i-1, i+1) x (j-1, j+1).
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, int lddy,
                                                           float* __restrict__ dx, int lddx, int N,
                                                           int Hi, int Wi, int Ho, int Wo, int C,
                                                           float sh, float sw) {
  __shared__ f32x4 red[256];
  const int pix = blockIdx.x;
  const int n = pix / (Hi * Wi);
  const int rem = pix - n * Hi * Wi;
  const int i = rem / Wi, j = rem - i * Wi;
  const int lanec = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int c = (blockIdx.y * 64 + lanec) * 4;
  // hi-res row range whose source row is in (i-1, i+1)
  int oh_lo = 0, oh_hi = Ho - 1, ow_lo = 0, ow_hi = Wo - 1;
  if (sh > 0.f) {
    oh_lo = (int)floorf((float)(i - 1) / sh);
    oh_hi = (int)ceilf((float)(i + 1) / sh);
  }
  if (sw > 0.f) {
    ow_lo = (int)floorf((float)(j - 1) / sw);
    ow_hi = (int)ceilf((float)(j + 1) / sw);
  }
  oh_lo = max(oh_lo, 0); ow_lo = max(ow_lo, 0);
  oh_hi = min(oh_hi, Ho - 1); ow_hi = min(ow_hi, Wo - 1);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const int nw = ow_hi - ow_lo + 1;
    const int cnt = (oh_hi - oh_lo + 1) * nw;
    for (int q = part + 4 * blockIdx.z; q < cnt; q += 4 * gridDim.z) {
      const int oh = oh_lo + q / nw, ow = ow_lo + q % nw;
      int h0, h1, w0, w1;
      float lh, lw;
      src_index(oh, Hi, sh, h0, h1, lh);
      src_index(ow, Wi, sw, w0, w1, lw);
      float wgt = 0.f;
      if (h0 == i && w0 == j) wgt += (1.f - lh) * (1.f - lw);
      if (h0 == i && w1 == j) wgt += (1.f - lh) * lw;
      if (h1 == i && w0 == j) wgt += lh * (1.f - lw);
      if (h1 == i && w1 == j) wgt += lh * lw;
      if (wgt != 0.f)
        acc += *reinterpret_cast<const f32x4*>(dy + ((size_t)(n * Ho + oh) * Wo + ow) * lddy + c) * wgt;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (part == 0 && c < C) {
    acc = red[lanec] + red[64 + lanec] + red[128 + lanec] + red[192 + lanec];
    float* o = dx + (size_t)pix * lddx + c;
    if (gridDim.z == 1) {
      *reinterpret_cast<f32x4*>(o) = acc;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) atomicAdd(o + k, acc[k]);
    }
  }
}

// NHWC logits (C channels, ld) -> NCHW bilinear-upsampled output (eval head, model/pspnet.py:95,105)
__global__ __launch_bounds__(256) void bilinear_to_nchw_kernel(const float* __restrict__ x, int ldx,
                                                               float* __restrict__ y, int N, int Hi,
                                                               int Wi, int Ho, int Wo, int C,
                                                               float sh, float sw) {
  const size_t total = (size_t)N * C * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ow = (int)(i % Wo);
    size_t t = i / Wo;
    const int oh = (int)(t % Ho); t /= Ho;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    int h0, h1, w0, w1;
    float lh, lw;
    src_index(oh, Hi, sh, h0, h1, lh);
    src_index(ow, Wi, sw, w0, w1, lw);
    const float* b = x + (size_t)n * Hi * Wi * ldx + c;
    const float v00 = b[((size_t)h0 * Wi + w0) * ldx], v01 = b[((size_t)h0 * Wi + w1) * ldx];
    const float v10 = b[((size_t)h1 * Wi + w0) * ldx], v11 = b[((size_t)h1 * Wi + w1) * ldx];
    y[i] = v00 * ((1.f - lh) * (1.f - lw)) + v01 * ((1.f - lh) * lw) + v10 * (lh * (1.f - lw)) +
           v11 * (lh * lw);
  }
}

inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

}  // namespace

extern "C" {

int semseg_maxpool3x3s2_fwd(const float* x, float* y, uint32_t* idx, int N, int H, int W, int C,
                            hipStream_t stream) {
  if (!x || !y || !idx || (C & 3)) return SEMSEG_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  maxpool_fwd_kernel<<<flat_grid((size_t)N * Ho * Wo * (C >> 2)), 256, 0, stream>>>(x, y, idx, N, H, W, C, Ho, Wo);
  return semseg_launch_status();
}

int semseg_maxpool3x3s2_bwd(const float* dy, const uint32_t* idx, float* dx, int N, int H, int W,
                            int C, hipStream_t stream) {
  if (!dy || !dx || !idx || (C & 3)) return SEMSEG_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  maxpool_bwd_kernel<<<flat_grid((size_t)N * H * W * (C >> 2)), 256, 0, stream>>>(dy, idx, dx, N, H, W, C, Ho, Wo);
  return semseg_launch_status();
}

// ---- PPM pooling in one pass over the feature map (deterministic) --------------------------------------
// A "slot" is one (bin, column-cell j) pair: sum of bins <= 16 slots.  Stage 1: one workgroup per image row
// accumulates, for every slot, the sum of the row's pixels that fall into that column window (windows of
// AdaptiveAvgPool2d may overlap by one pixel, so a pixel can feed two slots of a bin) -> rs[N*H][S][C].
// Stage 2: each pooled cell sums its slot over the rows of its row window.  x is read once (the per-bin
// kernel read it once per bin and merged partial windows with fp32 atomics).
#define POOL_SMAX 16
struct PoolSlots {
  int S;
  int w0[POOL_SMAX], w1[POOL_SMAX];   // column window of the slot
  int bin[POOL_SMAX], j[POOL_SMAX];   // its bin index (0..nb-1) and column cell
  int slot_start[5];
};

__global__ __launch_bounds__(256) void pool_rowsum_kernel(const float* __restrict__ x, int ldx,
                                                          float* __restrict__ rs, PoolSlots ps, int W,
                                                          int C) {
  const int nh = blockIdx.x;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  f32x4 acc[POOL_SMAX];
#pragma unroll
  for (int s = 0; s < POOL_SMAX; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* row = x + (size_t)nh * W * ldx + c;
  for (int w = 0; w < W; ++w) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + (size_t)w * ldx);
#pragma unroll
    for (int s = 0; s < POOL_SMAX; ++s)
      if (s < ps.S && w >= ps.w0[s] && w < ps.w1[s]) acc[s] += v;   // uniform branch
  }
#pragma unroll
  for (int s = 0; s < POOL_SMAX; ++s)
    if (s < ps.S) *reinterpret_cast<f32x4*>(rs + ((size_t)nh * ps.S + s) * C + c) = acc[s];
}

__global__ __launch_bounds__(256) void pool_gather_kernel(const float* __restrict__ rs,
                                                          float* __restrict__ y, PoolBins pb,
                                                          PoolSlots ps, int N, int H, int W, int C) {
  const int cells = pb.cell_start[pb.nb];
  const int n = blockIdx.x / cells;
  int cell = blockIdx.x - n * cells;
  int b = 0;
  while (b + 1 < pb.nb && cell >= pb.cell_start[b + 1]) ++b;
  cell -= pb.cell_start[b];
  const int bin = pb.bin[b];
  const int i = cell / bin, j = cell - i * bin;
  const int h0 = (i * H) / bin, h1 = ((i + 1) * H + bin - 1) / bin;
  const int w0 = (j * W) / bin, w1 = ((j + 1) * W + bin - 1) / bin;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  const int slot = ps.slot_start[b] + j;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int h = h0; h < h1; ++h)
    acc += *reinterpret_cast<const f32x4*>(rs + ((size_t)(n * H + h) * ps.S + slot) * C + c);
  acc *= 1.f / (float)((h1 - h0) * (w1 - w0));
  *reinterpret_cast<f32x4*>(y + pb.out_off[b] + ((size_t)(n * bin + i) * bin + j) * C + c) = acc;
}

// dx[n,h,:,c] for one image row: the row's contribution of every slot (sum over the row cells that contain
// h of dpool / area) is built once in registers, then each pixel adds the slots whose column window holds it.
__global__ __launch_bounds__(256) void pool_bwd_row_kernel(const float* __restrict__ base, int ldbase,
                                                           const float* __restrict__ dpool, PoolBins pb,
                                                           PoolSlots ps, float* __restrict__ dx, int lddx,
                                                           int H, int W, int C) {
  const int nh = blockIdx.x;
  const int n = nh / H, h = nh - n * H;
  const int c = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (c >= C) return;
  f32x4 g[POOL_SMAX];
#pragma unroll
  for (int s = 0; s < POOL_SMAX; ++s) {
    g[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s < ps.S) {
      const int b = ps.bin[s], bin = pb.bin[b], j = ps.j[s];
      for (int i = 0; i < bin; ++i) {
        const int h0 = (i * H) / bin, h1 = ((i + 1) * H + bin - 1) / bin;
        if (h < h0 || h >= h1) continue;
        const float inv = 1.f / (float)((h1 - h0) * (ps.w1[s] - ps.w0[s]));
        g[s] += *reinterpret_cast<const f32x4*>(dpool + pb.out_off[b] +
                                                ((size_t)(n * bin + i) * bin + j) * C + c) * inv;
      }
    }
  }
  for (int w = 0; w < W; ++w) {
    const size_t pix = (size_t)nh * W + w;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (base) acc = *reinterpret_cast<const f32x4*>(base + pix * ldbase + c);
#pragma unroll
    for (int s = 0; s < POOL_SMAX; ++s)
      if (s < ps.S && w >= ps.w0[s] && w < ps.w1[s]) acc += g[s];
    *reinterpret_cast<f32x4*>(dx + pix * lddx + c) = acc;
  }
}

// slots of a bin set; returns false when there are more than POOL_SMAX (callers fall back)
static bool make_slots(const int* bins, int nbins, int W, PoolSlots& ps) {
  int S = 0;
  for (int b = 0; b < nbins; ++b) {
    ps.slot_start[b] = S;
    for (int j = 0; j < bins[b]; ++j) {
      if (S >= POOL_SMAX) return false;
      ps.w0[S] = (j * W) / bins[b];
      ps.w1[S] = ((j + 1) * W + bins[b] - 1) / bins[b];
      ps.bin[S] = b;
      ps.j[S] = j;
      ++S;
    }
  }
  ps.slot_start[nbins] = S;
  for (int s2 = S; s2 < POOL_SMAX; ++s2) { ps.w0[s2] = ps.w1[s2] = 0; ps.bin[s2] = 0; ps.j[s2] = 0; }
  ps.S = S;
  return true;
}

static int make_bins(const int* bins, int nbins, int N, int C, PoolBins& pb) {
  if (nbins < 1 || nbins > 4) return SEMSEG_EINVAL;
  pb.nb = nbins;
  pb.cell_start[0] = 0;
  long long off = 0;
  for (int b = 0; b < nbins; ++b) {
    if (bins[b] < 1) return SEMSEG_EINVAL;
    pb.bin[b] = bins[b];
    pb.cell_start[b + 1] = pb.cell_start[b] + bins[b] * bins[b];
    pb.out_off[b] = off;
    off += (long long)N * bins[b] * bins[b] * C;
  }
  return SEMSEG_OK;
}

size_t semseg_adaptive_avgpool_scratch_floats(const int* bins, int nbins, int N, int H, int C) {
  size_t S = 0;
  for (int b = 0; b < nbins; ++b) S += (size_t)(bins[b] > 0 ? bins[b] : 0);
  return S > POOL_SMAX ? 0 : (size_t)N * H * S * C;
}

// y holds the nbins pooled maps back to back: [N,b0,b0,C][N,b1,b1,C]...
int semseg_adaptive_avgpool_fwd(const float* x, int ldx, float* y, const int* bins, int nbins,
                                int N, int H, int W, int C, float* scratch, size_t scratch_floats,
                                hipStream_t stream) {
  if (!x || !y || (C & 3) || (ldx & 3)) return SEMSEG_EINVAL;
  PoolBins pb;
  if (make_bins(bins, nbins, N, C, pb)) return SEMSEG_EINVAL;
  const int cells = N * pb.cell_start[nbins];
  PoolSlots ps;
  if (scratch && make_slots(bins, nbins, W, ps) && (size_t)N * H * ps.S * C <= scratch_floats) {
    // one pass over x (row sums per slot), then a gather over rows: deterministic, x read once
    dim3 g1(N * H, (C / 4 + 255) / 256);
    pool_rowsum_kernel<<<g1, 256, 0, stream>>>(x, ldx, scratch, ps, W, C);
    dim3 g2(cells, (C / 4 + 255) / 256);
    pool_gather_kernel<<<g2, 256, 0, stream>>>(scratch, y, pb, ps, N, H, W, C);
    return semseg_launch_status();
  }
  int nz = 1;
  if (cells * ((C / 4 + 255) / 256) < 512 && H >= 16) nz = 8;
  if (nz > 1) {
    size_t tot = 0;
    for (int b = 0; b < nbins; ++b) tot += (size_t)N * bins[b] * bins[b] * C;
    if (hipMemsetAsync(y, 0, tot * sizeof(float), stream) != hipSuccess) return SEMSEG_ELAUNCH;
  }
  dim3 grid(cells, (C / 4 + 255) / 256, nz);
  adaptive_pool_fwd_kernel<<<grid, 256, 0, stream>>>(x, ldx, y, pb, N, H, W, C);
  return semseg_launch_status();
}

int semseg_adaptive_avgpool_bwd(const float* base, int ldbase, const float* dpool, float* dx,
                                int lddx, const int* bins, int nbins, int N, int H, int W, int C,
                                hipStream_t stream) {
  if (!dpool || !dx || (C & 3) || (lddx & 3) || (base && (ldbase & 3))) return SEMSEG_EINVAL;
  PoolBins pb;
  if (make_bins(bins, nbins, N, C, pb)) return SEMSEG_EINVAL;
  PoolSlots ps;
  if (make_slots(bins, nbins, W, ps)) {
    dim3 grid(N * H, (C / 4 + 255) / 256);
    pool_bwd_row_kernel<<<grid, 256, 0, stream>>>(base, ldbase, dpool, pb, ps, dx, lddx, H, W, C);
    return semseg_launch_status();
  }
  dim3 grid(N * H * W, (C / 4 + 255) / 256);
  adaptive_pool_bwd_kernel<<<grid, 256, 0, stream>>>(base, ldbase, dpool, pb, dx, lddx, N, H, W, C);
  return semseg_launch_status();
}

int semseg_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int Hi, int Wi, int Ho,
                        int Wo, int C, hipStream_t stream) {
  if (!x || !y || (C & 3) || (ldx & 3) || (ldy & 3)) return SEMSEG_EINVAL;
  bilinear_fwd_kernel<<<flat_grid((size_t)N * Ho * Wo * (C >> 2)), 256, 0, stream>>>(
      x, ldx, y, ldy, N, Hi, Wi, Ho, Wo, C, ac_scale(Hi, Ho), ac_scale(Wi, Wo));
  return semseg_launch_status();
}

int semseg_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int Hi, int Wi,
                        int Ho, int Wo, int C, hipStream_t stream) {
  if (!dy || !dx || (C & 3) || (lddx & 3) || (lddy & 3)) return SEMSEG_EINVAL;
  // few low-res pixels with huge footprints (PPM bins): split the footprint over gridDim.z and
  // merge with fp32 atomics into the zeroed gradient (only valid for a dense dx, lddx == C)
  int nz = 1;
  const int blocks = N * Hi * Wi * ((C / 4 + 63) / 64);
  if (blocks < 1024 && lddx == C && Ho * Wo >= 64 * Hi * Wi) {
    nz = 1024 / blocks;
    if (nz > 64) nz = 64;
    if (nz < 1) nz = 1;
  }
  if (nz > 1 && hipMemsetAsync(dx, 0, (size_t)N * Hi * Wi * C * sizeof(float), stream) != hipSuccess)
    return SEMSEG_ELAUNCH;
  dim3 grid(N * Hi * Wi, (C / 4 + 63) / 64, nz);
  bilinear_bwd_kernel<<<grid, 256, 0, stream>>>(dy, lddy, dx, lddx, N, Hi, Wi, Ho, Wo, C,
                                               ac_scale(Hi, Ho), ac_scale(Wi, Wo));
  return semseg_launch_status();
}

int semseg_bilinear_nhwc_to_nchw(const float* x, int ldx, float* y, int N, int Hi, int Wi, int Ho,
                                 int Wo, int C, hipStream_t stream) {
  if (!x || !y) return SEMSEG_EINVAL;
  bilinear_to_nchw_kernel<<<flat_grid((size_t)N * C * Ho * Wo), 256, 0, stream>>>(
      x, ldx, y, N, Hi, Wi, Ho, Wo, C, ac_scale(Hi, Ho), ac_scale(Wi, Wo));
  return semseg_launch_status();
}

}  // extern "C"
