// Stem convolution 3->Co (3x3, stride 2, pad 1) reading the caller's NCHW input directly
// (reference: model/resnet.py:108 `conv3x3(3, 64, stride=2)` as layer0.0, model/pspnet.py:46).
// K = 27 is too thin for a matrix-core tile, so this is a direct VALU kernel; it is 0.04 % of the
// step's FLOPs.  The NCHW->NHWC layout change of the whole network happens here, once.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

// One thread: one output pixel x 16 output channels.  256 threads = 64 pixels x 4 channel groups.
template <int CO>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ w,  // [CO][3][3][3]
                                                       float* __restrict__ y, int N, int H, int W,
                                                       int Ho, int Wo) {
  __shared__ float ws[27 * CO];  // [k][co]
  for (int i = threadIdx.x; i < 27 * CO; i += 256) {
    const int co = i % CO, k = i / CO;
    ws[i] = w[co * 27 + k];
  }
  __syncthreads();
  constexpr int GROUPS = CO / 16;
  constexpr int PIX = 256 / GROUPS;
  const int cg = threadIdx.x % GROUPS;
  const int lp = threadIdx.x / GROUPS;
  const int M = N * Ho * Wo;
  for (int m = blockIdx.x * PIX + lp; m < M; m += gridDim.x * PIX) {
    const int n = m / (Ho * Wo);
    const int rem = m - n * Ho * Wo;
    const int oh = rem / Wo, ow = rem - oh * Wo;
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int ih = oh * 2 + r - 1, iw = ow * 2 + s - 1;
          float v = 0.f;
          if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[((size_t)(n * 3 + ci) * H + ih) * W + iw];
          const float* wk = &ws[(ci * 9 + r * 3 + s) * CO + cg * 16];
#pragma unroll
          for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, wk[c], acc[c]);
        }
    float* o = y + (size_t)m * CO + cg * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
      *reinterpret_cast<f32x4*>(o + q * 4) = v;
    }
  }
}

// dW[co][ci][r][s] = sum over pixels dy[m][co] * x[n][ci][2oh+r-1][2ow+s-1].
// Tile = SW consecutive output columns of one output row: its 3 channels x 3 input rows x (2 SW + 1) input columns go through LDS
// once (zero outside the image), lane = output channel, wave = every fourth pixel of the tile: per pixel ONE coalesced 256-byte
// read of dy and 27 same-address LDS reads.  A workgroup walks tiles with a grid stride, folds its four waves through LDS and stores ONE
// partial [CO][27] slab; stem_wgrad_fold_kernel sums the slabs in a fixed order (no atomics: the result does not depend on arrival order).
// (Rounds 1-5 and the first round-6 form read x from global memory per pixel and output channel, 27 dependent-latency loads per
// iteration on ~110-512 workgroups: 454 us at per-GPU batch 2 and 834 us at batch 16 for 29 / 230 MB of dy,
// profiles/r06_bs2_serial_family_stats.csv.)
constexpr int STEM_SW = 64;
template <int CO>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ dy,
                                                         float* __restrict__ partial, int N, int H,
                                                         int W, int Ho, int Wo, int tiles_w, int tiles) {
  constexpr int SW = STEM_SW, PW = 2 * SW + 1;
  __shared__ float patch[9 * PW + 3];
  __shared__ float red[256 * 27];
  const int co = threadIdx.x % CO;
  const int wave = threadIdx.x / CO;        // CO == 64: one wave per pixel sub-stream
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const int tw = t % tiles_w;
    const int row = t / tiles_w;            // n * Ho + oh
    const int n = row / Ho, oh = row - n * Ho;
    const int ow0 = tw * SW;
    __syncthreads();                        // the previous tile's readers are done with the patch
    for (int i = threadIdx.x; i < 9 * PW; i += 256) {
      const int cr = i / PW, j = i - cr * PW;
      const int ci = cr / 3, r = cr - ci * 3;
      const int ih = oh * 2 + r - 1, iw = ow0 * 2 + j - 1;
      float v = 0.f;
      if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[((size_t)(n * 3 + ci) * H + ih) * W + iw];
      patch[i] = v;
    }
    __syncthreads();
    const int npx = min(SW, Wo - ow0);
    const float* dyr = dy + ((size_t)row * Wo + ow0) * CO + co;
    for (int p0 = wave; p0 < npx; p0 += 16) {
      float g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = p0 + 4 * u < npx ? dyr[(size_t)(p0 + 4 * u) * CO] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int px = min(p0 + 4 * u, SW - 1);
#pragma unroll
        for (int cr = 0; cr < 9; ++cr)
#pragma unroll
          for (int s = 0; s < 3; ++s) acc[cr * 3 + s] = fmaf(g[u], patch[cr * PW + 2 * px + s], acc[cr * 3 + s]);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 27; ++k) red[k * 256 + threadIdx.x] = acc[k];
  __syncthreads();
  float* slab = partial + (size_t)blockIdx.x * (27 * CO);
  for (int i = threadIdx.x; i < 27 * CO; i += 256) {
    const int c = i % CO, k = i / CO;
    float v = 0.f;
    for (int pp = 0; pp < 256 / CO; ++pp) v += red[k * 256 + pp * CO + c];
    slab[c * 27 + k] = v;
  }
}

// dw[i] (=|+=) sum over the slabs of partial[slab][n]: 8 outputs per workgroup, 32 lanes per output each summing every 32nd slab (two
// loads in flight), then a fixed-order fold over the 32 lanes — the first form walked all slabs per thread: 237 dependent
// trips at per-GPU batch 2, 74 us for 6.5 MB.
__global__ __launch_bounds__(256) void stem_wgrad_fold_kernel(const float* __restrict__ partial, int slabs, int n,
                                                              float* __restrict__ dw, int accumulate) {
  const int l = threadIdx.x & 31;
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5);
  float v0 = 0.f, v1 = 0.f;
  if (i < n) {
    int s = l;
    for (; s + 32 < slabs; s += 64) {
      v0 += partial[(size_t)s * n + i];
      v1 += partial[(size_t)(s + 32) * n + i];
    }
    if (s < slabs) v0 += partial[(size_t)s * n + i];
  }
  float v = v0 + v1;
#pragma unroll
  for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  if (i < n && l == 0) dw[i] = accumulate ? dw[i] + v : v;
}

}  // namespace

extern "C" {

int semseg_stem_conv_fwd(const float* x_nchw, const float* w_oihw, float* y_nhwc, int N, int H,
                         int W, int Co, hipStream_t stream) {
  if (!x_nchw || !w_oihw || !y_nhwc || Co != 64) return SEMSEG_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int M = N * Ho * Wo;
  int grid = (M + 63) / 64;
  if (grid > 8192) grid = 8192;
  stem_fwd_kernel<64><<<grid, 256, 0, stream>>>(x_nchw, w_oihw, y_nhwc, N, H, W, Ho, Wo);
  return semseg_launch_status();
}

// scratch: >= semseg_stem_wgrad_scratch_floats(N, H, W) floats on the launch stream's arena (per-workgroup partial slabs).
static inline int stem_wgrad_grid(int N, int H, int W) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  long long tiles = (long long)N * Ho * ((Wo + STEM_SW - 1) / STEM_SW);
  long long g = (tiles + 1) / 2;          // two tiles per workgroup: the slab store and the fold stay small
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;
  return (int)g;
}

size_t semseg_stem_wgrad_scratch_floats(int N, int H, int W) { return (size_t)stem_wgrad_grid(N, H, W) * 27 * 64; }

int semseg_stem_conv_wgrad(const float* x_nchw, const float* dy_nhwc, float* dw_oihw, int N, int H,
                           int W, int Co, int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream) {
  if (!x_nchw || !dy_nhwc || !dw_oihw || Co != 64 || !scratch || N <= 0) return SEMSEG_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int grid = stem_wgrad_grid(N, H, W);
  if (scratch_floats < (size_t)grid * 27 * 64) return SEMSEG_EINVAL;
  const int tiles_w = (Wo + STEM_SW - 1) / STEM_SW;
  stem_wgrad_kernel<64><<<grid, 256, 0, stream>>>(x_nchw, dy_nhwc, scratch, N, H, W, Ho, Wo, tiles_w, N * Ho * tiles_w);
  stem_wgrad_fold_kernel<<<(27 * 64 + 7) / 8, 256, 0, stream>>>(scratch, grid, 27 * 64, dw_oihw, accumulate);
  return semseg_launch_status();
}

}  // extern "C"
