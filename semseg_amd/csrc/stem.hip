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
// Thread = (co, pixel sub-stream); 27 accumulators per thread; fp32 atomics merge blocks.
template <int CO>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x,
                                                         const float* __restrict__ dy,
                                                         float* __restrict__ dw, int N, int H,
                                                         int W, int Ho, int Wo) {
  constexpr int PARTS = 256 / CO;
  const int co = threadIdx.x % CO;
  const int part = threadIdx.x / CO;
  const int M = N * Ho * Wo;
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  for (int m = blockIdx.x * PARTS + part; m < M; m += gridDim.x * PARTS) {
    const int n = m / (Ho * Wo);
    const int rem = m - n * Ho * Wo;
    const int oh = rem / Wo, ow = rem - oh * Wo;
    const float g = dy[(size_t)m * CO + co];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int ih = oh * 2 + r - 1, iw = ow * 2 + s - 1;
          float v = 0.f;
          if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = x[((size_t)(n * 3 + ci) * H + ih) * W + iw];
          acc[ci * 9 + r * 3 + s] = fmaf(g, v, acc[ci * 9 + r * 3 + s]);
        }
  }
  __shared__ float red[256 * 27];
#pragma unroll
  for (int k = 0; k < 27; ++k) red[k * 256 + threadIdx.x] = acc[k];
  __syncthreads();
  for (int i = threadIdx.x; i < 27 * CO; i += 256) {
    const int c = i % CO, k = i / CO;
    float v = 0.f;
    for (int pp = 0; pp < PARTS; ++pp) v += red[k * 256 + pp * CO + c];
    atomicAdd(&dw[c * 27 + k], v);
  }
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

// dw_oihw must be zero-filled by the caller when accumulate == 0 semantics are wanted
// (the kernel always adds); the wrapper does the memset itself.
int semseg_stem_conv_wgrad(const float* x_nchw, const float* dy_nhwc, float* dw_oihw, int N, int H,
                           int W, int Co, int accumulate, hipStream_t stream) {
  if (!x_nchw || !dy_nhwc || !dw_oihw || Co != 64) return SEMSEG_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (!accumulate) {
    if (hipMemsetAsync(dw_oihw, 0, sizeof(float) * Co * 27, stream) != hipSuccess) return SEMSEG_ELAUNCH;
  }
  stem_wgrad_kernel<64><<<1024, 256, 0, stream>>>(x_nchw, dy_nhwc, dw_oihw, N, H, W, Ho, Wo);
  return semseg_launch_status();
}

}  // extern "C"
