// PSA head glue on the engine's NHWC layout (reference: model/psanet.py:53-98):
//  * psamask in pixel-major form: attention-conv output [N, H*W pixels, taps (ld)] -> affinity rows
//    A[n, q, p] (q = target pixel = row, p = source position = contiguous column), the operand layout
//    the MFMA contraction consumes directly (out[n,q,:] = sum_p A[n,q,p] * x[n,p,:], psanet.py:90-91).
//    Same index maps as lib/psa/src/cpu/psamask.cpp:11-113; out-of-window entries are written as 0
//    (the reference pre-zeroes, lib/psa/functions/psamask.py:17).
//  * softmax over p (F.softmax(y, dim=1), psanet.py:88-89) with the 1/normalization_factor of
//    psanet.py:90 folded in, and its backward.
//  * batched transpose (x[n]: [P,C] -> [C,P]) that puts the contraction's B operand K-contiguous.
// All HBM-bound; rows of P floats are contiguous so lanes walk consecutive addresses.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

// type 0 (collect):    A[n,(h,w),(h',w')] = M[n,(h,w), (h'-h+hh)*mW + (w'-w+hw)]
// type 1 (distribute): A[n,(h',w'),(h,w)] = M[n,(h,w), (h'-h+hh)*mW + (w'-w+hw)]
__global__ __launch_bounds__(256) void psamask_nhwc_fwd_kernel(const float* __restrict__ m, int ldm,
                                                               float* __restrict__ a, int lda,
                                                               int type, int N, int H, int W, int mH,
                                                               int mW, int hh, int hw) {
  const int HW = H * W;
  const size_t total = (size_t)N * HW * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int p = (int)(i % HW);
    size_t t = i / HW;
    const int q = (int)(t % HW);
    const int n = (int)(t / HW);
    int h, w, hp, wp;  // (h,w): pixel that predicted the mask; (hp,wp): shifted position
    if (type == 0) { h = q / W; w = q - h * W; hp = p / W; wp = p - hp * W; }
    else { hp = q / W; wp = q - hp * W; h = p / W; w = p - h * W; }
    const int hi = hp - h + hh, wi = wp - w + hw;
    float v = 0.f;
    if (hi >= 0 && hi < mH && wi >= 0 && wi < mW)
      v = m[((size_t)n * HW + h * W + w) * ldm + hi * mW + wi];
    a[((size_t)n * HW + q) * lda + p] = v;
  }
}

// dM[n,(h,w),(hi,wi)] = dA at the matching position, 0 when the shifted position is off the map
__global__ __launch_bounds__(256) void psamask_nhwc_bwd_kernel(const float* __restrict__ da, int lda,
                                                               float* __restrict__ dm, int ldm,
                                                               int type, int N, int H, int W, int mH,
                                                               int mW, int hh, int hw) {
  const int HW = H * W, T = mH * mW;
  const size_t total = (size_t)N * HW * T;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % T);
    size_t t = i / T;
    const int pix = (int)(t % HW);
    const int n = (int)(t / HW);
    const int h = pix / W, w = pix - h * W;
    const int hi = c / mW, wi = c - hi * mW;
    const int hp = h + hi - hh, wp = w + wi - hw;
    float v = 0.f;
    if (hp >= 0 && hp < H && wp >= 0 && wp < W) {
      const int sh = hp * W + wp;
      v = (type == 0) ? da[((size_t)n * HW + pix) * lda + sh] : da[((size_t)n * HW + sh) * lda + pix];
    }
    dm[((size_t)n * HW + pix) * ldm + c] = v;
  }
}

__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// one wave per row: y = alpha * softmax(x[0:P))   (softmax == 0: y = alpha * x)
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const float* __restrict__ x, int ldx,
                                                               float* __restrict__ y, int ldy,
                                                               int rows, int P, float alpha,
                                                               int softmax) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  float* yr = y + (size_t)row * ldy;
  if (!softmax) {
    for (int i = lane; i < P; i += 64) yr[i] = alpha * xr[i];
    return;
  }
  float mx = -INFINITY;
  for (int i = lane; i < P; i += 64) mx = fmaxf(mx, xr[i]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int i = lane; i < P; i += 64) s += expf(xr[i] - mx);
  s = wave_sum(s);
  const float inv = alpha / s;
  for (int i = lane; i < P; i += 64) yr[i] = expf(xr[i] - mx) * inv;
}

// dx = y * (dy - sum_j dy_j * y_j / alpha)   (softmax == 0: dx = alpha * dy); in place on dy allowed
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ y, int ldy,
                                                               const float* __restrict__ dy, int lddy,
                                                               float* __restrict__ dx, int lddx,
                                                               int rows, int P, float alpha,
                                                               int softmax) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* yr = y + (size_t)row * ldy;
  const float* gr = dy + (size_t)row * lddy;
  float* dr = dx + (size_t)row * lddx;
  if (!softmax) {
    for (int i = lane; i < P; i += 64) dr[i] = alpha * gr[i];
    return;
  }
  float dot = 0.f;
  for (int i = lane; i < P; i += 64) dot += gr[i] * yr[i];
  dot = wave_sum(dot) / alpha;
  for (int i = lane; i < P; i += 64) dr[i] = yr[i] * (gr[i] - dot);
}

// out[b][c][r] = in[b][r][c] for r < R, c < C; out columns r in [R, ldo) are zero-filled.
__global__ __launch_bounds__(256) void transpose_batched_kernel(const float* __restrict__ in,
                                                                int ldi, long long bsi,
                                                                float* __restrict__ out, int ldo,
                                                                long long bso, int R, int C) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* ib = in + (size_t)b * bsi;
  float* ob = out + (size_t)b * bso;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? ib[(size_t)r * ldi + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < C && r < ldo) ob[(size_t)c * ldo + r] = tile[tx][ty + 8 * k];
  }
}

inline int flat_grid(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" {

int semseg_psamask_nhwc_forward(int psa_type, const float* mask, int ldm, float* aff, int lda, int N,
                                int H, int W, int mH, int mW, hipStream_t stream) {
  if (!mask || !aff || lda < H * W || ldm < mH * mW) return SEMSEG_EINVAL;
  psamask_nhwc_fwd_kernel<<<flat_grid((size_t)N * H * W * H * W), 256, 0, stream>>>(
      mask, ldm, aff, lda, psa_type ? 1 : 0, N, H, W, mH, mW, (mH - 1) / 2, (mW - 1) / 2);
  return semseg_launch_status();
}

int semseg_psamask_nhwc_backward(int psa_type, const float* daff, int lda, float* dmask, int ldm,
                                 int N, int H, int W, int mH, int mW, hipStream_t stream) {
  if (!daff || !dmask || lda < H * W || ldm < mH * mW) return SEMSEG_EINVAL;
  psamask_nhwc_bwd_kernel<<<flat_grid((size_t)N * H * W * mH * mW), 256, 0, stream>>>(
      daff, lda, dmask, ldm, psa_type ? 1 : 0, N, H, W, mH, mW, (mH - 1) / 2, (mW - 1) / 2);
  return semseg_launch_status();
}

int semseg_softmax_rows_fwd(const float* x, int ldx, float* y, int ldy, int rows, int P,
                            float alpha, int softmax, hipStream_t stream) {
  if (!x || !y || ldx < P || ldy < P || rows <= 0) return SEMSEG_EINVAL;
  softmax_rows_fwd_kernel<<<(rows + 3) / 4, 256, 0, stream>>>(x, ldx, y, ldy, rows, P, alpha, softmax);
  return semseg_launch_status();
}

int semseg_softmax_rows_bwd(const float* y, int ldy, const float* dy, int lddy, float* dx,
                            int lddx, int rows, int P, float alpha, int softmax,
                            hipStream_t stream) {
  if (!y || !dy || !dx || ldy < P || lddy < P || lddx < P || rows <= 0) return SEMSEG_EINVAL;
  softmax_rows_bwd_kernel<<<(rows + 3) / 4, 256, 0, stream>>>(y, ldy, dy, lddy, dx, lddx, rows, P, alpha,
                                                             softmax);
  return semseg_launch_status();
}

int semseg_transpose_batched(const float* in, int ldi, long long batch_stride_in, float* out,
                             int ldo, long long batch_stride_out, int batch, int R, int C,
                             hipStream_t stream) {
  if (!in || !out || ldo < R || ldi < C || batch <= 0) return SEMSEG_EINVAL;
  dim3 grid((ldo + 31) / 32, (C + 31) / 32, batch);
  transpose_batched_kernel<<<grid, 256, 0, stream>>>(in, ldi, batch_stride_in, out, ldo,
                                                     batch_stride_out, R, C);
  return semseg_launch_status();
}

}  // extern "C"
