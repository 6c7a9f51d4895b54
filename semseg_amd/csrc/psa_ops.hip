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

// With a = the pixel that predicted the mask row and b = the shifted position it talks about,
//     G[n, a, b] = M[n, a, (h_b - h_a + hh) * mW + (w_b - w_a + hw)]     (0 outside the mask window)
// type 0 (collect) is A = G and type 1 (distribute) is A = G^T per image (lib/psa/src/cpu/psamask.cpp:26-29 vs
// 52-55).  One workgroup owns a 64 x 64 tile of G: lanes run along b, so for a fixed a the 64 source taps are
// runs of W contiguous floats of ONE mask row; the tile goes through LDS, which makes the store coalesced for both
// types (rows of A along b for collect, along a for distribute).  32-bit index arithmetic only, the per-lane b is
// decoded once per tile (the previous one-thread-per-element kernels spent their time in 64-bit div/mod).
constexpr int PT = 64;   // tile edge

struct PsaNhwcArgs {
  const float* src;
  float* dst;
  int lds_, ldd, type, N, H, W, mH, mW, hh, hw, tiles;
};

__global__ __launch_bounds__(256) void psamask_nhwc_fwd_kernel(const PsaNhwcArgs p) {
  __shared__ float tile[PT][PT + 1];
  const int HW = p.H * p.W;
  int blk = blockIdx.x;
  const int tb = blk % p.tiles; blk /= p.tiles;
  const int ta = blk % p.tiles;
  const int n = blk / p.tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = tb * PT + lane;
  const int hb = b / p.W, wb = b - hb * p.W;
  const float* mrow = p.src + (size_t)n * HW * p.lds_;
#pragma unroll 4
  for (int r = wave; r < PT; r += 4) {
    const int a = ta * PT + r;                      // wave-uniform
    const int ha = a / p.W, wa = a - ha * p.W;
    const int hi = hb - ha + p.hh, wi = wb - wa + p.hw;
    float v = 0.f;
    if (a < HW && b < HW && (unsigned)hi < (unsigned)p.mH && (unsigned)wi < (unsigned)p.mW)
      v = mrow[(size_t)a * p.lds_ + hi * p.mW + wi];
    tile[r][lane] = v;
  }
  __syncthreads();
  float* arow = p.dst + (size_t)n * HW * p.ldd;
  if (p.type == 0) {
#pragma unroll 4
    for (int r = wave; r < PT; r += 4) {
      const int a = ta * PT + r;
      if (a < HW && b < HW) arow[(size_t)a * p.ldd + b] = tile[r][lane];
    }
  } else {
    const int a = ta * PT + lane;
#pragma unroll 4
    for (int r = wave; r < PT; r += 4) {
      const int bb = tb * PT + r;
      if (a < HW && bb < HW) arow[(size_t)bb * p.ldd + a] = tile[lane][r];
    }
  }
}

// dM[n, a, (hi,wi)] = dG[n, a, b(a,hi,wi)], 0 when b is off the map; dG = dA (collect) or dA^T (distribute).
// Collect: one workgroup per (n, a), lanes along the taps: reads are runs of <= W contiguous floats of row a of dA,
// the store covers the whole tap row (zeros included) in order.
__global__ __launch_bounds__(256) void psamask_nhwc_bwd_collect_kernel(const PsaNhwcArgs p) {
  const int HW = p.H * p.W, T = p.mH * p.mW;
  const int a = blockIdx.x % HW, n = blockIdx.x / HW;
  const int ha = a / p.W, wa = a - ha * p.W;
  const float* src = p.src + ((size_t)n * HW + a) * p.lds_;
  float* dst = p.dst + ((size_t)n * HW + a) * p.ldd;
  for (int c = threadIdx.x; c < T; c += 256) {
    const int hi = c / p.mW, wi = c - hi * p.mW;
    const int hb = ha + hi - p.hh, wb = wa + wi - p.hw;
    float v = 0.f;
    if ((unsigned)hb < (unsigned)p.H && (unsigned)wb < (unsigned)p.W) v = src[hb * p.W + wb];
    dst[c] = v;
  }
}

// Distribute: the in-window taps of a 64 x 64 tile of dG = dA^T, staged through LDS so that the load runs along a
// (rows of dA) and the store along b (runs of contiguous taps).  The caller zero-fills dM first (out-of-window taps).
__global__ __launch_bounds__(256) void psamask_nhwc_bwd_distribute_kernel(const PsaNhwcArgs p) {
  __shared__ float tile[PT][PT + 1];
  const int HW = p.H * p.W;
  int blk = blockIdx.x;
  const int tb = blk % p.tiles; blk /= p.tiles;
  const int ta = blk % p.tiles;
  const int n = blk / p.tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* drow = p.src + (size_t)n * HW * p.lds_;
  {
    const int a = ta * PT + lane;
#pragma unroll 4
    for (int r = wave; r < PT; r += 4) {
      const int bb = tb * PT + r;
      tile[lane][r] = (a < HW && bb < HW) ? drow[(size_t)bb * p.lds_ + a] : 0.f;   // dG[a][b] = dA[b][a]
    }
  }
  __syncthreads();
  const int b = tb * PT + lane;
  const int hb = b / p.W, wb = b - hb * p.W;
  float* mrow = p.dst + (size_t)n * HW * p.ldd;
#pragma unroll 4
  for (int r = wave; r < PT; r += 4) {
    const int a = ta * PT + r;
    const int ha = a / p.W, wa = a - ha * p.W;
    const int hi = hb - ha + p.hh, wi = wb - wa + p.hw;
    if (a < HW && b < HW && (unsigned)hi < (unsigned)p.mH && (unsigned)wi < (unsigned)p.mW)
      mrow[(size_t)a * p.ldd + hi * p.mW + wi] = tile[r][lane];
  }
}

// Collect with a pre-zeroed destination (round 6): lanes run along the SOURCE positions b of row a of dA (one coalesced read of the
// row) and every lane stores its value at its tap — runs of W contiguous taps per image row; out-of-window taps are neither
// computed nor written (the engine's gradient buffer of the attention map is zeroed once: its out-of-window taps are the same
// elements every step).  Bytes moved = 2 * 4 * N * (HW)^2, the algorithmic figure (the full-row form writes N * HW * taps more).
__global__ __launch_bounds__(256) void psamask_nhwc_bwd_collect_sparse_kernel(const PsaNhwcArgs p) {
  const int HW = p.H * p.W;
  const int a = blockIdx.x % HW, n = blockIdx.x / HW;
  const int ha = a / p.W, wa = a - ha * p.W;
  const float* src = p.src + ((size_t)n * HW + a) * p.lds_;
  float* dst = p.dst + ((size_t)n * HW + a) * p.ldd;
  for (int b = threadIdx.x; b < HW; b += 256) {
    const int hb = b / p.W, wb = b - hb * p.W;
    const int hi = hb - ha + p.hh, wi = wb - wa + p.hw;
    if ((unsigned)hi < (unsigned)p.mH && (unsigned)wi < (unsigned)p.mW) dst[hi * p.mW + wi] = src[b];
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
  if (!mask || !aff || lda < H * W || ldm < mH * mW || N <= 0) return SEMSEG_EINVAL;
  const int tiles = (H * W + PT - 1) / PT;
  const long long grid = (long long)N * tiles * tiles;
  if (grid > 2147483647LL) return SEMSEG_EINVAL;
  PsaNhwcArgs a{mask, aff, ldm, lda, psa_type ? 1 : 0, N, H, W, mH, mW, (mH - 1) / 2, (mW - 1) / 2, tiles};
  psamask_nhwc_fwd_kernel<<<(int)grid, 256, 0, stream>>>(a);
  return semseg_launch_status();
}

int semseg_psamask_nhwc_backward(int psa_type, const float* daff, int lda, float* dmask, int ldm,
                                 int N, int H, int W, int mH, int mW, int dmask_prezeroed, hipStream_t stream) {
  if (!daff || !dmask || lda < H * W || ldm < mH * mW || N <= 0) return SEMSEG_EINVAL;
  const int tiles = (H * W + PT - 1) / PT;
  PsaNhwcArgs a{daff, dmask, lda, ldm, psa_type ? 1 : 0, N, H, W, mH, mW, (mH - 1) / 2, (mW - 1) / 2, tiles};
  if (!psa_type) {
    if (dmask_prezeroed) psamask_nhwc_bwd_collect_sparse_kernel<<<N * H * W, 256, 0, stream>>>(a);
    else psamask_nhwc_bwd_collect_kernel<<<N * H * W, 256, 0, stream>>>(a);
  } else {
    // out-of-window taps are zero: clear the [N*H*W, ldm] block (unless the caller keeps them zero), then scatter the in-window tiles
    if (!dmask_prezeroed && hipMemsetAsync(dmask, 0, (size_t)N * H * W * ldm * sizeof(float), stream) != hipSuccess) return SEMSEG_ELAUNCH;
    const long long grid = (long long)N * tiles * tiles;
    if (grid > 2147483647LL) return SEMSEG_EINVAL;
    psamask_nhwc_bwd_distribute_kernel<<<(int)grid, 256, 0, stream>>>(a);
  }
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
