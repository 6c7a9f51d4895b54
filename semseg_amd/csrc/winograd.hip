// Winograd F(2x2, 3x3) transforms for the stride-1 "same" 3x3 convolutions of the path (reference: every
// nn.Conv2d(kernel_size=3, stride=1, padding=dilation, dilation=dilation) of model/resnet.py:63-69 after the surgery of
// model/pspnet.py:49-58, and the head convs model/pspnet.py:65,73) — 67 % of the step's FLOPs.
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input patch, summed over input channels
//
// turns one 3x3 convolution into 16 independent GEMMs [tiles x Ci] x [Ci x Co] (one per position of the 4x4
// transformed patch) with 16 / 36 = 1 / 2.25 of the multiplications; the GEMMs run on the existing fp32 matrix-core
// kernels as ONE batched launch (semseg_gemm_rows_batched / semseg_gemm_kmajor_batched, batch = 16), the transforms are
// the HBM-bound kernels of this file.  Everything is fp32 (the transforms are additions and multiplications by 1/2).
//
// Dilation d: output pixel (oy, ox) only touches input pixels of its own phase (oy mod d, ox mod d), so the image
// splits into d*d sub-images on which the convolution is dense; tiles are cut per phase.  A tile is identified by
//   t = (((n * d + ry) * d + rx) * th + ty) * tw + tx,   th = ceil(ceil(H / d) / 2), tw likewise
// and covers sub-image rows 2*ty, 2*ty+1 (image rows d*(2*ty + a) + ry); patch rows are sub-image rows 2*ty-1 .. 2*ty+2.
// Rows / columns outside the image read as zero (that IS the convolution's zero padding) and are not written.
//
// Layouts (all fp32): V[16][T][C] transformed patches (GEMM A operand, K = C contiguous), U[16][Co_pad][Ci] transformed
// filters (GEMM B^T operand), M[16][T][Co] products, so that element e = 4*i + j of the 4x4 patch is batch item e.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

struct WinoGeo {
  int N, H, W, d, th, tw, T;
};

__host__ __device__ inline WinoGeo make_geo(int N, int H, int W, int d) {
  WinoGeo g;
  g.N = N; g.H = H; g.W = W; g.d = d;
  g.th = ((H + d - 1) / d + 1) / 2;
  g.tw = ((W + d - 1) / d + 1) / 2;
  g.T = N * d * d * g.th * g.tw;
  return g;
}

// tile index -> image, phase, tile position; returns the image row / column of sub-image coordinate 0 and the step d
__device__ __forceinline__ void decode_tile(const WinoGeo& g, int t, int& n, int& y0, int& x0) {
  const int tx = t % g.tw; t /= g.tw;
  const int ty = t % g.th; t /= g.th;
  const int rx = t % g.d; t /= g.d;
  const int ry = t % g.d;
  n = t / g.d;
  y0 = g.d * (2 * ty) + ry;   // image row of the tile's first OUTPUT row; patch rows are y0 + (i - 1) * d
  x0 = g.d * (2 * tx) + rx;
}

// V = B^T d B for one 4x4 patch of float4 (4 channels per thread).  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1].
__device__ __forceinline__ void bt_d_b(const f32x4 (&d)[4][4], f32x4 (&v)[4][4]) {
  f32x4 t[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = d[0][j] - d[2][j];
    t[1][j] = d[1][j] + d[2][j];
    t[2][j] = d[2][j] - d[1][j];
    t[3][j] = d[1][j] - d[3][j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i][0] = t[i][0] - t[i][2];
    v[i][1] = t[i][1] + t[i][2];
    v[i][2] = t[i][2] - t[i][1];
    v[i][3] = t[i][1] - t[i][3];
  }
}

// src NHWC [N][H][W][lds] (C channels, C % 4 == 0) -> V[16][T][C].  One thread = one tile x 4 channels.
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ src, int lds, float* __restrict__ V,
                                                         const WinoGeo g, int C) {
  const int C4 = C >> 2;
  const long long total = (long long)g.T * C4;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C4) * 4;
    const int t = (int)(idx / C4);
    int n, y0, x0;
    decode_tile(g, t, n, y0, x0);
    f32x4 d[4][4], v[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = y0 + (i - 1) * g.d;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = x0 + (j - 1) * g.d;
        const bool ok = (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
        d[i][j] = ok ? *reinterpret_cast<const f32x4*>(src + ((size_t)(n * g.H + y) * g.W + x) * lds + c)
                     : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    bt_d_b(d, v);
    const size_t plane = (size_t)g.T * C;
    float* o = V + (size_t)t * C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + j) * plane) = v[i][j];
  }
}

// Output-gradient transform of the weight gradient: Yh = A dY A^T (4x4 from the tile's 2x2 output gradients),
// A = [1 0; 1 1; 1 -1; 0 -1].  dy NHWC [N][H][W][lddy] -> Yh[16][T][ldo] (columns >= C are left untouched: the caller
// keeps them zero).
__global__ __launch_bounds__(256) void wino_dy_wgrad_kernel(const float* __restrict__ dy, int lddy, float* __restrict__ Yh,
                                                            int ldo, const WinoGeo g, int C) {
  const int C4 = C >> 2;
  const long long total = (long long)g.T * C4;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C4) * 4;
    const int t = (int)(idx / C4);
    int n, y0, x0;
    decode_tile(g, t, n, y0, x0);
    f32x4 q[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int y = y0 + a * g.d, x = x0 + b * g.d;
        const bool ok = y < g.H && x < g.W;
        q[a][b] = ok ? *reinterpret_cast<const f32x4*>(dy + ((size_t)(n * g.H + y) * g.W + x) * lddy + c)
                     : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    // r = A q (4x2), then Yh = r A^T (4x4)
    f32x4 r[4][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      r[0][b] = q[0][b];
      r[1][b] = q[0][b] + q[1][b];
      r[2][b] = q[0][b] - q[1][b];
      r[3][b] = -q[1][b];
    }
    const size_t plane = (size_t)g.T * ldo;
    float* o = Yh + (size_t)t * ldo + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 0) * plane) = r[i][0];
      *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 1) * plane) = r[i][0] + r[i][1];
      *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 2) * plane) = r[i][0] - r[i][1];
      *reinterpret_cast<f32x4*>(o + (size_t)(4 * i + 3) * plane) = -r[i][1];
    }
  }
}

// Y = A^T M A (2x2 from 4x4), A^T = [1 1 1 0; 0 1 -1 -1].  M[16][T][ldm] -> y NHWC [N][H][W][ldy] (+ add), optional fp64
// per-channel statistics [nslot][2C] of the stored values (before add), as the conv epilogue produces them; optional
// per-channel scale / shift and ReLU (the eval-mode BatchNorm + ReLU + residual epilogue of the direct kernel).
struct WinoOutArgs {
  const float* M;
  float* y;
  const float* add;
  double* stats;
  int ldm, ldy, ldadd, C, nslot;
  WinoGeo g;
  // Fused BatchNorm-backward reduction (data gradient of a conv whose input is a BatchNorm(+ReLU) output and whose
  // gradient this launch completes; same contract as semseg_conv_dgrad_bnreduce): what is stored is
  // g = (Y + add) * (act > 0), and stats += {sum g, sum g * (ybn - mean) * invstd}.  bnr == 0: off.
  // eval-mode epilogue (BatchNorm with running statistics, ReLU, residual folded in): y = [relu](Y * scale + shift + add)
  const float* scale;
  const float* shift;
  int relu;
  int bnr;
  const float* act;      // post-ReLU activation (nullptr: no ReLU)
  const unsigned* bits;  // the ReLU mask as bits (semseg_bn_apply's relu_bits; non-null: replaces act)
  int ldbits;
  const float* ybn;      // pre-BatchNorm tensor
  const float* mean;
  const float* invstd;
  int ldact, ldybn;
};

__global__ __launch_bounds__(256) void wino_output_kernel(const WinoOutArgs p) {
  // thread = (tile row chunk, 4 channels): blockDim 256 = tpr channel-threads x rpb tile-threads (tpr = min(C/4, 256))
  __shared__ double sred[256 * 8];
  const int C4 = p.C >> 2;
  int tpr = 1;
  while (tpr * 2 <= C4 && tpr * 2 <= 256) tpr *= 2;
  const int rpb = 256 / tpr;
  const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
  const int c4 = blockIdx.x * tpr + tc;
  const bool active = c4 < C4;
  const int c = c4 * 4;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const WinoGeo g = p.g;
  f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, bis = {0.f, 0.f, 0.f, 0.f};
  f32x4 esc = {1.f, 1.f, 1.f, 1.f}, esh = {0.f, 0.f, 0.f, 0.f};
  if (active && p.scale) esc = *reinterpret_cast<const f32x4*>(p.scale + c);
  if (active && p.shift) esh = *reinterpret_cast<const f32x4*>(p.shift + c);
  if (active && p.bnr) {
    bmu = *reinterpret_cast<const f32x4*>(p.mean + c);
    bis = *reinterpret_cast<const f32x4*>(p.invstd + c);
  }
  if (active) {
    const size_t plane = (size_t)g.T * p.ldm;
    for (int t = blockIdx.y * rpb + tr; t < g.T; t += gridDim.y * rpb) {
      int n, y0, x0;
      decode_tile(g, t, n, y0, x0);
      const float* m = p.M + (size_t)t * p.ldm + c;
      f32x4 s[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(m + (size_t)(0 + j) * plane);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(m + (size_t)(4 + j) * plane);
        const f32x4 m2 = *reinterpret_cast<const f32x4*>(m + (size_t)(8 + j) * plane);
        const f32x4 m3 = *reinterpret_cast<const f32x4*>(m + (size_t)(12 + j) * plane);
        s[0][j] = m0 + m1 + m2;
        s[1][j] = m1 - m2 - m3;
      }
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int y = y0 + a * g.d;
        if (y >= g.H) continue;
        const f32x4 o0 = s[a][0] + s[a][1] + s[a][2];
        const f32x4 o1 = s[a][1] - s[a][2] - s[a][3];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int x = x0 + b * g.d;
          if (x >= g.W) continue;
          f32x4 v = b == 0 ? o0 : o1;
          const size_t pix = (size_t)(n * g.H + y) * g.W + x;
          if (p.bnr) {
            if (p.add) v += *reinterpret_cast<const f32x4*>(p.add + pix * p.ldadd + c);
            if (p.bits) {
              const unsigned wq = p.bits[pix * p.ldbits + (c >> 5)] >> (c & 31);
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = ((wq >> k) & 1u) ? v[k] : 0.f;
            } else if (p.act) {
              const f32x4 a4 = *reinterpret_cast<const f32x4*>(p.act + pix * p.ldact + c);
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = a4[k] > 0.f ? v[k] : 0.f;
            }
            const f32x4 xh = (*reinterpret_cast<const f32x4*>(p.ybn + pix * p.ldybn + c) - bmu) * bis;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              acc[k] += (double)v[k];
              acc[4 + k] += (double)v[k] * (double)xh[k];
            }
            *reinterpret_cast<f32x4*>(p.y + pix * p.ldy + c) = v;
            continue;
          }
          v = v * esc + esh;
          if (p.stats) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const double dv = (double)v[k];
              acc[k] += dv;
              acc[4 + k] += dv * dv;
            }
          }
          if (p.add) v += *reinterpret_cast<const f32x4*>(p.add + pix * p.ldadd + c);
          if (p.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
          }
          *reinterpret_cast<f32x4*>(p.y + pix * p.ldy + c) = v;
        }
      }
    }
  }
  if (p.stats) {
    if (rpb > 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) sred[(tr * tpr + tc) * 8 + k] = acc[k];
      __syncthreads();
      if (tr == 0)
        for (int r = 1; r < rpb; ++r)
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] += sred[(r * tpr + tc) * 8 + k];
    }
    if (tr == 0 && active) {
      double* st = p.stats + (size_t)((blockIdx.x + blockIdx.y) % p.nslot) * 2 * p.C;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomic_add_f64(&st[c + k], acc[k]);
        atomic_add_f64(&st[p.C + c + k], acc[4 + k]);
      }
    }
  }
}

// U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].  w OIHW [Co][Ci][3][3].
//   flip == 0 (forward):        U[e][co][ci]  rows = output channels (Rows_pad >= Co), K = Ci
//   flip == 1 (data gradient):  U[e][ci][co]  with the taps rotated by 180 degrees, rows = input channels, K = Kc >= Co
// Rows >= the valid count and K columns >= the valid count are written as zero.
__global__ __launch_bounds__(256) void wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int Co, int Ci,
                                                          int rows_pad, int Kc, int flip) {
  const long long total = (long long)rows_pad * Kc;
  const size_t plane = (size_t)rows_pad * Kc;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int k = (int)(idx % Kc), row = (int)(idx / Kc);
    const int co = flip ? k : row, ci = flip ? row : k;
    float g[3][3];
    const bool ok = co < Co && ci < Ci;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int rr = flip ? 2 - r : r, ss = flip ? 2 - s : s;
        g[r][s] = ok ? w[((size_t)co * Ci + ci) * 9 + rr * 3 + ss] : 0.f;
      }
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      t[0][s] = g[0][s];
      t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
      t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
      t[3][s] = g[2][s];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      U[(size_t)(4 * i + 0) * plane + idx] = t[i][0];
      U[(size_t)(4 * i + 1) * plane + idx] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      U[(size_t)(4 * i + 2) * plane + idx] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      U[(size_t)(4 * i + 3) * plane + idx] = t[i][2];
    }
  }
}

// All filter transforms of a network in ONE launch (31 convs x 2 panels per PSPNet-101 training step): block b finds
// its panel by binary search over the block-start table, 256 (row, k) pairs per block.
__global__ __launch_bounds__(256) void wino_filter_multi_kernel(const SemsegWinoFilterDesc* __restrict__ descs,
                                                                const int* __restrict__ starts, int nseg) {
  int lo = 0, hi = nseg - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const SemsegWinoFilterDesc d = descs[lo];
  const long long idx = (long long)(b - starts[lo]) * 256 + threadIdx.x;
  const long long total = (long long)d.rows_pad * d.Kc;
  if (idx >= total) return;
  const size_t plane = (size_t)total;
  const int k = (int)(idx % d.Kc), row = (int)(idx / d.Kc);
  const int co = d.flip ? k : row, ci = d.flip ? row : k;
  float g[3][3];
  const bool ok = co < d.Co && ci < d.Ci;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int rr = d.flip ? 2 - r : r, ss = d.flip ? 2 - s : s;
      g[r][s] = ok ? d.w[((size_t)co * d.Ci + ci) * 9 + rr * 3 + ss] : 0.f;
    }
  float t[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t[0][s] = g[0][s];
    t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
    t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
    t[3][s] = g[2][s];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    d.U[(size_t)(4 * i + 0) * plane + idx] = t[i][0];
    d.U[(size_t)(4 * i + 1) * plane + idx] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
    d.U[(size_t)(4 * i + 2) * plane + idx] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
    d.U[(size_t)(4 * i + 3) * plane + idx] = t[i][2];
  }
}

// dW = G^T dU G (3x3 from 4x4): dU[16][Co][Ci] -> OIHW gradient [Co][Ci][3][3] (= or +=).
__global__ __launch_bounds__(256) void wino_filter_grad_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Co,
                                                               int Ci, int accumulate) {
  const long long total = (long long)Co * Ci;
  const size_t plane = (size_t)Co * Ci;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    float u[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) u[i][j] = dU[(size_t)(4 * i + j) * plane + idx];
    // t = G^T u (3x4): G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = u[0][j] + 0.5f * (u[1][j] + u[2][j]);
      t[1][j] = 0.5f * (u[1][j] - u[2][j]);
      t[2][j] = 0.5f * (u[1][j] + u[2][j]) + u[3][j];
    }
    float* o = dw + (size_t)idx * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float g0 = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
      const float g1 = 0.5f * (t[r][1] - t[r][2]);
      const float g2 = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
      if (accumulate) {
        o[r * 3 + 0] += g0; o[r * 3 + 1] += g1; o[r * 3 + 2] += g2;
      } else {
        o[r * 3 + 0] = g0; o[r * 3 + 1] = g1; o[r * 3 + 2] = g2;
      }
    }
  }
}

int launch_output(const WinoOutArgs& a, hipStream_t stream);

inline int grid_1d(long long total) {
  long long g = (total + 255) / 256;
  if (g > 65535 * 16) g = 65535 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" {

int semseg_wino_tiles(int N, int H, int W, int dil) {
  if (N <= 0 || H <= 0 || W <= 0 || dil <= 0) return -1;
  const long long T = (long long)make_geo(N, H, W, dil).T;
  return T > 2147483647LL ? -1 : (int)T;
}

int semseg_wino_input_transform(const float* src, int lds, float* V, int N, int H, int W, int C, int dil,
                                hipStream_t stream) {
  if (!src || !V || (C & 3) || (lds & 3) || lds < C || semseg_wino_tiles(N, H, W, dil) < 0) return SEMSEG_EINVAL;
  const WinoGeo g = make_geo(N, H, W, dil);
  wino_input_kernel<<<grid_1d((long long)g.T * (C >> 2)), 256, 0, stream>>>(src, lds, V, g, C);
  return semseg_launch_status();
}

int semseg_wino_dy_transform_wgrad(const float* dy, int lddy, float* Yh, int ldo, int N, int H, int W, int C, int dil,
                                   hipStream_t stream) {
  if (!dy || !Yh || (C & 3) || (lddy & 3) || (ldo & 3) || lddy < C || ldo < C || semseg_wino_tiles(N, H, W, dil) < 0)
    return SEMSEG_EINVAL;
  const WinoGeo g = make_geo(N, H, W, dil);
  wino_dy_wgrad_kernel<<<grid_1d((long long)g.T * (C >> 2)), 256, 0, stream>>>(dy, lddy, Yh, ldo, g, C);
  return semseg_launch_status();
}

int semseg_wino_output_transform(const float* M, int ldm, float* y, int ldy, const float* add, int ldadd, double* stats,
                                 int nslot, const float* scale, const float* shift, int relu, int N, int H, int W, int C,
                                 int dil, hipStream_t stream) {
  if (!M || !y || (C & 3) || (ldm & 3) || (ldy & 3) || ldm < C || ldy < C || (add && (ldadd & 3)) ||
      (stats && nslot < 1) || semseg_wino_tiles(N, H, W, dil) < 0)
    return SEMSEG_EINVAL;
  WinoOutArgs a;
  a.M = M; a.y = y; a.add = add; a.stats = stats; a.ldm = ldm; a.ldy = ldy; a.ldadd = ldadd; a.C = C;
  a.nslot = nslot > 0 ? nslot : 1;
  a.g = make_geo(N, H, W, dil);
  a.bnr = 0; a.act = nullptr; a.bits = nullptr; a.ldbits = 0; a.ybn = nullptr; a.mean = nullptr; a.invstd = nullptr; a.ldact = 0; a.ldybn = 0;
  a.scale = scale; a.shift = shift; a.relu = relu;
  return launch_output(a, stream);
}

int semseg_wino_output_transform_bnreduce(const float* M, int ldm, float* y, int ldy, const float* add, int ldadd,
                                          const float* act, int ldact, const unsigned* relu_bits, int ldbits,
                                          const float* ybn, int ldybn, const float* mean,
                                          const float* invstd, double* sums, int nslot, int N, int H, int W, int C,
                                          int dil, hipStream_t stream) {
  if (!M || !y || !ybn || !mean || !invstd || !sums || nslot < 1 || (C & 3) || (ldm & 3) || (ldy & 3) || ldm < C ||
      ldy < C || (add && (ldadd & 3)) || (act && (ldact & 3)) || (ldybn & 3) || semseg_wino_tiles(N, H, W, dil) < 0 ||
      (relu_bits && ((C & 31) || ldbits * 32 < C)))
    return SEMSEG_EINVAL;
  WinoOutArgs a;
  a.M = M; a.y = y; a.add = add; a.stats = sums; a.ldm = ldm; a.ldy = ldy; a.ldadd = ldadd; a.C = C; a.nslot = nslot;
  a.g = make_geo(N, H, W, dil);
  a.bnr = 1; a.act = relu_bits ? nullptr : act; a.bits = relu_bits; a.ldbits = ldbits; a.ybn = ybn; a.mean = mean; a.invstd = invstd; a.ldact = ldact; a.ldybn = ldybn;
  a.scale = nullptr; a.shift = nullptr; a.relu = 0;
  return launch_output(a, stream);
}

}  // extern "C"

namespace {
int launch_output(const WinoOutArgs& a, hipStream_t stream) {
  const int C = a.C;
  const double* stats = a.stats;
  const int C4 = C >> 2;
  int tpr = 1;
  while (tpr * 2 <= C4 && tpr * 2 <= 256) tpr *= 2;
  const int rpb = 256 / tpr, gx = (C4 + tpr - 1) / tpr;
  int gy = (a.g.T + rpb - 1) / rpb;
  // few blocks per channel group: every block ends in one fp64 atomic pair per channel, and same-address atomics
  // serialise (layer3's output transform, us, by the cap: 4096: 85, 2048: 69, 1024: 57, 512: 54, 256: 58)
  int cap = 512 / gx;
  if (cap < 1) cap = 1;
  if (!stats) cap = 65535;
  if (gy > cap) gy = cap;
  if (gy > 65535) gy = 65535;
  wino_output_kernel<<<dim3(gx, gy), 256, 0, stream>>>(a);
  return semseg_launch_status();
}
}  // namespace

extern "C" {

int semseg_wino_filter_transform(const float* w_oihw, float* U, int Co, int Ci, int rows_pad, int Kc, int flip,
                                 hipStream_t stream) {
  if (!w_oihw || !U || Co <= 0 || Ci <= 0) return SEMSEG_EINVAL;
  if (!flip && (rows_pad < Co || Kc < Ci)) return SEMSEG_EINVAL;
  if (flip && (rows_pad < Ci || Kc < Co)) return SEMSEG_EINVAL;
  wino_filter_kernel<<<grid_1d((long long)rows_pad * Kc), 256, 0, stream>>>(w_oihw, U, Co, Ci, rows_pad, Kc, flip);
  return semseg_launch_status();
}

int semseg_wino_filter_transform_multi(const SemsegWinoFilterDesc* descs_dev, const int* block_starts_dev, int npanels,
                                       int total_blocks, hipStream_t stream) {
  if (!descs_dev || !block_starts_dev || npanels < 1 || total_blocks < 1) return SEMSEG_EINVAL;
  wino_filter_multi_kernel<<<total_blocks, 256, 0, stream>>>(descs_dev, block_starts_dev, npanels);
  return semseg_launch_status();
}

int semseg_wino_filter_grad(const float* dU, float* dw_oihw, int Co, int Ci, int accumulate, hipStream_t stream) {
  if (!dU || !dw_oihw || Co <= 0 || Ci <= 0) return SEMSEG_EINVAL;
  wino_filter_grad_kernel<<<grid_1d((long long)Co * Ci), 256, 0, stream>>>(dU, dw_oihw, Co, Ci, accumulate);
  return semseg_launch_status();
}

}  // extern "C"
