// BatchNorm2d / SyncBatchNorm (train + eval), fused with ReLU, residual add, the downsample branch
// and Dropout2d, over NHWC fp32 activations.  Reference semantics: torch nn.BatchNorm2d as built at
// model/resnet.py:64,67,69,109-113,136 and model/pspnet.py:16,66,74 (eps 1e-5, momentum 0.1, biased
// variance for normalisation, unbiased into running_var), Bottleneck tail model/resnet.py:88-92,
// Dropout2d model/pspnet.py:68,76.
//
// All per-channel statistics are accumulated in fp64 (per-thread fp64 partials, fp64 hardware
// atomics) so E[x^2]-E[x]^2 has no cancellation problem; the [2C] fp64 sum vectors are also exactly
// what SyncBN all-reduces over RCCL (tool/train.py:142 behaviour).  These kernels are HBM-bound:
// every access is a 16-byte lane access, consecutive lanes on consecutive channels.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

struct Tiling {
  int tpr;    // threads along channels (power of two <= 256)
  int rpb;    // rows per block pass
  int gx;     // column groups
};

inline Tiling make_tiling(int CV) {
  Tiling t;
  int tpr = 1;
  while (tpr * 2 <= CV && tpr * 2 <= 256) tpr *= 2;
  t.tpr = tpr;
  t.rpb = 256 / tpr;
  t.gx = (CV + tpr - 1) / tpr;
  return t;
}

// Same-address fp64 atomics serialise at ~70 ns each, so the per-channel accumulators are replicated
// NSLOT times ([nslot][2C], slot = block % nslot) and combined by the consumer kernel.
inline int rows_grid(int M, int rpb, int gx) {
  int gy = (M + rpb * 4 - 1) / (rpb * 4);
  int cap = 768 / gx;
  if (cap < 1) cap = 1;
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  return gy;
}

// block-level reduce of NV doubles per thread across the rpb row-lanes, then fp64 atomics
template <int NV>
__device__ __forceinline__ void block_reduce_atomic(double (&v)[NV], int tpr, int rpb, int tr,
                                                    int tc, double* smem,
                                                    double* const (&dst)[NV], bool active) {
  // smem: [rpb][tpr][NV]
  if (rpb > 1) {
#pragma unroll
    for (int k = 0; k < NV; ++k) smem[(tr * tpr + tc) * NV + k] = v[k];
    __syncthreads();
    if (tr == 0) {
      for (int r = 1; r < rpb; ++r)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] += smem[(r * tpr + tc) * NV + k];
    }
  }
  if (tr == 0 && active) {
#pragma unroll
    for (int k = 0; k < NV; ++k) atomic_add_f64(dst[k], v[k]);
  }
}

__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, int ldx,
                                                            double* __restrict__ stats, int M,
                                                            int C, int tpr, int rpb, int nslot) {
  stats += (size_t)((blockIdx.x + blockIdx.y) % nslot) * 2 * C;
  extern __shared__ double sred[];
  const int CV = C >> 2;
  const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
  const int c4 = blockIdx.x * tpr + tc;
  const bool active = c4 < CV;
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (active) {
    constexpr int U = 4;
    const int step = gridDim.y * rpb;
    for (int mb = blockIdx.y * rpb + tr; mb < M; mb += U * step) {
      f32x4 a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int m = mb + u * step;
        a[u] = *reinterpret_cast<const f32x4*>(x + (size_t)(m < M ? m : mb) * ldx + c4 * 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (mb + u * step < M) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double d = (double)a[u][k];
            v[k] += d;
            v[4 + k] += d * d;
          }
        }
    }
  }
  double* dst[8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    dst[k] = stats + c4 * 4 + k;
    dst[4 + k] = stats + C + c4 * 4 + k;
  }
  block_reduce_atomic<8>(v, tpr, rpb, tr, tc, sred, dst, active);
}

// sums -> mean / invstd / scale / shift; running-stat update (train)
__global__ void bn_combine_kernel(const double* stats, int nslot, int C2, double* dst) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C2) return;
  double v = stats[c];
  for (int s = 1; s < nslot; ++s) v += stats[(size_t)s * C2 + c];
  dst[c] = v;
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, int nslot, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, long long* nbt,
                                   float momentum, float eps, float* __restrict__ mean,
                                   float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int s = 0; s < nslot; ++s) {
    s1 += stats[(size_t)s * 2 * C + c];
    s2 += stats[(size_t)s * 2 * C + C + c];
  }
  const double mu = s1 / count;
  double var = s2 / count - mu * mu;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float m = (float)mu;
  mean[c] = m;
  invstd[c] = is;
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - m * sc;
  if (running_mean) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

__global__ void bn_eval_params_kernel(const float* __restrict__ gamma,
                                      const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv,
                                      float eps, float* __restrict__ scale,
                                      float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

struct ApplyArgs {
  const float* y; const float* scale; const float* shift;
  const float* y2; const float* scale2; const float* shift2;
  const float* res; const float* dropmask;
  float* out;
  int ldy, ldy2, ldres, ldout;
  int M, C, HW, relu;
  unsigned* bits;      // optional: bit (c & 31) of word [m][c >> 5] = (value before the ReLU > 0); C % 32 == 0
  int ldbits;
};

__global__ __launch_bounds__(256) void bn_apply_kernel(const ApplyArgs p) {
  const int CV = p.C >> 2;
  const size_t total = (size_t)p.M * CV;
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t idx0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool fixed = stride % CV == 0;       // this thread keeps its 4 channels: per-channel vectors loaded once
  const int c0 = (int)(idx0 % CV) * 4;
  f32x4 sc0, sh0, sc20, sh20;
  if (fixed && idx0 < total) {
    sc0 = *reinterpret_cast<const f32x4*>(p.scale + c0);
    sh0 = *reinterpret_cast<const f32x4*>(p.shift + c0);
    if (p.y2) {
      sc20 = *reinterpret_cast<const f32x4*>(p.scale2 + c0);
      sh20 = *reinterpret_cast<const f32x4*>(p.shift2 + c0);
    }
  }
  for (size_t idx = idx0; idx < total; idx += stride) {
    const int m = (int)(idx / CV);
    const int c = fixed ? c0 : (int)(idx - (size_t)m * CV) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(p.y + (size_t)m * p.ldy + c);
    const f32x4 sc = fixed ? sc0 : *reinterpret_cast<const f32x4*>(p.scale + c);
    const f32x4 sh = fixed ? sh0 : *reinterpret_cast<const f32x4*>(p.shift + c);
    v = v * sc + sh;
    if (p.y2) {
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(p.y2 + (size_t)m * p.ldy2 + c);
      const f32x4 sc2 = fixed ? sc20 : *reinterpret_cast<const f32x4*>(p.scale2 + c);
      const f32x4 sh2 = fixed ? sh20 : *reinterpret_cast<const f32x4*>(p.shift2 + c);
      v += v2 * sc2 + sh2;
    }
    if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.ldres + c);
    if (p.bits) {
      // the 8 lanes that hold one 32-channel word of this pixel are adjacent and aligned (CV % 8 == 0, strides % 8 == 0):
      // each contributes its 4 bits, an OR over the group assembles the word, the first lane stores it
      const int sub = (c >> 2) & 7;
      unsigned w = ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u))
                   << (4 * sub);
      w |= __shfl_xor(w, 1);
      w |= __shfl_xor(w, 2);
      w |= __shfl_xor(w, 4);
      if (sub == 0) p.bits[(size_t)m * p.ldbits + (c >> 5)] = w;
    }
    if (p.relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
    }
    if (p.dropmask) {
      const int n = m / p.HW;
      v *= *reinterpret_cast<const f32x4*>(p.dropmask + (size_t)n * p.C + c);
    }
    *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.ldout + c) = v;
  }
}

struct BwdReduceArgs {
  const float* dout; const float* out; const float* dropmask;
  const float* y; const float* mean; const float* invstd;
  float* g; double* sums;
  int lddout, ldout, ldy, ldg;
  int M, C, HW, tpr, rpb, nslot;
};

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const BwdReduceArgs p) {
  extern __shared__ double sred[];
  const int CV = p.C >> 2;
  const int tc = threadIdx.x % p.tpr, tr = threadIdx.x / p.tpr;
  const int c4 = blockIdx.x * p.tpr + tc;
  const bool active = c4 < CV;
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (active) {
    const int c = c4 * 4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(p.invstd + c);
    // U rows in flight per thread: all loads of a group are issued before any is consumed
    constexpr int U = 4;
    const int step = gridDim.y * p.rpb;
    const f32x4 dmz = {1.f, 1.f, 1.f, 1.f};
    for (int mb = blockIdx.y * p.rpb + tr; mb < p.M; mb += U * step) {
      f32x4 gg[U], oo[U], yy[U], dd[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int m = mb + u * step;
        const int mc = m < p.M ? m : mb;
        gg[u] = *reinterpret_cast<const f32x4*>(p.dout + (size_t)mc * p.lddout + c);
        yy[u] = *reinterpret_cast<const f32x4*>(p.y + (size_t)mc * p.ldy + c);
        oo[u] = p.out ? *reinterpret_cast<const f32x4*>(p.out + (size_t)mc * p.ldout + c) : dmz;
        dd[u] = p.dropmask ? *reinterpret_cast<const f32x4*>(p.dropmask + (size_t)(mc / p.HW) * p.C + c) : dmz;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int m = mb + u * step;
        if (m < p.M) {
          f32x4 g = gg[u] * dd[u];
#pragma unroll
          for (int k = 0; k < 4; ++k) g[k] = oo[u][k] > 0.f ? g[k] : 0.f;
          if (p.g) *reinterpret_cast<f32x4*>(p.g + (size_t)m * p.ldg + c) = g;
          const f32x4 xh = (yy[u] - mu) * is;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v[k] += (double)g[k];
            v[4 + k] += (double)g[k] * (double)xh[k];
          }
        }
      }
    }
  }
  double* dst[8];
  double* sl = p.sums + (size_t)((blockIdx.x + blockIdx.y) % p.nslot) * 2 * p.C;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    dst[k] = sl + c4 * 4 + k;
    dst[4 + k] = sl + p.C + c4 * 4 + k;
  }
  block_reduce_atomic<8>(v, p.tpr, p.rpb, tr, tc, sred, dst, active);
}

struct BwdApplyArgs {
  const float* g; const float* y; const float* mean; const float* invstd; const float* gamma;
  const double* sums;
  float* dy;
  double inv_count;
  int ldg, ldy, lddy;
  int M, C;
};

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BwdApplyArgs p) {
  const int CV = p.C >> 2;
  const size_t total = (size_t)p.M * CV;
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t idx0 = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (stride % CV == 0) {
    // Every power-of-two channel count of the network: a thread keeps the same 4 channels over its whole row walk, so
    // the 8 fp64 sums and the 3 parameter vectors are read ONCE per thread instead of once per element.
    if (idx0 >= total) return;
    const int c = (int)(idx0 % CV) * 4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(p.invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
    f32x4 A, mg, AX;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mg[k] = (float)(p.sums[c + k] * p.inv_count);
      A[k] = ga[k] * is[k];
      AX[k] = A[k] * (float)(p.sums[p.C + c + k] * p.inv_count);
    }
    const int mstep = (int)(stride / CV);
    constexpr int U = 4;   // rows in flight per thread
    for (int m = (int)(idx0 / CV); m < p.M; m += U * mstep) {
      f32x4 g[U], yy[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int mm = m + u * mstep < p.M ? m + u * mstep : m;
        g[u] = *reinterpret_cast<const f32x4*>(p.g + (size_t)mm * p.ldg + c);
        yy[u] = *reinterpret_cast<const f32x4*>(p.y + (size_t)mm * p.ldy + c);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int mm = m + u * mstep;
        if (mm < p.M) {
          const f32x4 xh = (yy[u] - mu) * is;
          const f32x4 r = A * (g[u] - mg) - AX * xh;      // = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat))
          *reinterpret_cast<f32x4*>(p.dy + (size_t)mm * p.lddy + c) = r;
        }
      }
    }
    return;
  }
  for (size_t idx = idx0; idx < total; idx += stride) {
    const int m = (int)(idx / CV);
    const int c = (int)(idx - (size_t)m * CV) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(p.g + (size_t)m * p.ldg + c);
    const f32x4 yy = *reinterpret_cast<const f32x4*>(p.y + (size_t)m * p.ldy + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(p.invstd + c);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
    f32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float mg = (float)(p.sums[c + k] * p.inv_count);
      const float mgx = (float)(p.sums[p.C + c + k] * p.inv_count);
      const float xh = (yy[k] - mu[k]) * is[k];
      r[k] = ga[k] * is[k] * (g[k] - mg - xh * mgx);
    }
    *reinterpret_cast<f32x4*>(p.dy + (size_t)m * p.lddy + c) = r;
  }
}

// combines the nslot partial vectors into slot 0 (what bn_bwd_apply / the SyncBN all-reduce read)
// and emits the parameter gradients from the LOCAL sums
__global__ void bn_param_grads_kernel(double* sums, int nslot, float* dgamma, float* dbeta,
                                      int C, int accumulate, double* folded) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = sums[c], s2 = sums[C + c];
  for (int s = 1; s < nslot; ++s) {
    s1 += sums[(size_t)s * 2 * C + c];
    s2 += sums[(size_t)s * 2 * C + C + c];
  }
  folded[c] = s1;
  folded[C + c] = s2;
  const float dg = (float)s2, db = (float)s1;
  dgamma[c] = accumulate ? dgamma[c] + dg : dg;
  dbeta[c] = accumulate ? dbeta[c] + db : db;
}

// ---- Small-grid forms (round 6): the statistics -> scale / shift step (bn_finalize) folded into the apply kernel and the
// parameter gradients (bn_param_grads) into the backward apply kernel.  At a small per-GPU batch (M = 7 200 pixels per image
// pair: what each of 8 GPUs runs) a BatchNorm layer is four launches of 5-12 us, all latency; these forms make it two.  Every
// thread keeps 4 channels over a strided row walk and derives scale / shift (sums of g) for them from the [nslot][2C] fp64
// vector itself — the SAME expressions in the same order as bn_finalize_kernel / bn_bwd_apply_kernel, so the values are
// bit-identical to the separate kernels' — and the threads of the first row group write what only one thread may write:
// mean / invstd (backward reads them), the running statistics, num_batches_tracked; dgamma / dbeta.  The launcher bounds the
// grid (<= 512 workgroups) so that the per-thread derivation is amortised over >= 4 rows wherever the tensor has them; the
// engine selects these forms only where nslot <= 2 (few producer workgroups per address), see Engine.nslot_for.
struct ApplyTrainArgs {
  const float* y; const double* stats; const float* gamma; const float* beta;
  float* running_mean; float* running_var; long long* nbt;
  float* mean; float* invstd;
  const float* res; const float* dropmask;
  float* out;
  unsigned* bits;
  double count;
  float momentum, eps;
  int nslot, ldy, ldres, ldout, ldbits, M, C, HW, relu, tpr, rpb;
};

__global__ __launch_bounds__(256) void bn_apply_train_kernel(const ApplyTrainArgs p) {
  const int CV = p.C >> 2;
  const int tc = threadIdx.x % p.tpr, tr = threadIdx.x / p.tpr;
  const int c4 = blockIdx.x * p.tpr + tc;
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && p.nbt) *p.nbt += 1;
  if (c4 >= CV) return;
  const int c = c4 * 4;
  f32x4 sc, sh;
  const bool writer = blockIdx.y == 0 && tr == 0;
  const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
  const f32x4 be = *reinterpret_cast<const f32x4*>(p.beta + c);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double s1 = 0.0, s2 = 0.0;
    for (int s = 0; s < p.nslot; ++s) {
      s1 += p.stats[(size_t)s * 2 * p.C + c + k];
      s2 += p.stats[(size_t)s * 2 * p.C + p.C + c + k];
    }
    const double mu = s1 / p.count;
    double var = s2 / p.count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)p.eps));
    const float m = (float)mu;
    sc[k] = ga[k] * is;
    sh[k] = be[k] - m * sc[k];
    if (writer) {
      p.mean[c + k] = m;
      p.invstd[c + k] = is;
      if (p.running_mean) {
        const double unb = p.count > 1.0 ? var * p.count / (p.count - 1.0) : var;
        p.running_mean[c + k] = (1.f - p.momentum) * p.running_mean[c + k] + p.momentum * m;
        p.running_var[c + k] = (1.f - p.momentum) * p.running_var[c + k] + p.momentum * (float)unb;
      }
    }
  }
  constexpr int U = 4;
  const int step = gridDim.y * p.rpb;
  const int sub = c4 & 7;
  for (int mb = blockIdx.y * p.rpb + tr; mb < p.M; mb += U * step) {
    f32x4 v[U], rr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mb + u * step < p.M ? mb + u * step : mb;
      v[u] = *reinterpret_cast<const f32x4*>(p.y + (size_t)m * p.ldy + c);
      if (p.res) rr[u] = *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.ldres + c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mb + u * step;
      // (the 8 lanes of a 32-channel word share tr and therefore m: the group is uniformly in or out of range)
      if (m < p.M) {
        f32x4 o = v[u] * sc + sh;
        if (p.res) o += rr[u];
        if (p.bits) {
          unsigned w = ((o[0] > 0.f ? 1u : 0u) | (o[1] > 0.f ? 2u : 0u) | (o[2] > 0.f ? 4u : 0u) | (o[3] > 0.f ? 8u : 0u))
                       << (4 * sub);
          w |= __shfl_xor(w, 1);
          w |= __shfl_xor(w, 2);
          w |= __shfl_xor(w, 4);
          if (sub == 0) p.bits[(size_t)m * p.ldbits + (c >> 5)] = w;
        }
        if (p.relu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
        }
        if (p.dropmask) o *= *reinterpret_cast<const f32x4*>(p.dropmask + (size_t)(m / p.HW) * p.C + c);
        *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.ldout + c) = o;
      }
    }
  }
}

struct BwdApplyTrainArgs {
  const float* g; const float* y; const float* mean; const float* invstd; const float* gamma;
  const double* sums;
  float* dy; float* dgamma; float* dbeta;
  double inv_count, pscale;
  int nslot, ldg, ldy, lddy, M, C, tpr, rpb;
};

__global__ __launch_bounds__(256) void bn_bwd_apply_train_kernel(const BwdApplyTrainArgs p) {
  const int CV = p.C >> 2;
  const int tc = threadIdx.x % p.tpr, tr = threadIdx.x / p.tpr;
  const int c4 = blockIdx.x * p.tpr + tc;
  if (c4 >= CV) return;
  const int c = c4 * 4;
  const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mean + c);
  const f32x4 is = *reinterpret_cast<const f32x4*>(p.invstd + c);
  const f32x4 ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
  const bool writer = blockIdx.y == 0 && tr == 0 && p.dgamma;
  f32x4 A, mg, AX;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    double s1 = p.sums[c + k], s2 = p.sums[p.C + c + k];
    for (int s = 1; s < p.nslot; ++s) {
      s1 += p.sums[(size_t)s * 2 * p.C + c + k];
      s2 += p.sums[(size_t)s * 2 * p.C + p.C + c + k];
    }
    if (writer) {
      p.dgamma[c + k] = (float)(s2 * p.pscale);
      p.dbeta[c + k] = (float)(s1 * p.pscale);
    }
    mg[k] = (float)(s1 * p.inv_count);
    A[k] = ga[k] * is[k];
    AX[k] = A[k] * (float)(s2 * p.inv_count);
  }
  constexpr int U = 4;
  const int step = gridDim.y * p.rpb;
  for (int mb = blockIdx.y * p.rpb + tr; mb < p.M; mb += U * step) {
    f32x4 g[U], yy[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mb + u * step < p.M ? mb + u * step : mb;
      g[u] = *reinterpret_cast<const f32x4*>(p.g + (size_t)m * p.ldg + c);
      yy[u] = *reinterpret_cast<const f32x4*>(p.y + (size_t)m * p.ldy + c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = mb + u * step;
      if (m < p.M) {
        const f32x4 xh = (yy[u] - mu) * is;
        const f32x4 r = A * (g[u] - mg) - AX * xh;
        *reinterpret_cast<f32x4*>(p.dy + (size_t)m * p.lddy + c) = r;
      }
    }
  }
}

// grid of the two kernels above: gx column groups x gy row groups, <= 512 workgroups, >= 4 rows per thread where M allows
inline int train_rows_grid(int M, int rpb, int gx) {
  int gy = (M + rpb * 4 - 1) / (rpb * 4);
  int cap = 512 / gx;
  if (cap < 1) cap = 1;
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  return gy;
}

inline int flat_grid(size_t total) {
  size_t g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" {

int semseg_channel_stats(const float* x, int ldx, double* stats, int nslot, int M, int C,
                         hipStream_t stream) {
  if (!x || !stats || (C & 3) || (ldx & 3) || M <= 0 || nslot < 1) return SEMSEG_EINVAL;
  const Tiling t = make_tiling(C >> 2);
  dim3 grid(t.gx, rows_grid(M, t.rpb, t.gx));
  channel_stats_kernel<<<grid, 256, 256 * 8 * sizeof(double), stream>>>(x, ldx, stats, M, C, t.tpr, t.rpb,
                                                                       nslot);
  return semseg_launch_status();
}

int semseg_bn_combine(double* stats, int nslot, int C, double* dst, hipStream_t stream) {
  if (!stats || nslot < 1) return SEMSEG_EINVAL;
  bn_combine_kernel<<<(2 * C + 255) / 256, 256, 0, stream>>>(stats, nslot, 2 * C, dst ? dst : stats);
  return semseg_launch_status();
}

int semseg_bn_finalize(const double* stats, int nslot, double count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, long long* num_batches_tracked,
                       float momentum, float eps, float* mean, float* invstd, float* scale,
                       float* shift, int C, hipStream_t stream) {
  if (!stats || !gamma || !beta || !mean || !invstd || !scale || !shift || count <= 0 || nslot < 1)
    return SEMSEG_EINVAL;
  bn_finalize_kernel<<<(C + 255) / 256, 256, 0, stream>>>(stats, nslot, count, gamma, beta, running_mean,
                                                        running_var, num_batches_tracked, momentum,
                                                        eps, mean, invstd, scale, shift, C);
  return semseg_launch_status();
}

int semseg_bn_eval_params(const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, float* scale, float* shift, int C,
                          hipStream_t stream) {
  if (!gamma || !beta || !running_mean || !running_var || !scale || !shift) return SEMSEG_EINVAL;
  bn_eval_params_kernel<<<(C + 255) / 256, 256, 0, stream>>>(gamma, beta, running_mean, running_var,
                                                           eps, scale, shift, C);
  return semseg_launch_status();
}

int semseg_bn_apply(const float* y, int ldy, const float* scale, const float* shift,
                    const float* y2, int ldy2, const float* scale2, const float* shift2,
                    const float* res, int ldres, const float* dropmask, float* out, int ldout,
                    int M, int C, int HW, int relu, unsigned* relu_bits, int ldbits, hipStream_t stream) {
  if (!y || !scale || !shift || !out || (C & 3) || (ldy & 3) || (ldout & 3)) return SEMSEG_EINVAL;
  if (y2 && (!scale2 || !shift2 || (ldy2 & 3))) return SEMSEG_EINVAL;
  if (res && (ldres & 3)) return SEMSEG_EINVAL;
  if (relu_bits && ((C & 31) || ldbits * 32 < C)) return SEMSEG_EINVAL;
  ApplyArgs a{y, scale, shift, y2, scale2, shift2, res, dropmask, out,
              ldy, ldy2, ldres, ldout, M, C, HW, relu, relu_bits, ldbits};
  bn_apply_kernel<<<flat_grid((size_t)M * (C >> 2)), 256, 0, stream>>>(a);
  return semseg_launch_status();
}

int semseg_bn_apply_train(const float* y, int ldy, const double* stats, int nslot, double count, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                          float momentum, float eps, float* mean, float* invstd, const float* res, int ldres,
                          const float* dropmask, float* out, int ldout, int M, int C, int HW, int relu, unsigned* relu_bits,
                          int ldbits, hipStream_t stream) {
  if (!y || !stats || !gamma || !beta || !mean || !invstd || !out || (C & 3) || (ldy & 3) || (ldout & 3) || count <= 0 ||
      nslot < 1 || M <= 0)
    return SEMSEG_EINVAL;
  if (res && (ldres & 3)) return SEMSEG_EINVAL;
  if (relu_bits && ((C & 31) || ldbits * 32 < C)) return SEMSEG_EINVAL;
  const Tiling t = make_tiling(C >> 2);
  ApplyTrainArgs a{y, stats, gamma, beta, running_mean, running_var, num_batches_tracked, mean, invstd, res, dropmask, out,
                   relu_bits, count, momentum, eps, nslot, ldy, ldres, ldout, ldbits, M, C, HW, relu, t.tpr, t.rpb};
  dim3 grid(t.gx, train_rows_grid(M, t.rpb, t.gx));
  bn_apply_train_kernel<<<grid, 256, 0, stream>>>(a);
  return semseg_launch_status();
}

int semseg_bn_bwd_apply_train(const float* g, int ldg, const float* y, int ldy, const float* mean, const float* invstd,
                              const float* gamma, const double* sums, int nslot, double count, double param_scale,
                              float* dgamma, float* dbeta, float* dy, int lddy, int M, int C, hipStream_t stream) {
  if (!g || !y || !mean || !invstd || !gamma || !sums || !dy || (C & 3) || (ldg & 3) || (ldy & 3) || (lddy & 3) ||
      count <= 0 || nslot < 1 || M <= 0 || (dgamma && !dbeta))
    return SEMSEG_EINVAL;
  const Tiling t = make_tiling(C >> 2);
  BwdApplyTrainArgs a{g, y, mean, invstd, gamma, sums, dy, dgamma, dbeta, 1.0 / count, param_scale, nslot, ldg, ldy, lddy,
                      M, C, t.tpr, t.rpb};
  dim3 grid(t.gx, train_rows_grid(M, t.rpb, t.gx));
  bn_bwd_apply_train_kernel<<<grid, 256, 0, stream>>>(a);
  return semseg_launch_status();
}

int semseg_bn_bwd_reduce(const float* dout, int lddout, const float* out, int ldout,
                         const float* dropmask, int HW, const float* y, int ldy, const float* mean,
                         const float* invstd, float* g, int ldg, double* sums, int nslot, int M,
                         int C, hipStream_t stream) {
  if (!dout || !y || !mean || !invstd || !sums || (C & 3) || (lddout & 3) || (ldy & 3) || nslot < 1)
    return SEMSEG_EINVAL;
  const Tiling t = make_tiling(C >> 2);
  BwdReduceArgs a{dout, out, dropmask, y, mean, invstd, g, sums,
                  lddout, ldout, ldy, ldg, M, C, HW, t.tpr, t.rpb, nslot};
  dim3 grid(t.gx, rows_grid(M, t.rpb, t.gx));
  bn_bwd_reduce_kernel<<<grid, 256, 256 * 8 * sizeof(double), stream>>>(a);
  return semseg_launch_status();
}

int semseg_bn_bwd_apply(const float* g, int ldg, const float* y, int ldy, const float* mean,
                        const float* invstd, const float* gamma, const double* sums, double count,
                        float* dy, int lddy, int M, int C, hipStream_t stream) {
  if (!g || !y || !mean || !invstd || !gamma || !sums || !dy || (C & 3) || count <= 0) return SEMSEG_EINVAL;
  BwdApplyArgs a{g, y, mean, invstd, gamma, sums, dy, 1.0 / count, ldg, ldy, lddy, M, C};
  bn_bwd_apply_kernel<<<flat_grid((size_t)M * (C >> 2)), 256, 0, stream>>>(a);
  return semseg_launch_status();
}

int semseg_bn_param_grads(double* sums, int nslot, float* dgamma, float* dbeta, int C,
                          int accumulate, double* folded, hipStream_t stream) {
  if (!sums || !dgamma || !dbeta || nslot < 1) return SEMSEG_EINVAL;
  bn_param_grads_kernel<<<(C + 255) / 256, 256, 0, stream>>>(sums, nslot, dgamma, dbeta, C, accumulate,
                                                            folded ? folded : sums);
  return semseg_launch_status();
}

}  // extern "C"
