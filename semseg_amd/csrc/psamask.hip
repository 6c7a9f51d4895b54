// PSA mask collect / distribute, forward and backward — the reference's only native operator
// (lib/psa/src/cpu/psamask.cpp:11-113, lib/psa/src/gpu/psamask_cuda.cu:8-128), same NCHW fp32
// contract: caller allocates and zero-fills the destination, callee writes only the in-window
// elements.  Index maps (hh = half_mask_H, hw = half_mask_W, h' = h+hi-hh, w' = w+wi-hw):
//   collect    fwd: out[n, h'*W+w', h,  w ] = in[n, hi*mW+wi, h, w]      (psamask.cpp:26-29)
//   distribute fwd: out[n, h*W+w,   h', w'] = in[n, hi*mW+wi, h, w]      (psamask.cpp:52-55)
//   backward = the adjoint gathers                                       (psamask.cpp:63-113)
//
// The op is a pure permutation ("shear") => HBM-bound.  The reference GPU kernel gives each thread
// one (n,h,w) and walks <= mH*mW taps with H*W-strided, uncoalesced accesses.  Here one workgroup
// owns a slab (n, hi, h): the mW x W input slab is read in W-float contiguous rows into LDS and
// written back sheared, again in W-float contiguous rows, so both HBM directions move whole rows.
// LDS rows are padded to W+1 floats (odd stride) so the sheared LDS reads are conflict-free.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

struct PsaArgs {
  const float* src;
  float* dst;
  int N, H, W, mH, mW, hh, hw;
};

// mode 0: collect fwd, 1: distribute fwd, 2: collect bwd, 3: distribute bwd
template <int MODE>
__global__ __launch_bounds__(256) void psamask_kernel(const PsaArgs p) {
  extern __shared__ float sm[];
  const int H = p.H, W = p.W, mW = p.mW, HW = p.H * p.W;
  int b = blockIdx.x;
  const int h = b % H; b /= H;
  const int hi = b % p.mH;
  const int n = b / p.mH;
  const int hp = h + hi - p.hh;  // h'
  if (hp < 0 || hp >= H) return;
  const int ldW = W + 1;
  const size_t mask_base = ((size_t)(n * p.mH + hi) * mW) * HW + (size_t)h * W;  // + wi*HW + w
  const int tid = threadIdx.x;

  if (MODE == 0 || MODE == 1) {
    // stage mask slab [wi][w]
    for (int i = tid; i < mW * W; i += 256) {
      const int wi = i / W, w = i - wi * W;
      sm[wi * ldW + w] = p.src[mask_base + (size_t)wi * HW + w];
    }
    __syncthreads();
    if (MODE == 0) {
      // out[n, hp*W + w', h, w], rows w' contiguous in w
      const size_t ob = ((size_t)n * HW + (size_t)hp * W) * HW + (size_t)h * W;
      for (int i = tid; i < W * W; i += 256) {
        const int wp = i / W, w = i - wp * W;
        const int wi = wp - w + p.hw;
        if (wi >= 0 && wi < mW) p.dst[ob + (size_t)wp * HW + w] = sm[wi * ldW + w];
      }
    } else {
      // out[n, h*W + w, hp, w'], rows w contiguous in w'
      const size_t ob = ((size_t)n * HW + (size_t)h * W) * HW + (size_t)hp * W;
      for (int i = tid; i < W * W; i += 256) {
        const int w = i / W, wp = i - w * W;
        const int wi = wp - w + p.hw;
        if (wi >= 0 && wi < mW) p.dst[ob + (size_t)w * HW + wp] = sm[wi * ldW + w];
      }
    }
  } else {
    // stage buffer-gradient slab as [a][b] with b contiguous in memory
    if (MODE == 2) {
      const size_t ib = ((size_t)n * HW + (size_t)hp * W) * HW + (size_t)h * W;  // [w'][w]
      for (int i = tid; i < W * W; i += 256) {
        const int wp = i / W, w = i - wp * W;
        sm[wp * ldW + w] = p.src[ib + (size_t)wp * HW + w];
      }
    } else {
      const size_t ib = ((size_t)n * HW + (size_t)h * W) * HW + (size_t)hp * W;  // [w][w']
      for (int i = tid; i < W * W; i += 256) {
        const int w = i / W, wp = i - w * W;
        sm[w * ldW + wp] = p.src[ib + (size_t)w * HW + wp];
      }
    }
    __syncthreads();
    for (int i = tid; i < mW * W; i += 256) {
      const int wi = i / W, w = i - wi * W;
      const int wp = w + wi - p.hw;
      if (wp >= 0 && wp < W) {
        const float v = (MODE == 2) ? sm[wp * ldW + w] : sm[w * ldW + wp];
        p.dst[mask_base + (size_t)wi * HW + w] = v;
      }
    }
  }
}

template <int MODE>
int launch(const float* src, float* dst, int N, int H, int W, int mH, int mW, int hh, int hw,
           hipStream_t stream) {
  PsaArgs a{src, dst, N, H, W, mH, mW, hh, hw};
  const int rows = (MODE < 2) ? (mW > W ? mW : W) : W;
  const size_t lds = (size_t)rows * (W + 1) * sizeof(float);
  if (lds > 64 * 1024) return SEMSEG_EINVAL;
  const long long grid = (long long)N * mH * H;
  if (grid <= 0 || grid > 2147483647LL) return SEMSEG_EINVAL;
  psamask_kernel<MODE><<<(int)grid, 256, lds, stream>>>(a);
  return semseg_launch_status();
}


// ------------------------------------------------------------------------------------------
// Round 3: plane-group kernel.  The slab kernel above launches one workgroup per (n, hi, h): neighbouring slabs —
// which share every 128-byte line of the 120-byte rows they move — land on different XCDs (blockIdx round-robins
// over the 8 L2s), half of the workgroups exit at once (h' out of range), and each one is a single
// load -> barrier -> store round trip with nothing in flight behind it.
//
// Here a workgroup owns a whole plane group of the BUFFER side and walks its slabs in a loop:
//   collect:    (n, h')  -> buffer planes [h'*W, h'*W + W), loop over h  (hi = h' - h + hh)
//   distribute: (n, h)   -> buffer planes [h*W,  h*W + W),  loop over hi (h' = h + hi - hh)
// so (a) the 108 KB buffer region of a group (W planes x H x W floats) is read / written completely by ONE workgroup
// (every line is completed inside one L2 before it is evicted), (b) the mask rows of neighbouring groups — adjacent
// in memory — are touched by workgroups that sit next to each other in the logical order, which xcd_remap keeps on
// one XCD, (c) every slab does useful work (only valid (h, hi) pairs are walked), and (d) the loads of slab t + 2 are
// in flight (registers) while slab t goes through LDS: two LDS buffers, one barrier per slab.
// Mask-side elements outside the window (w' = w + wi - hw not in [0, W)) are neither fetched (forward) nor written
// (backward: the caller's zero fill stands, lib/psa/functions/psamask.py:31), per-thread element offsets / LDS
// addresses / validity bits are slab-independent and computed once.
// ------------------------------------------------------------------------------------------
struct PsaPlaneArgs {
  const float* src;
  float* dst;
  int N, H, W, mH, mW, hh, hw;
  int seg;      // the slab loop of a plane group is cut into `seg` contiguous pieces (blockIdx -> (n, piece, group))
  int ldm, ldb; // LDS row strides of the staged mask slab [wi][w] / buffer slab [r][c]
};

// MODE 0: collect fwd, 1: distribute fwd, 2: collect bwd, 3: distribute bwd.  NM / NB: mask-side / buffer-side
// elements per thread per slab (>= ceil(mW*W / 256), ceil(W*W / 256)).
template <int MODE, int NM, int NB>
__global__ __launch_bounds__(256) void psamask_plane_kernel(const PsaPlaneArgs p) {
  extern __shared__ float sm[];
  constexpr bool COLLECT = (MODE == 0 || MODE == 2);
  constexpr bool FWD = MODE < 2;
  const int H = p.H, W = p.W, mH = p.mH, mW = p.mW, HW = p.H * p.W;
  const int tid = threadIdx.x;
  int L = xcd_remap(blockIdx.x, gridDim.x);
  const int a = L % H; L /= H;            // plane group: h' (collect) or h (distribute)
  const int piece = L % p.seg;
  const int n = L / p.seg;
  // slab loop variable t: h (collect) or hi (distribute)
  int t_lo, t_hi;                          // inclusive range of valid t
  if (COLLECT) {
    t_lo = max(0, a + p.hh - (mH - 1));
    t_hi = min(H - 1, a + p.hh);
  } else {
    t_lo = max(0, p.hh - a);
    t_hi = min(mH - 1, H - 1 + p.hh - a);
  }
  {
    const int cnt = t_hi - t_lo + 1;
    if (cnt <= 0) return;
    const int per = (cnt + p.seg - 1) / p.seg;
    t_lo += piece * per;
    t_hi = min(t_hi, t_lo + per - 1);
    if (t_lo > t_hi) return;
  }

  // slab-independent per-thread tables
  int m_off[NM], m_lds[NM];
  unsigned m_ok = 0;
#pragma unroll
  for (int k = 0; k < NM; ++k) {
    const int i = tid + k * 256;
    const int wi = i / W, w = i - wi * W;
    const int wp = w + wi - p.hw;
    m_off[k] = wi * HW + w;
    m_lds[k] = FWD ? wi * p.ldm + w
                   : (COLLECT ? wp * p.ldb + w : w * p.ldb + wp);   // backward: where the staged buffer slab holds it
    if (i < mW * W && wp >= 0 && wp < W) m_ok |= 1u << k;
  }
  int b_off[NB], b_lds[NB];
  unsigned b_ok = 0;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int i = tid + k * 256;
    const int r = i / W, c = i - r * W;     // buffer slab row / column: (w', w) for collect, (w, w') for distribute
    const int wp = COLLECT ? r : c, w = COLLECT ? c : r;
    const int wi = wp - w + p.hw;
    b_off[k] = r * HW + c;
    b_lds[k] = FWD ? wi * p.ldm + w : r * p.ldb + c;
    if (i < W * W && wi >= 0 && wi < mW) b_ok |= 1u << k;
  }

  const float* __restrict__ src = p.src;
  float* __restrict__ dst = p.dst;
  auto mask_base = [&](int t) -> size_t {
    const int h = COLLECT ? t : a;
    const int hi = COLLECT ? a - t + p.hh : t;
    return ((size_t)(n * mH + hi) * mW) * HW + (size_t)h * W;
  };
  auto buf_base = [&](int t) -> size_t {
    if (COLLECT) return ((size_t)n * HW + (size_t)a * W) * HW + (size_t)t * W;            // planes h'*W.., row h
    return ((size_t)n * HW + (size_t)a * W) * HW + (size_t)(a + t - p.hh) * W;            // planes h*W.., row h'
  };

  constexpr int NL = FWD ? NM : NB;        // loads per thread per slab
  float reg[2][NL];
  auto load = [&](int t, float (&r)[NL]) {
    if (FWD) {
      const float* s = src + mask_base(t);
#pragma unroll
      for (int k = 0; k < NM; ++k) r[k] = ((m_ok >> k) & 1u) ? s[m_off[k]] : 0.f;
    } else {
      const float* s = src + buf_base(t);
#pragma unroll
      for (int k = 0; k < NB; ++k) r[k] = ((b_ok >> k) & 1u) ? s[b_off[k]] : 0.f;
    }
  };
  const int lds_slab = FWD ? mW * p.ldm : W * p.ldb;
  if (t_lo <= t_hi) load(t_lo, reg[0]);
  if (t_lo + 1 <= t_hi) load(t_lo + 1, reg[1]);
  for (int t = t_lo; t <= t_hi; ++t) {
    float* buf = sm + ((t - t_lo) & 1) * lds_slab;
    // (static register indexing: the two halves of the ring are written out as two code paths)
    if (((t - t_lo) & 1) == 0) {
#pragma unroll
      for (int k = 0; k < NL; ++k)
        if (FWD ? ((m_ok >> k) & 1u) : ((b_ok >> k) & 1u)) buf[FWD ? m_lds[k] : b_lds[k]] = reg[0][k];
      if (t + 2 <= t_hi) load(t + 2, reg[0]);
    } else {
#pragma unroll
      for (int k = 0; k < NL; ++k)
        if (FWD ? ((m_ok >> k) & 1u) : ((b_ok >> k) & 1u)) buf[FWD ? m_lds[k] : b_lds[k]] = reg[1][k];
      if (t + 2 <= t_hi) load(t + 2, reg[1]);
    }
    __syncthreads();   // slab t is in LDS; also: everybody is done reading the buffer slab t + 1 is about to overwrite
    if (FWD) {
      float* d = dst + buf_base(t);
#pragma unroll
      for (int k = 0; k < NB; ++k)
        if ((b_ok >> k) & 1u) d[b_off[k]] = buf[b_lds[k]];
    } else {
      float* d = dst + mask_base(t);
#pragma unroll
      for (int k = 0; k < NM; ++k)
        if ((m_ok >> k) & 1u) d[m_off[k]] = buf[m_lds[k]];
    }
  }
}

template <int MODE, int NM, int NB>
int launch_plane(const float* src, float* dst, int N, int H, int W, int mH, int mW, int hh, int hw,
                 hipStream_t stream) {
  PsaPlaneArgs a{src, dst, N, H, W, mH, mW, hh, hw, 1, 0, 0};
  // LDS strides: the sheared access must walk an odd number of banks per lane.  Forward reads the mask slab at
  // (w' - w + hw) * ldm + w: along w (collect) the step is 1 - ldm -> ldm even; along w' (distribute) it is ldm -> odd.
  // Backward reads the buffer slab at w' * ldb + w (collect) / w * ldb + w' (distribute) with w' = w + wi - hw: along w
  // the step is ldb + 1 -> ldb even.
  a.ldm = (MODE == 0) ? ((W & 1) ? W + 1 : W + 2) : ((W & 1) ? W : W + 1);
  a.ldb = (W & 1) ? W + 1 : W + 2;
  const size_t lds = 2 * sizeof(float) * (MODE < 2 ? (size_t)mW * a.ldm : (size_t)W * a.ldb);
  const long long groups = (long long)N * H;
  // pieces of the slab loop: enough workgroups to keep loads in flight on every CU.  Measured (N, H, mask): (16, 30, 59)
  // 39 / 31 / 33 / 30 us for 1 / 2 / 4 / 8 pieces, (2, 30, 59) 26 / 15 / 10 / 8 us, (16, 45, 89) 215 / 215 / 211 / 204 us
  // (that shape moves runs of ~90 contiguous bytes: 4.1 TB/s of bytes moved by PMC whatever the piece count)
  int seg = 1;
  while (groups * seg < 768 && seg < 8) seg *= 2;
  a.seg = seg;
  const long long grid = groups * seg;
  if (grid > 2147483647LL) return SEMSEG_EINVAL;
  psamask_plane_kernel<MODE, NM, NB><<<(int)grid, 256, lds, stream>>>(a);
  return semseg_launch_status();
}

// elements per thread per slab -> template instance; shapes beyond the largest instance (or its LDS) take the slab kernel
template <int MODE>
int launch_best(const float* src, float* dst, int N, int H, int W, int mH, int mW, int hh, int hw,
                hipStream_t stream) {
  const int nm = (mW * W + 255) / 256, nb = (W * W + 255) / 256;
  const size_t lds = 2 * sizeof(float) * (size_t)(mW > W ? mW : W) * (W + 2);
  if (lds <= 64 * 1024 && (long long)N * H * 8 < 2147483647LL) {
    if (nm <= 4 && nb <= 2) return launch_plane<MODE, 4, 2>(src, dst, N, H, W, mH, mW, hh, hw, stream);
    if (nm <= 8 && nb <= 4) return launch_plane<MODE, 8, 4>(src, dst, N, H, W, mH, mW, hh, hw, stream);
    if (nm <= 16 && nb <= 8) return launch_plane<MODE, 16, 8>(src, dst, N, H, W, mH, mW, hh, hw, stream);
    if (nm <= 32 && nb <= 16) return launch_plane<MODE, 32, 16>(src, dst, N, H, W, mH, mW, hh, hw, stream);
  }
  return launch<MODE>(src, dst, N, H, W, mH, mW, hh, hw, stream);
}

}  // namespace

extern "C" {

// Mirrors psamask_forward_cuda (lib/psa/src/gpu/operator.h:3, psamask_cuda.cu:108-117) with the
// at::Tensor arguments flattened to device pointers and an explicit stream.
int semseg_psamask_forward(int psa_type, const float* input, float* output, int num_,
                           int feature_H_, int feature_W_, int mask_H_, int mask_W_,
                           int half_mask_H_, int half_mask_W_, hipStream_t stream) {
  if (!input || !output || num_ <= 0 || feature_H_ <= 0 || feature_W_ <= 0 || mask_H_ <= 0 ||
      mask_W_ <= 0)
    return SEMSEG_EINVAL;
  if (psa_type == 0)
    return launch_best<0>(input, output, num_, feature_H_, feature_W_, mask_H_, mask_W_, half_mask_H_,
                     half_mask_W_, stream);
  return launch_best<1>(input, output, num_, feature_H_, feature_W_, mask_H_, mask_W_, half_mask_H_,
                   half_mask_W_, stream);
}

// Mirrors psamask_backward_cuda (lib/psa/src/gpu/operator.h:4, psamask_cuda.cu:119-128).
int semseg_psamask_backward(int psa_type, const float* grad_output, float* grad_input, int num_,
                            int feature_H_, int feature_W_, int mask_H_, int mask_W_,
                            int half_mask_H_, int half_mask_W_, hipStream_t stream) {
  if (!grad_output || !grad_input || num_ <= 0 || feature_H_ <= 0 || feature_W_ <= 0 ||
      mask_H_ <= 0 || mask_W_ <= 0)
    return SEMSEG_EINVAL;
  if (psa_type == 0)
    return launch_best<2>(grad_output, grad_input, num_, feature_H_, feature_W_, mask_H_, mask_W_,
                     half_mask_H_, half_mask_W_, stream);
  return launch_best<3>(grad_output, grad_input, num_, feature_H_, feature_W_, mask_H_, mask_W_,
                   half_mask_H_, half_mask_W_, stream);
}

}  // extern "C"
