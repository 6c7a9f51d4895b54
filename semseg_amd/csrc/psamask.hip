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
    return launch<0>(input, output, num_, feature_H_, feature_W_, mask_H_, mask_W_, half_mask_H_,
                     half_mask_W_, stream);
  return launch<1>(input, output, num_, feature_H_, feature_W_, mask_H_, mask_W_, half_mask_H_,
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
    return launch<2>(grad_output, grad_input, num_, feature_H_, feature_W_, mask_H_, mask_W_,
                     half_mask_H_, half_mask_W_, stream);
  return launch<3>(grad_output, grad_input, num_, feature_H_, feature_W_, mask_H_, mask_W_,
                   half_mask_H_, half_mask_W_, stream);
}

}  // extern "C"
