// Fused SGD(momentum, weight_decay) step over a flat fp32 parameter range — the optimizer update of
// the reference's train step (tool/train.py:140,276: torch.optim.SGD, momentum 0.9, weight decay
// 1e-4 applied to every parameter, no dampening, no Nesterov).  HBM-bound: 3 reads + 2 writes.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                  float* __restrict__ m, size_t n, float lr,
                                                  const float* lr_ptr, float momentum, float wd,
                                                  float gscale, int first, const int* __restrict__ skip) {
  if (skip && skip[0]) return;      // the step's statistics are known to be garbage (timed-out SyncBN exchange): leave the weights
  if (lr_ptr) lr = lr_ptr[0];
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 ww = reinterpret_cast<f32x4*>(w)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 d = gg * gscale + ww * wd;
    f32x4 b = first ? d : reinterpret_cast<f32x4*>(m)[i] * momentum + d;
    reinterpret_cast<f32x4*>(m)[i] = b;
    reinterpret_cast<f32x4*>(w)[i] = ww - b * lr;
  }
  if (blockIdx.x == 0) {
    for (size_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) {
      const float d = g[i] * gscale + w[i] * wd;
      const float b = first ? d : m[i] * momentum + d;
      m[i] = b;
      w[i] -= lr * b;
    }
  }
}
// Dropout2d keep/scale mask (model/pspnet.py:68,76: nn.Dropout2d(p) zeroes whole (n, c) planes and scales the kept
// ones by 1/(1-p)): one value per plane from a counter-based generator (splitmix64 of seed, call offset and plane
// index) — same distribution as torch's, not the same bit stream (semantics, not bit patterns).
__global__ void dropout2d_mask_kernel(float* __restrict__ mask, int n, float keep, float scale,
                                      unsigned long long seed, unsigned long long offset,
                                      const unsigned long long* __restrict__ offset_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (offset_dev) offset += offset_dev[0];      // the per-step part of the call counter of a replayed step (plan.hip)
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (offset * 0x100000000ull + (unsigned long long)i + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);   // 24 random bits -> [0, 1)
  mask[i] = u < keep ? scale : 0.f;
}
}  // namespace

extern "C" int semseg_dropout2d_mask(float* mask, int n, float p, unsigned long long seed, unsigned long long offset,
                                     const unsigned long long* offset_dev, hipStream_t stream) {
  if (!mask || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return SEMSEG_EINVAL;
  dropout2d_mask_kernel<<<(n + 255) / 256, 256, 0, stream>>>(mask, n, 1.f - p, 1.f / (1.f - p), seed, offset, offset_dev);
  return semseg_launch_status();
}

extern "C" int semseg_memset_zero(void* ptr, size_t bytes, hipStream_t stream) {
  if (!ptr) return SEMSEG_EINVAL;
  if (bytes == 0) return SEMSEG_OK;
  return hipMemsetAsync(ptr, 0, bytes, stream) == hipSuccess ? SEMSEG_OK : SEMSEG_ELAUNCH;
}

extern "C" int semseg_sgd_step(float* w, const float* g, float* mom, size_t n, float lr,
                               const float* lr_dev, float momentum, float weight_decay,
                               float grad_scale, int first_step, const int* skip_dev, hipStream_t stream) {
  if (!w || !g || !mom || ((uintptr_t)w & 15) || ((uintptr_t)g & 15) || ((uintptr_t)mom & 15))
    return SEMSEG_EINVAL;
  if (n == 0) return SEMSEG_OK;
  size_t grid = ((n >> 2) + 255) / 256;
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  sgd_kernel<<<(int)grid, 256, 0, stream>>>(w, g, mom, n, lr, lr_dev, momentum, weight_decay,
                                            grad_scale, first_step, skip_dev);
  return semseg_launch_status();
}
