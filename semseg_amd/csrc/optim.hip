// Fused SGD(momentum, weight_decay) step over a flat fp32 parameter range — the optimizer update of
// the reference's train step (tool/train.py:140,276: torch.optim.SGD, momentum 0.9, weight decay
// 1e-4 applied to every parameter, no dampening, no Nesterov).  HBM-bound: 3 reads + 2 writes.
#include "common.h"
#include "../../include/semseg_hip.h"

namespace {
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                  float* __restrict__ m, size_t n, float lr,
                                                  const float* lr_ptr, float momentum, float wd,
                                                  float gscale, int first) {
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
}  // namespace

extern "C" int semseg_sgd_step(float* w, const float* g, float* mom, size_t n, float lr,
                               const float* lr_dev, float momentum, float weight_decay,
                               float grad_scale, int first_step, hipStream_t stream) {
  if (!w || !g || !mom || ((uintptr_t)w & 15) || ((uintptr_t)g & 15) || ((uintptr_t)mom & 15))
    return SEMSEG_EINVAL;
  if (n == 0) return SEMSEG_OK;
  size_t grid = ((n >> 2) + 255) / 256;
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  sgd_kernel<<<(int)grid, 256, 0, stream>>>(w, g, mom, n, lr, lr_dev, momentum, weight_decay,
                                            grad_scale, first_step);
  return semseg_launch_status();
}
