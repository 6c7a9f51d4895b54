// Shared device helpers for the gfx950 (CDNA4) kernel library.
// Wavefront = 64 lanes; MFMA fragments follow the v_mfma_f32_32x32x2_f32 maps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SEMSEG_OK 0
#define SEMSEG_EINVAL (-1)
#define SEMSEG_ELAUNCH (-2)

static inline int semseg_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEMSEG_OK : SEMSEG_ELAUNCH;
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
  // hardware global_atomic_add_f64 on gfx950
  unsafeAtomicAdd(p, v);
}

__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask);
  hi = __shfl_xor(hi, mask);
  return __hiloint2double(hi, lo);
}

// XCD-aware bijective remap of a 1-D block id: hardware places block b on XCD b % 8,
// so give each XCD a contiguous chunk of the logical tile order (shared A/B panels hit
// that XCD's private L2).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
