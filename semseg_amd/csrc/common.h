// Shared device helpers for the gfx950 (CDNA4) kernel library.
// Wavefront = 64 lanes; MFMA fragments follow the v_mfma_f32_32x32x2_f32 maps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define SEMSEG_OK 0
#define SEMSEG_EINVAL (-1)
#define SEMSEG_ELAUNCH (-2)

static inline int semseg_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SEMSEG_OK : SEMSEG_ELAUNCH;
}

// A/B switch `name` of the environment variable SEMSEG_DEBUG = "name=value,name=value,..." (read per call: tests and tuning scripts
// switch inside one process): its value copied into buf, or nullptr.  Nothing a user needs lives there.
static inline const char* semseg_debug(const char* name, char* buf, size_t n) {
  const char* e = getenv("SEMSEG_DEBUG");
  if (!e) return nullptr;
  const size_t ln = strlen(name);
  while (*e) {
    while (*e == ',' || *e == ' ') ++e;
    const char* end = strchr(e, ',');
    if (!end) end = e + strlen(e);
    if ((size_t)(end - e) > ln && !strncmp(e, name, ln) && e[ln] == '=') {
      size_t l = (size_t)(end - (e + ln + 1));
      if (l >= n) l = n - 1;
      memcpy(buf, e + ln + 1, l);
      buf[l] = 0;
      return buf;
    }
    e = end;
  }
  return nullptr;
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

// bf16 -> fp32 of PACKED pieces by integer shifts / masks on the packed dwords (exact: a bf16 is the top half of an fp32).
// Written on the packed value on purpose: with __builtin_convertvector in both directions (x -> bf16 piece -> back, to form
// the remainder of the SEMSEG_ARITH_BF16X3 split) the compiler converted every element TWICE — once on its own for the
// remainder and once packed for the LDS store: 112 v_cvt_pk_bf16_f32 instead of 48 per 32 floats, 7.5 instead of 5.5 VALU
// instructions per split float, in kernels whose matrix pipe waits on VALU issue (round 4, ISA counts in DESIGN.md 8.5).
__device__ __forceinline__ f32x4 bf16x4_to_f32(const bf16x4 h) {
  const u32x2 u = __builtin_bit_cast(u32x2, h);
  f32x4 r;
  r[0] = __builtin_bit_cast(float, u[0] << 16);
  r[1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
  r[2] = __builtin_bit_cast(float, u[1] << 16);
  r[3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
  return r;
}
__device__ __forceinline__ f32x8 bf16x8_to_f32(const bf16x8 h) {
  const u32x4 u = __builtin_bit_cast(u32x4, h);
  f32x8 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r[2 * k] = __builtin_bit_cast(float, u[k] << 16);
    r[2 * k + 1] = __builtin_bit_cast(float, u[k] & 0xffff0000u);
  }
  return r;
}
