// Pieces shared by conv_igemm.hip (forward / data gradient) and conv_wgrad.hip (weight gradient).
#pragma once
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "../../include/semseg_hip.h"

namespace {

// exact n / d for 0 <= n < 2^31 by one 64-bit multiply: q = (n * mul) >> sh
struct FastDiv {
  unsigned mul, sh;
};
inline FastDiv make_fastdiv(int d) {
  int l = 0;
  while ((1LL << l) < d) ++l;
  FastDiv f;
  f.sh = 31 + l;
  f.mul = (unsigned)(((1ULL << f.sh) + (unsigned long long)d - 1) / (unsigned long long)d);
  return f;
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) {
  return (int)(((unsigned long long)(unsigned)n * f.mul) >> f.sh);
}

// Invalid gather lanes (padding taps, rows past M) read this zero line instead of being zeroed after
// the load: a select on the loaded value would make the compiler wait for the load before the MFMA
// block (measured: -13 % on every conv).
__device__ __attribute__((aligned(16))) float g_zero_line[32] = {0};

inline int grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

inline bool arith_ok(int a) { return a == SEMSEG_ARITH_F32 || a == SEMSEG_ARITH_BF16X3; }

}  // namespace
