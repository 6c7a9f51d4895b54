// Entry points of gemm_bf16split.hip used from other translation units of the library (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

// Forward of a 1x1, stride-1, unpadded convolution under SEMSEG_ARITH_BF16X3 on the 256 x 128 kernel (tile code 2128 of
// semseg_conv_fwd): y[M][Co] = x[M][Ci] * w_fwd[Co_pad][Ci]^T, stats (optional, [nslot][2 * Co] fp64) += {sum, sum of squares}
// of y per channel.  Ci % 16 == 0, ldx % 4 == 0, w_fwd padded to a multiple of 128 rows.  bm = rows per workgroup tile: 256 (tile
// code 2128, eight waves) or 128 (tile code 3128, four waves: grids of a small per-GPU batch).
int semseg_split_gemm_conv1x1_fwd(const float* x, int ldx, const float* w_fwd, float* y, int ldy, int M, int Ci, int Co,
                                  double* stats, int nslot, int bm, hipStream_t stream);

// Data gradient of the same kind of convolution (tile code 2128 of semseg_conv_dgrad[_bnreduce]): dx[M][Ci] = dy[M][Kc] *
// w_dgrad[Ci_pad][Kc]^T (+ add); ybn non-null: the fused BatchNorm-backward reduction of ONE layer (mask from relu_bits, else
// from act, else none), contract of semseg_conv_dgrad_bnreduce.  Ci % 128 == 0, Kc % 16 == 0, every ld % 4 == 0, 16-byte
// aligned bases.  The kernel accumulates one fp32 chain over Kc: the caller keeps Kc <= 1024 (conv_igemm.hip flushes chains
// longer than 576 into a second accumulator set; this kernel has no registers for one — dgrad_impl has the measured error).
int semseg_split_gemm_conv1x1_dgrad(const float* dy, int lddy, const float* w_dgrad, float* dx, int lddx, int M, int Kc, int Ci,
                                    const float* add, int ldadd, const float* act, int ldact, const unsigned* relu_bits,
                                    int ldbits, const float* ybn, int ldybn, const float* mean, const float* invstd,
                                    double* sums, int nslot, int bm, hipStream_t stream);
