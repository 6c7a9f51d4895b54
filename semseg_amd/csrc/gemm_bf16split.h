// Entry points of gemm_bf16split.hip used from other translation units of the library (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

// Forward of a 1x1, stride-1, unpadded convolution under SEMSEG_ARITH_BF16X3 on the 256 x 128 kernel (tile code 2128 of
// semseg_conv_fwd): y[M][Co] = x[M][Ci] * w_fwd[Co_pad][Ci]^T, stats (optional, [nslot][2 * Co] fp64) += {sum, sum of squares}
// of y per channel.  Ci % 16 == 0, ldx % 4 == 0, w_fwd padded to a multiple of 128 rows.
int semseg_split_gemm_conv1x1_fwd(const float* x, int ldx, const float* w_fwd, float* y, int ldy, int M, int Ci, int Co,
                                  double* stats, int nslot, hipStream_t stream);
