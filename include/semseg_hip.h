/* semseg_hip.h — C ABI of libsemseg_hip.so, the gfx950 (MI355X) kernel library behind the
 * PSPNet / PSANet train step of hszhao/semseg.
 *
 * Conventions (the reference's own native-op convention, lib/psa/src/gpu/operator.h:3-4 and
 * psamask_cuda.cu:108-128, made explicit): plain device pointers + int sizes, caller allocates every
 * output, callee launches asynchronously on `stream` and returns 0 (SEMSEG_OK) or a negative code
 * (-1 invalid argument, -2 launch failure).  No torch types cross this boundary.
 *
 * Activations are NHWC fp32 with an explicit channel stride `ld*` (elements per pixel) so channel
 * slices of concatenated buffers are addressed in place; `M` = N*H*W pixels.  Parameters cross in the
 * reference's state-dict layouts (conv weight OIHW, BN vectors [C]).
 */
#ifndef SEMSEG_HIP_H
#define SEMSEG_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- lib/psa native operator (reference: lib/psa/src/gpu/operator.h:3-4; CPU twin
 * lib/psa/src/cpu/operator.h:3-4; called from lib/psa/functions/psamask.py:18-22,32-35).
 * NCHW fp32, psa_type 0 = collect, else distribute; destination pre-zeroed by the caller. */
int semseg_psamask_forward(int psa_type, const float* input, float* output, int num_,
                           int feature_H_, int feature_W_, int mask_H_, int mask_W_,
                           int half_mask_H_, int half_mask_W_, hipStream_t stream);
int semseg_psamask_backward(int psa_type, const float* grad_output, float* grad_input, int num_,
                            int feature_H_, int feature_W_, int mask_H_, int mask_W_,
                            int half_mask_H_, int half_mask_W_, hipStream_t stream);

/* ---- Arithmetic of the matrix-core products: a PER-LAUNCH argument (`arith`) of every entry point below that runs
 * a GEMM on the matrix cores.  Operands, accumulators and results are fp32 in both cases.
 *   SEMSEG_ARITH_F32     exact fp32 products: v_mfma_f32_32x32x2_f32 (bit-wise an fmaf chain).
 *   SEMSEG_ARITH_BF16X3  every fp32 operand is cut IN FLIGHT (between the global load and the LDS store; HBM keeps
 *                        fp32) into three bf16 pieces h + m + l, round-to-nearest at every level, remainders exact in
 *                        fp32, so the pieces carry all 24 mantissa bits; the product is rebuilt from the six leading
 *                        cross products ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm on v_mfma_f32_32x32x16_bf16
 *                        (each exact in fp32; the three dropped ones are below 2^-24 of the result) with fp32
 *                        accumulation.  Measured error, PER OP against fp64: at or below the fp32 instruction's (tests/
 *                        test_ops_gpu.py, the in-situ tables of profiles/); at NETWORK level it is slightly above the exact path's
 *                        everywhere and inside every bound (PSPNet-101 473^2: eval logits 3.4e-6 vs 2.1e-6 of max, cls.4.weight
 *                        gradient 1.3e-4 vs 8.5e-5, 247 vs 217 argmax flips of 3.6 M pixels; profiles/r04_parity_report.txt).
 *                        Kernel instances without a split form (the generic tap walk of kernel sizes other than 1x1 / 3x3, the
 *                        PSA contraction) compute exact fp32 products under either value; the 64 x 64 weight-gradient tiles
 *                        HAVE a split form since round 4 (conv_wgrad_kernel<64,64,MODE,3>) and run it under BF16X3:
 *                        `arith` never makes a launch LESS precise than SEMSEG_ARITH_BF16X3.
 * Any other value: SEMSEG_EINVAL (-1).  No process-wide state is involved: concurrent launches on different streams /
 * host threads may use different values. */
enum { SEMSEG_ARITH_F32 = 0, SEMSEG_ARITH_BF16X3 = 3 };

/* ---- nn.Conv2d (reference call sites: model/resnet.py:63-69,108-112,134; model/pspnet.py:15,65,
 * 69,73,77; dilation surgery model/pspnet.py:49-58; model/psanet.py:25-48).
 * pack: OIHW -> K-contiguous panels consumed by fwd ([Co_pad][Ci*R*S]) and dgrad
 * ([Ci_pad][roundup32(Co)*R*S]); either destination may be NULL. */
int semseg_conv_pack_weights(const float* w_oihw, float* w_fwd, float* w_dgrad, int Co, int Ci,
                             int R, int S, int Co_pad, int Ci_pad, hipStream_t stream);
/* Every conv weight of a network in one launch.  descs_dev: device array of nconv descriptors;
 * block_starts_dev: 2*nconv ints, the first block (1024 elements each) of conv i's forward panel
 * ([2i]) and data-gradient panel ([2i+1]); total_blocks = one past the last block. */
typedef struct SemsegPackDesc {
  const float* w;   /* OIHW */
  float* w_fwd;     /* [Co_pad][Ci*RS] */
  float* w_dgrad;   /* [Ci_pad][Kc_dgrad*RS] */
  int Co, Ci, RS, Co_pad, Ci_pad, Kc_dgrad;
} SemsegPackDesc;
int semseg_conv_pack_weights_multi(const SemsegPackDesc* descs_dev, const int* block_starts_dev,
                                   int nconv, int total_blocks, hipStream_t stream);
/* y[M][Co] = [relu]( conv(x) (*scale) (+bias) (+add) ) — scale/bias/relu fold an eval-mode BatchNorm
 * (+ReLU, +residual) into the epilogue; stats (optional, [2*Co] fp64, caller-zeroed) receives the
 * per-channel sum and sum of squares of y for the following BatchNorm.  tile_n in {64, 128} = output columns of a 128-row tile, or 1064 / 1128 = 64-row tiles of 64 / 128 columns (launches whose
 * 128-row grid would not fill the chip), or 2128 / 3128 = the bf16x3 GEMM kernel of gemm_bf16split.hip with 256 x 128 / 128 x 128 tiles and a statistics / fused-reduction epilogue where the
 * call is a plain row GEMM — SEMSEG_ARITH_BF16X3, 1x1, stride 1, no padding, nothing folded into the epilogue, whole 128-column panels; data gradients also: at most
 * one fused BatchNorm layer and Co <= 1024 — and the 128 x 128 tile otherwise;
 * w_fwd must have Co_pad = roundup(Co, tile_n) rows.  Ci % 32 == 0.  scratch (optional) enables
 * split-K when the 128 x tile_n tile grid cannot fill the 256 CUs (small per-GPU batches). */
/* tile_counters (optional; semseg_conv_fwd / _dgrad / _dgrad_bnreduce): SEMSEG_TILE_COUNTERS 32-bit words, ZERO before the first
 * launch that is given them, owned by the stream the launches run on like scratch.  With them, tiles whose K range is split
 * are reduced INSIDE the launch: each slice stores its accumulators into its own slab of scratch and draws a ticket from its
 * tile's counter; the slice that draws the last ticket sums the slabs in slice order (the result does not depend on the arrival
 * order) and runs the epilogue — instead of a second launch that re-reads every partial sum.  The launch leaves the counters
 * zero.  NULL: the separate reduction launch of rounds 1-5. */
#define SEMSEG_TILE_COUNTERS 4096
int semseg_conv_fwd(const float* x, int ldx, const float* w_fwd, float* y, int ldy, int N, int H,
                    int W, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad,
                    int dil, const float* bias, const float* scale, int relu, const float* add,
                    int ldadd, double* stats, int stats_nslot, int tile_n, int arith, float* scratch,
                    size_t scratch_floats, unsigned int* tile_counters, hipStream_t stream);
/* dx[N*H*W][Ci] = conv_transpose(dy) (+add).  dy must be readable (zero padded) up to
 * roundup32(Co) channels; w_dgrad must have roundup(Ci, tile_n) rows. */
int semseg_conv_dgrad(const float* dy, int lddy, const float* w_dgrad, float* dx, int lddx, int N,
                      int H, int W, int Ci, int Ho, int Wo, int Co, int R, int S, int stride,
                      int pad, int dil, const float* add, int ldadd, int tile_n, int arith, float* scratch,
                      size_t scratch_floats, unsigned int* tile_counters, hipStream_t stream);
/* Data gradient + the BatchNorm-backward reduction (torch batch_norm backward for model/resnet.py:76-92) of the
 * layer(s) that PRODUCED this conv's input, in one kernel: dx = g = (dgrad (+ add)) * (act > 0), and
 * sums{0,1}[nslot][2*Ci] += {sum g, sum g * (y - mean) * invstd} in fp64 (slot replicas as in semseg_channel_stats).
 * Valid only when this data gradient is the last contribution to that activation's gradient.  act may be null (no
 * ReLU); relu_bits (optional, semseg_bn_apply's bit form of the same mask, Ci % 32 == 0) replaces act when given: the mask
 * operand shrinks 32-fold (act is the largest operand of the epilogue of a 1x1 data gradient).  bn_count 2 = bn3 + downsample
 * BN sharing g.  Ci % 4 == 0, every ld % 4 == 0. */
int semseg_conv_dgrad_bnreduce(const float* dy, int lddy, const float* w_dgrad, float* dx, int lddx, int N, int H,
                               int W, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad, int dil,
                               const float* add, int ldadd, int tile_n, int bn_count, const float* act, int ldact,
                               const unsigned* relu_bits, int ldbits,
                               const float* y0, int ldy0, const float* mean0, const float* invstd0, double* sums0,
                               const float* y1, int ldy1, const float* mean1, const float* invstd1, double* sums1,
                               int nslot, int arith, float* scratch, size_t scratch_floats, unsigned int* tile_counters,
                               hipStream_t stream);
/* dw_oihw[Co][Ci][R][S] (=|+=) sum over pixels; scratch holds the split-K partial slabs
 * (>= semseg_conv_wgrad_scratch_floats(...) floats; more slabs => more K parallelism AND shorter fp32
 * accumulation chains: with a single slab the whole pixel reduction is one chain and its rounding noise
 * is 3-10x a blocked CPU sum at M >= 57600 rows; with 64 Mi floats, as the engine passes, it is 0.9-1.3x).
 * dy must be readable (zero padded) up to roundup(Co, 64 or 128) channels.  Ci % 64 == 0. */
int semseg_conv_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw_oihw,
                      float* scratch, size_t scratch_floats, int N, int H, int W, int Ci, int Ho,
                      int Wo, int Co, int R, int S, int stride, int pad, int dil, int accumulate,
                      int arith, hipStream_t stream);
size_t semseg_conv_wgrad_scratch_floats(int Ci, int Co, int R, int S);

/* Batched GEMMs on the same two matrix-core kernels (one launch, blockIdx.y = batch item; strides in floats) — the
 * PSA point-affinity contraction torch.bmm(x, y) of model/psanet.py:90-91 and its two gradients:
 *   rows:   C[b][M][Nout] = A[b][M][K] * Bt[b][Nout_pad][K]^T   (K % 32 == 0; Bt rows >= Nout readable, zero)
 *   kmajor: out[b][Co][Ci] (=|+=) sum_k y[b][k][co] * x[b][k][ci]  (Ci % 64 == 0; scratch holds batch slab sets) */
int semseg_gemm_rows_batched(const float* a, int lda, long long a_bs, const float* bt, long long bt_bs, float* c,
                             int ldc, long long c_bs, int M, int K, int Nout, int batch, int arith,
                             hipStream_t stream);
int semseg_gemm_kmajor_batched(const float* x, int ldx, long long x_bs, const float* y, int ldy, long long y_bs,
                               float* out, long long out_bs, float* scratch, size_t scratch_floats, int K, int Ci,
                               int Co, int accumulate, int batch, int arith, hipStream_t stream);

/* semseg_gemm_rows_batched with SEMSEG_ARITH_BF16X3 products on a kernel of its own (256 x 128 tile, 8 waves,
 * chunk-swizzled unpadded piece planes; csrc/gemm_bf16split.hip) — what the engine runs for the 16 batched GEMMs of a
 * Winograd forward / data gradient.  Same operands and strides; nsplit = 3 (three pieces, six products: the arithmetic
 * above), bk = 16, K % 16 == 0, lda % 4 == 0, Bt rows readable up to roundup(Nout, 128).
 * nsplit = 2 (two pieces, three products, ~2^-16 per product; bk 16 | 32) is kept as a MEASUREMENT only: it fails the
 * per-op parity criteria and nothing in the engine selects it. */
int semseg_gemm_rows_batched_bf16split(const float* a, int lda, long long a_bs, const float* bt, long long bt_bs,
                                       float* c, int ldc, long long c_bs, int M, int K, int Nout, int batch, int nsplit,
                                       int bk, hipStream_t stream);

/* ---- Winograd F(2x2, 3x3) for the stride-1 "same" 3x3 convolutions (kernel 3, stride 1, padding = dilation:
 * model/resnet.py:63-69 as modified by model/pspnet.py:49-58; head convs model/pspnet.py:65,73).
 *   forward:        V = input_transform(x);  M[e] = V[e] * U[e]^T (semseg_gemm_rows_batched, batch 16);  y = output_transform(M)
 *   data gradient:  the same three steps on dy with the flipped, transposed filter (filter_transform flip = 1)
 *   weight gradient: Yh = dy_transform_wgrad(dy);  dU[e] = Yh[e]^T V[e] (semseg_gemm_kmajor_batched);  dw = filter_grad(dU)
 * T = semseg_wino_tiles(N, H, W, dil) 2x2 output tiles (cut per dilation phase); V is [16][T][C], U [16][rows_pad][Kc],
 * M [16][T][ldm], dU [16][Co][Ci].  1 / 2.25 of the direct convolution's multiplications, all in fp32. */
int semseg_wino_tiles(int N, int H, int W, int dil);
int semseg_wino_input_transform(const float* src, int lds, float* V, int N, int H, int W, int C, int dil,
                                hipStream_t stream);
int semseg_wino_dy_transform_wgrad(const float* dy, int lddy, float* Yh, int ldo, int N, int H, int W, int C, int dil,
                                   hipStream_t stream);
/* y = [relu]((A^T M A) * scale + shift + add); scale / shift optional per-channel vectors (eval-mode BatchNorm folded in,
 * as in semseg_conv_fwd); stats (optional, [nslot][2*C] fp64, caller-zeroed) += {sum, sum of squares} of the values
 * before add */
int semseg_wino_output_transform(const float* M, int ldm, float* y, int ldy, const float* add, int ldadd, double* stats,
                                 int nslot, const float* scale, const float* shift, int relu, int N, int H, int W, int C,
                                 int dil, hipStream_t stream);
/* The output transform of a DATA GRADIENT that completes the gradient of a BatchNorm(+ReLU) output, with that layer's
 * BatchNorm-backward reduction folded in (contract of semseg_conv_dgrad_bnreduce, one BatchNorm layer): stores
 * g = (A^T M A + add) * (act > 0) and accumulates sums[slot][2*C] += {sum g, sum g * (ybn - mean) * invstd} in fp64;
 * relu_bits (optional) replaces act as there. */
int semseg_wino_output_transform_bnreduce(const float* M, int ldm, float* y, int ldy, const float* add, int ldadd,
                                          const float* act, int ldact, const unsigned* relu_bits, int ldbits,
                                          const float* ybn, int ldybn, const float* mean,
                                          const float* invstd, double* sums, int nslot, int N, int H, int W, int C,
                                          int dil, hipStream_t stream);
/* flip 0: U[e][co][ci] (rows_pad >= Co, Kc >= Ci);  flip 1: U[e][ci][co] with taps rotated 180 degrees (rows_pad >= Ci,
 * Kc >= Co); padding rows / columns are written as zero */
int semseg_wino_filter_transform(const float* w_oihw, float* U, int Co, int Ci, int rows_pad, int Kc, int flip,
                                 hipStream_t stream);
/* every filter panel of a network in one launch: descs_dev[i] describes one semseg_wino_filter_transform call, panel i
 * owns blocks [block_starts_dev[i], block_starts_dev[i+1]) of 256 (row, k) pairs each */
typedef struct SemsegWinoFilterDesc {
  const float* w;   /* OIHW [Co][Ci][3][3] */
  float* U;         /* [16][rows_pad][Kc] */
  int Co, Ci, rows_pad, Kc, flip;
} SemsegWinoFilterDesc;
int semseg_wino_filter_transform_multi(const SemsegWinoFilterDesc* descs_dev, const int* block_starts_dev, int npanels,
                                       int total_blocks, hipStream_t stream);
int semseg_wino_filter_grad(const float* dU, float* dw_oihw, int Co, int Ci, int accumulate, hipStream_t stream);

/* Stem conv 3->64, 3x3 stride 2 pad 1, reading the caller's NCHW input (model/resnet.py:108). */
int semseg_stem_conv_fwd(const float* x_nchw, const float* w_oihw, float* y_nhwc, int N, int H,
                         int W, int Co, hipStream_t stream);
/* weight gradient: per-workgroup partial slabs in `scratch` (>= semseg_stem_wgrad_scratch_floats floats) + one fold launch; no atomics:
 * the result does not depend on the arrival order. */
size_t semseg_stem_wgrad_scratch_floats(int N, int H, int W);
int semseg_stem_conv_wgrad(const float* x_nchw, const float* dy_nhwc, float* dw_oihw, int N, int H,
                           int W, int Co, int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream);

/* ---- nn.BatchNorm2d / nn.SyncBatchNorm (+ReLU, +residual, +Dropout2d) — model/resnet.py:64-69,
 * 88-92,109-113,136; model/pspnet.py:16-17,66-68,74-76; tool/train.py:142.
 * stats / sums are [nslot][2*C] fp64 vectors (caller-zeroed): producers scatter their atomics over the
 * nslot replicas (same-address fp64 atomics serialise), consumers combine them; the combined slot 0
 * ([2*C]) is exactly what SyncBN all-reduces. */
int semseg_channel_stats(const float* x, int ldx, double* stats, int nslot, int M, int C,
                         hipStream_t stream);
/* dst[2*C] = sum over the nslot replicas (dst == NULL: into slot 0 of stats).  A SyncBN GROUP — BatchNorm layers whose
 * statistics do not depend on each other (bn3 + downsample BN of a projection block, the four PPM branches, the cls /
 * aux head BNs) — combines into adjacent pieces of ONE staging vector, which is all-reduced once. */
int semseg_bn_combine(double* stats, int nslot, int C, double* dst, hipStream_t stream);
int semseg_bn_finalize(const double* stats, int nslot, double count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, long long* num_batches_tracked,
                       float momentum, float eps, float* mean, float* invstd, float* scale,
                       float* shift, int C, hipStream_t stream);
int semseg_bn_eval_params(const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, float* scale, float* shift, int C,
                          hipStream_t stream);
/* out = [relu]( y*scale+shift (+ y2*scale2+shift2) (+ res) ) (* dropmask[n][c]).
 * relu_bits (optional, [M][ldbits] 32-bit words, C % 32 == 0, ldbits >= C / 32): bit (c & 31) of word [m][c >> 5] is set where
 * the value entering the ReLU is > 0 — the ReLU mask of torch's threshold backward at 1/32 of the activation's bytes; the fused
 * BatchNorm-backward reductions (semseg_conv_dgrad_bnreduce, semseg_wino_output_transform_bnreduce) read it instead of out. */
int semseg_bn_apply(const float* y, int ldy, const float* scale, const float* shift,
                    const float* y2, int ldy2, const float* scale2, const float* shift2,
                    const float* res, int ldres, const float* dropmask, float* out, int ldout,
                    int M, int C, int HW, int relu, unsigned* relu_bits, int ldbits, hipStream_t stream);
/* semseg_bn_finalize + semseg_bn_apply of ONE training-mode BatchNorm in one launch (round 6; what the engine runs where the
 * producers used nslot <= 2, i.e. at a small per-GPU batch, where a BatchNorm layer is otherwise four latency-bound launches):
 * every thread derives scale / shift of its 4 channels from stats ([nslot][2*C], after the SyncBN all-reduce when there is one)
 * with the expressions of semseg_bn_finalize — bit-identical values — and one row group also writes mean / invstd (read by
 * the backward kernels), the running statistics (NULL: not tracked) and *num_batches_tracked += 1. */
int semseg_bn_apply_train(const float* y, int ldy, const double* stats, int nslot, double count, const float* gamma,
                          const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                          float momentum, float eps, float* mean, float* invstd, const float* res, int ldres,
                          const float* dropmask, float* out, int ldout, int M, int C, int HW, int relu, unsigned* relu_bits,
                          int ldbits, hipStream_t stream);
/* semseg_bn_param_grads + semseg_bn_bwd_apply in one launch: sums is [nslot][2*C]; dgamma = param_scale * sum g*xhat,
 * dbeta = param_scale * sum g (dgamma NULL: not written).  Single process: the local sums, param_scale 1.  Under SyncBN the
 * call follows the all-reduce, sums are GLOBAL and param_scale = 1 / world: every rank then holds global / world, which is
 * what the gradient all-reduce (sum) followed by the 1 / world scale makes of torch's per-rank local gradients as well. */
int semseg_bn_bwd_apply_train(const float* g, int ldg, const float* y, int ldy, const float* mean, const float* invstd,
                              const float* gamma, const double* sums, int nslot, double count, double param_scale,
                              float* dgamma, float* dbeta, float* dy, int lddy, int M, int C, hipStream_t stream);
/* g = dout (*dropmask) (*[out>0]); sums += {sum g, sum g*xhat}; g optionally written. */
int semseg_bn_bwd_reduce(const float* dout, int lddout, const float* out, int ldout,
                         const float* dropmask, int HW, const float* y, int ldy, const float* mean,
                         const float* invstd, float* g, int ldg, double* sums, int nslot, int M,
                         int C, hipStream_t stream);
/* dy = gamma*invstd*(g - sum_g/count - xhat*sum_gx/count); sums = combined slot 0 */
int semseg_bn_bwd_apply(const float* g, int ldg, const float* y, int ldy, const float* mean,
                        const float* invstd, const float* gamma, const double* sums, double count,
                        float* dy, int lddy, int M, int C, hipStream_t stream);
/* combines the nslot replicas of sums into `folded` ([2*C]; NULL: slot 0 of sums) and writes dgamma = sum g*xhat,
 * dbeta = sum g from the LOCAL sums; `folded` is what the SyncBN backward all-reduces (per group, see above) */
int semseg_bn_param_grads(double* sums, int nslot, float* dgamma, float* dbeta, int C,
                          int accumulate, double* folded, hipStream_t stream);

/* ---- SyncBN statistics exchange through peer-mapped memory (csrc/xchg.hip; nn.SyncBatchNorm, tool/train.py:142): the
 * all-reduce of a small fp64 vector among the <= 8 GPUs of one node in ONE kernel per rank.  Every rank owns a buffer of
 * semseg_xchg_buffer_bytes(world) bytes in fine-grained device memory (semseg_xchg_alloc), exports it
 * (hipIpcMemHandle_t, 64 bytes) and maps every peer's (semseg_xchg_ipc_import); peer_bases = HOST array of the `world`
 * mapped base pointers in rank order (own buffer at [rank]).  allreduce: out[0:n] = sum over ranks of (sum over the nslot
 * replicas of in[nslot][n]), summed in rank order on every rank (bit-identical results); seq = 1, 2, ... must advance by
 * one per call, identically on every rank — by value, or (seq_dev non-null; seq ignored) kept in device memory and advanced
 * by the kernel itself, which leaves the launch without a per-call host argument: a recorded step replays it; n <= SEMSEG_XCHG_MAX_DOUBLES.  A rank whose peers do not arrive within timeout_ms
 * (<= 0: 20 000) sets *err_dev = 1 and returns garbage in out (the caller checks err_dev); once *err_dev is set every later
 * exchange gives up at once.  Verified with several processes on one GPU only: the host side takes this path after a start-up
 * self-test among the real peers and uses RCCL otherwise (semseg_amd/syncbn_xchg.py). */
#define SEMSEG_XCHG_MAX_DOUBLES 16384
size_t semseg_xchg_buffer_bytes(int world);
int semseg_xchg_alloc(int world, void** ptr);
int semseg_xchg_free(void* ptr);
int semseg_xchg_ipc_export(void* ptr, void* handle64);
int semseg_xchg_ipc_import(const void* handle64, void** ptr);
int semseg_xchg_ipc_close(void* ptr);
int semseg_xchg_allreduce_f64(const double* in, int nslot, int n, double* out, void* const* peer_bases, int world, int rank,
                              unsigned long long seq, unsigned long long* seq_dev, int* err_dev, int timeout_ms,
                              hipStream_t stream);

/* ---- spatial ops: MaxPool2d(3,2,1) model/resnet.py:115; AdaptiveAvgPool2d model/pspnet.py:14;
 * F.interpolate(bilinear, align_corners=True) model/pspnet.py:25,95,100; model/psanet.py:61,78,97. */
int semseg_maxpool3x3s2_fwd(const float* x, float* y, uint32_t* idx, int N, int H, int W, int C,
                            hipStream_t stream);
int semseg_maxpool3x3s2_bwd(const float* dy, const uint32_t* idx, float* dx, int N, int H, int W,
                            int C, hipStream_t stream);
/* y holds the pooled maps of all bins back to back.  With scratch >= semseg_adaptive_avgpool_scratch_floats()
 * floats the feature map is read once (row sums per (bin, column cell), then a gather over rows) and the
 * result is deterministic; without it every bin re-reads x and big windows are merged with fp32 atomics. */
int semseg_adaptive_avgpool_fwd(const float* x, int ldx, float* y, const int* bins, int nbins,
                                int N, int H, int W, int C, float* scratch, size_t scratch_floats,
                                hipStream_t stream);
size_t semseg_adaptive_avgpool_scratch_floats(const int* bins, int nbins, int N, int H, int C);
int semseg_adaptive_avgpool_bwd(const float* base, int ldbase, const float* dpool, float* dx,
                                int lddx, const int* bins, int nbins, int N, int H, int W, int C,
                                hipStream_t stream);
int semseg_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int Hi, int Wi, int Ho,
                        int Wo, int C, hipStream_t stream);
int semseg_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int Hi, int Wi,
                        int Ho, int Wo, int C, hipStream_t stream);
int semseg_bilinear_nhwc_to_nchw(const float* x, int ldx, float* y, int N, int Hi, int Wi, int Ho,
                                 int Wo, int C, hipStream_t stream);

/* ---- fused head: upsample + CrossEntropyLoss(ignore_index) + argmax — model/pspnet.py:95,100-103,
 * tool/train.py:121.  acc2 = THREE doubles {sum of losses, valid-pixel count, number of labels that are neither
 * ignore_index nor in [0, C)} (fp64, zeroed by the call), loss = acc2[0]/acc2[1].  Out-of-range labels contribute nothing;
 * torch's CrossEntropyLoss raises on them, so the caller must look at acc2[2] (semseg_amd.engine does, asynchronously). */
int semseg_ce_head_fwd(const float* scores, int ld, const long long* label, float* lse,
                       long long* pred, double* acc2, float* loss, int N, int h, int w, int H, int W,
                       int C, int ignore_index, hipStream_t stream);
int semseg_ce_head_bwd(const float* scores, int ld, const long long* label, const float* lse,
                       const double* acc2, const float* grad_loss, float grad_mul, float* dscores,
                       int lddz, int accumulate, int N, int h, int w, int H, int W, int C,
                       int ignore_index, float* scratch, size_t scratch_floats, hipStream_t stream);

/* Counts targets that are neither ignore_index nor in [0, C): torch's CrossEntropyLoss (tool/train.py:121) raises
 * on those; the fused head treats them as ignored, so callers validate (the engine does on its first step). */
int semseg_label_check(const long long* label, size_t n, int C, int ignore_index,
                       unsigned long long* bad_count_dev, hipStream_t stream);

/* ---- PSA head on the engine's pixel-major layout (model/psanet.py:53-98).
 * psamask_nhwc: attention map [N, H*W, taps(ldm)] <-> affinity rows aff[n, q, p] (lda >= H*W), same
 * index maps as semseg_psamask_* (lib/psa/src/cpu/psamask.cpp:11-113); out-of-window entries = 0.
 * softmax_rows: y = alpha*softmax(x[0:P)) per row (F.softmax(dim=1) + 1/normalization_factor,
 * psanet.py:88-91); transpose_batched: out[b][c][r] = in[b][r][c], columns r >= R zero-filled. */
int semseg_psamask_nhwc_forward(int psa_type, const float* mask, int ldm, float* aff, int lda, int N,
                                int H, int W, int mH, int mW, hipStream_t stream);
/* backward: dmask_prezeroed = 0: the whole [N*H*W, taps] block is defined (out-of-window taps written as 0: N * HW * taps * 4 bytes
 * on top of the algorithmic 2 * 4 * N * (HW)^2); 1: the caller guarantees that the out-of-window taps of dmask ARE zero (a buffer zeroed
 * once and written by nothing but this call: they are the same elements every time for a fixed geometry) and only the in-window taps
 * are written — what the reference's CUDA kernel does after its zeros_like (lib/psa/functions/psamask.py:33, psamask_cuda.cu:58-106). */
int semseg_psamask_nhwc_backward(int psa_type, const float* daff, int lda, float* dmask, int ldm,
                                 int N, int H, int W, int mH, int mW, int dmask_prezeroed, hipStream_t stream);
int semseg_softmax_rows_fwd(const float* x, int ldx, float* y, int ldy, int rows, int P,
                            float alpha, int softmax, hipStream_t stream);
int semseg_softmax_rows_bwd(const float* y, int ldy, const float* dy, int lddy, float* dx,
                            int lddx, int rows, int P, float alpha, int softmax,
                            hipStream_t stream);
int semseg_transpose_batched(const float* in, int ldi, long long batch_stride_in, float* out,
                             int ldo, long long batch_stride_out, int batch, int R, int C,
                             hipStream_t stream);

/* ---- test-time pipeline kept on the device (tool/test.py:122-204, tool/demo.py:106-189):
 * resize_linear_hwc = cv2.resize(float32 HWC, INTER_LINEAR) (test.py:201); crop_normalize_flip =
 * mean-padded crop + ToTensor/normalise + [x, flip(x)] batch (test.py:123-132,156,171; origins are
 * (y,x) pairs in unpadded coordinates, on the device); softmax_flip_accumulate = softmax + flip average
 * + canvas/count accumulation (test.py:139-141,172-173); resize_accumulate_chw = /count, un-pad,
 * cv2.resize to the original size, sum over scales (test.py:175-177,203); argmax_chw (test.py:204). */
int semseg_resize_linear_hwc(const float* src, int Hs, int Ws, float* dst, int Hd, int Wd, int C,
                             hipStream_t stream);
int semseg_crop_normalize_flip(const float* img_hwc, int H, int W, const int* origins_dev, int K,
                               int crop_h, int crop_w, const float* mean3, const float* std3,
                               float* out_nchw, hipStream_t stream);
int semseg_softmax_flip_accumulate(const float* logits_nchw, const int* pos_dev, int K, int C,
                                   int crop_h, int crop_w, float* canvas_chw, float* count, int Hc,
                                   int Wc, hipStream_t stream);
int semseg_resize_accumulate_chw(const float* canvas_chw, const float* count, int Hc, int Wc, int y0,
                                 int x0, int Hs, int Ws, float* dst_chw, int Hd, int Wd, int C,
                                 float weight, hipStream_t stream);
int semseg_argmax_chw(const float* prob_chw, long long* out, int C, int H, int W, hipStream_t stream);

/* ---- training input pipeline on the device: the reference's transform chain
 * util/transform.py:76-241 (composed at tool/train.py:194-201,209-212, applied per sample at util/dataset.py:67-69),
 * which runs cv2 on CPU workers.  The host plans every sample first (parameters depend only on sizes) and hands the
 * device one op descriptor per sample per round; a round is ONE launch for the whole batch.  Each op reads the
 * materialised region (ROI) of a virtual src_H x src_W image and writes the region dst_{y0,x0,h,w} of its virtual
 * dst_H x dst_W output: only pixels that can reach the final crop are ever computed.  Pointers are device addresses.
 *   RESIZE  cv2.resize INTER_LINEAR (float image) + INTER_NEAREST (uint8 label), transform.py:71-72,101-102;
 *           p[0], p[1] = 1/inv_scale_x, 1/inv_scale_y as OpenCV forms them
 *   ROTATE  cv2.warpAffine INTER_LINEAR / INTER_NEAREST, BORDER_CONSTANT, transform.py:192-194;
 *           p[0..5] = the inverted 2x3 matrix (source = M * destination)
 *   BLUR    cv2.GaussianBlur((k,k), 0), k in {1,3,5,7}, BORDER_REFLECT_101, transform.py:226 (label passes through)
 *   GATHER  a chain of index maps applied output -> source: cv2.flip (transform.py:204-205,215-216),
 *           copyMakeBorder + slice (Crop, transform.py:144-164), channel swap (transform.py:233,240); with out_chw
 *           also ToTensor (transform.py:24-41) and Normalize (transform.py:54-61): float [3,h,w] + int64 [h,w].
 * Image source is uint8 (src_u8 = 1, the decoded image) or float32 HWC; intermediate outputs are float32 HWC +
 * uint8 label, dense over the destination ROI. */
enum { SEMSEG_AUG_NONE = 0, SEMSEG_AUG_RESIZE = 1, SEMSEG_AUG_ROTATE = 2, SEMSEG_AUG_BLUR = 3, SEMSEG_AUG_GATHER = 4 };
#define SEMSEG_AUG_MAX_MAPS 6
typedef struct semseg_aug_map {
  int in_h, in_w;        /* size of the map's input image; outside it the output takes pad / pad_lab */
  int sy, oy, sx, ox;    /* input (y, x) = (sy*y + oy, sx*x + ox), sy, sx in {+1, -1} */
  int swap_rb;           /* output channel c reads input channel 2-c */
  int pad_lab;
  float pad[3];
  int reserved;
} semseg_aug_map;
typedef struct semseg_aug_op {
  int kind, src_u8;
  unsigned long long src_img, src_lab, dst_img, dst_lab;
  int src_H, src_W, src_y0, src_x0, src_h, src_w;
  int dst_H, dst_W, dst_y0, dst_x0, dst_h, dst_w;
  double p[6];
  float pad[3];
  int pad_lab;
  int ksize;
  int n_maps;
  int out_chw;           /* GATHER: 1 = write float CHW planes [3, dst_h, dst_w] + int64 label */
  int normalize;         /* with out_chw: 0 none, 1 subtract mean, 2 subtract mean and divide by std */
  float mean[3], std[3];
  int reserved[2];
  semseg_aug_map maps[SEMSEG_AUG_MAX_MAPS];
} semseg_aug_op;
/* ops_dev: n_samples descriptors of this round (kind NONE = the sample has no op in this round); max_pixels = the
 * largest dst_h*dst_w among them (sizes the grid). */
int semseg_augment_round(const semseg_aug_op* ops_dev, int n_samples, int max_pixels, hipStream_t stream);
int semseg_aug_op_size(void);   /* sizeof(semseg_aug_op), for binding self-checks */

/* ---- intersectionAndUnionGPU (util/util.py:55-67; tool/train.py:286,375): one pass over int64
 * prediction/target, hist3K = 3*K uint64 scratch, outputs fp32 [K] each like torch.histc returns. */
int semseg_intersection_and_union(const long long* pred, const long long* target, size_t n, int K,
                                  int ignore_index, unsigned long long* hist3K,
                                  float* area_intersection, float* area_union, float* area_target,
                                  hipStream_t stream);

/* Dropout2d(p) keep/scale mask, one value per (n, c) plane (model/pspnet.py:68,76): 1/(1-p) with probability 1-p,
 * else 0; counter-based generator keyed by (seed, offset, plane).  semseg_memset_zero: hipMemsetAsync on the stream. */
int semseg_dropout2d_mask(float* mask, int n, float p, unsigned long long seed, unsigned long long offset,
                          const unsigned long long* offset_dev, hipStream_t stream);   /* offset_dev (optional): offset += *offset_dev, read on the device */
int semseg_memset_zero(void* ptr, size_t bytes, hipStream_t stream);

/* ---- torch.optim.SGD step (tool/train.py:140,276) over a flat range.  skip_dev (may be NULL): a device int; when it is
 * non-zero at execution time the launch leaves w and mom untouched (the error flag of semseg_xchg_allreduce_f64: a step
 * whose SyncBN statistics timed out must not reach the weights). */
int semseg_sgd_step(float* w, const float* g, float* mom, size_t n, float lr, const float* lr_dev,
                    float momentum, float weight_decay, float grad_scale, int first_step,
                    const int* skip_dev, hipStream_t stream);

/* ---- Step plan: launch sequencing below the C ABI (csrc/plan.hip).  The loop body of the reference's train step
 * (tool/train.py:269-276: model(input, target), loss, zero_grad, backward, optimizer.step) is ~1 200 launches of the entry
 * points above with arguments that do not change from step to step.  A host driver records them once — semseg_plan_append:
 * entry point (semseg_plan_fn_id of its name) + one 64-bit slot per argument: pointers / integers / hipStream_t by value
 * (integers sign-extended), float / double by bit pattern in the low 32 / all 64 bits — and replays them from C:
 * semseg_plan_replay calls entries [first, last) in order and stops at the first non-zero return code (returned;
 * semseg_plan_failed_entry names the entry).  A range that contains a collective is replayed in
 * segments around it.  Per-step values live in device memory: semseg_step_state_set writes {lr, lr_head} (read by
 * semseg_sgd_step through lr_dev) and the dropout call offset (semseg_dropout2d_mask's offset_dev) on the stream.
 * A driver accepts a record only when the NEXT step, recorded the same way, holds the same calls (semseg_plan_compare): a
 * step whose launch sequence depends on anything but the recorded arguments is never replayed.
 * semseg_plan_* entry points themselves cannot be recorded.  semseg_stream_wait_stream(waiter, signaller): work enqueued on
 * `waiter` afterwards runs after the work enqueued on `signaller` so far (event record + wait).
 * semseg_host_probe: host-only, stores its arguments in host_out[0..5] and counts calls in host_out[6]; returns
 * SEMSEG_EINVAL for a == -12345 (tests of the slot encoding on machines without a GPU). */
int semseg_plan_create(void** plan);
int semseg_plan_destroy(void* plan);
int semseg_plan_fn_id(const char* name);
int semseg_plan_fn_nargs(int fn_id);
int semseg_plan_append(void* plan, int fn_id, int nargs, const unsigned long long* slots);
int semseg_plan_size(void* plan);
int semseg_plan_entry_fn(void* plan, int entry);
int semseg_plan_set_slot(void* plan, int entry, int arg, unsigned long long bits);
int semseg_plan_get_slot(void* plan, int entry, int arg, unsigned long long* bits);
int semseg_plan_compare(void* plan_a, void* plan_b, int ignore_fn, int ignore_arg, int* where);   /* 0 equal; 1 differ at entry where[0], argument where[1] (-1: entry point / count); argument ignore_arg of entry point ignore_fn is not compared */
int semseg_plan_replay(void* plan, int first, int last);
int semseg_plan_failed_entry(void* plan);
int semseg_stream_wait_stream(hipStream_t waiter, hipStream_t signaller);
int semseg_step_state_set(float* lr_dev2, float lr, float lr_head, unsigned long long* drop_dev,
                          unsigned long long drop_offset, hipStream_t stream);
/* semseg_step_state_set + the copies of the caller's batch (x) and targets (y) into the buffers the record points at, in ONE
 * launch (any byte counts, any alignment: 16-byte aligned pairs are copied in 16-byte words, 4-byte aligned ones in dwords). */
int semseg_step_begin(void* x_dst, const void* x_src, size_t x_bytes, void* y_dst, const void* y_src, size_t y_bytes,
                      float* lr_dev2, float lr, float lr_head, unsigned long long* drop_dev, unsigned long long drop_offset,
                      hipStream_t stream);
int semseg_host_probe(unsigned long long* host_out, int a, long long b, size_t c, float d, double e, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMSEG_HIP_H */
