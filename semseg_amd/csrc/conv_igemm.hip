// Implicit-GEMM convolution for gfx950: forward and data-gradient of the
// PSPNet/PSANet convolutions (reference: model/resnet.py:63-69,108-112,134; model/pspnet.py:15,
// 49-58,65,69,73,77; model/psanet.py:25-48 — all nn.Conv2d, fp32).
//
// Layout: activations NHWC fp32 with an explicit channel stride (ld) so channel-concatenated
// buffers (PPM/PSA concat) are addressed in place.  Weights are re-packed once per optimizer step
// (pack_weights below) into a K-contiguous "B^T" panel whose K order is
//   k = ((c / 32) * R*S + tap) * 32 + (c % 32)
// i.e. all taps of one 32-channel chunk are adjacent, so the 9 shifted re-reads of a 3x3 tap loop
// stay L1/L2-resident.  Arithmetic is exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32).
//
// Tile: 256 threads = 4 waves (2x2), block tile BM x BN x 32, each wave (BM/2)x(BN/2) as
// 32x32 MFMA blocks; global->register prefetch of K-step t+1 overlaps the MFMAs of step t.
// (The weight gradient lives in conv_wgrad.hip since round 4.)
#include <cstdlib>
#include "conv_common.h"
#include "gemm_bf16split.h"

namespace {

constexpr int BK = 32;   // K-step (floats)
constexpr int LDK = 36;  // LDS row stride (floats): 144 B keeps ds_read_b128 conflict-free

constexpr int CONV_OCC = 2;   // resident workgroups per CU the forward / data-gradient kernel is compiled for
#ifndef SPLITK_BATCH
#define SPLITK_BATCH 1   // split-K reductions: several slabs' loads in flight per trip (0 = one slab per trip; A/B builds)
#endif

struct ConvArgs {
  const float* x;   // A source: activations (fwd) or output-gradient (dgrad), NHWC
  const float* w;   // packed B^T panel [Nout_pad][KT*32]
  float* y;         // output [M][ldy]
  const float* bias;  // optional [Nout]
  const float* scale; // optional [Nout]: y = acc*scale + bias (eval-mode BatchNorm folded in)
  int relu;           // clamp at 0 after bias / add (eval-mode BN + ReLU (+ residual) epilogue)
  const float* add;   // optional [M][ldadd] added in the epilogue
  double* stats;      // optional [2*Nout]: per-channel sum, sum of squares (fp64 atomics)
  int ldx, ldy, ldadd;
  int N, Hin, Win;    // spatial dims of the A source
  int Hout, Wout;     // spatial dims of the output
  int Kc;             // channels of the A source per tap (multiple of 32)
  int Nout;           // valid output channels
  int R, S, stride, pad, dil;
  int M;              // N*Hout*Wout
  int tiles_n;
  int stats_nslot;    // stats is [nslot][2*Nout]; tile_m % nslot picks the replica
  // Split-K ("stream-K tail"): the first full_tiles tiles are computed whole; every later tile is
  // cut into ksplit K-ranges, workgroup (tile, ks) covering K-steps [ks*kt_per, (ks+1)*kt_per) and
  // writing a raw partial slab that splitk_epilogue_kernel reduces.  full_tiles == 0: all tiles split.
  int full_tiles, ksplit, kt_per;
  float* part;        // partial slabs [ksplit][M - tail_m0][ldpart]
  int ldpart, tail_m0;
  FastDiv div_hw, div_w, div_tn, div_ks;  // Hout*Wout, Wout, tiles_n, ksplit
  // Batched GEMM use (PSA point-affinity contraction, model/psanet.py:90-91): blockIdx.y = batch item, every
  // operand advances by its batch stride (floats).  batch == 1: plain convolution.
  int batch;
  long long x_bs, w_bs, y_bs, add_bs;
  // Fused BatchNorm-backward reduction (data gradient only; bnr_n = 0: off).  The tile this kernel produces is the
  // COMPLETE gradient dout of a BatchNorm(+ReLU) output, so the epilogue does what bn_bwd_reduce would do in a
  // separate pass over HBM: g = dout * (act > 0) is what gets stored, and sum g, sum g * xhat are accumulated in fp64
  // for up to two BatchNorm layers that share g (bn3 + the downsample BN of a bottleneck, model/resnet.py:88-92).
  int bnr_n;
  const float* bnr_mask;          // post-ReLU activation [M][bnr_ldm] (nullptr: no ReLU)
  int bnr_ldm;
  const unsigned* bnr_bits;       // the same mask as bits: bit (c & 31) of word [m][c >> 5] (non-null: replaces bnr_mask)
  int bnr_ldb;
  const float* bnr_y[2];          // pre-BN tensors [M][bnr_ldy]
  int bnr_ldy[2];
  const float* bnr_mean[2];
  const float* bnr_invstd[2];
  double* bnr_sums[2];            // [stats_nslot][2 * Nout]
  int gm;                         // tile rows per group of the workgroup order (decode_tile); <= 1: row-major
  // In-kernel reduction of split tiles (round 6; cnt != nullptr): every (tile, ks) workgroup stores its accumulators as they lie
  // in registers into slab [tile - full_tiles][ks] of `part` (write-through stores) and draws a ticket from cnt[tile - full_tiles];
  // the workgroup that draws the last one sums the ksplit slabs in slice order and runs the UNSPLIT epilogue on the sum — no
  // splitk_epilogue_kernel launch, no second trip of the partial sums through a [ksplit][M][N] array.  cnt is zero before the
  // launch and left zero.
  unsigned* cnt;
};

// Linear tile index -> (tile_m, tile_n).  Whole (unsplit) tiles are walked in groups of `gm` tile rows, row index
// fastest inside a group: the ~64 workgroups resident on an XCD then cover gm row blocks x 64/gm weight panels and
// stream K together, instead of 64/tiles_n rows x ALL weight panels (cls.0's data gradient: 32 panels = 75 MB of
// weights re-streamed from HBM for every pair of tile rows).  The stream-K tail keeps the row-major order its slab
// addressing assumes (tail tiles are the last tail rows).
__device__ __forceinline__ void decode_tile(const ConvArgs& p, int tile, int& tile_m, int& tile_n) {
  if (p.gm > 1 && tile < p.full_tiles) {
    const int rows_full = p.full_tiles / p.tiles_n;
    const int per_group = p.gm * p.tiles_n;
    const int g = tile / per_group;
    const int first = g * p.gm;
    const int gs = min(p.gm, rows_full - first);
    const int t = tile - g * per_group;
    tile_n = t / gs;
    tile_m = first + (t - tile_n * gs);
  } else {
    tile_m = fdiv(tile, p.div_tn);
    tile_n = tile - tile_m * p.tiles_n;
  }
}

// Fused BatchNorm-backward reduction of one 64-row x 32-column block of a wave (the block sits transposed in the wave's
// LDS slab `wl`): g = (acc (+ add)) * (act > 0) is stored and sum g, sum g * xhat accumulate per lane column in fp32 (`bs`,
// 8 rows per lane; fp64 across lanes / workgroups afterwards) for one or two BatchNorm layers sharing g.
// Shape of the loop: the operand tiles (ReLU mask, pre-BN tensor(s), residual gradient) are HBM /
// L2 reads with ~1-2 us of latency under load, and round 2-3 issued them one row group at a time (16 dependent round trips
// per wave and tile: the family sat at 0.59 of peak, bound by its epilogue in either arithmetic).  Here the loads of FOUR
// row groups are issued back to back before the first is used (2 round trips per block), and the loop exists once per
// (residual add, second BatchNorm) combination, so that absent operands are not loaded at all (round 2-3 aliased them to a
// present one: 4 loads per row where the two common cases need 2 and 3).  The registers come from the K loop's staging
// and second-level accumulators, which are dead here; the outer loop is NOT unrolled so that the compiler cannot hoist
// all eight row groups (223 VGPRs when it did).  Out-of-range rows / columns load from a clamped address and are dropped
// at the store.
#ifndef BNR_GROUP
#define BNR_GROUP 4      // row groups whose operand loads are in flight together (1 = the round-3 loop shape; A/B builds)
#endif
// BITS: the ReLU mask comes as one bit per element (written by semseg_bn_apply) instead of the post-ReLU activation itself:
// 1/32 of the bytes of the largest operand of this epilogue.
template <bool HAS_ADD, bool TWO, bool BITS>
__device__ __forceinline__ void bnr_rows(const ConvArgs& p, const float* wl, float* dst, int ldd, int mrow0, int colmax,
                                         int mb, int cg, int c4, int lane, const float* add_in, const f32x4 (&bmu)[2],
                                         const f32x4 (&bis)[2], float (&bs)[2][8]) {
  constexpr int G = BNR_GROUP;
  const bool has_mask = !BITS && p.bnr_mask != nullptr;
  const float* maskp = has_mask ? p.bnr_mask : p.bnr_y[0];     // no ReLU: any readable tile, neutralised below
  const int ldmp = has_mask ? p.bnr_ldm : p.bnr_ldy[0];
  const int cc = cg < colmax ? cg : 0;
#pragma nounroll
  for (int hb = 0; hb < 8 / G; ++hb) {
    f32x4 aa[BITS ? 1 : G], y0[G], dd[HAS_ADD ? G : 1], y1[TWO ? G : 1];
    unsigned wb[BITS ? G : 1];
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int m = mb + (lane >> 3) + 8 * (hb * G + q);
      const size_t mm = (size_t)(m < p.M ? m : p.M - 1);
      if constexpr (BITS) wb[q] = p.bnr_bits[mm * p.bnr_ldb + (cc >> 5)];
      else aa[q] = *reinterpret_cast<const f32x4*>(maskp + mm * ldmp + cc);
      y0[q] = *reinterpret_cast<const f32x4*>(p.bnr_y[0] + mm * p.bnr_ldy[0] + cc);
      if constexpr (HAS_ADD) dd[q] = *reinterpret_cast<const f32x4*>(add_in + mm * p.ldadd + cc);
      if constexpr (TWO) y1[q] = *reinterpret_cast<const f32x4*>(p.bnr_y[1] + mm * p.bnr_ldy[1] + cc);
    }
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int lr = (lane >> 3) + 8 * (hb * G + q);
      const int m = mb + lr;
      const bool ok = m < p.M && cg < colmax;
      f32x4 v = *reinterpret_cast<const f32x4*>(&wl[lr * LDK + c4]);
      if constexpr (HAS_ADD) v += dd[q];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bool keep;
        if constexpr (BITS) keep = (wb[q] >> ((cc & 31) + k)) & 1u;
        else keep = !has_mask || aa[q][k] > 0.f;
        v[k] = (keep && ok) ? v[k] : 0.f;
      }
      const f32x4 xh0 = (y0[q] - bmu[0]) * bis[0];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bs[0][k] += v[k];
        bs[0][4 + k] = fmaf(v[k], xh0[k], bs[0][4 + k]);
      }
      if constexpr (TWO) {
        const f32x4 xh1 = (y1[q] - bmu[1]) * bis[1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bs[1][k] += v[k];
          bs[1][4 + k] = fmaf(v[k], xh1[k], bs[1][4 + k]);
        }
      }
      if (ok) *reinterpret_cast<f32x4*>(dst + (size_t)(m - mrow0) * ldd + cg) = v;
    }
  }
}

// ---- epilogue shared by the register-staged and the direct-to-LDS kernels ----
// STATS: the fp64 per-channel statistics of the forward epilogue exist (forward kernels); data-gradient kernels never take
// statistics, and dropping the per-element fp64 conversion / accumulation from them is ~190 fp64 VALU ops per lane and block
// BNR: the fused BatchNorm-backward reduction is compiled in (data-gradient kernels only: in the forward kernels it was dead
// code that cost them ~30 VGPRs and, under bf16x3, 5 % of their time)
template <int BM, int BN, bool STATS = true, bool BNR = true>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[2][BN / 64], float* smem, bool split,
                                              int ks, int m0, int n0, int tile_m, double* red2, float* y_out,
                                              const float* add_in) {
  constexpr int NT = BM * 2;
  constexpr int MREP = 2, NREP = BN / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;
  // The MFMA accumulator layout gives each lane ONE column and 32 rows of its wave's 64 x 32 block: a
  // direct store is 32 dword stores per lane and is store-issue bound (~16k cycles per tile, 30 % of a
  // K = 256 tile).  Each wave therefore transposes block by block through its private 64 x 36-float LDS
  // slab and stores 16-byte lanes (8 per block instead of 32); bias / residual add ride along.
  // Statistics are taken from the accumulators (+bias) in registers, in fp64.
  const int Nout4 = (p.Nout + 3) & ~3;
  float* const dst = split ? p.part + (size_t)ks * (p.M - p.tail_m0) * p.ldpart : y_out;
  const int ldd = split ? p.ldpart : p.ldy;
  const int mrow0 = split ? p.tail_m0 : 0;           // partial slabs start at the tail's first row
  const int colmax = split ? p.ldpart : Nout4;       // widest column a 16-byte store may touch
  const bool wide = split || ((p.ldy & 3) == 0 && p.ldy >= Nout4 && (!add_in || (p.ldadd & 3) == 0));
  float* wl = smem + wave * (64 * LDK);              // this wave's slab (needs >= NT/64 * 64 * LDK floats)
  double* red = reinterpret_cast<double*>(smem);     // [BM/64 (wm)][BN][2], used after the stores
  const bool bnr = BNR && !split && p.bnr_n > 0;     // fused BatchNorm-backward reduction (wide stores only)
  double st1[NREP], st2[NREP];
#pragma unroll
  for (int j = 0; j < NREP; ++j) {
    const int lcol = wn * (BN / 2) + j * 32 + l31;
    const int col = n0 + lcol;
    const bool cok = col < p.Nout;
    const float bv = (!split && p.bias && cok) ? p.bias[col] : 0.f;
    const float sv = (!split && p.scale && cok) ? p.scale[col] : 1.f;
    const bool relu = !split && p.relu;
    double s1 = 0.0, s2 = 0.0;
    f32x4 bmu[2], bis[2];
    float bs[2][8];
    if (bnr) {
      const int cb = n0 + wn * (BN / 2) + j * 32 + (lane & 7) * 4;      // this lane's 4 columns in the stores below
#pragma unroll
      for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int k = 0; k < 8; ++k) bs[b][k] = 0.f;
        if (b < p.bnr_n && cb < Nout4) {
          bmu[b] = *reinterpret_cast<const f32x4*>(p.bnr_mean[b] + cb);
          bis[b] = *reinterpret_cast<const f32x4*>(p.bnr_invstd[b] + cb);
        } else {
          bmu[b] = f32x4{0.f, 0.f, 0.f, 0.f};
          bis[b] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    if (wide) {
#pragma unroll
      for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int lr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          const float v = acc[i][j][e] * sv + bv;
          wl[lr * LDK + l31] = v;
          if constexpr (STATS) {
            if (m0 + wm * 64 + lr < p.M && cok) {
              const double dv = (double)v;
              s1 += dv;
              s2 += dv * dv;
            }
          }
        }
      // same-wave LDS traffic is ordered: no barrier needed between the writes above and these reads
      if (bnr) {
        const int c4 = (lane & 7) * 4;
        const int cg = n0 + wn * (BN / 2) + j * 32 + c4;
        const int mb = m0 + wm * 64;
        const bool has_add = add_in != nullptr, two = p.bnr_n > 1;
        // one specialised copy of the row loop per (residual add, second BatchNorm) combination: a wave-uniform switch
        // outside the loop instead of aliased or guarded loads inside it
#define BNR_CALL(A_, T_, B_) bnr_rows<A_, T_, B_>(p, wl, dst, ldd, mrow0, colmax, mb, cg, c4, lane, add_in, bmu, bis, bs)
        if (p.bnr_bits) {
          if (has_add && two) BNR_CALL(true, true, true);
          else if (has_add) BNR_CALL(true, false, true);
          else if (two) BNR_CALL(false, true, true);
          else BNR_CALL(false, false, true);
        } else {
          if (has_add && two) BNR_CALL(true, true, false);
          else if (has_add) BNR_CALL(true, false, false);
          else if (two) BNR_CALL(false, true, false);
          else BNR_CALL(false, false, false);
        }
#undef BNR_CALL
      } else {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int idx = lane + 64 * t;
        const int lr = idx >> 3, c4 = (idx & 7) * 4;
        const int m = m0 + wm * 64 + lr;
        const int cg = n0 + wn * (BN / 2) + j * 32 + c4;
        f32x4 v = *reinterpret_cast<const f32x4*>(&wl[lr * LDK + c4]);
        if (m < p.M && cg < colmax) {
          if (!split && add_in) v += *reinterpret_cast<const f32x4*>(add_in + (size_t)m * p.ldadd + cg);
          if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
          }
          *reinterpret_cast<f32x4*>(dst + (size_t)(m - mrow0) * ldd + cg) = v;
        }
      }
      }
      if (bnr) {
        // lanes with equal (lane & 7) hold the same 4 columns for different rows: fold them, lanes 0-7 keep the totals
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (b < p.bnr_n) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              double t = (double)bs[b][k];
              t += shfl_xor_f64(t, 8);
              t += shfl_xor_f64(t, 16);
              t += shfl_xor_f64(t, 32);
              if (lane < 8) red2[((b * (BM / 64) + wm) * BN + wn * (BN / 2) + j * 32 + lane * 4 + (k & 3)) * 2 + (k >> 2)] = t;
            }
          }
      }
    } else {
#pragma unroll
      for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
          if (m < p.M && cok) {
            float v = acc[i][j][e] * sv + bv;
            if constexpr (STATS) {
              const double dv = (double)v;
              s1 += dv;
              s2 += dv * dv;
            }
            if (add_in) v += add_in[(size_t)m * p.ldadd + col];
            if (relu) v = fmaxf(v, 0.f);
            y_out[(size_t)m * p.ldy + col] = v;
          }
        }
    }
    st1[j] = s1;
    st2[j] = s2;
  }
  if (STATS && !split && p.stats) {
    __syncthreads();  // every wave is done with its transposition slab
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const int lcol = wn * (BN / 2) + j * 32 + l31;
      const double s1 = st1[j] + shfl_xor_f64(st1[j], 32);
      const double s2 = st2[j] + shfl_xor_f64(st2[j], 32);
      if (lhi == 0) {
        red[(wm * BN + lcol) * 2 + 0] = s1;
        red[(wm * BN + lcol) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int col = n0 + tid;
      if (col < p.Nout) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int r = 0; r < BM / 64; ++r) {
          s1 += red[(r * BN + tid) * 2 + 0];
          s2 += red[(r * BN + tid) * 2 + 1];
        }
        double* st = p.stats + (size_t)(tile_m % p.stats_nslot) * 2 * p.Nout;
        atomic_add_f64(&st[col], s1);
        atomic_add_f64(&st[p.Nout + col], s2);
      }
    }
  }
  if (bnr) {
    __syncthreads();   // red2 holds every wave's column sums: [bn][wm][BN][2]
    if (tid < BN) {
      const int col = n0 + tid;
      if (col < p.Nout) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (b < p.bnr_n) {
            double s1 = 0.0, s2 = 0.0;
#pragma unroll
            for (int r = 0; r < BM / 64; ++r) {
              s1 += red2[((b * (BM / 64) + r) * BN + tid) * 2 + 0];
              s2 += red2[((b * (BM / 64) + r) * BN + tid) * 2 + 1];
            }
            double* st = p.bnr_sums[b] + (size_t)(tile_m % p.stats_nslot) * 2 * p.Nout;
            atomic_add_f64(&st[col], s1);
            atomic_add_f64(&st[p.Nout + col], s2);
          }
      }
    }
  }
}

// RS_T == 0: generic tap walk with global loads.  RS_T == 1 / 9 (1x1 / 3x3): the tap loop is unrolled
// and the gather uses buffer loads with per-(row,tap) byte offsets precomputed in VGPRs (invalid taps
// carry an out-of-range offset, which the buffer unit returns as 0) plus one scalar offset per K-step —
// no per-K-step address VALU between the MFMAs at all.
//
// SP = 3 (SEMSEG_ARITH_BF16X3, include/semseg_hip.h) cuts each fp32 operand into three
// bf16 pieces between the global load and the LDS store and forms the product from six v_mfma_f32_32x32x16_bf16 per
// 16 K instead of eight v_mfma_f32_32x32x2_f32 — same gather, same accumulators, same epilogues.

template <int BM, int BN, bool TR, int RS_T, bool TL, int SP = 0>
__global__ __launch_bounds__(BM * 2, CONV_OCC) void conv_igemm_kernel(const ConvArgs pin) {
  ConvArgs p = pin;
  if (p.batch > 1) {
    const long long bz = blockIdx.y;
    p.x += bz * p.x_bs;
    p.w += bz * p.w_bs;
    p.y += bz * p.y_bs;
    if (p.add) p.add += bz * p.add_bs;
  }
  // BM/64 x 2 waves, each a 64 x (BN/2) sub-tile of 32x32 MFMA blocks
  constexpr int NT = BM * 2;                 // threads
  constexpr int RSTEP = NT / 8;              // tile rows staged per pass (8 lanes x 16 B per row)
  constexpr int MREP = 2, NREP = BN / 64;
  constexpr int A_PER = BM / RSTEP, B_PER = BN / RSTEP;
  constexpr int STAGE = SP ? SP * (BM + BN) * (BK / 2) : (BM + BN) * LDK;   // SP: bf16 piece planes [piece][row][32]
  constexpr int EPI = (NT / 64) * 64 * LDK;  // per-wave transposition slabs of the epilogue
  constexpr int SMEM_BASE = STAGE > EPI ? STAGE : EPI;
  constexpr int RED2_F = TR ? 2 * (BM / 64) * BN * 2 * 2 : 0;   // fp64 column sums of the fused BatchNorm-backward reduction
  // they live behind the epilogue's transposition slabs — inside the staging area when that is larger than the slabs (the
  // bf16x3 128 x 128 instances: 48 KB instead of 56 KB, i.e. three workgroups per CU instead of two), else behind it
  constexpr int RED2_OFF = (EPI + RED2_F <= SMEM_BASE) ? EPI : SMEM_BASE;
  constexpr int SMEM_F = (RED2_OFF + RED2_F > SMEM_BASE) ? RED2_OFF + RED2_F : SMEM_BASE;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_F];
  float* As = smem;
  float* Bs = smem + BM * LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  int ks = 0, tile;
  bool split = false;
  if ((int)blockIdx.x < p.full_tiles) {
    tile = xcd_remap(blockIdx.x, p.full_tiles);
  } else {
    const int u = xcd_remap(blockIdx.x - p.full_tiles, gridDim.x - p.full_tiles);
    const int uq = fdiv(u, p.div_ks);
    ks = u - uq * p.ksplit;
    tile = p.full_tiles + uq;
    split = p.ksplit > 1;
  }
  int tile_m, tile_n;
  decode_tile(p, tile, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int RS = p.R * p.S;
  const int nchunk = p.Kc / BK;
  const int KT_all = RS * nchunk;
  const size_t wK = (size_t)KT_all * BK;  // packed panel row length
  const int kt0 = split ? ks * p.kt_per : 0;
  const int KT = split ? min(KT_all, kt0 + p.kt_per) : KT_all;  // this workgroup walks [kt0, KT)

  // ---- per-thread staging assignment ----
  // The gather address of (row, tap) is separable: off = rowbase[row] + tapoff(tap) + channel, and
  // its validity is one bit of a per-row tap mask, both computed once here; the K loop then walks
  // (chunk, r, s) with scalar counters only (no divisions, no divergent branches around the loads).
  const int kq = tid & 7;     // which float4 of the 32-float K-step
  const int lrow = tid >> 3;  // 0..RSTEP-1
  int a_base[A_PER];
  unsigned a_mask[A_PER];
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int m = m0 + lrow + RSTEP * i;
    const bool rok = m < p.M;
    const int mm = rok ? m : 0;
    const int hw = p.Hout * p.Wout;
    const int n = fdiv(mm, p.div_hw);
    const int rem = mm - n * hw;
    const int oh = fdiv(rem, p.div_w);
    const int ow = rem - oh * p.Wout;
    unsigned mask = 0;
    int bh, bw;  // tap-independent part of the source coordinate
    if (!TR) {
      bh = oh * p.stride - p.pad;
      bw = ow * p.stride - p.pad;
    } else {
      bh = (oh + p.pad) / p.stride;
      bw = (ow + p.pad) / p.stride;
    }
    for (int r = 0; r < p.R; ++r)
      for (int s = 0; s < p.S; ++s) {
        bool ok = rok;
        int ih, iw;
        if (!TR) {
          ih = bh + r * p.dil;
          iw = bw + s * p.dil;
        } else {
          const int rd = r * p.dil, sd = s * p.dil;
          ih = bh - rd / p.stride;
          iw = bw - sd / p.stride;
          ok = ok && ((oh + p.pad) % p.stride == rd % p.stride) && ((ow + p.pad) % p.stride == sd % p.stride);
        }
        ok = ok && ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Win;
        if (ok) mask |= 1u << (r * p.S + s);
      }
    a_mask[i] = mask;
    a_base[i] = ((n * p.Hin + bh) * p.Win + bw) * p.ldx + kq * 4;
  }
  const float* bptr[B_PER];
#pragma unroll
  for (int i = 0; i < B_PER; ++i) bptr[i] = p.w + (size_t)(n0 + lrow + RSTEP * i) * wK + kq * 4;

  f32x4 ra[A_PER], rb[B_PER];
  // scalar K-walk state of the NEXT tile to prefetch
  int pf_tap = kt0 % RS, pf_coff = (kt0 / RS) * BK;
  int pf_r = pf_tap / p.S, pf_s = pf_tap - (pf_tap / p.S) * p.S;

  auto prefetch = [&](int kt) {
    int toff;
    if (!TR)
      toff = (pf_r * p.dil * p.Win + pf_s * p.dil) * p.ldx;
    else
      toff = -(((pf_r * p.dil) / p.stride) * p.Win + (pf_s * p.dil) / p.stride) * p.ldx;
    toff += pf_coff;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const bool ok = (a_mask[i] >> pf_tap) & 1u;
      const float* src = ok ? p.x + (a_base[i] + toff) : g_zero_line;
      ra[i] = *reinterpret_cast<const f32x4*>(src);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      rb[i] = *reinterpret_cast<const f32x4*>(bptr[i] + (size_t)kt * BK);
    // advance (tap fastest, then channel chunk)
    ++pf_tap;
    if (++pf_s == p.S) {
      pf_s = 0;
      if (++pf_r == p.R) {
        pf_r = 0;
        pf_tap = 0;
        pf_coff += BK;
      }
    }
  };

  f32x16 acc[MREP][NREP];
#pragma unroll
  for (int i = 0; i < MREP; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // Two-level accumulation: the MFMA chain sums sequentially along K, so its rounding noise grows ~sqrt(K)
  // (rms 2.3e-6 at K = 36864, 7x a blocked CPU sum).  Every ~1024 K the chain is flushed into a second
  // accumulator set, which bounds the chain length.  TL is set by the launcher whenever K > 576 (240 VGPRs on the
  // 128 x 128 tile, still occupancy 2); shorter reductions keep the leaner kernel.
  constexpr bool TWO_LEVEL = TL;
  f32x16 acc2[TWO_LEVEL ? MREP : 1][TWO_LEVEL ? NREP : 1];
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
      for (int j = 0; j < NREP; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
  }
  auto flush = [&] {
    if constexpr (TWO_LEVEL) {
#pragma unroll
      for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            acc2[i][j][e] += acc[i][j][e];
            acc[i][j][e] = 0.f;
          }
    }
  };

  // SP layout: rows of 32 bf16 (64 bytes), the four 16-byte chunks of a row XOR-swizzled with (row / 4) % 4 — the 16-byte
  // fragment reads (16 rows per LDS cycle) and the 8-byte staging stores (128 contiguous bytes per 16 lanes) are both
  // conflict-free without padding.  Piece planes: A pieces first, then B pieces.
  auto sp_off = [](int row, int chunk) { return row * BK + ((chunk ^ ((row >> 2) & 3)) << 3); };
  auto sp_split_store = [&](__bf16* plane0, int plane_elems, int row, const f32x4 v) {
    f32x4 r = v;
#pragma unroll
    for (int c = 0; c < (SP ? SP : 1); ++c) {
      const bf16x4 pc = __builtin_convertvector(r, bf16x4);
      if (c + 1 < SP) r -= bf16x4_to_f32(pc);
      *reinterpret_cast<bf16x4*>(&plane0[c * plane_elems + sp_off(row, kq >> 1) + (kq & 1) * 4]) = pc;
    }
  };
  auto stage_store_from = [&](float* A_, float* B_, const f32x4 (&qa)[A_PER], const f32x4 (&qb)[B_PER]) {
    if constexpr (SP) {
      __bf16* Ap = reinterpret_cast<__bf16*>(smem);
      __bf16* Bp = Ap + SP * BM * BK;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) sp_split_store(Ap, BM * BK, lrow + RSTEP * i, qa[i]);
#pragma unroll
      for (int i = 0; i < B_PER; ++i) sp_split_store(Bp, BN * BK, lrow + RSTEP * i, qb[i]);
    } else {
#pragma unroll
      for (int i = 0; i < A_PER; ++i)
        *reinterpret_cast<f32x4*>(&A_[(lrow + RSTEP * i) * LDK + kq * 4]) = qa[i];
#pragma unroll
      for (int i = 0; i < B_PER; ++i)
        *reinterpret_cast<f32x4*>(&B_[(lrow + RSTEP * i) * LDK + kq * 4]) = qb[i];
    }
  };
  auto stage_store = [&](float* A_, float* B_) { stage_store_from(A_, B_, ra, rb); };
  // do_pf: issue the next tile's global loads after the first MFMA group, so their address VALU and
  // issue slots hide in the shadow of this wave's own MFMAs instead of preceding them
  auto compute = [&](const float* A_, const float* B_, auto&& issue_next) {
    if constexpr (SP) {
      const __bf16* Ap = reinterpret_cast<const __bf16*>(smem);
      const __bf16* Bp = Ap + SP * BM * BK;
#pragma unroll
      for (int k16 = 0; k16 < 2; ++k16) {
        bf16x8 a[MREP][SP ? SP : 1], b[NREP][SP ? SP : 1];
#pragma unroll
        for (int c = 0; c < SP; ++c) {
#pragma unroll
          for (int i = 0; i < MREP; ++i)
            a[i][c] = *reinterpret_cast<const bf16x8*>(&Ap[c * BM * BK + sp_off(wm * 64 + i * 32 + l31, k16 * 2 + lhi)]);
#pragma unroll
          for (int j = 0; j < NREP; ++j)
            b[j][c] = *reinterpret_cast<const bf16x8*>(&Bp[c * BN * BK + sp_off(wn * (BN / 2) + j * 32 + l31, k16 * 2 + lhi)]);
        }
        // small terms first, the leading product last; consecutive MFMAs go to different accumulators
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[q]], b[j][PB[q]], acc[i][j], 0, 0, 0);
        if (k16 == 0) issue_next();
      }
      return;
    }
#pragma unroll
    for (int k8 = 0; k8 < 4; ++k8) {
      f32x4 a[MREP], b[NREP];
#pragma unroll
      for (int i = 0; i < MREP; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(
            &A_[(wm * 64 + i * 32 + l31) * LDK + k8 * 8 + lhi * 4]);
#pragma unroll
      for (int j = 0; j < NREP; ++j)
        b[j] = *reinterpret_cast<const f32x4*>(
            &B_[(wn * (BN / 2) + j * 32 + l31) * LDK + k8 * 8 + lhi * 4]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
          for (int j = 0; j < NREP; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
      if (k8 == 0) issue_next();
    }
  };

  if constexpr (RS_T > 0) {
    // ---------------- buffer-load path: unrolled taps, zero address VALU in the K loop -------------
    constexpr unsigned OOB = 0x80000000u;          // >= num_records: the buffer unit returns 0
    const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x80000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x80000000, 0x00020000);
    // per-tap byte offsets (uniform -> SGPRs)
    int toffs[RS_T];
#pragma unroll
    for (int t = 0; t < RS_T; ++t) {
      const int r = t / p.S, s_ = t - (t / p.S) * p.S;
      if (!TR)
        toffs[t] = (r * p.dil * p.Win + s_ * p.dil) * p.ldx * 4;
      else
        toffs[t] = -(((r * p.dil) / p.stride) * p.Win + (s_ * p.dil) / p.stride) * p.ldx * 4;
    }
    unsigned voffB[B_PER];
    unsigned baseA[A_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) baseA[i] = (unsigned)(a_base[i] * 4);
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      voffB[i] = (unsigned)(((size_t)(n0 + lrow + RSTEP * i) * wK + kq * 4) * 4);
    const int c_begin = kt0 / RS_T, c_end = KT / RS_T;
    auto load_tile = [&](auto tapc, int c) {
      constexpr int t = decltype(tapc)::value;
      const int so_a = c * (BK * 4);
      const int so_b = (c * RS_T + t) * (BK * 4);
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const unsigned vo = ((a_mask[i] >> t) & 1u) ? baseA[i] + (unsigned)toffs[t] : OOB;
        ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx_, vo, so_a, 0));
      }
#pragma unroll
      for (int i = 0; i < B_PER; ++i)
        rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw_, voffB[i], so_b, 0));
    };
    auto step = [&](auto tapc, int c) {
      constexpr int t = decltype(tapc)::value;
      stage_store(As, Bs);
      __syncthreads();
      compute(As, Bs, [&] {
        if constexpr (t + 1 < RS_T) {
          load_tile(std::integral_constant<int, t + 1>{}, c);
        } else {
          if (c + 1 < c_end) load_tile(std::integral_constant<int, 0>{}, c + 1);
        }
      });
      __syncthreads();
    };
    if (c_begin < c_end) load_tile(std::integral_constant<int, 0>{}, c_begin);
    for (int c = c_begin; c < c_end; ++c) {
      step(std::integral_constant<int, 0>{}, c);
      if constexpr (RS_T == 9) {
        step(std::integral_constant<int, 1>{}, c);
        step(std::integral_constant<int, 2>{}, c);
        step(std::integral_constant<int, 3>{}, c);
        step(std::integral_constant<int, 4>{}, c);
        step(std::integral_constant<int, 5>{}, c);
        step(std::integral_constant<int, 6>{}, c);
        step(std::integral_constant<int, 7>{}, c);
        step(std::integral_constant<int, 8>{}, c);
      }
      // every 2 channel blocks x 9 taps (576 K) / every 16 channel blocks (512 K)
      if (((c - c_begin) & (RS_T == 9 ? 1 : 15)) == (RS_T == 9 ? 1 : 15)) flush();
    }
  } else {
  prefetch(kt0);
  for (int kt = kt0; kt < KT; ++kt) {
    stage_store(As, Bs);
    __syncthreads();
    compute(As, Bs, [&] { if (kt + 1 < KT) prefetch(kt + 1); });
    __syncthreads();
    if (((kt - kt0) & 15) == 15) flush();
  }
  }  // RS_T == 0
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
      for (int j = 0; j < NREP; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += acc2[i][j][e];
  }

  if (split && p.cnt) {
    // hand-off of cdna_hip_programming.md guideline 16 (counter form): write-through (sc1) payload stores, every wave drains them,
    // barrier, ONE lane draws a relaxed agent-scope ticket; the last arriver issues ONE agent-scope acquire and reads the slabs back
    constexpr int QN = MREP * NREP * 4;                         // float4 per thread
    const int st = tile - p.full_tiles;
    const __amdgpu_buffer_rsrc_t rp_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.part, 0, 0x80000000, 0x00020000);
    const unsigned slab_bytes = (unsigned)(BM * BN * 4);
    const unsigned my = ((unsigned)st * (unsigned)p.ksplit + (unsigned)ks) * slab_bytes + (unsigned)tid * 16u;
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
      for (int j = 0; j < NREP; ++j)
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const f32x4 v = {acc[i][j][4 * e4], acc[i][j][4 * e4 + 1], acc[i][j][4 * e4 + 2], acc[i][j][4 * e4 + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp_, my + (unsigned)(((i * NREP + j) * 4 + e4) * NT * 16), 0, 16);
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      const unsigned tk = __hip_atomic_fetch_add(p.cnt + st, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = tk == (unsigned)p.ksplit - 1u;
      if (last) {
        __hip_atomic_store(p.cnt + st, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      *flag = last;
    }
    __syncthreads();
    const int last = *flag;
    __syncthreads();                 // the epilogue reuses this LDS
    if (!last) return;
    // slice order, every slab read back (this workgroup's own included: the sum does not depend on who arrives last)
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
      for (int j = 0; j < NREP; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const unsigned base0 = (unsigned)st * (unsigned)p.ksplit * slab_bytes + (unsigned)tid * 16u;
    for (int k = 0; k < p.ksplit; ++k) {
      f32x4 t[QN];
#pragma unroll
      for (int q = 0; q < QN; ++q)
        t[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rp_, base0 + (unsigned)k * slab_bytes + (unsigned)(q * NT * 16), 0, 16));
#pragma unroll
      for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[i][j][4 * e4 + c] += t[(i * NREP + j) * 4 + e4][c];
    }
    split = false;
    ks = 0;
  }
  conv_epilogue<BM, BN, !TR, TR>(p, acc, smem, split, ks, m0, n0, tile_m, reinterpret_cast<double*>(smem + RED2_OFF), p.y, p.add);
}

// Split-K epilogue: y = sum_ks part[ks] (+bias) (+add); optional fp64 channel statistics.
// Thread = one float4 of channels, rows strided over the grid (same tiling as the BN reductions).
struct BnrArgs {   // fused BatchNorm-backward reduction (see ConvArgs::bnr_*); pointers already at the first row handled
  int n;
  const float* mask; int ldm;
  const unsigned* bits; int ldb;     // the mask as bits (non-null: replaces mask)
  const float* y[2]; int ldy[2];
  const float* mean[2]; const float* invstd[2];
  double* sums[2];
};

__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const float* __restrict__ part,
                                                              int ksplit, int ldpart, float* y,
                                                              int ldy, const float* bias,
                                                              const float* scale, int relu,
                                                              const float* add, int ldadd,
                                                              double* stats, int nslot, int M,
                                                              int Nout, int tpr, int rpb, const BnrArgs bn) {
  __shared__ double sred[256 * 8];
  const int CV = (Nout + 3) >> 2;
  const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
  const int c4 = blockIdx.x * tpr + tc;
  const bool active = c4 < CV;
  const int c = c4 * 4;
  double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double w2[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // second BatchNorm layer of the fused reduction
  f32x4 bmu[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, bis[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (active && bn.n > 0) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
      if (b < bn.n) {
        bmu[b] = *reinterpret_cast<const f32x4*>(bn.mean[b] + c);
        bis[b] = *reinterpret_cast<const f32x4*>(bn.invstd[b] + c);
      }
  }
  if (active) {
    f32x4 bv = {0.f, 0.f, 0.f, 0.f}, sv = {1.f, 1.f, 1.f, 1.f};
    if (bias) {
#pragma unroll
      for (int k = 0; k < 4; ++k) bv[k] = (c + k < Nout) ? bias[c + k] : 0.f;
    }
    if (scale) {
#pragma unroll
      for (int k = 0; k < 4; ++k) sv[k] = (c + k < Nout) ? scale[c + k] : 1.f;
    }
    const size_t slab = (size_t)M * ldpart;
    constexpr int U = 2;
    const int step = gridDim.y * rpb;
    for (int mb = blockIdx.y * rpb + tr; mb < M; mb += U * step) {
      f32x4 a[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int m = mb + u * step < M ? mb + u * step : mb;
        a[u] = *reinterpret_cast<const f32x4*>(part + (size_t)m * ldpart + c);
      }
      // The slabs are summed in slab order (deterministic), but their loads are independent: four slabs' worth are
      // requested before the first is added — with one slab per trip the loop was a chain of ksplit dependent L2 / HBM
      // round trips (25 us per launch at per-GPU batch 2 for 2-4 us of traffic).
      int k = 1;
      for (; SPLITK_BATCH && k + 3 < ksplit; k += 4) {
        f32x4 t[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int m = mb + u * step < M ? mb + u * step : mb;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            t[u][q] = *reinterpret_cast<const f32x4*>(part + (size_t)(k + q) * slab + (size_t)m * ldpart + c);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) a[u] += t[u][q];
      }
      for (; k < ksplit; ++k) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int m = mb + u * step < M ? mb + u * step : mb;
          a[u] += *reinterpret_cast<const f32x4*>(part + k * slab + (size_t)m * ldpart + c);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int m = mb + u * step;
        if (m < M) {
          f32x4 r = a[u] * sv + bv;
          if (bn.n == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const double d = (double)r[k];
              v[k] += d;
              v[4 + k] += d * d;
            }
          }
          if (add) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (c + k < Nout) r[k] += add[(size_t)m * ldadd + c + k];
          }
          if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = fmaxf(r[k], 0.f);
          }
          if (bn.n > 0) {
            if (bn.bits) {
              const unsigned wq = bn.bits[(size_t)m * bn.ldb + (c >> 5)] >> (c & 31);
#pragma unroll
              for (int k = 0; k < 4; ++k) r[k] = ((wq >> k) & 1u) ? r[k] : 0.f;
            } else if (bn.mask) {
              const f32x4 a4 = *reinterpret_cast<const f32x4*>(bn.mask + (size_t)m * bn.ldm + c);
#pragma unroll
              for (int k = 0; k < 4; ++k) r[k] = a4[k] > 0.f ? r[k] : 0.f;
            }
            const f32x4 y0 = *reinterpret_cast<const f32x4*>(bn.y[0] + (size_t)m * bn.ldy[0] + c);
            const f32x4 xh0 = (y0 - bmu[0]) * bis[0];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v[k] += (double)r[k];
              v[4 + k] += (double)r[k] * (double)xh0[k];
            }
            if (bn.n > 1) {
              const f32x4 y1 = *reinterpret_cast<const f32x4*>(bn.y[1] + (size_t)m * bn.ldy[1] + c);
              const f32x4 xh1 = (y1 - bmu[1]) * bis[1];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                w2[k] += (double)r[k];
                w2[4 + k] += (double)r[k] * (double)xh1[k];
              }
            }
          }
          *reinterpret_cast<f32x4*>(y + (size_t)m * ldy + c) = r;
        }
      }
    }
  }
  if (bn.n > 0) {
    // same block reduction + slot as the statistics path, once per BatchNorm layer
#pragma unroll
    for (int b = 0; b < 2; ++b)
      if (b < bn.n) {
        double* acc8 = b == 0 ? v : w2;
        if (rpb > 1) {
          __syncthreads();
#pragma unroll
          for (int k = 0; k < 8; ++k) sred[(tr * tpr + tc) * 8 + k] = acc8[k];
          __syncthreads();
          if (tr == 0)
            for (int r = 1; r < rpb; ++r)
#pragma unroll
              for (int k = 0; k < 8; ++k) acc8[k] += sred[(r * tpr + tc) * 8 + k];
        }
        if (tr == 0 && active) {
          double* st = bn.sums[b] + (size_t)((blockIdx.x + blockIdx.y) % nslot) * 2 * Nout;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (c + k < Nout) {
              atomic_add_f64(&st[c + k], acc8[k]);
              atomic_add_f64(&st[Nout + c + k], acc8[4 + k]);
            }
        }
      }
    return;
  }
  if (stats) {
    if (rpb > 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) sred[(tr * tpr + tc) * 8 + k] = v[k];
      __syncthreads();
      if (tr == 0)
        for (int r = 1; r < rpb; ++r)
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] += sred[(r * tpr + tc) * 8 + k];
    }
    if (tr == 0 && active) {
      double* st = stats + (size_t)((blockIdx.x + blockIdx.y) % nslot) * 2 * Nout;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c + k < Nout) {
          atomic_add_f64(&st[c + k], v[k]);
          atomic_add_f64(&st[Nout + c + k], v[4 + k]);
        }
    }
  }
}

// OIHW weights -> packed forward panel [Co_pad][KT*32], k = ((ci/32)*RS + tap)*32 + ci%32
// (rows >= Co and channels >= Ci are zero).
__global__ void pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ out, int Co,
                                int Co_pad, int Ci, int Kc, int RS) {
  const size_t total = (size_t)Co_pad * Kc * RS;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c32 = (int)(idx % 32);
    size_t t = idx / 32;
    const int tap = (int)(t % RS);
    t /= RS;
    const int chunk = (int)(t % (Kc / 32));
    const int co = (int)(t / (Kc / 32));
    const int ci = chunk * 32 + c32;
    float v = 0.f;
    if (co < Co && ci < Ci) v = w[((size_t)co * Ci + ci) * RS + tap];
    out[idx] = v;
  }
}

// OIHW weights -> packed dgrad panel [Ci_pad][KT*32] with K running over output channels:
// k = ((co/32)*RS + tap)*32 + co%32.
__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int Co,
                                  int Kc, int Ci, int Ci_pad, int RS) {
  const size_t total = (size_t)Ci_pad * Kc * RS;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int c32 = (int)(idx % 32);
    size_t t = idx / 32;
    const int tap = (int)(t % RS);
    t /= RS;
    const int chunk = (int)(t % (Kc / 32));
    const int ci = (int)(t / (Kc / 32));
    const int co = chunk * 32 + c32;
    float v = 0.f;
    if (co < Co && ci < Ci) v = w[((size_t)co * Ci + ci) * RS + tap];
    out[idx] = v;
  }
}

// All conv weights of a network in ONE launch: block b finds its (conv, panel) by binary search over
// the block-start table; 1024 elements per block.
__global__ __launch_bounds__(256) void pack_multi_kernel(const SemsegPackDesc* __restrict__ descs,
                                                         const int* __restrict__ starts, int nseg) {
  // starts[2*i] = first block of conv i's forward panel, starts[2*i+1] = first block of its dgrad panel
  int lo = 0, hi = nseg - 1;
  const int b = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (starts[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const SemsegPackDesc d = descs[lo >> 1];
  const bool dgrad = lo & 1;
  const size_t base = (size_t)(b - starts[lo]) * 1024;
  const int RS = d.RS;
  const int Kc = dgrad ? d.Kc_dgrad : d.Ci;
  const size_t total = (size_t)(dgrad ? d.Ci_pad : d.Co_pad) * Kc * RS;
  float* out = dgrad ? d.w_dgrad : d.w_fwd;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const size_t idx = base + t * 256 + threadIdx.x;
    if (idx >= total) break;
    const int c32 = (int)(idx % 32);
    size_t q = idx / 32;
    const int tap = (int)(q % RS);
    q /= RS;
    const int chunk = (int)(q % (Kc / 32));
    const int row = (int)(q / (Kc / 32));
    const int k = chunk * 32 + c32;
    const int co = dgrad ? k : row, ci = dgrad ? row : k;
    float v = 0.f;
    if (co < d.Co && ci < d.Ci) v = d.w[((size_t)co * d.Ci + ci) * RS + tap];
    out[idx] = v;
  }
}

}  // namespace

extern "C" {


int semseg_conv_pack_weights(const float* w_oihw, float* w_fwd, float* w_dgrad, int Co, int Ci,
                             int R, int S, int Co_pad, int Ci_pad, hipStream_t stream) {
  if (!w_oihw || Co <= 0 || Ci <= 0 || R <= 0 || S <= 0) return SEMSEG_EINVAL;
  const int RS = R * S;
  if (w_fwd) {
    if (Ci % 32 != 0 || Co_pad < Co) return SEMSEG_EINVAL;
    const size_t total = (size_t)Co_pad * Ci * RS;
    pack_fwd_kernel<<<grid_for(total, 256), 256, 0, stream>>>(w_oihw, w_fwd, Co, Co_pad, Ci, Ci, RS);
  }
  if (w_dgrad) {
    const int Kc = (Co + 31) / 32 * 32;
    if (Ci_pad < Ci) return SEMSEG_EINVAL;
    const size_t total = (size_t)Ci_pad * Kc * RS;
    pack_dgrad_kernel<<<grid_for(total, 256), 256, 0, stream>>>(w_oihw, w_dgrad, Co, Kc, Ci, Ci_pad, RS);
  }
  return semseg_launch_status();
}

int semseg_conv_pack_weights_multi(const SemsegPackDesc* descs_dev, const int* block_starts_dev,
                                   int nconv, int total_blocks, hipStream_t stream) {
  if (!descs_dev || !block_starts_dev || nconv < 1 || total_blocks < 1) return SEMSEG_EINVAL;
  pack_multi_kernel<<<total_blocks, 256, 0, stream>>>(descs_dev, block_starts_dev, 2 * nconv);
  return semseg_launch_status();
}

// tile code of the C ABI: 128 / 64 = 128-row tiles, that many output columns; 1128 / 1064 = 64-row tiles (2 waves), for
// launches whose 128-row grid would not fill the chip (small per-GPU batch) — 4x the tiles of a 128 x 128 grid without
// splitting K, so no partial slabs and no separate epilogue pass
// 2128: the 256 x 128 bf16x3 GEMM kernel of gemm_bf16split.hip for eligible 1x1 convs (forward: semseg_conv_fwd, data gradient:
// dgrad_impl), else 128
// 3128: the same kernel with 128 x 128 tiles (four waves), for grids of a small per-GPU batch
static inline bool tile_code_ok(int t) { return t == 64 || t == 128 || t == 1064 || t == 1128 || t == 2128 || t == 3128; }


// arith (include/semseg_hip.h): SEMSEG_ARITH_BF16X3 selects the SP = 3 instances of the 1x1 / 3x3 buffer-load kernels
// (products from three-way split bf16 pieces); the generic tap walk (RS_T = 0) has no split form and stays exact fp32.
static int conv_launch(bool transposed, const ConvArgs& a, int tile_code, int arith, float* scratch,
                       size_t scratch_floats, unsigned* tile_counters, hipStream_t stream) {
  const bool sp3 = arith == SEMSEG_ARITH_BF16X3;
  const int BMr = tile_code >= 1000 ? 64 : 128;
  const int BN = tile_code % 1000;
  const int tiles_m = (a.M + BMr - 1) / BMr;
  ConvArgs p = a;
  if (p.batch > 1) { scratch = nullptr; scratch_floats = 0; }   // batched GEMM: the batch fills the chip, no split-K
  p.tiles_n = (a.Nout + BN - 1) / BN;
  const int tiles = tiles_m * p.tiles_n;
  const int KT = a.R * a.S * (a.Kc / BK);
  // Split K (a) for every tile when the grid cannot fill 256 CUs (small per-GPU batch), (b) for the
  // last partial wave of a large grid ("stream-K tail": 900 tiles on 256 CUs = 3.5 rounds, the half
  // round is cut into K-slices so it costs ~0.6 instead of 1 round).  Each (tile, ks) workgroup
  // writes a raw partial slab, splitk_epilogue_kernel reduces + applies the epilogue.
  const int P = 256;
  int ksplit = 1, full_tiles = tiles, tail_mt = 0;
  p.ldpart = p.tiles_n * BN;
  const bool can_split = scratch && (a.ldy & 3) == 0 && a.ldy >= ((a.Nout + 3) & ~3) && KT >= 8;
  if (can_split && tiles < 384) {
    const int target = 448;   // workgroups aimed at; swept 320 ... 768 at batch 2 / 4
    ksplit = (target + tiles - 1) / tiles;
    if (ksplit > KT / 4) ksplit = KT / 4;
    if (ksplit > 16) ksplit = 16;
    full_tiles = 0;
    tail_mt = tiles_m;
  } else if (can_split) {
    const int tcap = 16;   // cap of the tail's K split
    const int rem = tiles % P;
    if (tcap > 1 && rem != 0 && rem <= 208) {
      tail_mt = (rem + p.tiles_n - 1) / p.tiles_n;
      if (tail_mt > tiles_m) tail_mt = tiles_m;
      full_tiles = tiles - tail_mt * p.tiles_n;
      ksplit = tcap;
      while (ksplit > 1 && ksplit > KT / 4) ksplit >>= 1;
      if (ksplit == 1) { full_tiles = tiles; tail_mt = 0; }
    }
  }
  p.tail_m0 = (tiles_m - tail_mt) * BMr;
  if (ksplit > 1) {
    const size_t slab = (size_t)(a.M - p.tail_m0) * p.ldpart;
    while (ksplit > 1 && slab * ksplit > scratch_floats) --ksplit;
  }
  if (ksplit <= 1) { ksplit = 1; full_tiles = tiles; tail_mt = 0; p.tail_m0 = 0; }
  p.kt_per = (KT + ksplit - 1) / ksplit;
  {
    const int RSr = a.R * a.S;  // split on channel-chunk boundaries (whole tap groups)
    p.kt_per = (p.kt_per + RSr - 1) / RSr * RSr;
  }
  ksplit = (KT + p.kt_per - 1) / p.kt_per;
  p.ksplit = ksplit;
  p.full_tiles = ksplit > 1 ? full_tiles : tiles;
  p.part = scratch;
  // In-kernel reduction of the split tiles (ConvArgs::cnt) when the caller gave tile counters: the last slice of a tile to finish
  // sums the others' register-layout slabs — a serial read of ksplit x 32-64 KB by one workgroup, against a separate launch that
  // re-reads every partial sum from a [ksplit][M][N] array: taken up to 16 slices per tile (SEMSEG_DEBUG fused_split_max).
  char dbg_fm[16];
  const char* fm_s = semseg_debug("fused_split_max", dbg_fm, sizeof(dbg_fm));
  const int fused_max = fm_s ? atoi(fm_s) : 16;
  p.cnt = nullptr;
  if (ksplit > 1 && tile_counters && ksplit <= fused_max && (tiles - p.full_tiles) <= SEMSEG_TILE_COUNTERS &&
      (size_t)(tiles - p.full_tiles) * ksplit * BMr * BN <= scratch_floats &&
      (size_t)(tiles - p.full_tiles) * ksplit * BMr * BN * 4 < 0x7FFF0000ull)
    p.cnt = tile_counters;
  p.div_hw = make_fastdiv(a.Hout * a.Wout);
  p.div_w = make_fastdiv(a.Wout);
  p.div_tn = make_fastdiv(p.tiles_n);
  p.div_ks = make_fastdiv(ksplit);
  p.gm = 8;   // tile rows per group of the workgroup order (decode_tile)
  const int grid = p.full_tiles + (tiles - p.full_tiles) * ksplit;
  // buffer-load kernels need 1x1 / 3x3 taps, split points on chunk boundaries and < 2 GB operands
  const int RSv = a.R * a.S;
  const size_t x_bytes = (size_t)a.N * a.Hin * a.Win * a.ldx * 4, w_bytes = (size_t)p.tiles_n * BN * KT * BK * 4;
  const bool bl = (RSv == 1 || RSv == 9) && (p.kt_per % RSv == 0) &&
                  x_bytes < 0x7FFF0000ull && w_bytes < 0x7FFF0000ull;
  // Two-level accumulation whenever the reduction is longer than one flush interval (K > 576).  Round 1 used it for
  // K >= 4096 only; the in-situ backward test of round 2 showed single chains of K = 1152 ... 2304 (aux.0 /
  // layer0.6 data gradients) at 2.7-2.9x the rounding noise (rms) of the CPU's blocked sums, 6x in the maximum.
  // what counts is the chain one workgroup accumulates: the whole reduction for unsplit tiles, one K slice when
  // every tile is split (small per-GPU batch: slices are <= 18 K-steps there and the leaner kernel is 5 % faster
  // over the whole bs-2 step)
  const int chain = p.full_tiles > 0 ? KT : p.kt_per;
  const bool tl = chain > 18;
#define LAUNCH_CONV_(BM_, BN_, TR_, RS_, TL_) \
  conv_igemm_kernel<BM_, BN_, TR_, RS_, TL_><<<dim3(grid, p.batch), BM_ * 2, 0, stream>>>(p)
#define LAUNCH_CONV(BM_, BN_, TR_, RS_)                                        \
  do {                                                                         \
    if (tl) LAUNCH_CONV_(BM_, BN_, TR_, RS_, true);                            \
    else LAUNCH_CONV_(BM_, BN_, TR_, RS_, false);                              \
  } while (0)
#define LAUNCH_RS(BM_, BN_, TR_)                                   \
  do {                                                             \
    if (bl && RSv == 9 && sp3) {                     \
      if (tl) conv_igemm_kernel<BM_, BN_, TR_, 9, true, 3><<<dim3(grid, p.batch), BM_ * 2, 0, stream>>>(p);   \
      else conv_igemm_kernel<BM_, BN_, TR_, 9, false, 3><<<dim3(grid, p.batch), BM_ * 2, 0, stream>>>(p);     \
    } else if (bl && RSv == 9) LAUNCH_CONV(BM_, BN_, TR_, 9);      \
    else if (bl && sp3) {                            \
      if (tl) conv_igemm_kernel<BM_, BN_, TR_, 1, true, 3><<<dim3(grid, p.batch), BM_ * 2, 0, stream>>>(p);   \
      else conv_igemm_kernel<BM_, BN_, TR_, 1, false, 3><<<dim3(grid, p.batch), BM_ * 2, 0, stream>>>(p);     \
    } else if (bl) LAUNCH_CONV(BM_, BN_, TR_, 1);                  \
    else LAUNCH_CONV(BM_, BN_, TR_, 0);                            \
  } while (0)
  if (BMr == 128 && BN == 128) {
    if (transposed) LAUNCH_RS(128, 128, true); else LAUNCH_RS(128, 128, false);
  } else if (BMr == 128) {
    if (transposed) LAUNCH_RS(128, 64, true); else LAUNCH_RS(128, 64, false);
  } else if (BN == 128) {
    if (transposed) LAUNCH_RS(64, 128, true); else LAUNCH_RS(64, 128, false);
  } else {
    if (transposed) LAUNCH_RS(64, 64, true); else LAUNCH_RS(64, 64, false);
  }
#undef LAUNCH_RS
#undef LAUNCH_CONV
#undef LAUNCH_CONV_
  if (ksplit > 1 && !p.cnt) {
    const int Mt = a.M - p.tail_m0;  // rows covered by split tiles
    const int CV = (a.Nout + 3) / 4;
    int tpr = 1;
    while (tpr * 2 <= CV && tpr * 2 <= 256) tpr *= 2;
    const int rpb = 256 / tpr, gx = (CV + tpr - 1) / tpr;
    int gy = (Mt + 2 * rpb - 1) / (2 * rpb);
    int cap = 512 / gx;  // more blocks only add fp64 statistic atomics (1024 / 2048: forward +11 / +16 % at bs 2)
    if (cap < 1) cap = 1;
    if (gy > cap) gy = cap;
    BnrArgs bn;
    bn.n = a.bnr_n;
    bn.mask = (a.bnr_n && a.bnr_mask) ? a.bnr_mask + (size_t)p.tail_m0 * a.bnr_ldm : nullptr;
    bn.ldm = a.bnr_ldm;
    bn.bits = (a.bnr_n && a.bnr_bits) ? a.bnr_bits + (size_t)p.tail_m0 * a.bnr_ldb : nullptr;
    bn.ldb = a.bnr_ldb;
    for (int b = 0; b < 2; ++b) {
      const bool on = b < a.bnr_n;
      bn.y[b] = on ? a.bnr_y[b] + (size_t)p.tail_m0 * a.bnr_ldy[b] : nullptr;
      bn.ldy[b] = on ? a.bnr_ldy[b] : 0;
      bn.mean[b] = on ? a.bnr_mean[b] : nullptr;
      bn.invstd[b] = on ? a.bnr_invstd[b] : nullptr;
      bn.sums[b] = on ? a.bnr_sums[b] : nullptr;
    }
    splitk_epilogue_kernel<<<dim3(gx, gy), 256, 0, stream>>>(
        scratch, ksplit, p.ldpart, a.y + (size_t)p.tail_m0 * a.ldy, a.ldy, a.bias, a.scale, a.relu,
        a.add ? a.add + (size_t)p.tail_m0 * a.ldadd : nullptr, a.ldadd, a.stats, a.stats_nslot, Mt, a.Nout,
        tpr, rpb, bn);
  }
  return semseg_launch_status();
}

int semseg_conv_fwd(const float* x, int ldx, const float* w_fwd, float* y, int ldy, int N, int H,
                    int W, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad,
                    int dil, const float* bias, const float* scale, int relu, const float* add,
                    int ldadd, double* stats, int stats_nslot, int tile_n, int arith, float* scratch,
                    size_t scratch_floats, unsigned int* tile_counters, hipStream_t stream) {
  if (!x || !w_fwd || !y || Ci % 32 != 0 || (ldx & 3) || !tile_code_ok(tile_n) || !arith_ok(arith))
    return SEMSEG_EINVAL;
  ConvArgs a;
  a.x = x; a.w = w_fwd; a.y = y; a.bias = bias; a.scale = scale; a.relu = relu; a.add = add; a.stats = stats;
  a.ldx = ldx; a.ldy = ldy; a.ldadd = ldadd;
  a.N = N; a.Hin = H; a.Win = W; a.Hout = Ho; a.Wout = Wo;
  a.Kc = Ci; a.Nout = Co; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil;
  a.M = N * Ho * Wo; a.tiles_n = 0; a.stats_nslot = stats_nslot > 0 ? stats_nslot : 1;
  a.batch = 1; a.x_bs = a.w_bs = a.y_bs = a.add_bs = 0; a.bnr_n = 0; a.bnr_mask = nullptr; a.bnr_bits = nullptr;
  if (tile_n == 2128 || tile_n == 3128) {
    // the 256 x 128 (3128: 128 x 128) bf16x3 GEMM kernel (gemm_bf16split.hip) for what is a plain row GEMM with statistics: 1x1, stride 1, no
    // padding, nothing folded into the epilogue, whole 128-column panels; anything else runs the 128 x 128 implicit-GEMM tile
    const bool plain = R == 1 && S == 1 && stride == 1 && pad == 0 && Ho == H && Wo == W && !bias && !scale && !relu && !add;
    if (arith == SEMSEG_ARITH_BF16X3 && plain && Co % 128 == 0 && (ldy & 3) == 0 && ldy >= Co && (((size_t)y | (size_t)x) & 15) == 0)
      return semseg_split_gemm_conv1x1_fwd(x, ldx, w_fwd, y, ldy, a.M, Ci, Co, stats, a.stats_nslot, tile_n == 3128 ? 128 : 256, stream);
    tile_n = 128;
  }
  return conv_launch(false, a, tile_n, arith, scratch, scratch_floats, tile_counters, stream);
}

static int dgrad_impl(const float* dy, int lddy, const float* w_dgrad, float* dx, int lddx, int N,
                      int H, int W, int Ci, int Ho, int Wo, int Co, int R, int S, int stride,
                      int pad, int dil, const float* add, int ldadd, int tile_n, int arith, const ConvArgs* bnr,
                      float* scratch, size_t scratch_floats, unsigned* tile_counters, hipStream_t stream) {
  if (!dy || !w_dgrad || !dx || (lddy & 3) || !tile_code_ok(tile_n) || !arith_ok(arith)) return SEMSEG_EINVAL;
  const int Kc = (Co + 31) / 32 * 32;
  if (lddy < Kc) return SEMSEG_EINVAL;
  if (tile_n == 2128 || tile_n == 3128) {
    // the 256 x 128 (3128: 128 x 128) bf16x3 GEMM kernel (gemm_bf16split.hip): 1x1, stride 1, no padding, whole 128-column panels, at most one
    // fused BatchNorm layer, and a reduction of at most 1024: that kernel accumulates ONE fp32 chain (no registers for the
    // second accumulator set this file flushes chains longer than 576 into).  Measured in situ (PSPNet-101 473^2, every
    // eligible layer forced onto it, K up to 2048): rms error 1.7 x the CPU-fp32 recompute's at worst (K = 2048), 2.6 x in the
    // maximum, against 1.2 x / 1.6 x with flushing — inside the 3 x / 5 x criterion, and K <= 1024 keeps it to the layers
    // where the kernel pays (layer3's conv3: 215 -> 178 us).  Anything else runs the 128 x 128 implicit-GEMM tile.
    const bool plain = R == 1 && S == 1 && stride == 1 && pad == 0 && Ho == H && Wo == W;
    const bool al = (lddx & 3) == 0 && lddx >= Ci && (((size_t)dx | (size_t)dy) & 15) == 0 && (!add || ((ldadd & 3) == 0 && ((size_t)add & 15) == 0));
    if (arith == SEMSEG_ARITH_BF16X3 && plain && al && Ci % 128 == 0 && Kc <= 1024 && (!bnr || bnr->bnr_n == 1)) {
      const bool on = bnr != nullptr;
      return semseg_split_gemm_conv1x1_dgrad(dy, lddy, w_dgrad, dx, lddx, N * H * W, Kc, Ci, add, ldadd,
                                             on ? bnr->bnr_mask : nullptr, on ? bnr->bnr_ldm : 0,
                                             on ? bnr->bnr_bits : nullptr, on ? bnr->bnr_ldb : 0,
                                             on ? bnr->bnr_y[0] : nullptr, on ? bnr->bnr_ldy[0] : 0,
                                             on ? bnr->bnr_mean[0] : nullptr, on ? bnr->bnr_invstd[0] : nullptr,
                                             on ? bnr->bnr_sums[0] : nullptr, on ? bnr->stats_nslot : 1,
                                             tile_n == 3128 ? 128 : 256, stream);
    }
    tile_n = 128;
  }
  ConvArgs a;
  a.x = dy; a.w = w_dgrad; a.y = dx; a.bias = nullptr; a.scale = nullptr; a.relu = 0; a.add = add; a.stats = nullptr;
  a.ldx = lddy; a.ldy = lddx; a.ldadd = ldadd;
  a.N = N; a.Hin = Ho; a.Win = Wo; a.Hout = H; a.Wout = W;
  a.Kc = Kc; a.Nout = Ci; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil;
  a.M = N * H * W; a.tiles_n = 0; a.stats_nslot = 1;
  a.batch = 1; a.x_bs = a.w_bs = a.y_bs = a.add_bs = 0; a.bnr_n = 0; a.bnr_mask = nullptr; a.bnr_bits = nullptr;
  if (bnr) {
    a.bnr_n = bnr->bnr_n; a.bnr_mask = bnr->bnr_mask; a.bnr_ldm = bnr->bnr_ldm; a.bnr_bits = bnr->bnr_bits; a.bnr_ldb = bnr->bnr_ldb; a.stats_nslot = bnr->stats_nslot;
    for (int b = 0; b < 2; ++b) {
      a.bnr_y[b] = bnr->bnr_y[b]; a.bnr_ldy[b] = bnr->bnr_ldy[b]; a.bnr_mean[b] = bnr->bnr_mean[b];
      a.bnr_invstd[b] = bnr->bnr_invstd[b]; a.bnr_sums[b] = bnr->bnr_sums[b];
    }
  }
  return conv_launch(true, a, tile_n, arith, scratch, scratch_floats, tile_counters, stream);
}

int semseg_conv_dgrad(const float* dy, int lddy, const float* w_dgrad, float* dx, int lddx, int N,
                      int H, int W, int Ci, int Ho, int Wo, int Co, int R, int S, int stride,
                      int pad, int dil, const float* add, int ldadd, int tile_n, int arith, float* scratch,
                      size_t scratch_floats, unsigned int* tile_counters, hipStream_t stream) {
  return dgrad_impl(dy, lddy, w_dgrad, dx, lddx, N, H, W, Ci, Ho, Wo, Co, R, S, stride, pad, dil, add, ldadd, tile_n,
                    arith, nullptr, scratch, scratch_floats, tile_counters, stream);
}

// Data gradient + the BatchNorm-backward reduction of the layer(s) that PRODUCED this conv's input, in one kernel:
// dx receives g = (dgrad (+ add)) * (act > 0) and sums{0,1}[slot][2*Ci] += {sum g, sum g * (y - mean) * invstd} (fp64).
// Replaces the separate pass of semseg_bn_bwd_reduce (model/resnet.py:76-92 backward); only valid when this data
// gradient is the LAST contribution to that activation's gradient.  act may be null (no ReLU).
int semseg_conv_dgrad_bnreduce(const float* dy, int lddy, const float* w_dgrad, float* dx, int lddx, int N, int H,
                               int W, int Ci, int Ho, int Wo, int Co, int R, int S, int stride, int pad, int dil,
                               const float* add, int ldadd, int tile_n, int bn_count, const float* act, int ldact,
                               const unsigned* relu_bits, int ldbits,
                               const float* y0, int ldy0, const float* mean0, const float* invstd0, double* sums0,
                               const float* y1, int ldy1, const float* mean1, const float* invstd1, double* sums1,
                               int nslot, int arith, float* scratch, size_t scratch_floats, unsigned int* tile_counters,
                               hipStream_t stream) {
  if (bn_count < 1 || bn_count > 2 || !y0 || !mean0 || !invstd0 || !sums0 || nslot < 1) return SEMSEG_EINVAL;
  if (bn_count == 2 && (!y1 || !mean1 || !invstd1 || !sums1)) return SEMSEG_EINVAL;
  // the fused path lives in the 16-byte store phase of the epilogue: everything 4-float aligned, channels % 4 == 0
  if ((Ci & 3) || (lddx & 3) || lddx < Ci || (ldy0 & 3) || (act && (ldact & 3)) || (add && (ldadd & 3)) ||
      (bn_count == 2 && (ldy1 & 3)) || (relu_bits && ((Ci & 31) || ldbits * 32 < Ci)))
    return SEMSEG_EINVAL;
  ConvArgs b;
  b.bnr_n = bn_count; b.bnr_mask = relu_bits ? nullptr : act; b.bnr_ldm = ldact; b.stats_nslot = nslot;
  b.bnr_bits = relu_bits; b.bnr_ldb = ldbits;
  b.bnr_y[0] = y0; b.bnr_ldy[0] = ldy0; b.bnr_mean[0] = mean0; b.bnr_invstd[0] = invstd0; b.bnr_sums[0] = sums0;
  b.bnr_y[1] = y1; b.bnr_ldy[1] = ldy1; b.bnr_mean[1] = mean1; b.bnr_invstd[1] = invstd1; b.bnr_sums[1] = sums1;
  return dgrad_impl(dy, lddy, w_dgrad, dx, lddx, N, H, W, Ci, Ho, Wo, Co, R, S, stride, pad, dil, add, ldadd, tile_n,
                    arith, &b, scratch, scratch_floats, tile_counters, stream);
}

// Batched GEMMs on the two matrix-core kernels — the PSA point-affinity contraction (torch.bmm at
// model/psanet.py:90-91) and its two gradients as ONE launch each instead of a per-image loop.
//   rows:   C[b][M][Nout] (+= add) = A[b][M][K] * Bt[b][Nout_pad][K]^T      (K % 32 == 0, Bt rows zero padded)
//   kmajor: out[b][Co][Ci] (=|+=) sum_k y[b][k][co] * x[b][k][ci]           (Ci % 64 == 0)
int semseg_gemm_rows_batched(const float* a, int lda, long long a_bs, const float* bt, long long bt_bs, float* c,
                             int ldc, long long c_bs, int M, int K, int Nout, int batch, int arith,
                             hipStream_t stream) {
  if (!a || !bt || !c || K % 32 != 0 || (lda & 3) || batch < 1 || batch > 65535 || !arith_ok(arith)) return SEMSEG_EINVAL;
  ConvArgs g;
  g.x = a; g.w = bt; g.y = c; g.bias = nullptr; g.scale = nullptr; g.relu = 0; g.add = nullptr; g.stats = nullptr;
  g.ldx = lda; g.ldy = ldc; g.ldadd = 0;
  g.N = 1; g.Hin = M; g.Win = 1; g.Hout = M; g.Wout = 1;
  g.Kc = K; g.Nout = Nout; g.R = 1; g.S = 1; g.stride = 1; g.pad = 0; g.dil = 1;
  g.M = M; g.tiles_n = 0; g.stats_nslot = 1;
  g.batch = batch; g.x_bs = a_bs; g.w_bs = bt_bs; g.y_bs = c_bs; g.add_bs = 0; g.bnr_n = 0; g.bnr_mask = nullptr; g.bnr_bits = nullptr;
  return conv_launch(false, g, Nout >= 128 ? 128 : 64, arith, nullptr, 0, nullptr, stream);
}

}  // extern "C"
