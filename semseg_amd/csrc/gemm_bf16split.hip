// SEMSEG_ARITH_BF16X3 (include/semseg_hip.h) on a kernel of its own: what the engine runs for
// the batched row GEMM of the Winograd path,  C[b][M][Nout] = A[b][M][K] * Bt[b][Nout_pad][K]^T  (same operands,
// same layouts and strides as semseg_gemm_rows_batched), with each fp32 operand split on the fly into bf16 pieces
// and the product rebuilt from bf16 matrix-core instructions (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate)
// with fp32 accumulation:
//   nsplit 3 ("x6"):  x = h + m + l,  a*b ~ ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm  (6 MFMAs, ~2^-23) — the arithmetic
//   nsplit 2 ("x3"):  x = h + l,      a*b ~ ah*bh + ah*bl + al*bh     (3 MFMAs, ~2^-16 per product) — a MEASUREMENT only, it
//                     fails the per-op parity criteria and nothing in the engine selects it
// The operands stay fp32 in HBM; the split happens in registers between the global load and the LDS store, so the
// kernel is a drop-in for the fp32 one.  What it costs in accuracy is measured by scripts/split_bf16_probe.py and the
// in-situ test; DESIGN.md section 8.4 has the numbers.
//
// Tile: 256 rows x 128 columns per workgroup of 8 waves (4 x 2, 64 x 64 each = 2 x 2 MFMA blocks), K staged BK at a
// time: global -> registers (prefetched one stage ahead) -> split -> LDS [rows][BK] bf16 per piece (chunk-swizzled) ->
// fragments.  Measured and set aside in round 3 (git history, commit 9e79bfe: csrc/experiments/gemm_bf16split_256sq.inc): a
// 256 x 256 tile with a deeper pipeline ran the six-product form in the same time.
#include <cstring>

#include "common.h"
#include "gemm_bf16split.h"
#include "../../include/semseg_hip.h"

namespace {


constexpr int SB_BN = 128;     // output columns per workgroup; rows: template parameter BM_ (256: 8 waves, 128: 4 waves)
#ifndef SB_NBUF
#define SB_NBUF 2      // 1: the single-buffered K loop of rounds 3-4 (two barriers per K step; A/B builds only)
#endif
#ifndef EPI_WIDE
#define EPI_WIDE 1     // 0: direct dword stores from the accumulator layout (the round-3 epilogue; A/B builds only)
#endif

struct SplitArgs {
  const float* a;
  const float* bt;
  float* c;
  long long a_bs, bt_bs, c_bs;
  int lda, ldc, M, K, Nout, tiles_m, tiles_n, total;
  double* stats;        // EPI 1: [nslot][2 * Nout] fp64 {sum, sum of squares} per column of C (semseg_conv_fwd's statistics)
  int nslot;
  // EPI 2 (data gradient of a 1x1 conv): C = product (+ add); with bnr_n = 1 the fused BatchNorm-backward reduction of
  // semseg_conv_dgrad_bnreduce: C = g = (product (+ add)) * mask, sums[slot][2 * Nout] += {sum g, sum g * (ybn - mean) * invstd}
  const float* add; int ldadd;
  int bnr_n;
  const float* mask; int ldm;          // post-ReLU activation (null: no ReLU or bits given)
  const unsigned* bits; int ldb;       // the same mask as bits (semseg_bn_apply's relu_bits)
  const float* ybn; int ldybn;
  const float* mean; const float* invstd;
  double* sums;
};

// Rows rr, rr + 8, rr + 16, rr + 24 of one 32 x 32 block (already transposed into the wave's slab) through the fused
// reduction, two rows' operands in flight at a time (the kernel has no registers to spare: 128 VGPRs = two workgroups per CU).
// MASK: 0 none, 1 post-ReLU activation, 2 bits.  bs[0..3] += g, bs[4..7] += g * xhat (fp32 over the lane's 8 rows of a column
// block, fp64 above that — the accumulation scheme of conv_igemm.hip's bnr_rows).
template <bool HAS_ADD, int MASK>
__device__ __forceinline__ void split_bnr_block(const SplitArgs& p, const float* slab, int slab_ld, float* C, int rowbase,
                                                int colb, bool colok, int lane, const f32x4 bmu, const f32x4 bis,
                                                float (&bs)[8]) {
  const int rr = lane >> 3, c4 = (lane & 7) * 4;
  const int cc = colok ? colb : 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x4 y0[2], dd[HAS_ADD ? 2 : 1], aa[MASK == 1 ? 2 : 1];
    unsigned wb[MASK == 2 ? 2 : 1];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = rowbase + rr + 8 * (2 * h + q);
      const size_t mm = (size_t)(row < p.M ? row : p.M - 1);
      y0[q] = *reinterpret_cast<const f32x4*>(p.ybn + mm * p.ldybn + cc);
      if constexpr (HAS_ADD) dd[q] = *reinterpret_cast<const f32x4*>(p.add + mm * p.ldadd + cc);
      if constexpr (MASK == 1) aa[q] = *reinterpret_cast<const f32x4*>(p.mask + mm * p.ldm + cc);
      if constexpr (MASK == 2) wb[q] = p.bits[mm * p.ldb + (cc >> 5)];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = rr + 8 * (2 * h + q);
      const int row = rowbase + r;
      const bool ok = row < p.M && colok;
      f32x4 v = *reinterpret_cast<const f32x4*>(&slab[r * slab_ld + c4]);
      if constexpr (HAS_ADD) v += dd[q];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bool keep = true;
        if constexpr (MASK == 1) keep = aa[q][k] > 0.f;
        if constexpr (MASK == 2) keep = (wb[q] >> ((cc & 31) + k)) & 1u;
        v[k] = (keep && ok) ? v[k] : 0.f;
      }
      const f32x4 xh = (y0[q] - bmu) * bis;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bs[k] += v[k];
        bs[4 + k] = fmaf(v[k], xh[k], bs[4 + k]);
      }
      if (ok) *reinterpret_cast<f32x4*>(&C[(size_t)row * p.ldc + colb]) = v;
    }
  }
}

// x -> NS bf16 pieces (round to nearest even at every level; the remainders are exact in fp32)
template <int NS>
__device__ __forceinline__ void split4(const f32x4 v, bf16x4 (&out)[NS]) {
  f32x4 r = v;
#pragma unroll
  for (int c = 0; c < NS; ++c) {
    out[c] = __builtin_convertvector(r, bf16x4);
    if (c + 1 < NS) r -= bf16x4_to_f32(out[c]);
  }
}

// EPI 0: C is stored (the batched row GEMM of the Winograd path).  EPI 1: a 1x1 stride-1 convolution's forward — the same GEMM on
// the NHWC activation and the packed forward panel — with the per-channel fp64 statistics of the following BatchNorm taken
// from the accumulators, as conv_igemm_kernel's epilogue takes them.
// BM_ = 128 (round 5): the same kernel with four waves (2 x 2) for grids whose 256-row tiles would leave most of the chip idle —
// the 1x1 convs of a small per-GPU batch (57 tiles of 256 x 128 for a [7200 x 256] output on 256 CUs); three workgroups per CU.
template <int NS, int BK, int EPI = 0, int BM_ = 256>
__global__ __launch_bounds__(BM_ * 2, BM_ == 256 ? 4 : 3) void gemm_rows_bf16split_kernel(const SplitArgs p) {
  constexpr int SB_BM = BM_, SB_THREADS = BM_ * 2, WM = BM_ / 64;      // WM row-waves x 2 column-waves of 64 x 64
  // LDS rows are BK bf16 wide, unpadded; the 16-byte chunks of a row are XOR-swizzled with the index of the 256-byte
  // group the row sits in, which makes both the 16-byte fragment reads (16 rows per LDS cycle, 64 banks) and the 8-byte
  // staging stores (128 contiguous bytes per 16 lanes, 32 banks) conflict-free.  (The first version padded rows by 16
  // bytes: reads were clean but every store was a 2-way conflict, 30 % of the LDS cycles by SQ_LDS_BANK_CONFLICT.)
  constexpr int RP = 128 / BK, CPR = BK / 8;        // rows per 256 bytes, 16-byte chunks per row
  auto off = [](int row, int chunk) { return row * BK + ((chunk ^ ((row / RP) & (CPR - 1))) << 3); };
  constexpr int K4 = BK / 4;                        // float4 per tile row
  constexpr int RPP = SB_THREADS / K4;              // rows covered by one pass of the workgroup
  constexpr int A_PER = SB_BM / RPP, B_PER = SB_BN / RPP;
  static_assert(A_PER >= 1 && B_PER >= 1, "stage shape");
  // one allocation per stage: [NS][SB_BM * BK] A pieces, then [NS][SB_BN * BK] B pieces; the epilogue reuses it as per-wave slabs.
  // NBUF = 2 (three pieces, BK 16: 2 x 36 KB, still two workgroups per CU in 160 KB): the pieces of stage kt + 1 are written into
  // the other buffer while stage kt is being multiplied, ONE barrier per K step (round 5; the single-buffered loop had two, with
  // the split + store phase of every wave of a workgroup sitting between them while no wave of it could multiply).
  constexpr int NBUF = (NS == 3 && BK == 16) ? SB_NBUF : 1;
  constexpr int SLAB_LD = 36;                                           // floats per slab row (conflict-free ds_read_b128)
  constexpr int STAGE_BYTES = NS * (SB_BM + SB_BN) * BK * 2;
  constexpr int SLAB_BYTES = (SB_THREADS / 64) * 32 * SLAB_LD * 4;
  constexpr int RED2_BYTES = EPI == 2 ? WM * SB_BN * 2 * 8 : 0;         // fp64 column sums of the fused reduction: [WM (wm)][SB_BN][2]
  constexpr int EPI_BYTES = SLAB_BYTES + RED2_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[NBUF * STAGE_BYTES > EPI_BYTES ? NBUF * STAGE_BYTES : EPI_BYTES];
  auto sA = [&](int buf, int c) { return reinterpret_cast<__bf16*>(smem_raw + buf * STAGE_BYTES) + c * (SB_BM * BK); };
  auto sB = [&](int buf, int c) { return reinterpret_cast<__bf16*>(smem_raw + buf * STAGE_BYTES + NS * SB_BM * BK * 2) + c * (SB_BN * BK); };

  int t = xcd_remap(blockIdx.x, p.total);
  const int tn = t % p.tiles_n; t /= p.tiles_n;     // column tiles of one row panel are XCD neighbours
  const int tm = t % p.tiles_m;
  const int bi = t / p.tiles_m;
  const int m0 = tm * SB_BM, n0 = tn * SB_BN;
  const float* A = p.a + (size_t)bi * p.a_bs;
  const float* Bt = p.bt + (size_t)bi * p.bt_bs;
  float* C = p.c + (size_t)bi * p.c_bs;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int kq = tid % K4, r0 = tid / K4;

  const float* ga[A_PER];
  const float* gb[B_PER];
#pragma unroll
  for (int j = 0; j < A_PER; ++j) ga[j] = A + (size_t)min(m0 + r0 + j * RPP, p.M - 1) * p.lda + kq * 4;
#pragma unroll
  for (int j = 0; j < B_PER; ++j) gb[j] = Bt + (size_t)(n0 + r0 + j * RPP) * p.K + kq * 4;   // panel rows are padded

  f32x4 ra[A_PER], rb[B_PER];
  auto stage_load = [&](int kt) {
#pragma unroll
    for (int j = 0; j < A_PER; ++j) ra[j] = *(const f32x4*)(ga[j] + (size_t)kt * BK);
#pragma unroll
    for (int j = 0; j < B_PER; ++j) rb[j] = *(const f32x4*)(gb[j] + (size_t)kt * BK);
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_PER; ++j) {
      bf16x4 pc[NS];
      split4<NS>(ra[j], pc);
#pragma unroll
      for (int c = 0; c < NS; ++c) *(bf16x4*)&sA(buf, c)[off(r0 + j * RPP, kq >> 1) + (kq & 1) * 4] = pc[c];
    }
#pragma unroll
    for (int j = 0; j < B_PER; ++j) {
      bf16x4 pc[NS];
      split4<NS>(rb[j], pc);
#pragma unroll
      for (int c = 0; c < NS; ++c) *(bf16x4*)&sB(buf, c)[off(r0 + j * RPP, kq >> 1) + (kq & 1) * 4] = pc[c];
    }
  };
  stage_load(0);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int frow = lane & 31, fk8 = lane >> 5;
  auto multiply = [&](int buf) {
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fb[2][NS];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < NS; ++c)
          fb[j][c] = *(const bf16x8*)&sB(buf, c)[off(wn * 64 + j * 32 + frow, ks * 2 + fk8)];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bf16x8 fa[NS];
#pragma unroll
        for (int c = 0; c < NS; ++c) fa[c] = *(const bf16x8*)&sA(buf, c)[off(wm * 64 + i * 32 + frow, ks * 2 + fk8)];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // small terms first, the leading product last
          if (NS == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[j][0], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j][0], acc[i][j], 0, 0, 0);
        }
      }
    }
  };
  const int KT = p.K / BK;
  if constexpr (NBUF == 1) {
    for (int kt = 0; kt < KT; ++kt) {
      stage_store(0);
      __syncthreads();
      if (kt + 1 < KT) stage_load(kt + 1);
      multiply(0);
      __syncthreads();
    }
  } else {
    stage_store(0);
    __syncthreads();
    if (KT > 1) stage_load(1);
    // straight-line body (the last step is peeled, the look-ahead load is clamped instead of branched around) so that the
    // compiler can interleave the split + store of stage kt + 1 with the matrix-core instructions of stage kt
    for (int kt = 0; kt + 1 < KT; ++kt) {
      const int cur = kt & 1;
      stage_store(cur ^ 1);      // buffer cur ^ 1 was last read in step kt - 1, which every wave left through the barrier below
      stage_load(min(kt + 2, KT - 1));
      multiply(cur);
      __syncthreads();
    }
    multiply((KT - 1) & 1);
    __syncthreads();
  }

  // C/D map of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  Stored directly
  // that is 16 dword stores per lane and block (64 per lane and tile): store-issue bound, as long as the whole K loop of a
  // K = 256 tile (the 46 layer3 GEMMs of a PSPNet-101 step).  Each wave therefore transposes block by block through a
  // private 32 x 36-float slab of the (now idle) staging LDS and stores 16-byte lanes: 4 stores per lane and block.
  // Same-wave LDS traffic is ordered, so no barrier is needed between a block's writes, its reads and the next block.
  if constexpr (EPI == 1) {
    if (p.stats) {
      // lane = one column of block j, 32 of the wave's 64 rows in its registers: fp64 sums over them, the two lane halves
      // combined by a shuffle, the four row-waves through LDS (the staging area is idle: the K loop ended on a barrier),
      // then one atomic pair per column and workgroup into the slot replica of this row tile
      double* red = reinterpret_cast<double*>(smem_raw);          // [WM (wm)][SB_BN][2]
      const int l31 = lane & 31, lhi = lane >> 5;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
            const double d = row < p.M ? (double)acc[i][j][e] : 0.0;
            s1 += d;
            s2 += d * d;
          }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (lhi == 0) {
          red[(wm * SB_BN + wn * 64 + j * 32 + l31) * 2 + 0] = s1;
          red[(wm * SB_BN + wn * 64 + j * 32 + l31) * 2 + 1] = s2;
        }
      }
      __syncthreads();
      if (tid < 2 * SB_BN) {
        const int col = tid >> 1, which = tid & 1;
        if (n0 + col < p.Nout) {
          double v = 0.0;
#pragma unroll
          for (int r = 0; r < WM; ++r) v += red[(r * SB_BN + col) * 2 + which];
          atomic_add_f64(&p.stats[(size_t)(tm % p.nslot) * 2 * p.Nout + (size_t)which * p.Nout + n0 + col], v);
        }
      }
      __syncthreads();       // the slabs below reuse the same LDS
    }
  }
  if constexpr (EPI == 2) {
    // the launcher guarantees the 16-byte path: ldc % 4 == 0, Nout % 128 == 0, aligned bases
    float* slab = reinterpret_cast<float*>(smem_raw) + wave * (32 * SLAB_LD);
    double* red2 = reinterpret_cast<double*>(smem_raw + SLAB_BYTES);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
    const bool bnr = p.bnr_n > 0;
    const int mode = p.bits ? 2 : (p.mask ? 1 : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int colb = n0 + wn * 64 + j * 32 + c4;
      const bool colok = colb < p.Nout;
      f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, bis = {0.f, 0.f, 0.f, 0.f};
      if (bnr && colok) {
        bmu = *reinterpret_cast<const f32x4*>(p.mean + colb);
        bis = *reinterpret_cast<const f32x4*>(p.invstd + colb);
      }
      float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * lhi) * SLAB_LD + l31] = acc[i][j][e];
        const int rowbase = m0 + wm * 64 + i * 32;
        if (bnr) {
#define SPLIT_BNR(A_, M_) split_bnr_block<A_, M_>(p, slab, SLAB_LD, C, rowbase, colb, colok, lane, bmu, bis, bs)
          if (p.add) {
            if (mode == 2) SPLIT_BNR(true, 2); else if (mode == 1) SPLIT_BNR(true, 1); else SPLIT_BNR(true, 0);
          } else {
            if (mode == 2) SPLIT_BNR(false, 2); else if (mode == 1) SPLIT_BNR(false, 1); else SPLIT_BNR(false, 0);
          }
#undef SPLIT_BNR
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int r = rr + 8 * t;
            const int row = rowbase + r;
            f32x4 v = *reinterpret_cast<const f32x4*>(&slab[r * SLAB_LD + c4]);
            if (row < p.M && colok) {
              if (p.add) v += *reinterpret_cast<const f32x4*>(p.add + (size_t)row * p.ldadd + colb);
              *reinterpret_cast<f32x4*>(&C[(size_t)row * p.ldc + colb]) = v;
            }
          }
        }
      }
      if (bnr) {
        // lanes with equal (lane & 7) hold the same 4 columns for different rows: fold them in fp64, lanes 0-7 keep the totals
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          double t = (double)bs[k];
          t += __shfl_xor(t, 8);
          t += __shfl_xor(t, 16);
          t += __shfl_xor(t, 32);
          if (lane < 8) red2[((wm * SB_BN) + wn * 64 + j * 32 + lane * 4 + (k & 3)) * 2 + (k >> 2)] = t;
        }
      }
    }
    if (bnr) {
      __syncthreads();
      if (tid < SB_BN && n0 + tid < p.Nout) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int r = 0; r < WM; ++r) {
          s1 += red2[(r * SB_BN + tid) * 2 + 0];
          s2 += red2[(r * SB_BN + tid) * 2 + 1];
        }
        double* st = p.sums + (size_t)(tm % p.nslot) * 2 * p.Nout;
        atomic_add_f64(&st[n0 + tid], s1);
        atomic_add_f64(&st[p.Nout + n0 + tid], s2);
      }
    }
    return;
  }
  const bool wide = EPI_WIDE && (p.ldc & 3) == 0 && (p.Nout & 3) == 0 && ((((size_t)C) & 15) == 0);
  if (wide) {
    float* slab = reinterpret_cast<float*>(smem_raw) + wave * (32 * SLAB_LD);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int rr = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * lhi) * SLAB_LD + l31] = acc[i][j][e];
        const int colb = n0 + wn * 64 + j * 32 + c4;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int r = rr + 8 * t;
          const f32x4 v = *reinterpret_cast<const f32x4*>(&slab[r * SLAB_LD + c4]);
          const int row = m0 + wm * 64 + i * 32 + r;
          if (row < p.M && colb < p.Nout) *reinterpret_cast<f32x4*>(&C[(size_t)row * p.ldc + colb]) = v;
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (col >= p.Nout) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (row < p.M) C[(size_t)row * p.ldc + col] = acc[i][j][e];
      }
    }
}

}  // namespace

extern "C" int semseg_gemm_rows_batched_bf16split(const float* a, int lda, long long a_bs, const float* bt,
                                                  long long bt_bs, float* c, int ldc, long long c_bs, int M, int K,
                                                  int Nout, int batch, int nsplit, int bk, hipStream_t stream) {
  if (!a || !bt || !c || M <= 0 || Nout <= 0 || batch <= 0 || K <= 0) return SEMSEG_EINVAL;
  if ((nsplit != 2 && nsplit != 3) || (bk != 16 && bk != 32) || K % bk != 0 || (lda & 3) || (K & 3)) return SEMSEG_EINVAL;
  if (nsplit == 3 && bk != 16) return SEMSEG_EINVAL;      // three pieces at BK 32 would not leave two workgroups per CU
  SplitArgs p;
  std::memset(&p, 0, sizeof(p));
  p.stats = nullptr; p.nslot = 1;
  p.a = a; p.bt = bt; p.c = c;
  p.a_bs = a_bs; p.bt_bs = bt_bs; p.c_bs = c_bs;
  p.lda = lda; p.ldc = ldc; p.M = M; p.K = K; p.Nout = Nout;
  p.tiles_n = (Nout + SB_BN - 1) / SB_BN;
  // 128-row tiles (four waves, three workgroups per CU) where 256-row tiles would occupy at most half of the 256 CUs
  const bool small = nsplit == 3 && (long long)((M + 255) / 256) * p.tiles_n * batch <= 128;
  const int bm = small ? 128 : 256;
  p.tiles_m = (M + bm - 1) / bm;
  p.total = p.tiles_m * p.tiles_n * batch;
  if (nsplit == 2 && bk == 32) gemm_rows_bf16split_kernel<2, 32><<<p.total, 512, 0, stream>>>(p);
  else if (nsplit == 2) gemm_rows_bf16split_kernel<2, 16><<<p.total, 512, 0, stream>>>(p);
  else if (small) gemm_rows_bf16split_kernel<3, 16, 0, 128><<<p.total, 256, 0, stream>>>(p);
  else gemm_rows_bf16split_kernel<3, 16><<<p.total, 512, 0, stream>>>(p);
  return semseg_launch_status();
}

// tile codes 2128 (bm 256) / 3128 (bm 128) of semseg_conv_fwd (conv_igemm.hip): y[M][Co] = x[M][Ci] * w_fwd[Co_pad][Ci]^T + statistics, on the kernel above
int semseg_split_gemm_conv1x1_fwd(const float* x, int ldx, const float* w_fwd, float* y, int ldy, int M, int Ci, int Co,
                                  double* stats, int nslot, int bm, hipStream_t stream) {
  SplitArgs p;
  std::memset(&p, 0, sizeof(p));
  p.a = x; p.bt = w_fwd; p.c = y;
  p.a_bs = p.bt_bs = p.c_bs = 0;
  p.lda = ldx; p.ldc = ldy; p.M = M; p.K = Ci; p.Nout = Co;
  if (bm != 128 && bm != 256) return SEMSEG_EINVAL;
  p.tiles_m = (M + bm - 1) / bm;
  p.tiles_n = (Co + SB_BN - 1) / SB_BN;
  p.total = p.tiles_m * p.tiles_n;
  p.stats = stats; p.nslot = nslot > 0 ? nslot : 1;
  if (bm == 128) gemm_rows_bf16split_kernel<3, 16, 1, 128><<<p.total, 256, 0, stream>>>(p);
  else gemm_rows_bf16split_kernel<3, 16, 1><<<p.total, 512, 0, stream>>>(p);
  return semseg_launch_status();
}

// tile codes 2128 (bm 256) / 3128 (bm 128) of semseg_conv_dgrad / semseg_conv_dgrad_bnreduce: dx[M][Ci] = dy[M][Kc] * w_dgrad[Ci_pad][Kc]^T (+ add), with
// the fused BatchNorm-backward reduction of ONE layer when ybn is given
int semseg_split_gemm_conv1x1_dgrad(const float* dy, int lddy, const float* w_dgrad, float* dx, int lddx, int M, int Kc, int Ci,
                                    const float* add, int ldadd, const float* act, int ldact, const unsigned* relu_bits,
                                    int ldbits, const float* ybn, int ldybn, const float* mean, const float* invstd,
                                    double* sums, int nslot, int bm, hipStream_t stream) {
  if (bm != 128 && bm != 256) return SEMSEG_EINVAL;
  SplitArgs p;
  std::memset(&p, 0, sizeof(p));
  p.a = dy; p.bt = w_dgrad; p.c = dx;
  p.lda = lddy; p.ldc = lddx; p.M = M; p.K = Kc; p.Nout = Ci;
  p.tiles_m = (M + bm - 1) / bm;
  p.tiles_n = (Ci + SB_BN - 1) / SB_BN;
  p.total = p.tiles_m * p.tiles_n;
  p.nslot = nslot > 0 ? nslot : 1;
  p.add = add; p.ldadd = ldadd;
  p.bnr_n = ybn ? 1 : 0;
  p.mask = relu_bits ? nullptr : act; p.ldm = ldact; p.bits = relu_bits; p.ldb = ldbits;
  p.ybn = ybn; p.ldybn = ldybn; p.mean = mean; p.invstd = invstd; p.sums = sums;
  if (bm == 128) gemm_rows_bf16split_kernel<3, 16, 2, 128><<<p.total, 256, 0, stream>>>(p);
  else gemm_rows_bf16split_kernel<3, 16, 2><<<p.total, 512, 0, stream>>>(p);
  return semseg_launch_status();
}
