// Weight gradient of the PSPNet/PSANet convolutions for gfx950 (reference: the autograd of every nn.Conv2d of
// model/resnet.py:63-69,108-112,134; model/pspnet.py:15,65,69,73,77; model/psanet.py:25-48) and the batched K-major GEMM
// built on the same kernels (Winograd weight gradients, the PSA contraction's gradient): the register-staged kernel
// (64 x 64 / 128 x 128 tiles, fp32 and bf16x3 forms), the direct-to-LDS ring (fp32 form and the bf16x3 form that splits at
// fragment time), the deterministic split-K reduction into OIHW.  Split out of conv_igemm.hip in round 4.
#include "conv_common.h"

namespace {

#ifndef SPLITK_BATCH
#define SPLITK_BATCH 1   // split-K reductions: several slabs' loads in flight per trip (0 = one slab per trip; A/B builds)
#endif
#ifndef WGRAD_SP_POLICY
#define WGRAD_SP_POLICY 10
#endif
#ifndef WIDE_NSTAGE
#define WIDE_NSTAGE 4     // ring depth of the 128 x 256 kernel (24 KB per stage): 4 and 5 run the step equally fast, 6 (144 KB) loses the
                          // whole gain — the LDS it leaves is what lets the main stream's kernels share the CU
#endif
#ifndef SP_PIPE
#define SP_PIPE 1      // bf16x3 direct-to-LDS weight gradient: 0 = read -> split -> MFMA in sequence inside a stage (A/B builds)
#endif

// ------------------------------------------------------------------------------------------
// Weight gradient: dW[co][tap][ci] = sum_m dy[m][co] * x[pix(m,tap)][ci]   (K = pixels)
// One workgroup per (co tile, ci tile, tap, K split); partial sums go to a [ksplit] slab that
// wgrad_reduce_unpack sums deterministically while converting to the OIHW layout of .grad.
// ------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* x;
  const float* dy;
  float* dw;  // [ksplit][Co_pad][RS][Ci]
  int ldx, lddy;
  int N, Hin, Win, Ho, Wo;
  int Ci, Co_pad;
  int R, S, stride, pad, dil;
  int M;      // N*Ho*Wo
  int ksplit, kper;  // kper: pixels per split (multiple of 32)
  int tiles_co, tiles_ci;
  FastDiv div_hw, div_wo;
  // batched K-major GEMM (blockIdx.y = batch item): operand / slab strides in floats
  int batch;
  long long x_bs, dy_bs, dw_bs;
  int order;  // direct-to-LDS kernel: workgroup order inside a K slice (see conv_wgrad_dma_kernel)
};

// MODE 0: generic gather (any stride / padding).  MODE 1: 1x1, stride 1, pad 0 — the gathered row IS row m,
// no decode, always valid.  MODE 2: stride 1 with Ho x Wo == Hin x Win ("same" 3x3, any dilation) — the
// gathered row is m + tap offset (linear); the pixel is decoded only for the border test.
// SP = 3 (SEMSEG_ARITH_BF16X3, include/semseg_hip.h) — each thread
// stages FOUR CONSECUTIVE pixels of its four channels, cuts them into three bf16 pieces and stores them pixel-contiguous
// ([channel][32 pixels] planes, the layout the bf16 matrix-core instruction wants for a K-major operand), and the
// product is formed from six v_mfma_f32_32x32x16_bf16 per 16 pixels.  128 x 128: every thread stages both operands;
// 64 x 64 (the layers with 64 input or output channels, and every layer of a small per-GPU batch): a stage is half the
// data, so waves 0-1 stage dy and waves 2-3 stage x (HALF below).
template <int TM, int TN, int MODE, int SP = 0>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs pin) {
  static_assert(SP == 0 || (TM == TN && (TM == 128 || TM == 64)), "split mode needs 4 float4 per staging thread and operand");
  constexpr bool HALF = SP != 0 && TM == 64;
  WgradArgs p = pin;
  if (p.batch > 1) {
    const long long bz = blockIdx.y;
    p.x += bz * p.x_bs;
    p.dy += bz * p.dy_bs;
    p.dw += bz * p.dw_bs;
  }
  constexpr int MREP = TM / 64, NREP = TN / 64;
  constexpr int YV = TM / 4, XV = TN / 4;          // float4 per k-row
  constexpr int Y_PER = HALF ? 4 : 32 * YV / 256, X_PER = HALF ? 4 : 32 * XV / 256;
  constexpr int YROWS = 256 / YV, XROWS = 256 / XV;  // k-rows covered per pass
  __shared__ __attribute__((aligned(16))) float smem[SP ? SP * 16 * (TM + TN) : 32 * (TM + TN)];
  float* Ys = smem;            // [32][TM]
  float* Xs = smem + 32 * TM;  // [32][TN]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int RS = p.R * p.S;
  // XCD-aware order: workgroups that share a pixel range (same ks) and neighbouring taps / tiles read the
  // same x and dy rows; give each XCD a contiguous chunk of the logical order so those rows are fetched
  // into one L2 instead of all eight.
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int tci = b % p.tiles_ci; b /= p.tiles_ci;
  const int tco = b % p.tiles_co; b /= p.tiles_co;
  const int tap = b % RS;
  const int ks = b / RS;
  const int r = tap / p.S, s = tap - r * p.S;
  const int co0 = tco * TM, ci0 = tci * TN;
  const int kbeg = ks * p.kper;
  const int kend = min(p.M, kbeg + p.kper);

  // SP: the pixel group (4 consecutive pixels) is the fastest thread index, so that the 8 lanes of one channel quad
  // fill one 64-byte LDS row per store
  const int st = HALF ? (tid & 127) : tid;          // index among the threads staging one operand
  const bool stage_y = !HALF || tid < 128, stage_x = !HALF || tid >= 128;   // wave-uniform
  const int yc = SP ? st >> 3 : tid % YV, yr = SP ? st & 7 : tid / YV;
  const int xc = SP ? st >> 3 : tid % XV, xr = SP ? st & 7 : tid / XV;
  const int hw = p.Ho * p.Wo;

  f32x4 ry[Y_PER], rx[X_PER];
  const int tapoff = ((r * p.dil - p.pad) * p.Win + (s * p.dil - p.pad)) * p.ldx + ci0 + xc * 4;
  // linear modes: row m of the gather is x + (m + tap offset) * ldx (the offset may be negative)
  const float* const xlin = p.x + (ptrdiff_t)tapoff;
  // Measured on this kernel (scripts/tune_conv.py): without the in-loop global loads it
  // runs 13 % faster (WGRAD_ABL=1), yet none of these moved it: buffer loads (neutral/negative), dropping the
  // decode entirely for 1x1 (MODE 1, +2 %), issuing the prefetch at kp 0 instead of 3 (neutral), a two-step
  // deep register prefetch (occupancy 3 -> 2, -5.6 %), XCD-aware block order (neutral in time), K-contiguous
  // LDS with ds_read_b128 fragments (-1 %).
  auto prefetch_into = [&](int kb, f32x4 (&ry)[Y_PER], f32x4 (&rx)[X_PER]) {
    if (stage_y)
#pragma unroll
    for (int i = 0; i < Y_PER; ++i) {
      const int m = SP ? kb + yr * Y_PER + i : kb + yr + i * YROWS;
      const float* src = m < kend ? p.dy + (size_t)m * p.lddy + co0 + yc * 4 : g_zero_line;
      ry[i] = *reinterpret_cast<const f32x4*>(src);
    }
    if (stage_x)
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
      const int m = SP ? kb + xr * X_PER + i : kb + xr + i * XROWS;
      if constexpr (MODE == 1) {
        const float* src = m < kend ? xlin + (size_t)m * p.ldx : g_zero_line;
        rx[i] = *reinterpret_cast<const f32x4*>(src);
      } else {
        const int mm = m < kend ? m : kbeg;
        const int n = fdiv(mm, p.div_hw);
        const int rem = mm - n * hw;
        const int oh = fdiv(rem, p.div_wo);
        const int ow = rem - oh * p.Wo;
        const int ih = oh * p.stride + r * p.dil - p.pad;
        const int iw = ow * p.stride + s * p.dil - p.pad;
        const bool ok = m < kend && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
        const float* src;
        if constexpr (MODE == 2)
          src = ok ? xlin + (size_t)m * p.ldx : g_zero_line;
        else
          src = ok ? p.x + (((n * p.Hin + oh * p.stride) * p.Win + ow * p.stride) * p.ldx + tapoff)
                   : g_zero_line;
        rx[i] = *reinterpret_cast<const f32x4*>(src);
      }
    }
  };
  auto prefetch = [&](int kb) { prefetch_into(kb, ry, rx); };

  f32x16 acc[MREP][NREP];
#pragma unroll
  for (int i = 0; i < MREP; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // WGRAD_ABL (measurement only, results are wrong): 1 = no global loads in the loop, 2 = additionally no
  // LDS stores / barriers per step, 3 = additionally fragments from registers (pure MFMA ceiling)
  if (kbeg < kend) prefetch(kbeg);
  if constexpr (SP) {
    __bf16* Yp = reinterpret_cast<__bf16*>(smem);      // [piece][TM channels][32 pixels], 16-byte chunks swizzled
    __bf16* Xp = Yp + SP * TM * 32;
    // rows of odd channel quads are stored pairwise swapped (row ^ 1): the 16 lanes of one 8-byte store cycle write the
    // same channel of two neighbouring quads, and without the swap both rows start on the same 16 of the 32 store banks
    auto sp_off = [](int row, int chunk) {
      const int pr = row ^ ((row >> 2) & 1);
      return pr * 32 + ((chunk ^ ((pr >> 2) & 3)) << 3);
    };
    auto sp_store = [&](__bf16* plane0, int plane_elems, int quad, int pg, const f32x4 (&rr)[4]) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x4 v = {rr[0][c], rr[1][c], rr[2][c], rr[3][c]};     // four consecutive pixels of one channel
#pragma unroll
        for (int pc = 0; pc < SP; ++pc) {
          const bf16x4 h = __builtin_convertvector(v, bf16x4);
          if (pc + 1 < SP) v -= bf16x4_to_f32(h);
          *reinterpret_cast<bf16x4*>(&plane0[pc * plane_elems + sp_off(quad * 4 + c, pg >> 1) + (pg & 1) * 4]) = h;
        }
      }
    };
    for (int kb = kbeg; kb < kend; kb += 32) {
      if (stage_y) sp_store(Yp, TM * 32, yc, yr, ry);
      if (stage_x) sp_store(Xp, TN * 32, xc, xr, rx);
      __syncthreads();
#pragma unroll
      for (int k16 = 0; k16 < 2; ++k16) {
        bf16x8 fa[MREP][SP], fb[NREP][SP];
#pragma unroll
        for (int pc = 0; pc < SP; ++pc) {
#pragma unroll
          for (int i = 0; i < MREP; ++i)
            fa[i][pc] = *reinterpret_cast<const bf16x8*>(&Yp[pc * TM * 32 + sp_off(wm * (TM / 2) + i * 32 + l31, k16 * 2 + lhi)]);
#pragma unroll
          for (int j = 0; j < NREP; ++j)
            fb[j][pc] = *reinterpret_cast<const bf16x8*>(&Xp[pc * TN * 32 + sp_off(wn * (TN / 2) + j * 32 + l31, k16 * 2 + lhi)]);
        }
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[q]], fb[j][PB[q]], acc[i][j], 0, 0, 0);
        if (k16 == 0 && kb + 32 < kend) prefetch(kb + 32);
      }
      __syncthreads();
    }
  } else {
    for (int kb = kbeg; kb < kend; kb += 32) {
  #pragma unroll
      for (int i = 0; i < Y_PER; ++i)
        *reinterpret_cast<f32x4*>(&Ys[(yr + i * YROWS) * TM + yc * 4]) = ry[i];
  #pragma unroll
      for (int i = 0; i < X_PER; ++i)
        *reinterpret_cast<f32x4*>(&Xs[(xr + i * XROWS) * TN + xc * 4]) = rx[i];
      __syncthreads();
      // fragments of k-pair kp+1 are read from LDS while the MFMAs of k-pair kp issue (the compiler
      // otherwise waits for every ds_read right in front of its 4 MFMAs)
      float fa[2][MREP], fb[2][NREP];
  #pragma unroll
      for (int i = 0; i < MREP; ++i) fa[0][i] = Ys[(lhi)*TM + wm * (TM / 2) + i * 32 + l31];
  #pragma unroll
      for (int j = 0; j < NREP; ++j) fb[0][j] = Xs[(lhi)*TN + wn * (TN / 2) + j * 32 + l31];
  #pragma unroll
      for (int kp = 0; kp < 16; ++kp) {
        if (kp + 1 < 16) {
  #pragma unroll
          for (int i = 0; i < MREP; ++i)
            fa[(kp + 1) & 1][i] = Ys[(2 * (kp + 1) + lhi) * TM + wm * (TM / 2) + i * 32 + l31];
  #pragma unroll
          for (int j = 0; j < NREP; ++j)
            fb[(kp + 1) & 1][j] = Xs[(2 * (kp + 1) + lhi) * TN + wn * (TN / 2) + j * 32 + l31];
        }
  #pragma unroll
        for (int i = 0; i < MREP; ++i)
  #pragma unroll
          for (int j = 0; j < NREP; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kp & 1][i], fb[kp & 1][j], acc[i][j], 0, 0, 0);
        if (kp == 3 && kb + 32 < kend) prefetch(kb + 32);
      }
      __syncthreads();
    }

  }
  float* out = p.dw + (size_t)ks * p.Co_pad * RS * p.Ci;
#pragma unroll
  for (int i = 0; i < MREP; ++i)
#pragma unroll
    for (int j = 0; j < NREP; ++j) {
      const int ci = ci0 + wn * (TN / 2) + j * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + wm * (TM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
        if (co < p.Co_pad && ci < p.Ci) out[((size_t)co * RS + tap) * p.Ci + ci] = acc[i][j][e];
      }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient, direct-to-LDS variant (128 x 128 tiles).  Both operands are K-major in memory exactly as the
// MFMA wants them in LDS ([k][row], a k-row of 128 floats = 512 contiguous bytes of one pixel), so a tile is a
// lane-linear image: every wave streams its k-rows with `buffer_load_dwordx4 ... lds` (LDS-DMA, 1 KiB per
// wave-instruction) into an NSTAGE-deep LDS ring — no staging VGPRs, no ds_write pass, and the prefetch depth is
// set by the ring, not by registers.  Invalid rows (padding taps, rows past M) carry an out-of-range buffer offset:
// the buffer unit returns 0 for them, which is what lands in LDS.  One raw s_barrier per K-step; the DMA queue is
// drained with a COUNTED vmcnt so that NSTAGE-2 stages stay in flight across the barrier.
// Fragments are read as ds_read_b64: lane l31 takes tile rows (2*l31, 2*l31+1), i.e. MFMA block i holds the rows
// 2*r + i — a relabelling that only the epilogue sees (and which turns its stores into 8-byte lanes).
// ------------------------------------------------------------------------------------------
// 16 bytes per lane from a raw buffer straight into LDS at (wave-uniform lds + lane * 16).  The builtin only exists
// for the device pass: on the host pass of a TEMPLATE kernel it silently suppresses the launch stub (ROCm 7.2), hence
// the guard.
__device__ __forceinline__ void dma16_to_lds(__amdgpu_buffer_rsrc_t rsrc, float* lds, unsigned voffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, lds, 16, voffset, 0, 0, 0);
#endif
}

// SP = 3 (SEMSEG_ARITH_BF16X3, round 4): the same ring, the same DMA stream, but the fragments are formed for the bf16
// matrix-core instruction.  LDS-DMA cannot transform what it moves, so the stage stays fp32 and K-major, and the split
// happens at FRAGMENT time: a lane reads the 8 pixels of its k-group for its two rows (8 ds_read_b64, the even / odd row
// relabelling of the fp32 kernel), cuts the 16 floats into three bf16 pieces each in registers and packs them along K —
// which is exactly the transposition v_mfma_f32_32x32x16_bf16 needs for a K-major operand.  Each tile element is split by
// the two waves that use it (2x the conversion VALU of the register-staged SP kernel, ~350 VALU cycles against 768
// matrix-pipe cycles per stage and wave), in exchange for: no staging VGPRs, no ds_write pass, ONE barrier per K-step and a
// prefetch depth set by the ring instead of one K-step (the register-staged SP kernel ran at 160 TFLOP/s fp32-equivalent,
// latency-bound on its one-step prefetch).  KS must be 16 (one MFMA K-group per stage).

template <int MODE, int KS, int NSTAGE, int OCC, bool TL2, int SP = 0>
__global__ __launch_bounds__(256, OCC) void conv_wgrad_dma_kernel(const WgradArgs pin) {
  static_assert(SP == 0 || (SP == 3 && KS == 16), "split form: three pieces, one 16-pixel K-group per stage");
  WgradArgs p = pin;
  if (p.batch > 1) {
    const long long bz = blockIdx.y;
    p.x += bz * p.x_bs;
    p.dy += bz * p.dy_bs;
    p.dw += bz * p.dw_bs;
  }
  constexpr int TM = 128, TN = 128;
  constexpr int STAGE_F = KS * (TM + TN);
  constexpr int RPW = KS / 4;   // k-rows each wave streams per operand per stage
  constexpr int NI = RPW / 2;   // DMA instructions per operand per wave per stage (2 k-rows = 1 KiB each)
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE_F];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int RS = p.R * p.S;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  // Order inside a K slice decides which operand block the ~64 workgroups resident on an XCD share through its L2
  // (a workgroup streams one x block [pixels, 128 ci] and one dy block [pixels, 128 co]):
  //   0: ci tile fastest, then co tile, then tap   (64 neighbours: 1 tap, 2 co tiles, 32 ci tiles -> 34 blocks)
  //   1: tap fastest, then co tile, ci tile slowest (the 9 taps of a tile pair are neighbours and share BOTH blocks;
  //      64 neighbours of a 4 x 32 x 9 grid touch 2 x blocks + 4 dy blocks)
  //   2: tap fastest, then ci tile, co tile slowest
  const int per_ks = p.tiles_ci * p.tiles_co * RS;
  const int ks = b / per_ks;
  b -= ks * per_ks;
  int tci, tco, tap;
  if (p.order == 1) {
    tap = b % RS; b /= RS;
    tco = b % p.tiles_co;
    tci = b / p.tiles_co;
  } else if (p.order == 2) {
    tap = b % RS; b /= RS;
    tci = b % p.tiles_ci;
    tco = b / p.tiles_ci;
  } else {
    tci = b % p.tiles_ci; b /= p.tiles_ci;
    tco = b % p.tiles_co;
    tap = b / p.tiles_co;
  }
  const int r = tap / p.S, s = tap - r * p.S;
  const int co0 = tco * TM, ci0 = tci * TN;
  const int kbeg = ks * p.kper;
  const int kend = min(p.M, kbeg + p.kper);
  const int hw = p.Ho * p.Wo;

  // raw buffers: rows past the end of dy / x are out of range by construction (num_records), so the M tail needs
  // no test at all for dy and for the 1x1 case
  const __amdgpu_buffer_rsrc_t ry_ = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.dy, 0, (int)min((size_t)0x7FFFFFFFu, (size_t)p.M * p.lddy * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.x, 0, (int)min((size_t)0x7FFFFFFFu, (size_t)p.N * p.Hin * p.Win * p.ldx * 4), 0x00020000);

  // this lane's k-row inside a stage, per DMA instruction, and its 16-byte column
  const int rrow = wave * RPW + lhi;   // + 2*i
  const int col = l31 * 4;
  unsigned yoff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) yoff[i] = (unsigned)(((kbeg + rrow + 2 * i) * p.lddy + co0 + col) * 4);
  const unsigned ystep = (unsigned)(KS * p.lddy * 4);
  // x: byte offset of (row m, this lane's column) is m * ldx * 4 + xcol (+ the tap shift in the linear modes)
  const int tapoff = ((r * p.dil - p.pad) * p.Win + (s * p.dil - p.pad)) * p.ldx + ci0 + col;
  unsigned xoff[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) xoff[i] = (unsigned)(((kbeg + rrow + 2 * i) * p.ldx + tapoff) * 4);
  const unsigned xstep = (unsigned)(KS * p.ldx * 4);

  auto issue = [&](int kb, int slot) {
    float* Ysl = smem + slot * STAGE_F;
    float* Xsl = Ysl + KS * TM;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      dma16_to_lds(ry_, Ysl + (wave * RPW + 2 * i) * TM, yoff[i]);
      yoff[i] += ystep;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      unsigned vo;
      if constexpr (MODE == 1) {
        vo = xoff[i];
      } else {
        const int m = kb + rrow + 2 * i;
        const int mm = m < kend ? m : kbeg;
        const int n = fdiv(mm, p.div_hw);
        const int rem = mm - n * hw;
        const int oh = fdiv(rem, p.div_wo);
        const int ow = rem - oh * p.Wo;
        const int ih = oh * p.stride + r * p.dil - p.pad;
        const int iw = ow * p.stride + s * p.dil - p.pad;
        const bool ok = m < kend && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
        if constexpr (MODE == 2)
          vo = ok ? xoff[i] : OOB;
        else
          vo = ok ? (unsigned)((((n * p.Hin + ih) * p.Win + iw) * p.ldx + ci0 + col) * 4) : OOB;
      }
      dma16_to_lds(rx_, Xsl + (wave * RPW + 2 * i) * TN, vo);
      xoff[i] += xstep;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // Two-level accumulation (rings that run 2 workgroups per CU have the registers for it): the MFMA chain is flushed
  // into a second accumulator set every 512 pixels, which bounds its rounding noise (it grows ~sqrt(chain length);
  // a K split of a bs-16 layer is 1200-14400 pixels long).
  constexpr bool TWO_LEVEL = TL2;
  constexpr int FLUSH_STEPS = 512 / KS;
  f32x16 acc2[TWO_LEVEL ? 2 : 1][TWO_LEVEL ? 2 : 1];
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
  }

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 fa[2], fb[2];
  // the first fragments of a stage are requested BEFORE the next stage's DMA is issued, so their LDS latency
  // hides under that issue code instead of sitting between the barrier and the first MFMA
  auto first_frags = [&](int slot) {
    const float* Ya = smem + slot * STAGE_F + lhi * TM + wm * 64 + 2 * l31;
    const float* Xa = smem + slot * STAGE_F + KS * TM + lhi * TN + wn * 64 + 2 * l31;
    fa[0] = *reinterpret_cast<const f32x2*>(Ya);
    fb[0] = *reinterpret_cast<const f32x2*>(Xa);
  };
  auto compute = [&](int slot) {
    const float* Ya = smem + slot * STAGE_F + lhi * TM + wm * 64 + 2 * l31;
    const float* Xa = smem + slot * STAGE_F + KS * TM + lhi * TN + wn * 64 + 2 * l31;
#pragma unroll
    for (int kp = 0; kp < KS / 2; ++kp) {
      if (kp + 1 < KS / 2) {
        fa[(kp + 1) & 1] = *reinterpret_cast<const f32x2*>(Ya + 2 * (kp + 1) * TM);
        fb[(kp + 1) & 1] = *reinterpret_cast<const f32x2*>(Xa + 2 * (kp + 1) * TN);
      }
      const f32x2 a = fa[kp & 1], bb = fb[kp & 1];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bb[0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], bb[1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], bb[0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], bb[1], acc[1][1], 0, 0, 0);
    }
    // pin the issue order "fragments of k-pair kp+1, then the 4 MFMAs of k-pair kp" (the scheduler otherwise
    // batches the reads of two k-pairs and waits for them right in front of 8 MFMAs)
#pragma unroll
    for (int kp = 0; kp < KS / 2; ++kp) {
      if (kp + 1 < KS / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
  };

  // SP: raw fp32 fragments of a stage (8 pixels x 2 rows per operand and lane), their split into bf16 pieces, the 24 MFMAs
  auto sp_read = [&](int slot, f32x2 (&ya)[8], f32x2 (&xa)[8]) {
    const float* Ya = smem + slot * STAGE_F + (lhi * 8) * TM + wm * 64 + 2 * l31;
    const float* Xa = smem + slot * STAGE_F + KS * TM + (lhi * 8) * TN + wn * 64 + 2 * l31;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      ya[kk] = *reinterpret_cast<const f32x2*>(Ya + kk * TM);
      xa[kk] = *reinterpret_cast<const f32x2*>(Xa + kk * TN);
    }
  };
  auto sp_split = [&](const f32x2 (&ya)[8], const f32x2 (&xa)[8], bf16x8 (&fa2)[2][3], bf16x8 (&fb2)[2][3]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x8 va, vb;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        va[kk] = ya[kk][i];
        vb[kk] = xa[kk][i];
      }
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) {
        const bf16x8 ha = __builtin_convertvector(va, bf16x8);
        const bf16x8 hb = __builtin_convertvector(vb, bf16x8);
        fa2[i][pc] = ha;
        fb2[i][pc] = hb;
        if (pc < 2) {
          va -= bf16x8_to_f32(ha);
          vb -= bf16x8_to_f32(hb);
        }
      }
    }
  };
  auto sp_mfma = [&](const bf16x8 (&fa2)[2][3], const bf16x8 (&fb2)[2][3]) {
    // small terms first, the leading product last; consecutive MFMAs go to different accumulators
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa2[i][PA[q]], fb2[j][PB[q]], acc[i][j], 0, 0, 0);
  };
  auto compute_sp = [&](int slot) {
    f32x2 ya[8], xa[8];
    bf16x8 fa2[2][3], fb2[2][3];
    sp_read(slot, ya, xa);
    sp_split(ya, xa, fa2, fb2);
    sp_mfma(fa2, fb2);
  };

  const int nsteps = (kend - kbeg + KS - 1) / KS;
  // Stages past the end of this split are issued too (their rows are either another split's valid memory or out
  // of range): the DMA count per iteration stays constant, which is what the counted vmcnt relies on.
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) issue(kbeg + st * KS, st);
  int slot = 0, pslot = NSTAGE - 1;
  if constexpr (SP != 0 && SP_PIPE != 0 && NSTAGE >= 4) {
    // Software-pipelined form: the raw fragments of stage t + 1 are read and split WHILE the 24 matrix-core instructions
    // of stage t issue (an in-order wave hides ~5 single-issue instructions behind each 32-cycle MFMA, and the split is
    // ~180 VALU instructions per stage), instead of read -> wait -> split -> MFMAs in sequence.  For that, stage t + 1
    // must have landed when iteration t starts: the counted wait moves one stage earlier (NSTAGE - 3 stages stay in
    // flight across the barrier instead of NSTAGE - 2).  The barrier of iteration t says that every wave has finished
    // reading stage t (it did so in iteration t - 1), so the DMA of stage t + NSTAGE - 1 may overwrite slot (t - 1).
    bf16x8 fca[2][3], fcb[2][3];
    {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI * (NSTAGE - 2)) : "memory");
      __builtin_amdgcn_s_barrier();
      f32x2 ya[8], xa[8];
      sp_read(0, ya, xa);
      sp_split(ya, xa, fca, fcb);
    }
    int nslot_ = 1, pslot_ = NSTAGE - 1;
    for (int t = 0; t < nsteps; ++t) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI * (NSTAGE - 3)) : "memory");
      __builtin_amdgcn_s_barrier();
      issue(kbeg + (t + NSTAGE - 1) * KS, pslot_);
      f32x2 ya[8], xa[8];
      bf16x8 fna[2][3], fnb[2][3];
      sp_read(nslot_, ya, xa);        // stage t + 1 (past the end of the split: rows of another split or zeros, never used)
      sp_mfma(fca, fcb);
      sp_split(ya, xa, fna, fnb);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
          fca[i][pc] = fna[i][pc];
          fcb[i][pc] = fnb[i][pc];
        }
      pslot_ = pslot_ + 1 == NSTAGE ? 0 : pslot_ + 1;
      nslot_ = nslot_ + 1 == NSTAGE ? 0 : nslot_ + 1;
    }
  } else
  for (int t = 0; t < nsteps; ++t) {
    // stage t has landed for this wave once at most NSTAGE-2 younger stages are still outstanding ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI * (NSTAGE - 2)) : "memory");
    // ... and for every wave after the barrier, which also says: everybody is done reading slot (t-1) % NSTAGE
    __builtin_amdgcn_s_barrier();
    if constexpr (SP) {
      issue(kbeg + (t + NSTAGE - 1) * KS, pslot);
      compute_sp(slot);
    } else {
      first_frags(slot);
      issue(kbeg + (t + NSTAGE - 1) * KS, pslot);
      compute(slot);
    }
    pslot = slot;
    slot = slot + 1 == NSTAGE ? 0 : slot + 1;
    if constexpr (TWO_LEVEL) {
      if ((t & (FLUSH_STEPS - 1)) == FLUSH_STEPS - 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              acc2[i][j][e] += acc[i][j][e];
              acc[i][j][e] = 0.f;
            }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may land in LDS after the workgroup has retired
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += acc2[i][j][e];
  }

  float* out = p.dw + (size_t)ks * p.Co_pad * RS * p.Ci;
  const int ci = ci0 + wn * 64 + 2 * l31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rr = (e & 3) + 8 * (e >> 2) + 4 * lhi;
      const int co = co0 + wm * 64 + 2 * rr + i;
      f32x2 v = {acc[i][0][e], acc[i][1][e]};
      *reinterpret_cast<f32x2*>(out + ((size_t)co * RS + tap) * p.Ci + ci) = v;
    }
}

// ------------------------------------------------------------------------------------------
// bf16x3 weight gradient with a 64 x 128 wave tile (128 x 256 workgroup tile, 4 waves, ONE workgroup per CU).
// The 128 x 128 kernel above splits 32 floats per lane for 24 matrix-core instructions per stage (2 + 2 fragments); its
// matrix pipe is busy 0.51 by PMC and it waits on VALU issue.  With a 64 x 128 wave tile a lane splits 48 floats for 48
// instructions (2 + 4 fragments): two thirds of the conversion work per product.  Price: 128 accumulator registers and a
// 24 KB stage, i.e. one wave per SIMD — every VALU instruction of the split has to hide behind the matrix-core instructions of
// the SAME wave (the software-pipelined loop of the kernel above: fragments of stage t + 1 are read and split while the 48
// instructions of stage t issue), nothing else of THIS kernel is resident to cover a stall.  Same ring, same DMA stream, same
// slab layout.  Measured: alone it runs at the rate of the 128 x 128 kernel (172 vs 169 TFLOP/s fp32-equivalent on layer3's
// shapes, 180 vs 181 on cls.0 — both at what the chip sustains for six-product work), but the STEP is 3.8 % faster with it
// (118.8-119.0 -> 114.4-114.5 ms): weight gradients run on the side stream next to the data-gradient / BatchNorm chain, and
// four waves and 96 KB per CU leave that chain the issue slots and the LDS that eight waves and 128 KB did not.
// x rows are 256 floats = 1 KiB = one wave-instruction per pixel, so the gather decode of a DMA instruction is wave-uniform.
// ------------------------------------------------------------------------------------------
#ifndef WGRAD_WIDE_INTERLEAVE
#define WGRAD_WIDE_INTERLEAVE 1     // 0: sp_mfma + sp_split in the compiler's own order (rounds 4-5; A/B builds)
#endif
#ifndef WGRAD_ACC2_MIN_M
#define WGRAD_ACC2_MIN_M 16384     // 1x1 weight gradients with a longer reduction run the two-accumulator instance (ACC2; see the kernel); 0x7fffffff: never (A/B builds)
#endif
template <int MODE, int NSTAGE, int ACC2 = 0>
__global__ __launch_bounds__(256, 1) void conv_wgrad_dma_wide_kernel(const WgradArgs pin) {
  WgradArgs p = pin;
  if (p.batch > 1) {
    const long long bz = blockIdx.y;
    p.x += bz * p.x_bs;
    p.dy += bz * p.dy_bs;
    p.dw += bz * p.dw_bs;
  }
  constexpr int KS = 16, TM = 128, TN = 256;
  constexpr int STAGE_F = KS * (TM + TN);
  constexpr int NIY = 2, NIX = 4;          // DMA instructions per wave and stage: dy (2 k-rows each), x (1 k-row each)
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) float smem[NSTAGE * STAGE_F];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int RS = p.R * p.S;
  int b = xcd_remap(blockIdx.x, gridDim.x);
  const int per_ks = p.tiles_ci * p.tiles_co * RS;
  const int ks = b / per_ks;
  b -= ks * per_ks;
  int tci, tco, tap;
  if (p.order == 1) {
    tap = b % RS; b /= RS;
    tco = b % p.tiles_co;
    tci = b / p.tiles_co;
  } else if (p.order == 2) {
    tap = b % RS; b /= RS;
    tci = b % p.tiles_ci;
    tco = b / p.tiles_ci;
  } else {
    tci = b % p.tiles_ci; b /= p.tiles_ci;
    tco = b % p.tiles_co;
    tap = b / p.tiles_co;
  }
  const int r = tap / p.S, s = tap - r * p.S;
  const int co0 = tco * TM, ci0 = tci * TN;
  const int kbeg = ks * p.kper;
  const int kend = min(p.M, kbeg + p.kper);
  const int hw = p.Ho * p.Wo;

  const __amdgpu_buffer_rsrc_t ry_ = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.dy, 0, (int)min((size_t)0x7FFFFFFFu, (size_t)p.M * p.lddy * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.x, 0, (int)min((size_t)0x7FFFFFFFu, (size_t)p.N * p.Hin * p.Win * p.ldx * 4), 0x00020000);

  // dy: this lane's k-row inside a stage is wave * 4 + lhi (+ 2 i), 16-byte column l31 * 4
  unsigned yoff[NIY];
#pragma unroll
  for (int i = 0; i < NIY; ++i) yoff[i] = (unsigned)(((kbeg + wave * 4 + lhi + 2 * i) * p.lddy + co0 + l31 * 4) * 4);
  const unsigned ystep = (unsigned)(KS * p.lddy * 4);
  // x: k-row wave * 4 + i, 16-byte column lane * 4
  const int tapoff = ((r * p.dil - p.pad) * p.Win + (s * p.dil - p.pad)) * p.ldx + ci0 + lane * 4;
  unsigned xoff[NIX];
#pragma unroll
  for (int i = 0; i < NIX; ++i) xoff[i] = (unsigned)(((kbeg + wave * 4 + i) * p.ldx + tapoff) * 4);
  const unsigned xstep = (unsigned)(KS * p.ldx * 4);

  auto issue = [&](int kb, int slot) {
    float* Ysl = smem + slot * STAGE_F;
    float* Xsl = Ysl + KS * TM;
#pragma unroll
    for (int i = 0; i < NIY; ++i) {
      dma16_to_lds(ry_, Ysl + (wave * 4 + 2 * i) * TM, yoff[i]);
      yoff[i] += ystep;
    }
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
      unsigned vo;
      if constexpr (MODE == 1) {
        vo = xoff[i];
      } else {
        const int m = kb + wave * 4 + i;          // one pixel per instruction: wave-uniform
        const int mm = m < kend ? m : kbeg;
        const int n = fdiv(mm, p.div_hw);
        const int rem = mm - n * hw;
        const int oh = fdiv(rem, p.div_wo);
        const int ow = rem - oh * p.Wo;
        const int ih = oh * p.stride + r * p.dil - p.pad;
        const int iw = ow * p.stride + s * p.dil - p.pad;
        const bool ok = m < kend && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
        if constexpr (MODE == 2)
          vo = ok ? xoff[i] : OOB;
        else
          vo = ok ? (unsigned)((((n * p.Hin + ih) * p.Win + iw) * p.ldx + ci0 + lane * 4) * 4) : OOB;
      }
      dma16_to_lds(rx_, Xsl + (wave * 4 + i) * TN, vo);
      xoff[i] += xstep;
    }
  };

  // ACC2 = 1 (round 6; the instance 1x1 weight gradients with reductions longer than WGRAD_ACC2_MIN_M pixels run): the LEADING product
  // (high piece x high piece) of every K step goes into `acc`, the five small cross products (2^-9 ... 2^-18 of it) into a second
  // set `accs`, added once at the end.  Why: the bf16 matrix-core instruction costs the accumulator it adds into ~0.3 ulp of noise
  // per PRODUCT, whatever the size of the product, so with one set the six instructions of a K step put sqrt(6) x the leading
  // product's noise on the sum.  Measured (scripts/wgrad_noise.py, 55 696-pixel reductions, random operands, rms against fp64):
  // exact fp32 products 3.1e-7, bf16x3 with one set 5.6e-7 (Ci 512) / 7.5e-7 (Ci 1024), with two sets 2.0e-7 / 2.2e-7 — below the
  // exact path.  One set put five 1x1 weight gradients of PSANet-101 at batch 16 at 3.1-4.2 x the CPU-fp32 noise in situ (criterion
  // 3 x).  Price: 256 accumulator registers (one wave per SIMD: they are there), +9 % per launch — paid where
  // the accumulator noise is what limits the result (long single reductions), not by the batched Winograd-domain launches (1.3 x).
  f32x16 acc[2][4];
  f32x16 accs[ACC2 ? 2 : 1][ACC2 ? 4 : 1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[i][j][e] = 0.f;
        if constexpr (ACC2) accs[i][j][e] = 0.f;
      }

  typedef float f32x2 __attribute__((ext_vector_type(2)));
  // raw fp32 fragments of a stage: the 8 pixels of this lane's k-group for its 2 dy rows (2 l31 + i) and 4 x columns (4 l31 + j)
  auto sp_read = [&](int slot, f32x2 (&ya)[8], f32x4 (&xa)[8]) {
    const float* Ya = smem + slot * STAGE_F + (lhi * 8) * TM + wm * 64 + 2 * l31;
    const float* Xa = smem + slot * STAGE_F + KS * TM + (lhi * 8) * TN + wn * 128 + 4 * l31;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      ya[kk] = *reinterpret_cast<const f32x2*>(Ya + kk * TM);
      xa[kk] = *reinterpret_cast<const f32x4*>(Xa + kk * TN);
    }
  };
  auto split8 = [&](f32x8 v, bf16x8 (&out)[3]) {
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
      out[pc] = __builtin_convertvector(v, bf16x8);
      if (pc < 2) v -= bf16x8_to_f32(out[pc]);
    }
  };
  auto sp_split = [&](const f32x2 (&ya)[8], const f32x4 (&xa)[8], bf16x8 (&fa2)[2][3], bf16x8 (&fb2)[4][3]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x8 va;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) va[kk] = ya[kk][i];
      split8(va, fa2[i]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x8 vb;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) vb[kk] = xa[kk][j];
      split8(vb, fb2[j]);
    }
  };
  auto sp_mfma = [&](const bf16x8 (&fa2)[2][3], const bf16x8 (&fb2)[4][3]) {
    // small terms first, the leading product last; consecutive instructions go to different accumulators
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ACC2 && q < 5) accs[ACC2 ? i : 0][ACC2 ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa2[i][PA[q]], fb2[j][PB[q]], accs[ACC2 ? i : 0][ACC2 ? j : 0], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa2[i][PA[q]], fb2[j][PB[q]], acc[i][j], 0, 0, 0);
        }
  };

  const int nsteps = (kend - kbeg + KS - 1) / KS;
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st) issue(kbeg + st * KS, st);
  bf16x8 fca[2][3], fcb[4][3];
  {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NIY + NIX) * (NSTAGE - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    f32x2 ya[8];
    f32x4 xa[8];
    sp_read(0, ya, xa);
    sp_split(ya, xa, fca, fcb);
  }
  int nslot_ = 1, pslot_ = NSTAGE - 1;
#if WGRAD_WIDE_INTERLEAVE
  // One wave per SIMD: nothing but this wave's OWN VALU work can run under its matrix-core instructions, and a wave issues in order —
  // the 48 matrix-core instructions of a stage (32 cycles each) and the ~290 VALU instructions that split the next stage's fragments
  // (4 cycles each) overlap only where they ALTERNATE in the instruction stream.  Left to itself the compiler emits the 48 in runs of
  // 11-18 and ~215 VALU instructions behind the last one (ISA of rounds 4-5: the matrix pipe was busy 0.48 of the kernel); a
  // sched_group_barrier pipeline is dropped by its solver after ~11 groups.  So the stage is written as 48 slots — one matrix-core
  // instruction + one ~10-instruction piece of the split — with a scheduling barrier behind each slot.  The split of a vector of 8
  // floats is cut into 5 pieces (first / second bf16 piece of either half: convert, widen, subtract; third piece: convert): 30 pieces,
  // five in every eight slots.  The two fragment sets swap roles from stage to stage (no copies).
  auto split_piece = [&](int c, f32x2 (&ya)[8], f32x4 (&xa)[8], bf16x4 (&pa)[2][3][2], bf16x4 (&pb)[4][3][2]) {
    if (c < 24) {
      const int ph = c / 12, cc = c - 12 * ph, h = cc / 6, v = cc - 6 * h;
      f32x4 x;
      if (v < 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = ya[4 * h + k][v];
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = xa[4 * h + k][v - 2];
      }
      bf16x4 pc = __builtin_convertvector(x, bf16x4);
      f32x4 r = x - bf16x4_to_f32(pc);
      // (an empty volatile asm on the results: the optimiser otherwise sinks this arithmetic to its first use — the NEXT stage's
      // matrix-core instructions — long before the machine scheduler sees the scheduling barriers)
      asm volatile("" : "+v"(pc), "+v"(r));
      if (v < 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ya[4 * h + k][v] = r[k];
        pa[v][ph][h] = pc;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) xa[4 * h + k][v - 2] = r[k];
        pb[v - 2][ph][h] = pc;
      }
    } else if (c < 30) {
      const int v = c - 24;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f32x4 x;
        if (v < 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] = ya[4 * h + k][v];
          bf16x4 pc = __builtin_convertvector(x, bf16x4);
          asm volatile("" : "+v"(pc));
          pa[v][2][h] = pc;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] = xa[4 * h + k][v - 2];
          bf16x4 pc = __builtin_convertvector(x, bf16x4);
          asm volatile("" : "+v"(pc));
          pb[v - 2][2][h] = pc;
        }
      }
    }
  };
  auto stage_body = [&](int t, const bf16x8 (&ca)[2][3], const bf16x8 (&cb)[4][3], bf16x8 (&na)[2][3], bf16x8 (&nb)[4][3]) {
    // stage t + 1 has landed for this wave once at most NSTAGE - 3 younger stages are outstanding, for every wave after the
    // barrier — which also says that everybody finished reading stage t (in iteration t - 1): its slot may be overwritten
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NIY + NIX) * (NSTAGE - 3)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(kbeg + (t + NSTAGE - 1) * KS, pslot_);
    f32x2 ya[8];
    f32x4 xa[8];
    bf16x4 pa[2][3][2], pb[4][3][2];
    sp_read(nslot_, ya, xa);        // stage t + 1 (past the end of the split: another split's rows or zeros, never used)
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};
#pragma unroll
    for (int sl = 0; sl < 48; ++sl) {
      const int q = sl / 8, i = (sl & 7) >> 2, j = sl & 3;
      if (ACC2 && q < 5) accs[ACC2 ? i : 0][ACC2 ? j : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[i][PA[q]], cb[j][PB[q]], accs[ACC2 ? i : 0][ACC2 ? j : 0], 0, 0, 0);
      else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[i][PA[q]], cb[j][PB[q]], acc[i][j], 0, 0, 0);
      // 30 pieces over 48 slots, five per eight (a piece is ~40 cycles of VALU work, a matrix-core instruction 32): slots 0, 2, 4, 5, 7 of every eight
      constexpr int PRE[8] = {0, 1, 1, 2, 2, 3, 4, 4};
      const int r8 = sl & 7;
      if (r8 == 0 || r8 == 2 || r8 == 4 || r8 == 5 || r8 == 7) split_piece((sl >> 3) * 5 + PRE[r8], ya, xa, pa, pb);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
      for (int i = 0; i < 2; ++i) na[i][pc] = __builtin_shufflevector(pa[i][pc][0], pa[i][pc][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int j = 0; j < 4; ++j) nb[j][pc] = __builtin_shufflevector(pb[j][pc][0], pb[j][pc][1], 0, 1, 2, 3, 4, 5, 6, 7);
    }
    pslot_ = pslot_ + 1 == NSTAGE ? 0 : pslot_ + 1;
    nslot_ = nslot_ + 1 == NSTAGE ? 0 : nslot_ + 1;
  };
  {
    bf16x8 fda[2][3], fdb[4][3];
    int t = 0;
    for (; t + 1 < nsteps; t += 2) {
      stage_body(t, fca, fcb, fda, fdb);
      stage_body(t + 1, fda, fdb, fca, fcb);
    }
    if (t < nsteps) stage_body(t, fca, fcb, fda, fdb);
  }
#else
  for (int t = 0; t < nsteps; ++t) {
    // stage t + 1 has landed for this wave once at most NSTAGE - 3 younger stages are outstanding, for every wave after the
    // barrier — which also says that everybody finished reading stage t (in iteration t - 1): its slot may be overwritten
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NIY + NIX) * (NSTAGE - 3)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(kbeg + (t + NSTAGE - 1) * KS, pslot_);
    f32x2 ya[8];
    f32x4 xa[8];
    bf16x8 fna[2][3], fnb[4][3];
    sp_read(nslot_, ya, xa);        // stage t + 1 (past the end of the split: another split's rows or zeros, never used)
    sp_mfma(fca, fcb);
    sp_split(ya, xa, fna, fnb);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
      for (int i = 0; i < 2; ++i) fca[i][pc] = fna[i][pc];
#pragma unroll
      for (int j = 0; j < 4; ++j) fcb[j][pc] = fnb[j][pc];
    }
    pslot_ = pslot_ + 1 == NSTAGE ? 0 : pslot_ + 1;
    nslot_ = nslot_ + 1 == NSTAGE ? 0 : nslot_ + 1;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may land in LDS after the workgroup has retired
  if constexpr (ACC2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += accs[i][j][e];
  }

  float* out = p.dw + (size_t)ks * p.Co_pad * RS * p.Ci;
  const int ci = ci0 + wn * 128 + 4 * l31;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rr = (e & 3) + 8 * (e >> 2) + 4 * lhi;
      const int co = co0 + wm * 64 + 2 * rr + i;
      const f32x4 v = {acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]};
      *reinterpret_cast<f32x4*>(out + ((size_t)co * RS + tap) * p.Ci + ci) = v;
    }
}

// dW partial slabs [ksplit][Co_pad][RS][Ci] -> OIHW gradient [Co][Ci][R][S] (sum over ksplit, in slab order:
// deterministic).  Threads walk the SLAB layout, four input channels per thread: the ksplit reads (the bulk of the
// traffic) are 16-byte coalesced, the four OIHW stores are RS floats apart (round 1 walked the OIHW order: for 3x3
// kernels every slab read was a 4-byte access at a Ci-float stride, and every index was a 64-bit div/mod).
__global__ __launch_bounds__(256) void wgrad_reduce_unpack_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                  int ksplit, int Co, int Co_pad, int Ci, int RS,
                                                                  int accumulate, long long part_bs, long long dw_bs) {
  const size_t slab = (size_t)Co_pad * RS * Ci;
  part += (size_t)blockIdx.y * part_bs;   // batched K-major GEMM: blockIdx.y = batch item
  dw += (size_t)blockIdx.y * dw_bs;
  const int CV = Ci >> 2;                 // Ci % 4 == 0 (the weight-gradient kernels need Ci % 64 == 0)
  const int total = Co * RS * CV;         // float4 groups of the valid rows, slab order (co, tap, ci)
  for (int g = blockIdx.x * 256 + threadIdx.x; g < total; g += gridDim.x * 256) {
    const int c4 = g % CV;
    const int t = g / CV;
    const int tap = t % RS;
    const int co = t / RS;
    const size_t src = (size_t)g * 4;     // = ((co * RS + tap) * Ci + c4 * 4)
    f32x4 v = *reinterpret_cast<const f32x4*>(part + src);
    // slab order (deterministic); eight slabs' loads in flight per trip instead of one (a chain of ksplit dependent round
    // trips made this kernel 21 us per launch at per-GPU batch 2)
    int k = 1;
    for (; SPLITK_BATCH && k + 7 < ksplit; k += 8) {
      f32x4 t[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) t[q] = *reinterpret_cast<const f32x4*>(part + (size_t)(k + q) * slab + src);
#pragma unroll
      for (int q = 0; q < 8; ++q) v += t[q];
    }
    for (; SPLITK_BATCH && k + 1 < ksplit; k += 2) {
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(part + (size_t)k * slab + src);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(part + (size_t)(k + 1) * slab + src);
      v += t0;
      v += t1;
    }
    for (; k < ksplit; ++k) v += *reinterpret_cast<const f32x4*>(part + (size_t)k * slab + src);
    float* o = dw + ((size_t)co * Ci + c4 * 4) * RS + tap;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[(size_t)j * RS] = accumulate ? o[(size_t)j * RS] + v[j] : v[j];
  }
}

}  // namespace

extern "C" {

static int wgrad_launch(const float* x, int ldx, const float* dy, int lddy, float* dw_oihw,
                        float* scratch, size_t scratch_floats, int N, int H, int W, int Ci, int Ho,
                        int Wo, int Co, int R, int S, int stride, int pad, int dil, int accumulate,
                        int batch, long long x_bs, long long dy_bs, long long out_bs, int arith, hipStream_t stream) {
  if (!x || !dy || !dw_oihw || !scratch || (ldx & 3) || (lddy & 3) || Ci % 64 != 0 || batch < 1 || !arith_ok(arith))
    return SEMSEG_EINVAL;
  const int RS = R * S;
  const int M = N * Ho * Wo;
  if ((size_t)N * H * W * ldx >= 0x7FFF0000ull) return SEMSEG_EINVAL;  // 32-bit element offsets
  // 128 x 128 tiles need Ci % 128 == 0 and Co >= 128.  With few tiles AND few pixels (small per-GPU batch) K
  // would be split 20-50 ways to fill the chip, and the partial slabs (ksplit x |dW|) cost more to write and
  // reduce than the GEMM: 64 x 64 tiles (4x the tiles, 1/4 the slabs) win there and only there.  Measured
  // (scripts/conv_bench.py 2|4|8): 1x1 with 16 tiles -18 % at M = 7200, -12 % at 14400, -4 % at 28800;
  // 3x3 with 36 tiles -9 % at 7200, +2 % at 14400; cls.0 (1152 tiles) +10...18 % everywhere.
  const int t128 = ((Co + 127) / 128) * (Ci / 128) * RS;
  // (a batched launch multiplies the grid by the batch: the 16 GEMMs of a Winograd weight gradient fill the chip with
  // 128 x 128 tiles where a single GEMM of that size would not)
  const bool small_tiles = batch == 1 && ((RS == 1 && t128 <= 16 && M < 32768) || (RS > 1 && t128 <= 36 && M < 8192));
  char dbg_small[16], dbg_sp0[16], dbg_dma[64], dbg_sp64[16], dbg_sp[16];
  const char* small_s = semseg_debug("wgrad_small", dbg_small, sizeof(dbg_small));   // "0": never fall back to 64 x 64 tiles (tests / tuning)
  const bool allow_small = !(small_s && small_s[0] == '0');
  const bool big = (Ci % 128 == 0) && (Co >= 128) && !(allow_small && small_tiles);
  // bf16x3 variant 10: the 128 x 256 kernel (64 x 128 wave tiles, one workgroup per CU); needs 256-channel column tiles
  const char* sp_s0 = semseg_debug("wgrad_sp", dbg_sp0, sizeof(dbg_sp0));
  const int sp_env0 = sp_s0 ? atoi(sp_s0) : WGRAD_SP_POLICY;
  const bool wide = big && arith == SEMSEG_ARITH_BF16X3 && sp_env0 == 10 && Ci % 256 == 0 &&
                    (size_t)N * H * W * ldx * 4 < 0x7FFF0000ull && (size_t)M * lddy * 4 < 0x7FFF0000ull;
  const int TM = big ? 128 : 64, TN = wide ? 256 : (big ? 128 : 64);
  WgradArgs a;
  a.x = x; a.dy = dy; a.dw = scratch; a.ldx = ldx; a.lddy = lddy;
  a.N = N; a.Hin = H; a.Win = W; a.Ho = Ho; a.Wo = Wo; a.Ci = Ci;
  a.tiles_co = (Co + TM - 1) / TM;
  a.tiles_ci = Ci / TN;
  a.Co_pad = a.tiles_co * TM;
  if (lddy < a.Co_pad) return SEMSEG_EINVAL;  // dy rows must be readable (zero padded) up to Co_pad
  a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil; a.M = M;
  a.div_hw = make_fastdiv(Ho * Wo);
  a.div_wo = make_fastdiv(Wo);
  const int tiles = a.tiles_co * a.tiles_ci * RS;
  const int ksteps = (M + 31) / 32;
  a.batch = batch; a.x_bs = x_bs; a.dy_bs = dy_bs;
  // Workgroup order inside a K slice (direct-to-LDS kernel): taps fastest, then the operand dimension with FEWER
  // tiles, so that the ~64 workgroups an XCD holds at a time touch few distinct operand blocks.  Measured per shape
  // with FETCH_SIZE (scripts/wgrad_traffic.py): cls.0 34.2 -> 6.1 GB, aux.0 1.57 -> 0.67 GB,
  // layer4 3x3 1.25 -> 0.81 GB, layer4 1x1 0.9 -> 0.64 GB per launch; 2.79x -> 1.4x the algorithmic bytes over the
  // step's launch mix.
  a.order = RS > 1 ? 1 : (a.tiles_co < a.tiles_ci ? 1 : 0);
  scratch_floats /= batch;   // every batch item owns its own slab set
  // Direct-to-LDS variants of the 128 x 128 kernel (1..5 = K-step / ring depth / residency; 0 = register-staged
  // kernel).  They need byte offsets below 2^31 for both operands.
  // Variant choice.  SEMSEG_DEBUG wgrad_dma = 0..5 forces one variant (read per call: tuning scripts switch inside one
  // process); "a:b:t" = variant a for grids of <= t tiles, b above; unset = WGRAD_DMA_POLICY.  Measured at bs 16
  // (DESIGN.md section 8.2): kernel by kernel the three rings are within 1.5 % of each other (55.2-56.0 ms per
  // step vs 58.3 for the register-staged kernel; KS 32 best on the short 1x1 grids, KS 16 x 3 on cls.0), but inside
  // the step, where the weight gradients share the chip with the main stream, the 2-workgroup-per-CU ring (3) wins
  // for every layer (207.9 ms vs 210.1 for "1:3:32" and 212.2 for 2).
#ifndef WGRAD_DMA_POLICY
#define WGRAD_DMA_POLICY "6"
#endif


  const char* dma_s = semseg_debug("wgrad_dma", dbg_dma, sizeof(dbg_dma));
  if (!dma_s) dma_s = WGRAD_DMA_POLICY;
  int dma_env = atoi(dma_s);
  if (const char* c1 = strchr(dma_s, ':')) {
    const int vb = atoi(c1 + 1);
    const char* c2 = strchr(c1 + 1, ':');
    const int thr = c2 ? atoi(c2 + 1) : 32;
    if (tiles > thr) dma_env = vb;
  }
  const bool dma_ok = (size_t)N * H * W * ldx * 4 < 0x7FFF0000ull && (size_t)M * lddy * 4 < 0x7FFF0000ull;
  const bool sp = big && arith == SEMSEG_ARITH_BF16X3;   // 128 x 128 tiles with split-bf16 products
  // 64 x 64 tiles under bf16x3: the SP instance of the register-staged kernel (SEMSEG_DEBUG wgrad_sp64=0: the exact-fp32 kernel,
  // what rounds 3-4 ran these tiles on)
  const char* sp64_s = semseg_debug("wgrad_sp64", dbg_sp64, sizeof(dbg_sp64));
  const bool sp64 = !big && arith == SEMSEG_ARITH_BF16X3 && !(sp64_s && sp64_s[0] == '0');
  // bf16x3 variants: 8 / 9 = the direct-to-LDS ring with the split at fragment time (4 stages, 2 workgroups per CU / 3
  // stages, 3 per CU); 10 (the policy) = the 128 x 256 kernel with 64 x 128 wave tiles for layers with Ci % 256 == 0 and
  // variant 8 for the rest; 0 = the register-staged SP kernel of round 3.  SEMSEG_DEBUG wgrad_sp = 0 | 8 | 9 | 10 (A/B, tests).
  const char* sp_s = semseg_debug("wgrad_sp", dbg_sp, sizeof(dbg_sp));
  const int sp_env = sp_s ? atoi(sp_s) : WGRAD_SP_POLICY;
  const int sp_dma = wide ? 10 : (sp && dma_ok && (sp_env == 8 || sp_env == 9 || sp_env == 10)) ? (sp_env == 10 ? 8 : sp_env) : 0;
  const int dma = sp ? sp_dma : ((big && dma_ok && dma_env >= 0 && dma_env <= 7) ? dma_env : 0);
  static const int occ_of[11] = {3, 2, 3, 2, 5, 5, 2, 2, 2, 3, 1};
  // Fill whole residency rounds (256 CUs x resident workgroups per CU): among the K splits that give at most two
  // rounds, take the one with the best fill (ties: fewer splits = fewer slabs to write and reduce).
  const int ROUND = 256 * occ_of[dma];
  const int wg1 = tiles * batch;     // workgroups per K slice
  // Accuracy bound on the split (round 5): one fp32 accumulation chain covers at most WGRAD_MAX_CHAIN pixels.  The fill rule
  // alone gives the large weight matrices of a batch-16 step few, long slices (layer4's 1x1 convs: 32-64 tiles -> 4-8 slices of
  // 7 200-14 400 pixels), and the in-situ check at the headline batch measured their weight gradients at 4.3-5.1 x the
  // CPU-fp32 recompute's rms error (criterion 3 x; 1.2-1.9 x at per-GPU batch 2, where the same rule gives 1 800-pixel
  // chains; profiles/r05_insitu_b16.txt).  The error of a chain grows with its length; 2 048 pixels puts every layer of the step
  // inside the criterion (layer4: 1.4-1.8 x afterwards) for 16-32 partial slabs instead of 4-8 on those ~10 launches: 114.7 vs
  // 114.9 ms and 114.2 vs 115.1 ms per batch-16 step in two interleaved A/Bs (bound off / on).  Bounding the chains INSIDE the
  // 128 x 256 kernel instead was measured in two forms and rejected: a second accumulator set in registers (the compiler moves
  // ~300 instructions' worth of accumulators per stage: 116.1 -> 121.8-122.4 ms per step) and a flush of the accumulators into
  // the workgroup's own partial slab every 2 048 pixels (the code between two chunks changes the schedule of the stage loop:
  // 114.7 -> 119.5 ms).  Batched launches (the Winograd-domain weight gradients: 1.3 x in situ at batch 16) keep the fill rule.
#ifndef WGRAD_MAX_CHAIN
#define WGRAD_MAX_CHAIN 2048
#endif
  const int ks_min = batch == 1 ? (M + WGRAD_MAX_CHAIN - 1) / WGRAD_MAX_CHAIN : 1;
  int ksplit = ks_min;
  {
    double best = 0.0;
    const int ks_fill = wg1 > ROUND / 2 ? 8 : (2 * ROUND + wg1 - 1) / wg1;
    const int ks_max = ks_min > 1 ? ks_min + (ROUND + wg1 - 1) / wg1 : ks_fill;   // above the bound: the next full round
    for (int ks = ks_min; ks <= (ks_max > ks_fill ? ks_max : ks_fill); ++ks) {
      const int wgs = wg1 * ks;
      const double eff = (double)wgs / (double)(((wgs + ROUND - 1) / ROUND) * ROUND);
      if (eff > best + 0.02) { best = eff; ksplit = ks; }
    }
  }
  if (ksplit > ksteps / 8) ksplit = ksteps / 8;
  if (ksplit < 1) ksplit = 1;
  const size_t slab = (size_t)a.Co_pad * RS * Ci;
  while (ksplit > 1 && slab * ksplit > scratch_floats) --ksplit;
  if (slab * ksplit > scratch_floats) return SEMSEG_EINVAL;
  a.kper = ((ksteps + ksplit - 1) / ksplit) * 32;
  ksplit = (M + a.kper - 1) / a.kper;
  a.ksplit = ksplit;
  a.dw_bs = (long long)(slab * ksplit);
  const dim3 grid(tiles * ksplit, batch);
  const bool same = stride == 1 && Ho == H && Wo == W;
  const int mode = !same ? 0 : (RS == 1 && pad == 0) ? 1 : 2;
#define LAUNCH_WGRAD(TM_, TN_)                                                        \
  do {                                                                                \
    if (mode == 1) conv_wgrad_kernel<TM_, TN_, 1><<<grid, 256, 0, stream>>>(a);       \
    else if (mode == 2) conv_wgrad_kernel<TM_, TN_, 2><<<grid, 256, 0, stream>>>(a);  \
    else conv_wgrad_kernel<TM_, TN_, 0><<<grid, 256, 0, stream>>>(a);                 \
  } while (0)
#define LAUNCH_WGRAD_DMA(KS_, NST_, OCC_, TL_)                                                          \
  do {                                                                                                  \
    if (mode == 1) conv_wgrad_dma_kernel<1, KS_, NST_, OCC_, TL_><<<grid, 256, 0, stream>>>(a);         \
    else if (mode == 2) conv_wgrad_dma_kernel<2, KS_, NST_, OCC_, TL_><<<grid, 256, 0, stream>>>(a);    \
    else conv_wgrad_dma_kernel<0, KS_, NST_, OCC_, TL_><<<grid, 256, 0, stream>>>(a);                   \
  } while (0)
#define LAUNCH_WGRAD_DMA_SP(NST_, OCC_)                                                                    \
  do {                                                                                                    \
    if (mode == 1) conv_wgrad_dma_kernel<1, 16, NST_, OCC_, false, 3><<<grid, 256, 0, stream>>>(a);       \
    else if (mode == 2) conv_wgrad_dma_kernel<2, 16, NST_, OCC_, false, 3><<<grid, 256, 0, stream>>>(a);  \
    else conv_wgrad_dma_kernel<0, 16, NST_, OCC_, false, 3><<<grid, 256, 0, stream>>>(a);                 \
  } while (0)
  if (sp && sp_dma == 10) {
    // long single 1x1 reductions: the two-accumulator instance (mode 1 = the 1x1 gather; batched launches are the Winograd-domain GEMMs)
    if (mode == 1 && batch == 1 && M > WGRAD_ACC2_MIN_M) conv_wgrad_dma_wide_kernel<1, WIDE_NSTAGE, 1><<<grid, 256, 0, stream>>>(a);
    else if (mode == 1) conv_wgrad_dma_wide_kernel<1, WIDE_NSTAGE><<<grid, 256, 0, stream>>>(a);
    else if (mode == 2) conv_wgrad_dma_wide_kernel<2, WIDE_NSTAGE><<<grid, 256, 0, stream>>>(a);
    else conv_wgrad_dma_wide_kernel<0, WIDE_NSTAGE><<<grid, 256, 0, stream>>>(a);
  } else if (sp && sp_dma == 8) LAUNCH_WGRAD_DMA_SP(4, 2);
  else if (sp && sp_dma == 9) LAUNCH_WGRAD_DMA_SP(3, 3);
  else if (sp) {
    if (mode == 1) conv_wgrad_kernel<128, 128, 1, 3><<<grid, 256, 0, stream>>>(a);
    else if (mode == 2) conv_wgrad_kernel<128, 128, 2, 3><<<grid, 256, 0, stream>>>(a);
    else conv_wgrad_kernel<128, 128, 0, 3><<<grid, 256, 0, stream>>>(a);
  } else if (big && dma == 1) LAUNCH_WGRAD_DMA(32, 2, 2, false);
  else if (big && dma == 2) LAUNCH_WGRAD_DMA(16, 3, 3, false);
  else if (big && dma == 3) LAUNCH_WGRAD_DMA(16, 4, 2, false);
  else if (big && dma == 4) LAUNCH_WGRAD_DMA(16, 2, 4, false);
  else if (big && dma == 5) LAUNCH_WGRAD_DMA(8, 4, 4, false);
  else if (big && dma == 6) LAUNCH_WGRAD_DMA(16, 4, 2, true);    // 3 + two-level accumulation
  else if (big && dma == 7) LAUNCH_WGRAD_DMA(32, 2, 2, true);    // 1 + two-level accumulation
  else if (big) LAUNCH_WGRAD(128, 128);
  else if (sp64) {
    if (mode == 1) conv_wgrad_kernel<64, 64, 1, 3><<<grid, 256, 0, stream>>>(a);
    else if (mode == 2) conv_wgrad_kernel<64, 64, 2, 3><<<grid, 256, 0, stream>>>(a);
    else conv_wgrad_kernel<64, 64, 0, 3><<<grid, 256, 0, stream>>>(a);
  } else LAUNCH_WGRAD(64, 64);
#undef LAUNCH_WGRAD_DMA_SP
#undef LAUNCH_WGRAD_DMA
#undef LAUNCH_WGRAD
  const size_t total = (size_t)Co * Ci * RS / 4;   // one thread per 4 input channels
  wgrad_reduce_unpack_kernel<<<dim3(grid_for(total, 256), batch), 256, 0, stream>>>(
      scratch, dw_oihw, ksplit, Co, a.Co_pad, Ci, RS, accumulate, a.dw_bs, out_bs);
  return semseg_launch_status();
}

int semseg_conv_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw_oihw,
                      float* scratch, size_t scratch_floats, int N, int H, int W, int Ci, int Ho,
                      int Wo, int Co, int R, int S, int stride, int pad, int dil, int accumulate,
                      int arith, hipStream_t stream) {
  return wgrad_launch(x, ldx, dy, lddy, dw_oihw, scratch, scratch_floats, N, H, W, Ci, Ho, Wo, Co, R, S, stride, pad,
                      dil, accumulate, 1, 0, 0, 0, arith, stream);
}

int semseg_gemm_kmajor_batched(const float* x, int ldx, long long x_bs, const float* y, int ldy, long long y_bs,
                               float* out, long long out_bs, float* scratch, size_t scratch_floats, int K, int Ci,
                               int Co, int accumulate, int batch, int arith, hipStream_t stream) {
  if (batch < 1 || batch > 65535) return SEMSEG_EINVAL;
  // every batch item needs at least one partial slab of the scratch arena: a batch that does not fit runs in chunks
  const size_t slab = semseg_conv_wgrad_scratch_floats(Ci, Co, 1, 1);
  const size_t fit = slab ? scratch_floats / slab : 0;
  if (fit < 1) return SEMSEG_EINVAL;
  for (int b0 = 0; b0 < batch; b0 += (int)fit) {
    const int nb = batch - b0 < (int)fit ? batch - b0 : (int)fit;
    const int rc = wgrad_launch(x + (long long)b0 * x_bs, ldx, y + (long long)b0 * y_bs, ldy, out + (long long)b0 * out_bs,
                                scratch, scratch_floats, 1, K, 1, Ci, K, 1, Co, 1, 1, 1, 0, 1, accumulate, nb, x_bs, y_bs,
                                out_bs, arith, stream);
    if (rc != SEMSEG_OK) return rc;
  }
  return SEMSEG_OK;
}

size_t semseg_conv_wgrad_scratch_floats(int Ci, int Co, int R, int S) {
  const bool big = (Ci % 128 == 0) && (Co >= 128);
  const int TM = big ? 128 : 64;
  const size_t Co_pad = (size_t)((Co + TM - 1) / TM) * TM;
  return Co_pad * R * S * Ci;  // one slab; callers size the arena as slabs * desired ksplit
}

}  // extern "C"
