"""Thin tensor-level wrappers over the C ABI (include/semseg_hip.h).  torch is used only for device
memory and the current HIP stream.  Every wrapper raises on a non-zero return code; nothing here
computes on the CPU."""
import torch

from ._lib import lib


class HipError(RuntimeError):
    pass


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ck(rc, name):
    if rc != 0:
        raise HipError("%s failed with code %d" % (name, rc))


def _f32(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda and t.dtype == torch.float32, "expected a CUDA fp32 tensor"


def roundup(x, m):
    return (x + m - 1) // m * m


def conv_out(h, k, s, p, d):
    return (h + 2 * p - d * (k - 1) - 1) // s + 1


# ---------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------
class PackedConv:
    """Device-side packed copies of one OIHW conv weight (fwd + dgrad panels)."""

    def __init__(self, Co, Ci, R, S, device, need_dgrad=True):
        self.Co, self.Ci, self.R, self.S = Co, Ci, R, S
        self.tile_fwd = 128 if Co >= 128 else 64
        self.tile_dgrad = 128 if Ci >= 128 else 64
        self.Co_pad = roundup(Co, self.tile_fwd)
        self.Ci_pad = roundup(Ci, self.tile_dgrad)
        self.Kc_dgrad = roundup(Co, 32)
        self.w_fwd = torch.empty(self.Co_pad * Ci * R * S, dtype=torch.float32, device=device)
        self.w_dgrad = (torch.empty(self.Ci_pad * self.Kc_dgrad * R * S, dtype=torch.float32,
                                    device=device) if need_dgrad else None)

    def pack(self, w_oihw):
        _f32(w_oihw)
        assert w_oihw.is_contiguous() and tuple(w_oihw.shape) == (self.Co, self.Ci, self.R, self.S)
        _ck(lib.semseg_conv_pack_weights(_p(w_oihw), _p(self.w_fwd), _p(self.w_dgrad), self.Co,
                                         self.Ci, self.R, self.S, self.Co_pad, self.Ci_pad,
                                         _stream()), "conv_pack_weights")


NSLOT = 8  # replicas of every fp64 statistics vector (see include/semseg_hip.h)

# Arithmetic of the matrix-core products, a per-launch argument of every GEMM-shaped entry point (include/semseg_hip.h):
ARITH_F32 = 0      # exact fp32 products (v_mfma_f32_32x32x2_f32)
ARITH_BF16X3 = 3   # fp32 operands cut into three bf16 pieces in flight, six cross products, fp32 accumulation

# ---------------------------------------------------------------------------------------------
# Tile width of the forward / data-gradient kernel: a COMMITTED per-shape table.
# Layers with >= 128 output columns can run 128 x 128 or 128 x 64 tiles on the same packed panels.  Which one wins is
# decided by residency-round quantisation and K-split overhead, not by the tile's own efficiency: at bs 16 the 900-tile
# layer3 launches are 5-10 % faster with 128 x 64 tiles while layer4 / cls.0 are not; at per-GPU batch 2-8 nearly every
# layer prefers the narrow tile to a 2-4 way K split.  No static rule covered batch
# 2 / 4 / 8 / 16 and 473 / 713 inputs, so the choice is MEASURED per shape — but offline: scripts/make_tile_table.py
# times both widths on the GPU for the BASELINE.json configurations and writes semseg_amd/tile_table.json, which is
# committed.  Every process (bench, profiles, tests, every rank of a job) therefore runs the same kernels and the same
# K splits (round 2 timed inside the first step: two runs could settle on different widths for a near-tie shape).
# A shape that is not in the table runs the static default (128 wide when the layer has >= 128 output columns).
# SEMSEG_TILE_TUNE=1 (used by that script only) times unknown shapes on first use and adds them to TILE_CHOICE.
# ---------------------------------------------------------------------------------------------
import json as _json
import os as _os
from ._lib import debug as _debug

TILE_TABLE_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tile_table.json")
TILE_TUNE = _os.environ.get("SEMSEG_TILE_TUNE", "0") == "1"
_FORCE_SPLIT_GEMM = _debug("force_split_gemm", "0") == "1"    # measurement: every eligible shape on tile code 2128


def tile_key(kind, N, H, W, Ci, Co, R, S, stride, pad, dil):
    """kind "fwd" | "dgrad"; N, H, W = batch and INPUT size of the convolution (both directions)."""
    return "%s|%d|%d|%d|%d|%d|%d|%d|%d|%d|%d" % (kind, N, H, W, Ci, Co, R, S, stride, pad, dil)


def load_tile_table(path=TILE_TABLE_PATH):
    if not _os.path.exists(path):
        return {}
    with open(path) as f:
        return {k: int(v) for k, v in _json.load(f)["tiles"].items()}


TILE_RUNNER_UP = {}     # key -> best measured implicit-GEMM tile of a shape whose table entry is 2128


def _load_tables():
    """Both committed tables in one dict (and, for every shape that runs the 256 x 128 GEMM kernel, the best measured
    implicit-GEMM tile beside it: what a launch of that shape takes when it is NOT a plain row GEMM — an eval forward with
    the BatchNorm / ReLU / residual folded in, a bias, two fused BatchNorm layers on the data gradient); a code whose column width exceeds what the packed panels of that layer are padded
    for (PackedConv: 64 columns for layers with fewer than 128 output columns) would make the kernel read past the
    panel, so such an entry is dropped here and the shape runs its default."""
    out = {}
    for path in (TILE_TABLE_PATH, TILE_TABLE_SP_PATH):
        for k, v in load_tile_table(path).items():
            f = k.split("|")
            ncols = int(f[5]) if f[0] == "fwd" else int(f[4])      # fwd: Co columns, dgrad: Ci columns
            if v not in TILE_CODES + SPLIT_GEMM_CODES or (ncols < 128 and v % 1000 > 64):
                continue
            if v in SPLIT_GEMM_CODES and not (k.endswith("|sp") and _split_gemm_eligible(k)):
                continue
            out[k] = v
        try:
            with open(path) as f:
                times = _json.load(f).get("ms_per_tile_code", {})
        except OSError:
            times = {}
        for k, v in out.items():
            if v in SPLIT_GEMM_CODES and k in times:
                ig = {int(c): ms for c, ms in times[k].items() if int(c) in TILE_CODES}
                if ig:
                    TILE_RUNNER_UP[k] = min(ig, key=ig.get)
    return out


def plain_tile(tile, key, dflt, plain):
    """Tile code 2128 is for plain row GEMMs; anything else of that shape runs its best measured implicit-GEMM tile."""
    if tile not in SPLIT_GEMM_CODES or plain:
        return tile
    return TILE_RUNNER_UP.get(key, dflt)


def _split_gemm_eligible(key):
    """The shapes tile code 2128 exists for: a 1x1, stride-1, unpadded conv; forward with whole 128-column panels of output
    channels, data gradient with whole 128-column panels of input channels and a reduction (output channels, padded to 32) of
    at most 1024 (that kernel runs one fp32 accumulation chain; conv_igemm.hip: dgrad_impl has the measured error)."""
    f = key.split("|")
    if f[6:10] != ["1", "1", "1", "0"]:
        return False
    if f[0] == "fwd":
        return int(f[5]) % 128 == 0
    return f[0] == "dgrad" and int(f[4]) % 128 == 0 and roundup(int(f[5]), 32) <= 1024


TILE_CODES = (128, 64, 1128, 1064)   # 128x128, 128x64, 64x128, 64x64 (rows x columns; include/semseg_hip.h)
# 2128: semseg_conv_fwd runs the 256 x 128 bf16x3 GEMM kernel (gemm_bf16split.hip, the Winograd path's GEMM) with the statistics
# epilogue instead of the implicit-GEMM kernel — bf16x3 table only, eligible shapes only (the library falls back to 128)
TILE_SPLIT_GEMM = 2128
TILE_SPLIT_GEMM_128 = 3128     # the same kernel with 128 x 128 tiles (four waves): grids of a small per-GPU batch (round 5)
SPLIT_GEMM_CODES = (TILE_SPLIT_GEMM, TILE_SPLIT_GEMM_128)
_FORCE_SPLIT_CODE = TILE_SPLIT_GEMM_128 if _debug("force_split_gemm_bm") == "128" else TILE_SPLIT_GEMM   # with _FORCE_SPLIT_GEMM
TILE_TIMES = {}   # key -> {tile code: ms per launch}, filled in tuning mode only
# tile shapes measured with the SEMSEG_ARITH_BF16X3 instances of the kernels: keys suffixed "|sp"
TILE_TABLE_SP_PATH = _debug("tile_table_sp") or TILE_TABLE_PATH.replace("tile_table.json", "tile_table_sp.json")
TILE_CHOICE = _load_tables()


def _tuned_tile(key, dflt, device, out_floats, launch, arith=ARITH_F32):
    """Table lookup; with SEMSEG_TILE_TUNE=1 an unknown shape is timed once (launch(tile, out_tensor) -> return code of a
    side-effect-free launch of this shape into a scratch output on the operands' device)."""
    codes = TILE_CODES if dflt == 128 else (64, 1064)
    if arith == ARITH_BF16X3:
        key = key + "|sp"
        if dflt == 128 and _split_gemm_eligible(key):
            codes = codes + SPLIT_GEMM_CODES
            if _FORCE_SPLIT_GEMM:
                return _FORCE_SPLIT_CODE
    t = TILE_CHOICE.get(key)
    if t is not None:
        return t if (dflt == 128 or t % 1000 == 64) else dflt
    if not TILE_TUNE:
        return dflt
    tmp = torch.empty(out_floats, dtype=torch.float32, device=device)
    torch.cuda.synchronize(device)
    best = {}
    for rnd in range(3):
        for tile in codes:
            _ck(launch(tile, tmp), "tile tuning")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                _ck(launch(tile, tmp), "tile tuning")
            e1.record()
            e1.synchronize()
            best[tile] = min(best.get(tile, 1e30), e0.elapsed_time(e1))
    # the default shape (128 x 128, or 128 x 64 for layers with < 128 output columns) unless another wins by >= 3 %; the
    # 256 x 128 GEMM kernel (code 2128) by >= 2 %: it never needs the K-split epilogue launch, and taking it at 2-3 % measured
    # alone is worth 0.4 ms of the batch-16 step (interleaved A/B of the two tables, three rounds)
    t = min(codes, key=lambda c: best[c])
    if best[t] >= (0.98 if t in SPLIT_GEMM_CODES else 0.97) * best[dflt]:
        t = dflt
    TILE_CHOICE[key] = t
    TILE_TIMES[key] = {c: round(best[c] / 3, 4) for c in codes}
    return t


def _scr(scratch):
    return (None, 0) if scratch is None else (scratch.data_ptr(), scratch.numel())


TILE_COUNTERS = 4096      # SEMSEG_TILE_COUNTERS of include/semseg_hip.h
_CNT = {}
FUSED_SPLIT = _debug("fused_split", "1") != "0"      # 0: split tiles reduced by the separate launch of rounds 1-5 (A/B)


def _cnt(scratch):
    """The tile counters that go with a split-K scratch arena (same owner: the stream the arena belongs to): zeroed once, left zero
    by every launch.  Allocated on first use of the arena — in the launch-by-launch steps in front of a recorded step plan."""
    if scratch is None or not FUSED_SPLIT:
        return None
    key = (scratch.device.index, scratch.data_ptr())
    t = _CNT.get(key)
    if t is None:
        t = _CNT[key] = torch.zeros(TILE_COUNTERS, dtype=torch.int32, device=scratch.device)
    return t.data_ptr()


def conv_pack_weights_multi(descs_dev, starts_dev, nconv, total_blocks):
    _ck(lib.semseg_conv_pack_weights_multi(_p(descs_dev), _p(starts_dev), nconv, total_blocks, _stream()),
        "conv_pack_weights_multi")


def conv_fwd(x, ldx, pk, y, ldy, N, H, W, stride, pad, dil, bias=None, add=None, ldadd=0, stats=None,
             nslot=1, scratch=None, scale=None, relu=False, arith=ARITH_F32):
    Ho = conv_out(H, pk.R, stride, pad, dil)
    Wo = conv_out(W, pk.S, stride, pad, dil)
    ldt = roundup(pk.Co, 4)
    tstats = torch.zeros(NSLOT * 2 * pk.Co, dtype=torch.float64, device=x.device) if TILE_TUNE else None   # timed with statistics
    tile = _tuned_tile(tile_key("fwd", N, H, W, pk.Ci, pk.Co, pk.R, pk.S, stride, pad, dil), pk.tile_fwd, x.device,
                       N * Ho * Wo * ldt,
                       lambda t, out: lib.semseg_conv_fwd(
                           _p(x), ldx, _p(pk.w_fwd), _p(out), ldt, N, H, W, pk.Ci, Ho, Wo, pk.Co, pk.R, pk.S, stride,
                           pad, dil, None, None, 0, None, 0, _p(tstats), NSLOT, t, arith, *_scr(scratch), _cnt(scratch), _stream()), arith)
    tile = plain_tile(tile, tile_key("fwd", N, H, W, pk.Ci, pk.Co, pk.R, pk.S, stride, pad, dil) + "|sp", pk.tile_fwd,
                      bias is None and scale is None and not relu and add is None)
    _ck(lib.semseg_conv_fwd(_p(x), ldx, _p(pk.w_fwd), _p(y), ldy, N, H, W, pk.Ci, Ho, Wo, pk.Co,
                            pk.R, pk.S, stride, pad, dil, _p(bias), _p(scale), int(relu), _p(add), ldadd,
                            _p(stats), nslot, tile, arith, *_scr(scratch), _cnt(scratch), _stream()), "conv_fwd")
    return Ho, Wo


def chosen_tile(kind, pk, N, H, W, stride, pad, dil, ld_in, ld_out, arith=ARITH_F32, plain=True):
    """Tile width the (already measured) shape runs with; kind "fwd" | "dgrad"; plain = the launch is a plain row GEMM (nothing
    folded into a forward's epilogue, at most one fused BatchNorm layer on a data gradient).  For kernel-family labels."""
    dflt = pk.tile_fwd if kind == "fwd" else pk.tile_dgrad
    key = tile_key(kind, N, H, W, pk.Ci, pk.Co, pk.R, pk.S, stride, pad, dil) + ("|sp" if arith == ARITH_BF16X3 else "")
    if _FORCE_SPLIT_GEMM and dflt == 128 and arith == ARITH_BF16X3 and _split_gemm_eligible(key):
        return plain_tile(_FORCE_SPLIT_CODE, key, dflt, plain)
    t = TILE_CHOICE.get(key, dflt)
    return plain_tile(t if (dflt == 128 or t % 1000 == 64) else dflt, key, dflt, plain)


def _dgrad_tile(dy, lddy, pk, lddx, N, H, W, Ho, Wo, stride, pad, dil, scratch, arith=ARITH_F32):
    ldt = roundup(pk.Ci, 4)
    key = tile_key("dgrad", N, H, W, pk.Ci, pk.Co, pk.R, pk.S, stride, pad, dil)
    if TILE_TUNE and arith == ARITH_BF16X3 and pk.tile_dgrad == 128 and _split_gemm_eligible(key) and \
            (key + "|sp") not in TILE_CHOICE:
        # shapes the 256 x 128 GEMM kernel can take are timed in the form the engine mostly runs them in: the fused
        # BatchNorm-backward reduction with the bit mask and a residual gradient added (their epilogue is most of the launch)
        M, C = N * H * W, pk.Ci
        dev = dy.device
        ybn = torch.randn(M, C, device=dev)
        addt = torch.randn(M, C, device=dev)
        bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (M, C // 32), dtype=torch.int32, device=dev)
        mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        sums = torch.zeros(NSLOT * 2 * C, dtype=torch.float64, device=dev)
        return _tuned_tile(key, pk.tile_dgrad, dev, M * ldt,
                           lambda t, out: lib.semseg_conv_dgrad_bnreduce(
                               _p(dy), lddy, _p(pk.w_dgrad), _p(out), ldt, N, H, W, pk.Ci, Ho, Wo, pk.Co, pk.R, pk.S, stride,
                               pad, dil, _p(addt), C, t, 1, None, 0, _p(bits), C // 32, _p(ybn), C, _p(mean), _p(invstd),
                               _p(sums), None, 0, None, None, None, NSLOT, arith, *_scr(scratch), _cnt(scratch), _stream()), arith)
    return _tuned_tile(key, pk.tile_dgrad, dy.device,
                       N * H * W * ldt,
                       lambda t, out: lib.semseg_conv_dgrad(
                           _p(dy), lddy, _p(pk.w_dgrad), _p(out), ldt, N, H, W, pk.Ci, Ho, Wo, pk.Co, pk.R, pk.S,
                           stride, pad, dil, None, 0, t, arith, *_scr(scratch), _cnt(scratch), _stream()), arith)


def conv_dgrad(dy, lddy, pk, dx, lddx, N, H, W, stride, pad, dil, add=None, ldadd=0, scratch=None, arith=ARITH_F32):
    Ho = conv_out(H, pk.R, stride, pad, dil)
    Wo = conv_out(W, pk.S, stride, pad, dil)
    tile = _dgrad_tile(dy, lddy, pk, lddx, N, H, W, Ho, Wo, stride, pad, dil, scratch, arith)
    _ck(lib.semseg_conv_dgrad(_p(dy), lddy, _p(pk.w_dgrad), _p(dx), lddx, N, H, W, pk.Ci, Ho, Wo,
                              pk.Co, pk.R, pk.S, stride, pad, dil, _p(add), ldadd, tile, arith,
                              *_scr(scratch), _cnt(scratch), _stream()), "conv_dgrad")


def conv_dgrad_bnreduce(dy, lddy, pk, dx, lddx, N, H, W, stride, pad, dil, act, ldact, bns, nslot, add=None, ldadd=0,
                        scratch=None, arith=ARITH_F32, relu_bits=None):
    """Data gradient fused with the BatchNorm-backward reduction of the layer(s) that produced the conv's input.
    bns: 1 or 2 tuples (y, ldy, mean, invstd, sums[nslot][2*Ci]).  relu_bits: bn_apply's bit mask, read instead of act."""
    Ho = conv_out(H, pk.R, stride, pad, dil)
    Wo = conv_out(W, pk.S, stride, pad, dil)
    b0 = bns[0]
    b1 = bns[1] if len(bns) > 1 else (None, 0, None, None, None)
    tile = plain_tile(_dgrad_tile(dy, lddy, pk, lddx, N, H, W, Ho, Wo, stride, pad, dil, scratch, arith),
                      tile_key("dgrad", N, H, W, pk.Ci, pk.Co, pk.R, pk.S, stride, pad, dil) + "|sp", pk.tile_dgrad, len(bns) <= 1)
    _ck(lib.semseg_conv_dgrad_bnreduce(_p(dy), lddy, _p(pk.w_dgrad), _p(dx), lddx, N, H, W, pk.Ci, Ho, Wo, pk.Co, pk.R,
                                       pk.S, stride, pad, dil, _p(add), ldadd, tile, len(bns), _p(act), ldact,
                                       _p(relu_bits), 0 if relu_bits is None else relu_bits.shape[-1],
                                       _p(b0[0]), b0[1], _p(b0[2]), _p(b0[3]), _p(b0[4]),
                                       _p(b1[0]), b1[1], _p(b1[2]), _p(b1[3]), _p(b1[4]), nslot, arith, *_scr(scratch),
                                       _cnt(scratch), _stream()), "conv_dgrad_bnreduce")


# ---------------------------------------------------------------------------------------------
# Winograd F(2x2, 3x3): 3x3, stride 1, padding = dilation (include/semseg_hip.h)
# ---------------------------------------------------------------------------------------------
class WinoConv:
    """Transformed filter panels of one 3x3 conv: U_fwd [16][Co_pad][Ci], U_dgrad [16][Ci_pad][Kc] (Kc = roundup(Co, 32))."""

    def __init__(self, Co, Ci, device, need_dgrad=True):
        self.Co, self.Ci = Co, Ci
        self.Co_pad = roundup(Co, 128 if Co >= 128 else 64)
        self.Ci_pad = roundup(Ci, 128 if Ci >= 128 else 64)
        self.Kc = roundup(Co, 32)
        self.U_fwd = torch.empty(16 * self.Co_pad * Ci, dtype=torch.float32, device=device)
        self.U_dgrad = (torch.empty(16 * self.Ci_pad * self.Kc, dtype=torch.float32, device=device)
                        if need_dgrad else None)

    def transform(self, w_oihw):
        _f32(w_oihw)
        assert w_oihw.is_contiguous() and tuple(w_oihw.shape) == (self.Co, self.Ci, 3, 3)
        _ck(lib.semseg_wino_filter_transform(_p(w_oihw), _p(self.U_fwd), self.Co, self.Ci, self.Co_pad, self.Ci, 0,
                                             _stream()), "wino_filter_transform")
        if self.U_dgrad is not None:
            _ck(lib.semseg_wino_filter_transform(_p(w_oihw), _p(self.U_dgrad), self.Co, self.Ci, self.Ci_pad, self.Kc, 1,
                                                 _stream()), "wino_filter_transform")


def wino_filter_transform_multi(descs_dev, starts_dev, npanels, total_blocks):
    _ck(lib.semseg_wino_filter_transform_multi(_p(descs_dev), _p(starts_dev), npanels, total_blocks, _stream()),
        "wino_filter_transform_multi")


def wino_tiles(N, H, W, dil):
    t = int(lib.semseg_wino_tiles(N, H, W, dil))
    if t < 0:
        raise HipError("wino_tiles: bad geometry")
    return t


def wino_input_transform(src, lds, V, N, H, W, C, dil):
    _ck(lib.semseg_wino_input_transform(_p(src), lds, _p(V), N, H, W, C, dil, _stream()), "wino_input_transform")


def wino_dy_transform_wgrad(dy, lddy, Yh, ldo, N, H, W, C, dil):
    _ck(lib.semseg_wino_dy_transform_wgrad(_p(dy), lddy, _p(Yh), ldo, N, H, W, C, dil, _stream()),
        "wino_dy_transform_wgrad")


def wino_output_transform(M, ldm, y, ldy, N, H, W, C, dil, add=None, ldadd=0, stats=None, nslot=1, scale=None,
                          shift=None, relu=False):
    _ck(lib.semseg_wino_output_transform(_p(M), ldm, _p(y), ldy, _p(add), ldadd, _p(stats), nslot, _p(scale), _p(shift),
                                         int(relu), N, H, W, C, dil, _stream()), "wino_output_transform")


def wino_output_transform_bnreduce(M, ldm, y, ldy, N, H, W, C, dil, act, ldact, ybn, ldybn, mean, invstd, sums, nslot,
                                   add=None, ldadd=0, relu_bits=None):
    _ck(lib.semseg_wino_output_transform_bnreduce(_p(M), ldm, _p(y), ldy, _p(add), ldadd, _p(act), ldact, _p(relu_bits),
                                                  0 if relu_bits is None else relu_bits.shape[-1], _p(ybn), ldybn,
                                                  _p(mean), _p(invstd), _p(sums), nslot, N, H, W, C, dil, _stream()),
        "wino_output_transform_bnreduce")


def wino_filter_grad(dU, dw, Co, Ci, accumulate=False):
    _ck(lib.semseg_wino_filter_grad(_p(dU), _p(dw), Co, Ci, int(accumulate), _stream()), "wino_filter_grad")


def wino_conv_fwd(x, ldx, wc, y, ldy, N, H, W, dil, V, Mbuf, stats=None, nslot=1, add=None, ldadd=0, arith=ARITH_F32):
    """y[N,H,W,ldy] = conv3x3(x, w; stride 1, padding = dilation = dil).  V [16*T*Ci] receives the transformed input
    (kept by the caller for the weight gradient), Mbuf [>= 16*T*Co] is scratch."""
    T = wino_tiles(N, H, W, dil)
    Ci, Co = wc.Ci, wc.Co
    wino_input_transform(x, ldx, V, N, H, W, Ci, dil)
    gemm_rows_batched(V, Ci, T * Ci, wc.U_fwd, wc.Co_pad * Ci, Mbuf, Co, T * Co, T, Ci, Co, 16, arith=arith)
    wino_output_transform(Mbuf, Co, y, ldy, N, H, W, Co, dil, add=add, ldadd=ldadd, stats=stats, nslot=nslot)
    return T


def wino_conv_dgrad(dy, lddy, wc, dx, lddx, N, H, W, dil, Vdy, Mbuf, add=None, ldadd=0, arith=ARITH_F32):
    """dx[N,H,W,lddx] (= | + add) = the data gradient of the same convolution: a 3x3 convolution of dy with the
    transposed filter, taps rotated by 180 degrees.  dy must be readable (zero) up to Kc = roundup(Co, 32) channels."""
    T = wino_tiles(N, H, W, dil)
    Ci, Kc = wc.Ci, wc.Kc
    assert lddy >= Kc
    wino_input_transform(dy, lddy, Vdy, N, H, W, Kc, dil)
    gemm_rows_batched(Vdy, Kc, T * Kc, wc.U_dgrad, wc.Ci_pad * Kc, Mbuf, Ci, T * Ci, T, Kc, Ci, 16, arith=arith)
    wino_output_transform(Mbuf, Ci, dx, lddx, N, H, W, Ci, dil, add=add, ldadd=ldadd)
    return T


def wino_conv_wgrad(V, dy, lddy, wc, dw, N, H, W, dil, Yh, dU, scratch, accumulate=False, arith=ARITH_F32):
    """dw[Co][Ci][3][3] from the kept transformed input V and dy.  Yh [16*T*roundup(Co,128)] zero-initialised scratch
    (its padding columns must stay zero), dU [16*Co*Ci] scratch."""
    T = wino_tiles(N, H, W, dil)
    Ci, Co = wc.Ci, wc.Co
    ldo = roundup(Co, 128)
    wino_dy_transform_wgrad(dy, lddy, Yh, ldo, N, H, W, roundup(Co, 4), dil)
    gemm_kmajor_batched(V, Ci, T * Ci, Yh, ldo, T * ldo, dU, Co * Ci, scratch, T, Ci, Co, 16, arith=arith)
    wino_filter_grad(dU, dw, Co, Ci, accumulate)


def wgrad_scratch_floats(Ci, Co, R, S):
    return int(lib.semseg_conv_wgrad_scratch_floats(Ci, Co, R, S))


def conv_wgrad(x, ldx, dy, lddy, dw, scratch, N, H, W, Ci, Co, R, S, stride, pad, dil,
               accumulate=False, arith=ARITH_F32):
    Ho = conv_out(H, R, stride, pad, dil)
    Wo = conv_out(W, S, stride, pad, dil)
    _ck(lib.semseg_conv_wgrad(_p(x), ldx, _p(dy), lddy, _p(dw), _p(scratch), scratch.numel(), N, H,
                              W, Ci, Ho, Wo, Co, R, S, stride, pad, dil, int(accumulate), arith, _stream()),
        "conv_wgrad")


def stem_conv_fwd(x_nchw, w, y, N, H, W):
    _ck(lib.semseg_stem_conv_fwd(_p(x_nchw), _p(w), _p(y), N, H, W, w.shape[0], _stream()),
        "stem_conv_fwd")


def stem_conv_wgrad(x_nchw, dy, dw, N, H, W, accumulate=False, scratch=None):
    if scratch is None:
        scratch = torch.empty(int(lib.semseg_stem_wgrad_scratch_floats(N, H, W)), dtype=torch.float32, device=dy.device)
    _ck(lib.semseg_stem_conv_wgrad(_p(x_nchw), _p(dy), _p(dw), N, H, W, dw.shape[0], int(accumulate),
                                   *_scr(scratch), _stream()), "stem_conv_wgrad")


# ---------------------------------------------------------------------------------------------
# batch norm family
# ---------------------------------------------------------------------------------------------
def channel_stats(x, ldx, stats, M, C, nslot=1):
    _ck(lib.semseg_channel_stats(_p(x), ldx, _p(stats), nslot, M, C, _stream()), "channel_stats")


def bn_combine(stats, nslot, C, dst=None):
    _ck(lib.semseg_bn_combine(_p(stats), nslot, C, _p(dst), _stream()), "bn_combine")


def bn_finalize(stats, count, gamma, beta, rm, rv, nbt, momentum, eps, mean, invstd, scale, shift, C,
                nslot=1):
    _ck(lib.semseg_bn_finalize(_p(stats), nslot, float(count), _p(gamma), _p(beta), _p(rm), _p(rv), _p(nbt),
                               momentum, eps, _p(mean), _p(invstd), _p(scale), _p(shift), C,
                               _stream()), "bn_finalize")


def bn_eval_params(gamma, beta, rm, rv, eps, scale, shift, C):
    _ck(lib.semseg_bn_eval_params(_p(gamma), _p(beta), _p(rm), _p(rv), eps, _p(scale), _p(shift), C,
                                  _stream()), "bn_eval_params")


def bn_apply(y, ldy, scale, shift, out, ldout, M, C, HW, relu, y2=None, ldy2=0, scale2=None,
             shift2=None, res=None, ldres=0, dropmask=None, relu_bits=None):
    """relu_bits: optional int32 [M][C / 32] tensor receiving the ReLU mask as bits (include/semseg_hip.h)."""
    _ck(lib.semseg_bn_apply(_p(y), ldy, _p(scale), _p(shift), _p(y2), ldy2, _p(scale2), _p(shift2),
                            _p(res), ldres, _p(dropmask), _p(out), ldout, M, C, HW, int(relu),
                            _p(relu_bits), 0 if relu_bits is None else relu_bits.shape[-1], _stream()), "bn_apply")


def bn_apply_train(y, ldy, stats, nslot, count, gamma, beta, rm, rv, nbt, momentum, eps, mean, invstd, out, ldout, M, C, HW,
                   relu, res=None, ldres=0, dropmask=None, relu_bits=None):
    """semseg_bn_finalize + semseg_bn_apply in one launch (small grids; include/semseg_hip.h)."""
    _ck(lib.semseg_bn_apply_train(_p(y), ldy, _p(stats), nslot, float(count), _p(gamma), _p(beta), _p(rm), _p(rv), _p(nbt),
                                  momentum, eps, _p(mean), _p(invstd), _p(res), ldres, _p(dropmask), _p(out), ldout, M, C, HW,
                                  int(relu), _p(relu_bits), 0 if relu_bits is None else relu_bits.shape[-1], _stream()),
        "bn_apply_train")


def bn_bwd_apply_train(g, ldg, y, ldy, mean, invstd, gamma, sums, nslot, count, param_scale, dgamma, dbeta, dy, lddy, M, C):
    """semseg_bn_param_grads + semseg_bn_bwd_apply in one launch (small grids)."""
    _ck(lib.semseg_bn_bwd_apply_train(_p(g), ldg, _p(y), ldy, _p(mean), _p(invstd), _p(gamma), _p(sums), nslot, float(count),
                                      float(param_scale), _p(dgamma), _p(dbeta), _p(dy), lddy, M, C, _stream()),
        "bn_bwd_apply_train")


def bn_bwd_reduce(dout, lddout, out, ldout, dropmask, HW, y, ldy, mean, invstd, g, ldg, sums, M, C,
                  nslot=1):
    _ck(lib.semseg_bn_bwd_reduce(_p(dout), lddout, _p(out), ldout, _p(dropmask), HW, _p(y), ldy,
                                 _p(mean), _p(invstd), _p(g), ldg, _p(sums), nslot, M, C, _stream()),
        "bn_bwd_reduce")


def bn_bwd_apply(g, ldg, y, ldy, mean, invstd, gamma, sums, count, dy, lddy, M, C):
    _ck(lib.semseg_bn_bwd_apply(_p(g), ldg, _p(y), ldy, _p(mean), _p(invstd), _p(gamma), _p(sums),
                                float(count), _p(dy), lddy, M, C, _stream()), "bn_bwd_apply")


def bn_param_grads(sums, dgamma, dbeta, C, accumulate=False, nslot=1, folded=None):
    _ck(lib.semseg_bn_param_grads(_p(sums), nslot, _p(dgamma), _p(dbeta), C, int(accumulate), _p(folded),
                                  _stream()), "bn_param_grads")


# ---------------------------------------------------------------------------------------------
# spatial ops
# ---------------------------------------------------------------------------------------------
def maxpool_fwd(x, y, idx, N, H, W, C):
    _ck(lib.semseg_maxpool3x3s2_fwd(_p(x), _p(y), _p(idx), N, H, W, C, _stream()), "maxpool_fwd")


def maxpool_bwd(dy, idx, dx, N, H, W, C):
    _ck(lib.semseg_maxpool3x3s2_bwd(_p(dy), _p(idx), _p(dx), N, H, W, C, _stream()), "maxpool_bwd")


_bins_cache = {}


def _bins(bins):
    import ctypes
    key = tuple(bins)
    if key not in _bins_cache:
        _bins_cache[key] = (ctypes.c_int * len(key))(*key)
    return ctypes.addressof(_bins_cache[key])


def adaptive_avgpool_scratch_floats(bins, N, H, C):
    return int(lib.semseg_adaptive_avgpool_scratch_floats(_bins(bins), len(bins), N, H, C))


def adaptive_avgpool_fwd(x, ldx, y, bins, N, H, W, C, scratch=None):
    _ck(lib.semseg_adaptive_avgpool_fwd(_p(x), ldx, _p(y), _bins(bins), len(bins), N, H, W, C,
                                        *_scr(scratch), _stream()), "adaptive_avgpool_fwd")


def adaptive_avgpool_bwd(base, ldbase, dpool, dx, lddx, bins, N, H, W, C):
    _ck(lib.semseg_adaptive_avgpool_bwd(_p(base), ldbase, _p(dpool), _p(dx), lddx, _bins(bins),
                                        len(bins), N, H, W, C, _stream()), "adaptive_avgpool_bwd")


def bilinear_fwd(x, ldx, y, ldy, N, Hi, Wi, Ho, Wo, C):
    _ck(lib.semseg_bilinear_fwd(_p(x), ldx, _p(y), ldy, N, Hi, Wi, Ho, Wo, C, _stream()),
        "bilinear_fwd")


def bilinear_bwd(dy, lddy, dx, lddx, N, Hi, Wi, Ho, Wo, C):
    _ck(lib.semseg_bilinear_bwd(_p(dy), lddy, _p(dx), lddx, N, Hi, Wi, Ho, Wo, C, _stream()),
        "bilinear_bwd")


def bilinear_nhwc_to_nchw(x, ldx, y, N, Hi, Wi, Ho, Wo, C):
    _ck(lib.semseg_bilinear_nhwc_to_nchw(_p(x), ldx, _p(y), N, Hi, Wi, Ho, Wo, C, _stream()),
        "bilinear_nhwc_to_nchw")


# ---------------------------------------------------------------------------------------------
# fused CE head, SGD, psamask
# ---------------------------------------------------------------------------------------------
def ce_head_fwd(scores, ld, label, lse, pred, acc2, loss, N, h, w, H, W, C, ignore_index):
    assert label.dtype == torch.int64
    _ck(lib.semseg_ce_head_fwd(_p(scores), ld, _p(label), _p(lse), _p(pred), _p(acc2), _p(loss), N, h,
                               w, H, W, C, ignore_index, _stream()), "ce_head_fwd")


def ce_head_bwd(scores, ld, label, lse, acc2, grad_loss, grad_mul, dscores, lddz, accumulate, N, h, w,
                H, W, C, ignore_index, scratch=None):
    _ck(lib.semseg_ce_head_bwd(_p(scores), ld, _p(label), _p(lse), _p(acc2), _p(grad_loss),
                               float(grad_mul), _p(dscores), lddz, int(accumulate), N, h, w, H, W, C,
                               ignore_index, *_scr(scratch), _stream()), "ce_head_bwd")


def dropout2d_mask(mask, p, seed, offset, offset_dev=None):
    """offset_dev: optional device uint64 added to `offset` on the device (the per-step counter of a replayed step plan)."""
    _ck(lib.semseg_dropout2d_mask(_p(mask), mask.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset),
                                  _p(offset_dev), _stream()), "dropout2d_mask")


def stream_wait(waiter, signaller):
    """torch.cuda.Stream.wait_stream through the C ABI (semseg_stream_wait_stream), so that a recorded step plan holds it."""
    _ck(lib.semseg_stream_wait_stream(waiter.cuda_stream, signaller.cuda_stream), "stream_wait_stream")


def zero_(t):
    """hipMemsetAsync on the current stream (no torch fill kernel on the path)."""
    assert t.is_contiguous()
    _ck(lib.semseg_memset_zero(_p(t), t.numel() * t.element_size(), _stream()), "memset_zero")


def label_check(label, C, ignore_index):
    """Number of targets that are neither ignore_index nor a class id (synchronises)."""
    assert label.dtype == torch.int64 and label.is_cuda and label.is_contiguous()
    bad = torch.zeros(1, dtype=torch.int64, device=label.device)
    _ck(lib.semseg_label_check(_p(label), label.numel(), C, ignore_index, _p(bad), _stream()), "label_check")
    return int(bad.item())


def sgd_step(w, g, mom, n, lr, momentum, weight_decay, grad_scale=1.0, first_step=False, lr_dev=None, skip_dev=None):
    _ck(lib.semseg_sgd_step(_p(w), _p(g), _p(mom), n, float(lr), _p(lr_dev), momentum, weight_decay,
                            grad_scale, int(first_step), _p(skip_dev), _stream()), "sgd_step")


def psamask_forward(psa_type, inp, out, num, fH, fW, mH, mW, hH, hW):
    _f32(inp, out)
    _ck(lib.semseg_psamask_forward(psa_type, _p(inp), _p(out), num, fH, fW, mH, mW, hH, hW, _stream()),
        "psamask_forward")


def psamask_backward(psa_type, gout, gin, num, fH, fW, mH, mW, hH, hW):
    _f32(gout, gin)
    _ck(lib.semseg_psamask_backward(psa_type, _p(gout), _p(gin), num, fH, fW, mH, mW, hH, hW,
                                    _stream()), "psamask_backward")


# ---------------------------------------------------------------------------------------------
# PSA head (pixel-major layout) + raw GEMM entry points used for the point-affinity contraction
# ---------------------------------------------------------------------------------------------
def psamask_nhwc_forward(psa_type, mask, ldm, aff, lda, N, H, W, mH, mW):
    _ck(lib.semseg_psamask_nhwc_forward(psa_type, _p(mask), ldm, _p(aff), lda, N, H, W, mH, mW, _stream()),
        "psamask_nhwc_forward")


def psamask_nhwc_backward(psa_type, daff, lda, dmask, ldm, N, H, W, mH, mW, prezeroed=False):
    _ck(lib.semseg_psamask_nhwc_backward(psa_type, _p(daff), lda, _p(dmask), ldm, N, H, W, mH, mW, int(prezeroed),
                                         _stream()), "psamask_nhwc_backward")


def softmax_rows_fwd(x, ldx, y, ldy, rows, P, alpha, softmax):
    _ck(lib.semseg_softmax_rows_fwd(_p(x), ldx, _p(y), ldy, rows, P, float(alpha), int(softmax),
                                    _stream()), "softmax_rows_fwd")


def softmax_rows_bwd(y, ldy, dy, lddy, dx, lddx, rows, P, alpha, softmax):
    _ck(lib.semseg_softmax_rows_bwd(_p(y), ldy, _p(dy), lddy, _p(dx), lddx, rows, P, float(alpha),
                                    int(softmax), _stream()), "softmax_rows_bwd")


def transpose_batched(inp, ldi, bsi, out, ldo, bso, batch, R, C):
    _ck(lib.semseg_transpose_batched(_p(inp), ldi, bsi, _p(out), ldo, bso, batch, R, C, _stream()),
        "transpose_batched")


def gemm_rows(a_ptr, lda, bt_ptr, c_ptr, ldc, M, K, Nout, add_ptr=None, ldadd=0, arith=ARITH_F32):
    """C[M][Nout] = A[M][K] * B, B given K-contiguous as bt[Nout_pad][K] (K % 32 == 0): the 1x1
    implicit-GEMM kernel on raw device pointers."""
    tile = 128 if Nout >= 128 else 64
    _ck(lib.semseg_conv_fwd(a_ptr, lda, bt_ptr, c_ptr, ldc, 1, M, 1, K, M, 1, Nout, 1, 1, 1, 0, 1, None,
                            None, 0, add_ptr, ldadd, None, 1, tile, arith, None, 0, None, _stream()), "gemm_rows")


def gemm_kmajor(x_ptr, ldx, y_ptr, ldy, out_ptr, scratch, K, Ci, Co, accumulate=False, arith=ARITH_F32):
    """out[Co][Ci] (=|+=) sum_k y[k][co] * x[k][ci] — the weight-gradient kernel as a K-major GEMM."""
    _ck(lib.semseg_conv_wgrad(x_ptr, ldx, y_ptr, ldy, out_ptr, _p(scratch), scratch.numel(), 1, K, 1, Ci,
                              K, 1, Co, 1, 1, 1, 0, 1, int(accumulate), arith, _stream()), "gemm_kmajor")


def gemm_rows_batched(a, lda, a_bs, bt, bt_bs, c, ldc, c_bs, M, K, Nout, batch, arith=ARITH_F32):
    """C[b][M][Nout] = A[b][M][K] * Bt[b][Nout_pad][K]^T in ONE launch (tensors or raw pointers; strides in floats)."""
    _ck(lib.semseg_gemm_rows_batched(_ptr(a), lda, a_bs, _ptr(bt), bt_bs, _ptr(c), ldc, c_bs, M, K, Nout, batch, arith,
                                     _stream()), "gemm_rows_batched")


def gemm_rows_batched_bf16split(a, lda, a_bs, bt, bt_bs, c, ldc, c_bs, M, K, Nout, batch, nsplit=3, bk=16):
    """gemm_rows_batched on the split-bf16 kernel of csrc/gemm_bf16split.hip: nsplit 3 = ARITH_BF16X3 (what the engine runs
    for the Winograd GEMMs); nsplit 2 (two pieces, three products) is a measurement only and fails the per-op parity
    criteria.  Bt rows must be readable up to roundup(Nout, 128)."""
    _ck(lib.semseg_gemm_rows_batched_bf16split(_ptr(a), lda, a_bs, _ptr(bt), bt_bs, _ptr(c), ldc, c_bs, M, K, Nout,
                                               batch, nsplit, bk, _stream()), "gemm_rows_batched_bf16split")


def gemm_kmajor_batched(x, ldx, x_bs, y, ldy, y_bs, out, out_bs, scratch, K, Ci, Co, batch, accumulate=False,
                        arith=ARITH_F32):
    """out[b][Co][Ci] (=|+=) sum_k y[b][k][co] * x[b][k][ci] in ONE launch of the weight-gradient kernel."""
    _ck(lib.semseg_gemm_kmajor_batched(_ptr(x), ldx, x_bs, _ptr(y), ldy, y_bs, _ptr(out), out_bs, _p(scratch),
                                       scratch.numel(), K, Ci, Co, int(accumulate), batch, arith, _stream()),
        "gemm_kmajor_batched")


def _ptr(t):
    return t if isinstance(t, int) else t.data_ptr()


# ---------------------------------------------------------------------------------------------
# test-time pipeline (tool/test.py:122-204) kept on the device
# ---------------------------------------------------------------------------------------------
def resize_linear_hwc(src, Hs, Ws, dst, Hd, Wd, C):
    _ck(lib.semseg_resize_linear_hwc(_p(src), Hs, Ws, _p(dst), Hd, Wd, C, _stream()), "resize_linear_hwc")


def crop_normalize_flip(img, H, W, origins_dev, K, ch, cw, mean, std, out):
    import ctypes
    m = (ctypes.c_float * 3)(*mean)
    s_ = (ctypes.c_float * 3)(*std)
    _ck(lib.semseg_crop_normalize_flip(_p(img), H, W, _p(origins_dev), K, ch, cw, ctypes.addressof(m),
                                       ctypes.addressof(s_), _p(out), _stream()), "crop_normalize_flip")


def softmax_flip_accumulate(logits, pos_dev, K, C, ch, cw, canvas, count, Hc, Wc):
    _ck(lib.semseg_softmax_flip_accumulate(_p(logits), _p(pos_dev), K, C, ch, cw, _p(canvas), _p(count), Hc,
                                           Wc, _stream()), "softmax_flip_accumulate")


def resize_accumulate_chw(canvas, count, Hc, Wc, y0, x0, Hs, Ws, dst, Hd, Wd, C, weight):
    _ck(lib.semseg_resize_accumulate_chw(_p(canvas), _p(count), Hc, Wc, y0, x0, Hs, Ws, _p(dst), Hd, Wd, C,
                                         float(weight), _stream()), "resize_accumulate_chw")


def argmax_chw(prob, out, C, H, W):
    _ck(lib.semseg_argmax_chw(_p(prob), _p(out), C, H, W, _stream()), "argmax_chw")


# ---------------------------------------------------------------------------------------------
# training input pipeline (util/transform.py) on the device; planned by semseg_amd/transform.py
# ---------------------------------------------------------------------------------------------
def augment_round(ops_dev, n_samples, max_pixels):
    """ops_dev: uint8 CUDA tensor holding n_samples semseg_aug_op descriptors."""
    assert ops_dev.is_cuda and ops_dev.dtype == torch.uint8
    assert ops_dev.numel() >= n_samples * lib.semseg_aug_op_size()
    _ck(lib.semseg_augment_round(_p(ops_dev), n_samples, max_pixels, _stream()), "augment_round")
