"""GPU parity of every C-ABI kernel against a plain torch CPU reference of the same op (fp64 where a
tighter reference is useful).  Error metric: max|a-b| / max|b| (relative to the tensor's scale), the
same normalisation BASELINE.json uses for the logits."""
import os
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def relerr(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def relu_bits(act_nhwc):
    """The ReLU mask of an NHWC activation as semseg_bn_apply writes it: int32 [M][C / 32], bit (c & 31) of word [m][c >> 5]."""
    C = act_nhwc.shape[-1]
    m = (act_nhwc.reshape(-1, C // 32, 32) > 0).to(torch.int64)
    w = (m << torch.arange(32, device=m.device)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # N, H, W, Ci, Co, k, stride, pad, dil
    (2, 17, 19, 64, 64, 3, 1, 1, 1),
    (2, 15, 15, 64, 128, 1, 1, 0, 1),
    (1, 20, 20, 128, 256, 3, 1, 2, 2),
    (1, 23, 23, 256, 128, 3, 1, 4, 4),
    (2, 21, 21, 128, 128, 3, 2, 1, 1),
    (2, 21, 21, 256, 512, 1, 2, 0, 1),
    (2, 9, 9, 512, 150, 1, 1, 0, 1),
    (3, 7, 7, 2048, 512, 1, 1, 0, 1),
    (1, 12, 12, 512, 64, 3, 1, 1, 1),
    (7, 60, 60, 64, 256, 3, 1, 2, 2),     # 197 m-tiles x 2 = 394 tiles: stream-K tail path
]


@pytest.mark.parametrize("case", [
    # N, H, W, Ci, Co, k, stride, pad, dil, nbn, with_add, split
    (2, 20, 20, 128, 256, 3, 1, 2, 2, 1, False, False),
    (2, 20, 20, 128, 256, 3, 1, 2, 2, 1, False, True),      # K-split tiles: fused reduction in the split-K epilogue
    (3, 15, 15, 256, 64, 1, 1, 0, 1, 2, True, False),       # bn3 + downsample BN sharing g, residual add, M = 675
    (3, 15, 15, 256, 64, 1, 1, 0, 1, 2, True, True),
    (7, 60, 60, 128, 64, 1, 1, 0, 1, 1, True, True),        # 197 m-tiles: full tiles AND a split tail in one launch
    (2, 21, 21, 64, 128, 3, 2, 1, 1, 1, False, False),      # 64-wide tile, strided conv
])
@pytest.mark.parametrize("mask", ["act", "bits"])
def test_conv_dgrad_fused_bn_backward_reduce(case, mask, report):
    """semseg_conv_dgrad_bnreduce: dx = (dgrad (+ add)) * (act > 0) and fp64 [sum g, sum g * xhat] per channel for one
    or two BatchNorm layers, against the unfused formula in fp64; the ReLU mask given as the activation or as bits."""
    from semseg_amd import ops
    N, H, W, Ci, Co, k, s, p, d, nbn, with_add, split = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    Ho, Wo = ops.conv_out(H, k, s, p, d), ops.conv_out(W, k, s, p, d)
    w = torch.randn(Co, Ci, k, k, generator=g) * (1.0 / (Co * k * k) ** 0.5)
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    act = torch.relu(torch.randn(N, Ci, H, W, generator=g))
    add = torch.randn(N, Ci, H, W, generator=g) if with_add else None
    ys = [torch.randn(N, Ci, H, W, generator=g) * 2 + 0.5 for _ in range(nbn)]
    means = [torch.randn(Ci, generator=g) for _ in range(nbn)]
    invs = [torch.rand(Ci, generator=g) + 0.5 for _ in range(nbn)]
    dx64 = torch.nn.grad.conv2d_input((N, Ci, H, W), w.double(), dy.double(), stride=s, padding=p, dilation=d)
    if with_add:
        dx64 = dx64 + add.double()
    g64 = dx64 * (act > 0)
    pk = ops.PackedConv(Co, Ci, k, k, DEV)
    pk.pack(w.to(DEV))
    ldy = ops.roundup(Co, 128)
    dyb = torch.zeros(N, Ho, Wo, ldy, device=DEV)
    dyb[..., :Co] = nhwc(dy).to(DEV)
    dxb = nhwc(add).to(DEV).contiguous() if with_add else torch.full((N, H, W, Ci), float("nan"), device=DEV)
    NS = ops.NSLOT
    sums = [torch.zeros(NS * 2 * Ci, dtype=torch.float64, device=DEV) for _ in range(nbn)]
    bns = [(nhwc(ys[b]).to(DEV).contiguous(), Ci, means[b].to(DEV), invs[b].to(DEV), sums[b]) for b in range(nbn)]
    scratch = torch.empty(16 * 1024 * 1024, device=DEV) if split else None
    actb = nhwc(act).to(DEV).contiguous()
    ops.conv_dgrad_bnreduce(dyb, ldy, pk, dxb, Ci, N, H, W, s, p, d, actb if mask == "act" else None, Ci, bns, NS,
                            add=dxb if with_add else None, ldadd=Ci, scratch=scratch,
                            relu_bits=relu_bits(actb) if mask == "bits" else None)
    e_g = relerr(nchw(dxb), g64)
    errs = []
    for b in range(nbn):
        tot = sums[b].view(NS, 2 * Ci).sum(0).cpu()
        xh = (ys[b].double() - means[b].double().view(1, -1, 1, 1)) * invs[b].double().view(1, -1, 1, 1)
        errs.append(relerr(tot[:Ci], g64.sum((0, 2, 3))))
        errs.append(relerr(tot[Ci:], (g64 * xh).sum((0, 2, 3))))
    report("fused dgrad + BN-backward reduce %s mask=%s: g %.2e sums %s" % (case, mask, e_g, " ".join("%.1e" % e for e in errs)))
    assert e_g < 2e-5 and max(errs) < 2e-5


@pytest.mark.parametrize("case", [
    # N, H, W, Ci, Co
    (2, 30, 30, 256, 1024),     # layer3 conv3 at a small batch; M = 1800: a partial 256-row tile at the end
    (3, 15, 15, 1024, 256),     # layer3 conv1: K = 1024
    (1, 17, 19, 64, 128),       # one column tile, M = 323
    (2, 20, 20, 512, 384),      # three column tiles
])
@pytest.mark.parametrize("code", [2128, 3128])
def test_conv_fwd_1x1_on_the_split_gemm_kernel(case, code, report, monkeypatch):
    """Tile codes 2128 (256 x 128 tiles) / 3128 (128 x 128 tiles, round 5) of semseg_conv_fwd: the forward of a 1x1 stride-1 conv under SEMSEG_ARITH_BF16X3 on the 256 x 128 GEMM
    kernel of gemm_bf16split.hip with the fp64 statistics epilogue; operands in wider buffers, replicated statistics slots.
    Same bounds as the implicit-GEMM kernel; an ineligible call (bias in the epilogue) silently takes the 128 x 128 tile."""
    from semseg_amd import ops
    N, H, W, Ci, Co = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 1, 1, generator=g) * (1.0 / Ci ** 0.5)
    y64 = F.conv2d(x.double(), w.double())
    pk = ops.PackedConv(Co, Ci, 1, 1, DEV)
    pk.pack(w.to(DEV))
    ldx, ldy = Ci + 64, Co + 128
    xb = torch.randn(N, H, W, ldx, device=DEV)
    xb[..., 32:32 + Ci] = nhwc(x).to(DEV)
    NS = ops.NSLOT
    monkeypatch.setattr(ops, "_FORCE_SPLIT_GEMM", True)
    monkeypatch.setattr(ops, "_FORCE_SPLIT_CODE", code)
    assert ops.chosen_tile("fwd", pk, N, H, W, 1, 0, 1, ldx, ldy, ops.ARITH_BF16X3) == code
    yb = torch.full((N, H, W, ldy), float("nan"), device=DEV)
    st = torch.zeros(NS * 2 * Co, dtype=torch.float64, device=DEV)
    ops.conv_fwd(xb[..., 32:], ldx, pk, yb, ldy, N, H, W, 1, 0, 1, stats=st, nslot=NS, arith=ops.ARITH_BF16X3)
    assert torch.isnan(yb[..., Co:]).all()
    e_f = relerr(nchw(yb[..., :Co]), y64)
    tot = st.view(NS, 2, Co).sum(0).cpu()
    e_s = max(relerr(tot[0], y64.sum((0, 2, 3))), relerr(tot[1], (y64 * y64).sum((0, 2, 3))))
    # with a bias the same call falls back to the implicit-GEMM kernel
    bias = torch.randn(Co, generator=g)
    yb2 = torch.empty(N, H, W, Co, device=DEV)
    ops.conv_fwd(xb[..., 32:], ldx, pk, yb2, Co, N, H, W, 1, 0, 1, bias=bias.to(DEV), arith=ops.ARITH_BF16X3)
    e_b = relerr(nchw(yb2), y64 + bias.double().view(1, -1, 1, 1))
    report("1x1 forward on the bf16x3 GEMM kernel, tile code %d %s: y %.2e stats %.2e (fallback with bias %.2e)" % (code, case, e_f, e_s, e_b))
    assert max(e_f, e_b) < 2e-5 and e_s < 1e-5


@pytest.mark.parametrize("mask", ["none", "act", "bits"])
@pytest.mark.parametrize("case", [
    # N, H, W, Ci, Co, with_add
    (2, 30, 30, 1024, 256, True),      # layer3 conv1's data gradient at a small batch: K = 256, eight column tiles, M = 1800
    (3, 15, 15, 256, 64, False),       # K = 64, M = 675
    (1, 17, 19, 128, 512, True),       # K = 512, one column tile, M = 323 (a partial 256-row tile)
    (2, 16, 16, 256, 1024, False),     # layer3 conv3's data gradient: K = 1024, the longest chain the kernel is given
])
@pytest.mark.parametrize("code", [2128, 3128])
def test_conv_dgrad_1x1_on_the_split_gemm_kernel(case, mask, code, report, monkeypatch):
    """Tile codes 2128 / 3128 of semseg_conv_dgrad / semseg_conv_dgrad_bnreduce: the data gradient of a 1x1 stride-1 conv under
    SEMSEG_ARITH_BF16X3 on the 256 x 128 GEMM kernel, plain (+ add) and with the fused BatchNorm-backward reduction of one layer
    (mask none / activation / bits), against fp64; same bounds as the implicit-GEMM kernel."""
    from semseg_amd import ops
    N, H, W, Ci, Co, with_add = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    w = torch.randn(Co, Ci, 1, 1, generator=g) * (1.0 / Co ** 0.5)
    dy = torch.randn(N, Co, H, W, generator=g)
    act = torch.relu(torch.randn(N, Ci, H, W, generator=g))
    add = torch.randn(N, Ci, H, W, generator=g) if with_add else None
    ybn = torch.randn(N, Ci, H, W, generator=g) * 2 + 0.5
    mean, inv = torch.randn(Ci, generator=g), torch.rand(Ci, generator=g) + 0.5
    dx64 = torch.nn.grad.conv2d_input((N, Ci, H, W), w.double(), dy.double())
    if with_add:
        dx64 = dx64 + add.double()
    g64 = dx64 * (act > 0) if mask != "none" else dx64
    pk = ops.PackedConv(Co, Ci, 1, 1, DEV)
    pk.pack(w.to(DEV))
    monkeypatch.setattr(ops, "_FORCE_SPLIT_GEMM", True)
    monkeypatch.setattr(ops, "_FORCE_SPLIT_CODE", code)
    assert ops.chosen_tile("dgrad", pk, N, H, W, 1, 0, 1, 0, 0, ops.ARITH_BF16X3) == code
    ldy = ops.roundup(Co, 128)
    dyb = torch.zeros(N, H, W, ldy, device=DEV)
    dyb[..., :Co] = nhwc(dy).to(DEV)
    NS = ops.NSLOT
    # plain data gradient (+ add)
    dxp = nhwc(add).to(DEV).contiguous() if with_add else torch.full((N, H, W, Ci), float("nan"), device=DEV)
    ops.conv_dgrad(dyb, ldy, pk, dxp, Ci, N, H, W, 1, 0, 1, add=dxp if with_add else None, ldadd=Ci, arith=ops.ARITH_BF16X3)
    e_p = relerr(nchw(dxp), dx64)
    # fused reduction
    dxb = nhwc(add).to(DEV).contiguous() if with_add else torch.full((N, H, W, Ci), float("nan"), device=DEV)
    sums = torch.zeros(NS * 2 * Ci, dtype=torch.float64, device=DEV)
    actb = nhwc(act).to(DEV).contiguous()
    bns = [(nhwc(ybn).to(DEV).contiguous(), Ci, mean.to(DEV), inv.to(DEV), sums)]
    ops.conv_dgrad_bnreduce(dyb, ldy, pk, dxb, Ci, N, H, W, 1, 0, 1, actb if mask == "act" else None, Ci, bns, NS,
                            add=dxb if with_add else None, ldadd=Ci, arith=ops.ARITH_BF16X3,
                            relu_bits=relu_bits(actb) if mask == "bits" else None)
    e_g = relerr(nchw(dxb), g64)
    tot = sums.view(NS, 2 * Ci).sum(0).cpu()
    xh = (ybn.double() - mean.double().view(1, -1, 1, 1)) * inv.double().view(1, -1, 1, 1)
    e_s = max(relerr(tot[:Ci], g64.sum((0, 2, 3))), relerr(tot[Ci:], (g64 * xh).sum((0, 2, 3))))
    report("1x1 data gradient on the bf16x3 GEMM kernel, tile code %d %s mask=%s: plain %.2e fused g %.2e sums %.2e"
           % (code, case, mask, e_p, e_g, e_s))
    assert max(e_p, e_g, e_s) < 2e-5


WGRAD_BIG_CASES = [
    # N, H, W, Ci, Co, k, stride, pad, dil  (Ci % 128 == 0, Co >= 128: the 128 x 128 weight-gradient tile)
    (2, 13, 13, 128, 128, 3, 1, 2, 2),     # "same" dilated 3x3: linear gather with border taps out of range
    (2, 17, 17, 128, 256, 3, 2, 1, 1),     # strided 3x3: generic gather
    (5, 9, 9, 256, 150, 1, 1, 0, 1),       # 1x1, Co not a tile multiple (dy zero padded), M = 405 (K tail)
    (2, 21, 21, 256, 512, 1, 2, 0, 1),     # strided 1x1 (downsample)
    (3, 30, 30, 128, 128, 3, 1, 1, 1),     # M = 2700: several K splits
]


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("case", WGRAD_BIG_CASES)
def test_wgrad_128_tile_variants(case, variant, report, monkeypatch):
    """Every variant of the 128 x 128 weight-gradient kernel (0 register-staged, 1-5 direct-to-LDS rings with
    out-of-range buffer offsets for padding taps / the K tail) against fp64; operands live in wider buffers."""
    from semseg_amd import ops
    monkeypatch.setenv("SEMSEG_DEBUG", "wgrad_small=0,wgrad_dma=%s" % variant)
    N, H, W, Ci, Co, k, s, p, d = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(N, Ci, H, W, generator=g)
    Ho, Wo = ops.conv_out(H, k, s, p, d), ops.conv_out(W, k, s, p, d)
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, k, k), dy.double(), stride=s, padding=p, dilation=d)
    ldx = Ci + 64
    xb = torch.randn(N, H, W, ldx, device=DEV)
    xb[..., 32:32 + Ci] = nhwc(x).to(DEV)
    ldy = ops.roundup(Co, 128) + 128
    dyb = torch.zeros(N, Ho, Wo, ldy, device=DEV)
    dyb[..., :Co] = nhwc(dy).to(DEV)
    scratch = torch.empty(ops.wgrad_scratch_floats(Ci, Co, k, k) * 4, device=DEV)
    dw = torch.full((Co, Ci, k, k), float("nan"), device=DEV)
    ops.conv_wgrad(xb[..., 32:], ldx, dyb, ldy, dw, scratch, N, H, W, Ci, Co, k, k, s, p, d)
    e = relerr(dw, ref)
    report("wgrad 128-tile variant %d %s: %.2e" % (variant, case, e))
    assert e < 2e-5


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, report):
    from semseg_amd import ops
    N, H, W, Ci, Co, k, s, p, d = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * (1.0 / (Ci * k * k) ** 0.5)
    bias = torch.randn(Co, generator=g)
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, s, p, d)
    Ho, Wo = y64.shape[2:]
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    y64.backward(dy.double())

    pk = ops.PackedConv(Co, Ci, k, k, DEV)
    wd = w.to(DEV)
    pk.pack(wd)
    # activations live in a wider buffer to exercise the channel stride
    ldx = Ci + 64
    xb = torch.zeros(N, H, W, ldx, device=DEV)
    xb[..., 32:32 + Ci] = nhwc(x).to(DEV)
    xv = xb[..., 32:]
    ldy = ops.roundup(Co, 128) + 128
    yb = torch.zeros(N, Ho, Wo, ldy, device=DEV)
    stats = torch.zeros(2 * Co, dtype=torch.float64, device=DEV)
    ops.conv_fwd(xv, ldx, pk, yb, ldy, N, H, W, s, p, d, stats=stats)
    y = nchw(yb[..., :Co])
    e_f = relerr(y, y64)
    e_s1 = relerr(stats[:Co], y64.sum((0, 2, 3)))
    e_s2 = relerr(stats[Co:], (y64 * y64).sum((0, 2, 3)))
    assert float(yb[..., Co:].abs().max()) == 0.0

    # bias + residual add epilogue
    addb = torch.randn(N, Ho, Wo, Co, generator=g).to(DEV)
    yb2 = torch.zeros(N, Ho, Wo, Co, device=DEV)
    ops.conv_fwd(xv, ldx, pk, yb2, Co, N, H, W, s, p, d, bias=bias.to(DEV), add=addb, ldadd=Co)
    ref2 = y64.detach() + bias.double().view(1, -1, 1, 1) + nchw(addb.cpu()).double()
    e_b = relerr(nchw(yb2), ref2)

    # dgrad: dy lives in a zero-padded buffer (ld >= roundup(Co, 128))
    dyb = torch.zeros(N, Ho, Wo, ldy, device=DEV)
    dyb[..., :Co] = nhwc(dy).to(DEV)
    dxb = torch.zeros(N, H, W, Ci, device=DEV)
    ops.conv_dgrad(dyb, ldy, pk, dxb, Ci, N, H, W, s, p, d)
    e_d = relerr(nchw(dxb), x64.grad)

    # wgrad
    dw = torch.empty(Co, Ci, k, k, device=DEV)
    scratch = torch.empty(ops.wgrad_scratch_floats(Ci, Co, k, k) * 4, device=DEV)
    ops.conv_wgrad(xv, ldx, dyb, ldy, dw, scratch, N, H, W, Ci, Co, k, k, s, p, d)
    e_w = relerr(dw, w64.grad)
    # split-K path (what small per-GPU batches take) with replicated statistics slots
    NS = ops.NSLOT
    st8 = torch.zeros(NS * 2 * Co, dtype=torch.float64, device=DEV)
    yb3 = torch.zeros(N, Ho, Wo, ldy, device=DEV)
    ops.conv_fwd(xv, ldx, pk, yb3, ldy, N, H, W, s, p, d, stats=st8, nslot=NS, scratch=scratch)
    ops.bn_combine(st8, NS, Co)
    e_sf = relerr(nchw(yb3[..., :Co]), y64)
    e_ss = max(relerr(st8[:Co], y64.sum((0, 2, 3))), relerr(st8[Co:2 * Co], (y64 * y64).sum((0, 2, 3))))
    yb4 = torch.zeros(N, Ho, Wo, Co, device=DEV)
    ops.conv_fwd(xv, ldx, pk, yb4, Co, N, H, W, s, p, d, bias=bias.to(DEV), add=addb, ldadd=Co, scratch=scratch)
    e_sb = relerr(nchw(yb4), ref2)
    dxb2 = torch.randn(N, H, W, Ci, device=DEV)
    base = dxb2.clone()
    ops.conv_dgrad(dyb, ldy, pk, dxb2, Ci, N, H, W, s, p, d, add=dxb2, ldadd=Ci, scratch=scratch)
    e_sd = relerr(nchw(dxb2 - base), x64.grad)
    assert float(yb3[..., Co:].abs().max()) == 0.0
    report("conv %s fwd %.2e stats %.2e/%.2e bias+add %.2e dgrad %.2e wgrad %.2e | split-K fwd %.2e stats %.2e "
           "bias+add %.2e dgrad+add %.2e" % (case, e_f, e_s1, e_s2, e_b, e_d, e_w, e_sf, e_ss, e_sb, e_sd))
    assert max(e_f, e_b, e_d, e_w, e_sf, e_sb, e_sd) < 2e-5
    assert max(e_s1, e_s2, e_ss) < 1e-5


def test_stem(report):
    from semseg_amd import ops
    N, H, W = 2, 41, 33
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.2
    w64 = w.double().requires_grad_(True)
    y64 = F.conv2d(x.double(), w64, None, 2, 1)
    Ho, Wo = y64.shape[2:]
    dy = torch.randn(N, 64, Ho, Wo, generator=g)
    y64.backward(dy.double())
    xd, wd = x.to(DEV), w.to(DEV)
    y = torch.empty(N, Ho, Wo, 64, device=DEV)
    ops.stem_conv_fwd(xd, wd, y, N, H, W)
    dw = torch.empty(64, 3, 3, 3, device=DEV)
    ops.stem_conv_wgrad(xd, nhwc(dy).to(DEV), dw, N, H, W)
    e1, e2 = relerr(nchw(y), y64), relerr(dw, w64.grad)
    report("stem fwd %.2e wgrad %.2e" % (e1, e2))
    assert e1 < 1e-5 and e2 < 1e-4


@pytest.mark.parametrize("C,HW,mode", [(64, 13, "plain"), (256, 9, "res"), (1024, 5, "ds"),
                                       (512, 6, "drop"), (2048, 3, "plain")])
def test_bn_train_fwd_bwd(C, HW, mode, report):
    from semseg_amd import ops
    N = 3
    M = N * HW * HW
    g = torch.Generator().manual_seed(C + HW)
    y = (torch.randn(N, C, HW, HW, generator=g) * 3 + 1.5).double().requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).double().requires_grad_(True)
    beta = torch.randn(C, generator=g).double().requires_grad_(True)
    rm, rv = torch.randn(C, generator=g).double(), (torch.rand(C, generator=g) + 0.5).double()
    rm0, rv0 = rm.clone(), rv.clone()
    out = F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5)
    extra = {}
    if mode == "res":
        res = torch.randn(N, C, HW, HW, generator=g).double().requires_grad_(True)
        out = out + res
        extra["res"] = res
    if mode == "ds":
        y2 = torch.randn(N, C, HW, HW, generator=g).double().requires_grad_(True)
        g2 = (torch.rand(C, generator=g) + 0.5).double().requires_grad_(True)
        b2 = torch.randn(C, generator=g).double().requires_grad_(True)
        out = out + F.batch_norm(y2, None, None, g2, b2, True, 0.1, 1e-5)
    out = F.relu(out)
    dm = None
    if mode == "drop":
        dm = (torch.rand(N, C, generator=g) > 0.3).double() / 0.7
        out = out * dm.view(N, C, 1, 1)
    dout = torch.randn(N, C, HW, HW, generator=g).double()
    out.backward(dout)

    f = lambda t: t.detach().float().to(DEV)
    yd = nhwc(f(y))
    NS = ops.NSLOT
    stats = torch.zeros(NS * 2 * C, dtype=torch.float64, device=DEV)
    ops.channel_stats(yd, C, stats, M, C, nslot=NS)
    mean, invstd, scale, shift = (torch.empty(C, device=DEV) for _ in range(4))
    rmd, rvd = f(rm0), f(rv0)
    nbt = torch.zeros((), dtype=torch.int64, device=DEV)
    ops.bn_finalize(stats, M, f(gamma), f(beta), rmd, rvd, nbt, 0.1, 1e-5, mean, invstd, scale, shift, C, nslot=NS)
    outd = torch.empty(N, HW, HW, C, device=DEV)
    kw = {}
    if mode == "res":
        kw = dict(res=nhwc(f(extra["res"])), ldres=C)
    if mode == "ds":
        y2d = nhwc(f(y2))
        st2 = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
        ops.channel_stats(y2d, C, st2, M, C)
        mean2, invstd2, scale2, shift2 = (torch.empty(C, device=DEV) for _ in range(4))
        ops.bn_finalize(st2, M, f(g2), f(b2), None, None, None, 0.1, 1e-5, mean2, invstd2, scale2, shift2, C)
        kw = dict(y2=y2d, ldy2=C, scale2=scale2, shift2=shift2)
    dmd = f(dm) if dm is not None else None
    ops.bn_apply(yd, C, scale, shift, outd, C, M, C, HW * HW, True, dropmask=dmd, **kw)
    e_out = relerr(nchw(outd), out)
    e_rm, e_rv = relerr(rmd, rm), relerr(rvd, rv)
    assert int(nbt.item()) == 1
    # backward
    doutd = nhwc(f(dout))
    sums = torch.zeros(NS * 2 * C, dtype=torch.float64, device=DEV)
    gd = torch.empty(N, HW, HW, C, device=DEV)
    ops.bn_bwd_reduce(doutd, C, outd, C, dmd, HW * HW, yd, C, mean, invstd, gd, C, sums, M, C, nslot=NS)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_param_grads(sums, dg, db, C, nslot=NS)   # combines the replicas into slot 0
    dyd = torch.empty(N, HW, HW, C, device=DEV)
    ops.bn_bwd_apply(gd, C, yd, C, mean, invstd, f(gamma), sums, M, dyd, C, M, C)
    e_dy, e_dg, e_db = relerr(nchw(dyd), y.grad), relerr(dg, gamma.grad), relerr(db, beta.grad)
    errs = [e_out, e_rm, e_rv, e_dy, e_dg, e_db]
    if mode == "res":
        errs.append(relerr(nchw(gd), extra["res"].grad))
    if mode == "ds":
        s2 = torch.zeros(2 * C, dtype=torch.float64, device=DEV)
        ops.bn_bwd_reduce(gd, C, None, 0, None, HW * HW, y2d, C, mean2, invstd2, None, 0, s2, M, C)
        dy2 = torch.empty(N, HW, HW, C, device=DEV)
        ops.bn_bwd_apply(gd, C, y2d, C, mean2, invstd2, f(g2), s2, M, dy2, C, M, C)
        errs.append(relerr(nchw(dy2), y2.grad))
    report("bn C=%d HW=%d %s: %s" % (C, HW, mode, " ".join("%.2e" % e for e in errs)))
    assert max(errs) < 2e-5


@pytest.mark.parametrize("C,HW,N,ns", [(64, 13, 3, 1), (256, 60, 2, 1), (1024, 9, 2, 2), (2048, 5, 2, 2), (512, 6, 1, 1), (256, 60, 2, 8), (512, 30, 2, 8), (64, 119, 2, 8)])
@pytest.mark.parametrize("mode", ["plain", "res", "drop"])
def test_bn_fused_train_launches_equal_the_separate_ones(C, HW, N, ns, mode, report):
    """semseg_bn_apply_train == semseg_bn_finalize + semseg_bn_apply and semseg_bn_bwd_apply_train == semseg_bn_param_grads +
    semseg_bn_bwd_apply, BIT for bit (same expressions in the same order): activation, ReLU bits, mean / invstd, running
    statistics, num_batches_tracked, dy, dgamma, dbeta; param_scale (the SyncBN 1 / world form) scales the parameter gradients."""
    from semseg_amd import ops
    M = N * HW * HW
    g = torch.Generator().manual_seed(C + HW + ns)
    yd = (torch.randn(N, HW, HW, C, generator=g) * 2 + 0.7).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    rm0, rv0 = torch.randn(C, generator=g).to(DEV), (torch.rand(C, generator=g) + 0.5).to(DEV)
    res = torch.randn(N, HW, HW, C, generator=g).to(DEV) if mode == "res" else None
    dm = ((torch.rand(N, C, generator=g) > 0.3).float() / 0.7).to(DEV) if mode == "drop" else None
    stats = torch.zeros(ns * 2 * C, dtype=torch.float64, device=DEV)
    ops.channel_stats(yd, C, stats, M, C, nslot=ns)
    # separate launches
    mean, invstd, scale, shift = (torch.empty(C, device=DEV) for _ in range(4))
    rm, rv, nbt = rm0.clone(), rv0.clone(), torch.zeros((), dtype=torch.int64, device=DEV)
    ops.bn_finalize(stats, M, gamma, beta, rm, rv, nbt, 0.1, 1e-5, mean, invstd, scale, shift, C, nslot=ns)
    out, bits = torch.empty_like(yd), torch.zeros(M, C // 32, dtype=torch.int32, device=DEV)
    ops.bn_apply(yd, C, scale, shift, out, C, M, C, HW * HW, True, res=res, ldres=C, dropmask=dm,
                 relu_bits=None if dm is not None else bits)
    # fused launch
    mean2, invstd2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    rm2, rv2, nbt2 = rm0.clone(), rv0.clone(), torch.zeros((), dtype=torch.int64, device=DEV)
    out2, bits2 = torch.empty_like(yd), torch.zeros(M, C // 32, dtype=torch.int32, device=DEV)
    ops.bn_apply_train(yd, C, stats, ns, M, gamma, beta, rm2, rv2, nbt2, 0.1, 1e-5, mean2, invstd2, out2, C, M, C, HW * HW, True,
                       res=res, ldres=C, dropmask=dm, relu_bits=None if dm is not None else bits2)
    torch.cuda.synchronize()
    assert torch.equal(out, out2) and torch.equal(bits, bits2)
    assert torch.equal(mean, mean2) and torch.equal(invstd, invstd2) and torch.equal(rm, rm2) and torch.equal(rv, rv2)
    assert int(nbt2.item()) == 1
    # backward
    gd = torch.randn(N, HW, HW, C, generator=g).to(DEV)
    sums = torch.zeros(ns * 2 * C, dtype=torch.float64, device=DEV)
    ops.bn_bwd_reduce(gd, C, None, 0, None, HW * HW, yd, C, mean, invstd, None, 0, sums, M, C, nslot=ns)
    sums2 = sums.clone()
    dg, db, dy = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty_like(yd)
    ops.bn_param_grads(sums, dg, db, C, nslot=ns)
    ops.bn_bwd_apply(gd, C, yd, C, mean, invstd, gamma, sums, M, dy, C, M, C)
    dg2, db2, dy2 = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty_like(yd)
    ops.bn_bwd_apply_train(gd, C, yd, C, mean, invstd, gamma, sums2, ns, M, 1.0, dg2, db2, dy2, C, M, C)
    torch.cuda.synchronize()
    assert torch.equal(dy, dy2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    dg3, db3 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_bwd_apply_train(gd, C, yd, C, mean, invstd, gamma, sums2, ns, M, 0.25, dg3, db3, dy2, C, M, C)
    torch.cuda.synchronize()
    assert torch.equal(dy, dy2) and torch.allclose(dg3, dg * 0.25, rtol=1e-6, atol=0) and torch.allclose(db3, db * 0.25, rtol=1e-6, atol=0)
    report("fused BatchNorm launches C=%d %dx%d N=%d nslot=%d [%s]: bit-identical to finalize + apply / param_grads + bwd_apply" % (C, HW, HW, N, ns, mode))


@pytest.mark.parametrize("shape", [(3, 7, 9, 64), (2, 15, 15, 256), (1, 5, 5, 2048)])
@pytest.mark.parametrize("form", ["plain", "res", "two"])
def test_bn_apply_relu_bits(shape, form, report):
    """semseg_bn_apply's optional bit output: bit (c & 31) of word [m][c >> 5] = (the value entering the ReLU > 0), for the
    three forms the engine uses (BatchNorm+ReLU, + residual, bn3 + downsample BN); the activation itself is unchanged."""
    from semseg_amd import ops
    N, H, W, C = shape
    M = N * H * W
    g = torch.Generator().manual_seed(C + len(form))
    y = torch.randn(M, C, generator=g).to(DEV)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    y2 = torch.randn(M, C, generator=g).to(DEV) if form == "two" else None
    sc2, sh2 = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    res = torch.randn(M, C, generator=g).to(DEV) if form == "res" else None
    out0 = torch.empty(M, C, device=DEV)
    out1 = torch.empty(M, C, device=DEV)
    bits = torch.full((M, C // 32), -1, dtype=torch.int32, device=DEV)
    kw = dict(y2=y2, ldy2=C, scale2=sc2 if y2 is not None else None, shift2=sh2 if y2 is not None else None, res=res,
              ldres=C)
    ops.bn_apply(y, C, sc, sh, out0, C, M, C, H * W, True, **kw)
    ops.bn_apply(y, C, sc, sh, out1, C, M, C, H * W, True, relu_bits=bits, **kw)
    assert torch.equal(out0, out1)
    assert torch.equal(bits, relu_bits(out1))
    assert 0.2 < float((out1 > 0).float().mean()) < 0.8


def test_bn_eval(report):
    from semseg_amd import ops
    C, N, HW = 128, 2, 7
    g = torch.Generator().manual_seed(3)
    y = torch.randn(N, C, HW, HW, generator=g)
    gamma, beta, rm = torch.rand(C, generator=g) + .5, torch.randn(C, generator=g), torch.randn(C, generator=g)
    rv = torch.rand(C, generator=g) + .5
    ref = F.relu(F.batch_norm(y, rm, rv, gamma, beta, False, 0.1, 1e-5))
    scale, shift = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.bn_eval_params(gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV), 1e-5, scale, shift, C)
    out = torch.empty(N, HW, HW, C, device=DEV)
    ops.bn_apply(nhwc(y).to(DEV), C, scale, shift, out, C, N * HW * HW, C, HW * HW, True)
    e = relerr(nchw(out), ref)
    report("bn eval %.2e" % e)
    assert e < 1e-5


def test_maxpool(report):
    from semseg_amd import ops
    N, C, H, W = 2, 128, 23, 17
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, W, generator=g).double().requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    dy = torch.randn(y.shape, generator=g).double()
    y.backward(dy)
    Ho, Wo = y.shape[2:]
    xd = nhwc(x.detach().float()).to(DEV)
    yd = torch.empty(N, Ho, Wo, C, device=DEV)
    idx = torch.empty(N, Ho, Wo, C // 4, dtype=torch.int32, device=DEV)
    ops.maxpool_fwd(xd, yd, idx, N, H, W, C)
    dx = torch.empty(N, H, W, C, device=DEV)
    ops.maxpool_bwd(nhwc(dy.float()).to(DEV), idx, dx, N, H, W, C)
    e1, e2 = relerr(nchw(yd), y), relerr(nchw(dx), x.grad)
    report("maxpool fwd %.2e bwd %.2e" % (e1, e2))
    assert e1 == 0.0 and e2 < 1e-6


@pytest.mark.parametrize("H", [60, 10, 59])
def test_adaptive_pool(H, report):
    from semseg_amd import ops
    N, C, W = 2, 2048, H - 1 if H > 20 else H
    bins = (1, 2, 3, 6)
    g = torch.Generator().manual_seed(H)
    x = torch.randn(N, C, H, W, generator=g).double().requires_grad_(True)
    outs = [F.adaptive_avg_pool2d(x, b) for b in bins]
    dps = [torch.randn(o.shape, generator=g).double() for o in outs]
    base = torch.randn(N, C, H, W, generator=g).double()
    sum((o * d).sum() for o, d in zip(outs, dps)).backward()
    ld = C + 512
    xb = torch.zeros(N, H, W, ld, device=DEV)
    xb[..., :C] = nhwc(x.detach().float()).to(DEV)
    tot = sum(N * b * b * C for b in bins)
    y = torch.empty(tot, device=DEV)
    ops.adaptive_avgpool_fwd(xb, ld, y, bins, N, H, W, C)            # per-bin kernel (no scratch)
    off, es = 0, []
    for b, o in zip(bins, outs):
        n = N * b * b * C
        es.append(relerr(nchw(y[off:off + n].view(N, b, b, C)), o))
        off += n
    # single-pass path (what the engine runs): same result, and bit-identical from run to run
    scr = torch.empty(ops.adaptive_avgpool_scratch_floats(bins, N, H, C), device=DEV)
    assert scr.numel() == N * H * sum(bins) * C
    y1, y2 = torch.empty(tot, device=DEV), torch.empty(tot, device=DEV)
    ops.adaptive_avgpool_fwd(xb, ld, y1, bins, N, H, W, C, scratch=scr)
    ops.adaptive_avgpool_fwd(xb, ld, y2, bins, N, H, W, C, scratch=scr)
    assert torch.equal(y1, y2)
    off = 0
    for b, o in zip(bins, outs):
        n = N * b * b * C
        es.append(relerr(nchw(y1[off:off + n].view(N, b, b, C)), o))
        off += n
    dp = torch.cat([nhwc(d.float()).reshape(-1) for d in dps]).to(DEV)
    dx = torch.empty(N, H, W, C, device=DEV)
    ops.adaptive_avgpool_bwd(nhwc(base.float()).to(DEV), C, dp, dx, C, bins, N, H, W, C)
    e_b = relerr(nchw(dx), x.grad + base)
    report("adaptive pool H=%d fwd %s bwd %.2e" % (H, ["%.1e" % e for e in es], e_b))
    assert max(es) < 1e-5 and e_b < 1e-5


@pytest.mark.parametrize("hi,ho", [(1, 60), (2, 60), (3, 10), (6, 60), (30, 59), (59, 30)])
def test_bilinear(hi, ho, report):
    from semseg_amd import ops
    N, C = 2, 512
    wi, wo = hi, ho + (1 if ho > 20 else 0)
    g = torch.Generator().manual_seed(hi * 100 + ho)
    x = torch.randn(N, C, hi, wi, generator=g).double().requires_grad_(True)
    y = F.interpolate(x, (ho, wo), mode="bilinear", align_corners=True)
    dy = torch.randn(y.shape, generator=g).double()
    y.backward(dy)
    ldy = 2 * C
    yb = torch.zeros(N, ho, wo, ldy, device=DEV)
    ops.bilinear_fwd(nhwc(x.detach().float()).to(DEV), C, yb[..., C:], ldy, N, hi, wi, ho, wo, C)
    dyb = torch.zeros(N, ho, wo, ldy, device=DEV)
    dyb[..., C:] = nhwc(dy.float()).to(DEV)
    dx = torch.empty(N, hi, wi, C, device=DEV)
    ops.bilinear_bwd(dyb[..., C:], ldy, dx, C, N, hi, wi, ho, wo, C)
    e1, e2 = relerr(nchw(yb[..., C:]), y), relerr(nchw(dx), x.grad)
    report("bilinear %d->%d fwd %.2e bwd %.2e" % (hi, ho, e1, e2))
    assert e1 < 1e-5 and e2 < 1e-5 and float(yb[..., :C].abs().max()) == 0


@pytest.mark.parametrize("C,h,H", [(150, 9, 65), (19, 12, 89), (21, 5, 33)])
def test_ce_head(C, h, H, report):
    from semseg_amd import ops
    N, w, W = 2, h + 1, None
    W = (w - 1) * 8 + 1
    ld = ops.roundup(C, 64) if C > 64 else 64
    g = torch.Generator().manual_seed(C)
    z = (torch.randn(N, C, h, w, generator=g) * 3).double().requires_grad_(True)
    lab = torch.randint(0, C, (N, H, W), generator=g)
    lab[torch.rand(N, H, W, generator=g) < 0.1] = 255
    up = F.interpolate(z, (H, W), mode="bilinear", align_corners=True)
    loss = F.cross_entropy(up, lab, ignore_index=255)
    (loss * 0.4).backward()
    zb = torch.zeros(N, h, w, ld, device=DEV)
    zb[..., :C] = nhwc(z.detach().float()).to(DEV)
    labd = lab.to(DEV)
    lse = torch.empty(N, H, W, device=DEV)
    pred = torch.empty(N, H, W, dtype=torch.int64, device=DEV)
    acc = torch.zeros(3, dtype=torch.float64, device=DEV)
    lossd = torch.empty(1, device=DEV)
    ops.ce_head_fwd(zb, ld, labd, lse, pred, acc, lossd, N, h, w, H, W, C, 255)
    dz = torch.full((N, h, w, ld), 7.0, device=DEV)
    gl = torch.tensor([0.4], device=DEV)
    ops.ce_head_bwd(zb, ld, labd, lse, acc, gl, 1.0, dz, ld, False, N, h, w, H, W, C, 255)
    e_l = abs(float(lossd.item()) - float(loss)) / abs(float(loss))
    e_g = relerr(nchw(dz[..., :C]), z.grad)
    # cell-based backward (the path the engine takes: scratch given)
    dz2 = torch.full((N, h, w, ld), 7.0, device=DEV)
    scr = torch.empty(8 * 1024 * 1024, device=DEV)
    ops.ce_head_bwd(zb, ld, labd, lse, acc, gl, 1.0, dz2, ld, False, N, h, w, H, W, C, 255, scratch=scr)
    e_g2 = relerr(nchw(dz2[..., :C]), z.grad)
    assert e_g2 < 2e-5, e_g2
    assert float(dz2[..., C:ops.roundup(C, 4)].abs().max()) == 0.0 if ops.roundup(C, 4) > C else True
    agree = float((pred.cpu() == up.argmax(1)).float().mean())
    padz = float(dz[..., C:ops.roundup(C, 4)].abs().max()) if ops.roundup(C, 4) > C else 0.0
    report("ce head C=%d loss %.2e grad %.2e argmax-agree %.5f cnt %d/%d" %
           (C, e_l, e_g, agree, int(acc[1].item()), int((lab != 255).sum())))
    assert e_l < 1e-5 and e_g < 2e-5 and agree > 0.999 and padz == 0.0
    assert int(acc[1].item()) == int((lab != 255).sum())


def test_upsample_nchw(report):
    from semseg_amd import ops
    N, C, h, w, H, W = 2, 150, 9, 10, 65, 73
    g = torch.Generator().manual_seed(9)
    z = torch.randn(N, C, h, w, generator=g)
    ref = F.interpolate(z, (H, W), mode="bilinear", align_corners=True)
    zb = torch.zeros(N, h, w, 192, device=DEV)
    zb[..., :C] = nhwc(z).to(DEV)
    out = torch.empty(N, C, H, W, device=DEV)
    ops.bilinear_nhwc_to_nchw(zb, 192, out, N, h, w, H, W, C)
    e = relerr(out, ref)
    report("upsample->nchw %.2e" % e)
    assert e < 1e-5


def test_sgd(report):
    from semseg_amd import ops
    n = 100003
    g = torch.Generator().manual_seed(11)
    w = torch.randn(n, generator=g)
    p = torch.nn.Parameter(w.clone())
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9, weight_decay=1e-4)
    wd, mom = w.to(DEV), torch.zeros(n, device=DEV)
    for it in range(3):
        gr = torch.randn(n, generator=g)
        p.grad = gr.clone()
        opt.step()
        ops.sgd_step(wd, gr.to(DEV), mom, n, 0.01, 0.9, 1e-4, 1.0, it == 0)
    e = relerr(wd, p.data)
    report("sgd %.2e" % e)
    assert e < 1e-6


@pytest.mark.parametrize("H,W,mH,mW", [(5, 5, 9, 9), (5, 7, 9, 13), (6, 6, 5, 5), (4, 4, 3, 3), (7, 6, 5, 11),
                                      (3, 8, 9, 5), (1, 1, 1, 1), (30, 30, 59, 59), (45, 45, 89, 89), (70, 70, 9, 139)])
def test_psamask_vs_oracle(H, W, mH, mW, report):
    """Bit-exact against the C oracle (restatement of lib/psa/src/cpu/psamask.cpp): full-size, small, non-square and
    even-width masks, the two PSANet shapes (30^2 / 59^2 at 465 px, 45^2 / 89^2 at 705 px) and a shape past the
    plane-group kernel's largest instance (falls back to the slab kernel)."""
    import numpy as np
    from oracle import psamask as orc
    from semseg_amd import ops
    N = 2 if H < 40 else 1
    rng = np.random.default_rng(H * 1000 + mW)
    x = rng.standard_normal((N, mH * mW, H, W)).astype(np.float32)
    gy = rng.standard_normal((N, H * W, H, W)).astype(np.float32)
    for t in (0, 1):
        ref_f = orc.psa_mask_forward(x, t, mH, mW)
        ref_b = orc.psa_mask_backward(gy, t, mH, mW)
        out = torch.zeros(N, H * W, H, W, device=DEV)
        ops.psamask_forward(t, torch.from_numpy(x).to(DEV), out, N, H, W, mH, mW, (mH - 1) // 2, (mW - 1) // 2)
        gin = torch.zeros(N, mH * mW, H, W, device=DEV)
        ops.psamask_backward(t, torch.from_numpy(gy).to(DEV), gin, N, H, W, mH, mW, (mH - 1) // 2, (mW - 1) // 2)
        assert np.array_equal(out.cpu().numpy(), ref_f), "psamask fwd type %d" % t
        assert np.array_equal(gin.cpu().numpy(), ref_b), "psamask bwd type %d" % t
    report("psamask H=%d W=%d mask %dx%d bit-exact" % (H, W, mH, mW))


@pytest.mark.parametrize("H,W,mH,mW", [(5, 5, 9, 9), (5, 7, 9, 13), (6, 6, 5, 5), (8, 8, 15, 15)])
def test_psamask_nhwc_vs_oracle(H, W, mH, mW, report):
    """Pixel-major form used inside the engine == the reference op up to the layout permutation."""
    import numpy as np
    from oracle import psamask as orc
    from semseg_amd import ops
    N, HW, T = 2, H * W, mH * mW
    rng = np.random.default_rng(H * 100 + mW)
    x = rng.standard_normal((N, T, H, W)).astype(np.float32)       # reference layout
    gy = rng.standard_normal((N, HW, H, W)).astype(np.float32)
    ldm, lda = ops.roundup(T, 128), ops.roundup(HW, 128)
    for t in (0, 1):
        ref_f = orc.psa_mask_forward(x, t, mH, mW)                  # [N, p, h, w]
        ref_b = orc.psa_mask_backward(gy, t, mH, mW)                # [N, taps, h, w]
        m = torch.zeros(N * HW, ldm, device=DEV)
        m[:, :T] = torch.from_numpy(x).permute(0, 2, 3, 1).reshape(N * HW, T).to(DEV)
        a = torch.full((N * HW, lda), 5.0, device=DEV)
        ops.psamask_nhwc_forward(t, m, ldm, a, lda, N, H, W, mH, mW)
        got = a[:, :HW].view(N, HW, HW).permute(0, 2, 1).reshape(N, HW, H, W)   # [n, p, q]
        assert np.array_equal(got.cpu().numpy(), ref_f), "fwd type %d" % t
        da = torch.zeros(N * HW, lda, device=DEV)
        da[:, :HW] = torch.from_numpy(gy).reshape(N, HW, HW).permute(0, 2, 1).reshape(N * HW, HW).to(DEV)
        dm = torch.full((N * HW, ldm), 3.0, device=DEV)
        ops.psamask_nhwc_backward(t, da, lda, dm, ldm, N, H, W, mH, mW)
        gotb = dm[:, :T].view(N, H, W, T).permute(0, 3, 1, 2)
        assert np.array_equal(gotb.cpu().numpy(), ref_b), "bwd type %d" % t
        # pre-zeroed destination: only in-window taps are written, twice in a row (stale in-window values are overwritten)
        dz = torch.zeros(N * HW, ldm, device=DEV)
        ops.psamask_nhwc_backward(t, da * 2, lda, dz, ldm, N, H, W, mH, mW, prezeroed=True)
        ops.psamask_nhwc_backward(t, da, lda, dz, ldm, N, H, W, mH, mW, prezeroed=True)
        assert np.array_equal(dz[:, :T].view(N, H, W, T).permute(0, 3, 1, 2).cpu().numpy(), ref_b), "bwd type %d, pre-zeroed" % t
        assert float(dz[:, T:].abs().sum()) == 0.0
    report("psamask nhwc H=%d W=%d mask %dx%d bit-exact" % (H, W, mH, mW))


def test_softmax_rows_and_transpose(report):
    from semseg_amd import ops
    rows, P, ld = 37, 900, 1024
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(rows, P, generator=g) * 4).double().requires_grad_(True)
    y = torch.softmax(x, 1) * 0.25
    dy = torch.randn(rows, P, generator=g).double()
    y.backward(dy)
    xb = torch.zeros(rows, ld, device=DEV); xb[:, :P] = x.detach().float().to(DEV)
    yb = torch.zeros(rows, ld, device=DEV)
    ops.softmax_rows_fwd(xb, ld, yb, ld, rows, P, 0.25, True)
    dyb = torch.zeros(rows, ld, device=DEV); dyb[:, :P] = dy.float().to(DEV)
    ops.softmax_rows_bwd(yb, ld, dyb, ld, dyb, ld, rows, P, 0.25, True)
    e1, e2 = relerr(yb[:, :P], y), relerr(dyb[:, :P], x.grad)
    # transpose
    B, R, C = 3, 45, 70
    t = torch.randn(B, R, C, generator=g)
    out = torch.full((B, C, 64), 9.0, device=DEV)
    ops.transpose_batched(t.to(DEV), C, R * C, out, 64, C * 64, B, R, C)
    ok = torch.equal(out[:, :, :R].cpu(), t.transpose(1, 2)) and float(out[:, :, R:].abs().max()) == 0.0
    report("softmax rows fwd %.2e bwd %.2e transpose %s" % (e1, e2, ok))
    assert e1 < 1e-6 and e2 < 1e-5 and ok


def test_gemm_entry_points(report):
    """The raw-pointer GEMM forms used for the PSA point-affinity contraction (torch.bmm in
    model/psanet.py:90-91) and its two gradients."""
    from semseg_amd import ops
    g = torch.Generator().manual_seed(4)
    hw, C, P = 100, 512, 128
    A = torch.zeros(hw + 128, P); A[:hw, :hw] = torch.randn(hw, hw, generator=g)
    X = torch.zeros(hw + 128, C); X[:hw] = torch.randn(hw, C, generator=g)
    Z = A[:hw, :hw].double() @ X[:hw].double()
    Ad, Xd = A.to(DEV), X.to(DEV)
    XT = torch.zeros(C + 128, P, device=DEV)
    ops.transpose_batched(Xd, C, hw * C, XT, P, C * P, 1, hw, C)
    Zd = torch.zeros(hw, C, device=DEV)
    ops.gemm_rows(Ad.data_ptr(), P, XT.data_ptr(), Zd.data_ptr(), C, hw, P, C)
    dZ = torch.randn(hw, C, generator=g)
    dA = dZ.double() @ X[:hw].double().t()
    dX = A[:hw, :hw].double().t() @ dZ.double()
    dZd = dZ.to(DEV)
    dAd = torch.zeros(hw + 128, P, device=DEV)
    ops.gemm_rows(dZd.data_ptr(), C, Xd.data_ptr(), dAd.data_ptr(), P, hw, C, hw)
    dXd = torch.zeros(hw, C, device=DEV)
    scratch = torch.empty(8 * 1024 * 1024, device=DEV)
    ops.gemm_kmajor(dZd.data_ptr(), C, Ad.data_ptr(), P, dXd.data_ptr(), scratch, hw, C, hw)
    e = (relerr(Zd, Z), relerr(dAd[:hw, :hw], dA), relerr(dXd, dX))
    report("psa contraction gemms: fwd %.2e dA %.2e dX %.2e" % e)
    assert max(e) < 1e-5 and float(dAd[:hw, hw:].abs().max()) == 0.0


def test_batched_gemm_entry_points(report):
    """torch.bmm of model/psanet.py:90-91 and its two gradients as ONE launch per GEMM (blockIdx.y = image): the
    batched forms of the two matrix-core kernels vs fp64, including accumulation into an existing dx and the
    weight-gradient kernel variants (SEMSEG_DEBUG wgrad_dma)."""
    import os
    from semseg_amd import ops
    g = torch.Generator().manual_seed(5)
    B, hw, C = 3, 100, 512
    P = 128
    A = torch.zeros(B * hw + 128, P); X = torch.zeros(B * hw + 128, C)
    A[:B * hw, :hw] = torch.randn(B * hw, hw, generator=g)
    X[:B * hw] = torch.randn(B * hw, C, generator=g)
    A3, X3 = A[:B * hw, :hw].view(B, hw, hw).double(), X[:B * hw].view(B, hw, C).double()
    Z = torch.bmm(A3, X3)
    Ad, Xd = A.to(DEV), X.to(DEV)
    XT = torch.zeros(B * C + 128, P, device=DEV)
    ops.transpose_batched(Xd, C, hw * C, XT, P, C * P, B, hw, C)
    Zd = torch.zeros(B, hw, 2 * C, device=DEV)                      # written into a channel slice (ld = 2C)
    ops.gemm_rows_batched(Ad, P, hw * P, XT, C * P, Zd[..., C:], 2 * C, hw * 2 * C, hw, P, C, B)
    dZ = torch.randn(B, hw, C, generator=g)
    dA = torch.bmm(dZ.double(), X3.transpose(1, 2))
    dX0 = torch.randn(B, hw, C, generator=g)
    dX = torch.bmm(A3.transpose(1, 2), dZ.double()) + dX0.double()
    dZd = dZ.to(DEV)
    dAd = torch.zeros(B * hw + 128, P, device=DEV)
    ops.gemm_rows_batched(dZd, C, hw * C, Xd, hw * C, dAd, P, hw * P, hw, C, hw, B)
    scratch = torch.empty(16 * 1024 * 1024, device=DEV)
    errs = []
    for v in ("0", "2"):
        os.environ["SEMSEG_DEBUG"] = "wgrad_dma=" + v
        dXd = dX0.to(DEV).clone()
        ops.gemm_kmajor_batched(dZd, C, hw * C, Ad, P, hw * P, dXd, hw * C, scratch, hw, C, hw, B, accumulate=True)
        errs.append(relerr(dXd, dX))
    os.environ.pop("SEMSEG_DEBUG")
    e = (relerr(Zd[..., C:], Z), relerr(dAd[:B * hw, :hw].view(B, hw, hw), dA), max(errs))
    report("batched psa contraction gemms (%d images, one launch each): fwd %.2e dA %.2e dX %.2e" % ((B,) + e))
    assert max(e) < 1e-5 and float(Zd[..., :C].abs().max()) == 0.0 and float(dAd[:, hw:].abs().max()) == 0.0


def test_lib_psa_functional_dropin(report):
    """`lib.psa.functional.psa_mask` (the reference's Python op API) forward + autograd backward against
    the golden vectors produced by the reference's compiled CPU op."""
    import os
    import numpy as np
    import lib.psa.functional as PF
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "psamask_ref.npz"))
    keys = sorted({k.rsplit("_", 1)[0] for k in fx.files})
    for key in keys:
        parts = key.split("_")
        mH, mW = (int(v) for v in parts[2][1:].split("x"))
        t = int(parts[3][1:])
        x = torch.from_numpy(fx[key + "_x"]).to(DEV).requires_grad_(True)
        out = PF.psa_mask(x, t, mH, mW)
        out.backward(torch.from_numpy(fx[key + "_gy"]).to(DEV))
        assert np.array_equal(out.detach().cpu().numpy(), fx[key + "_out"]), key
        assert np.array_equal(x.grad.cpu().numpy(), fx[key + "_gin"]), key
    # default mask size = 2*feature-1 and the argument checks of functions/psamask.py:9-15
    x = torch.randn(1, 49, 4, 4, device=DEV)
    assert PF.psa_mask(x).shape == (1, 16, 4, 4)
    with pytest.raises(AssertionError):
        PF.psa_mask(x, 0, 6, 6)
    with pytest.raises(RuntimeError):
        PF.psa_mask(torch.randn(1, 49, 4, 4))
    report("lib.psa.functional.psa_mask == reference golden vectors (%d cases), checks ok" % len(keys))


def test_dropout2d_mask_statistics(report):
    """semseg_dropout2d_mask: one Bernoulli(1-p) per (n, c) plane scaled by 1/(1-p) (nn.Dropout2d, model/pspnet.py:68);
    deterministic in (seed, offset), different across offsets."""
    from semseg_amd import ops
    n, p = 1 << 16, 0.1
    a = torch.empty(n, device=DEV)
    b = torch.empty(n, device=DEV)
    c = torch.empty(n, device=DEV)
    ops.dropout2d_mask(a, p, 1234, 1)
    ops.dropout2d_mask(b, p, 1234, 1)
    ops.dropout2d_mask(c, p, 1234, 2)
    vals = torch.unique(a).cpu().tolist()
    keep = float((a > 0).float().mean())
    assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / 0.9) < 1e-6
    assert abs(keep - 0.9) < 0.005 and abs(float(a.mean()) - 1.0) < 0.01
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(((a > 0) & (c > 0)).float().mean()) - 0.81) < 0.01          # independent across offsets
    report("dropout2d mask: keep fraction %.4f (p = 0.1), mean %.4f, reproducible per (seed, offset)" % (keep, float(a.mean())))


def test_intersection_and_union(report):
    """util/util.py:40-52 (numpy form) is the oracle for the device kernel (util/util.py:55-67 form)."""
    import numpy as np
    from semseg_amd.metrics import intersectionAndUnionGPU
    K = 150
    g = torch.Generator().manual_seed(8)
    out = torch.randint(0, K, (4, 97, 113), generator=g)
    tgt = torch.randint(0, K, (4, 97, 113), generator=g)
    tgt[torch.rand(4, 97, 113, generator=g) < 0.1] = 255
    same = torch.rand(4, 97, 113, generator=g) < 0.5
    out[same] = tgt[same].clamp(max=K - 1)
    o, t = out.numpy().reshape(-1).copy(), tgt.numpy().reshape(-1)
    o[np.where(t == 255)[0]] = 255
    inter = o[np.where(o == t)[0]]
    ai, _ = np.histogram(inter, bins=np.arange(K + 1))
    ao, _ = np.histogram(o, bins=np.arange(K + 1))
    at, _ = np.histogram(t, bins=np.arange(K + 1))
    i, u, tt = intersectionAndUnionGPU(out.cuda(), tgt.cuda(), K, 255)
    assert np.array_equal(i.cpu().numpy(), ai.astype(np.float32))
    assert np.array_equal(u.cpu().numpy(), (ao + at - ai).astype(np.float32))
    assert np.array_equal(tt.cpu().numpy(), at.astype(np.float32))
    report("intersectionAndUnionGPU == numpy reference (exact)")


@pytest.mark.parametrize("Ci,k,rows_n", [(2048, 1, 2), (512, 3, 2), (4096, 3, 2)])
def test_conv_rounding_noise_vs_reduction_length(Ci, k, rows_n, report):
    """fp32 MFMA accumulates sequentially along K = Ci*R*S, so rounding noise grows ~sqrt(K) with the chain
    length: a single chain measured rms 2.3e-6 at K=36864 (7x torch CPU's blocked summation).  The kernel
    therefore flushes into a second accumulator set every ~1024 K when K >= 4096 (4.1e-7 at K=36864, 1.3x CPU);
    K=2048 unsplit stays a single chain (5.8e-7, 2.7x CPU).  Regression guard, bounds set from those
    measurements: split-K rms <= 3x CPU fp32 rms (+1e-7), unsplit <= 4x (+1e-7).  DESIGN.md section 2.1."""
    from semseg_amd import ops
    Co, N, H = 512, rows_n, 8
    g = torch.Generator().manual_seed(Ci + k)
    x = torch.relu(torch.randn(N, Ci, H, H, generator=g))
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, 1, k // 2)
    rms = lambda a: float(((a - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    e_cpu = rms(F.conv2d(x, w, None, 1, k // 2).double())
    pk = ops.PackedConv(Co, Ci, k, k, DEV)
    pk.pack(w.to(DEV))
    xd = nhwc(x).contiguous().to(DEV)
    errs = []
    for scratch in (None, torch.empty(64 * 1024 * 1024, device=DEV)):
        yb = torch.empty(N, H, H, Co, device=DEV)
        ops.conv_fwd(xd, Ci, pk, yb, Co, N, H, H, 1, k // 2, 1, scratch=scratch)
        torch.cuda.synchronize()
        errs.append(rms(nchw(yb).cpu().double()))
    report("conv noise K=%d: rms unsplit %.2e split-K %.2e torch-cpu-fp32 %.2e" % (Ci * k * k, errs[0], errs[1], e_cpu))
    assert errs[1] <= 3.0 * e_cpu + 1e-7 and errs[0] <= 4.0 * e_cpu + 1e-7


@pytest.mark.parametrize("code", [128, 64, 1128, 1064])
@pytest.mark.parametrize("case", [(2, 15, 15, 256, 256, 1, 1, 0, 1), (3, 13, 11, 128, 128, 3, 1, 2, 2), (2, 21, 21, 256, 512, 1, 2, 0, 1)])
def test_conv_tile_codes(case, code, report, monkeypatch):
    """Every tile shape of the forward / data-gradient kernel (128x128, 128x64, 64x128, 64x64) on the same operands,
    forced through the tile table: forward with statistics, data gradient, fused BatchNorm-backward reduction."""
    from semseg_amd import ops
    N, H, W, Ci, Co, k, s_, p_, d = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * (1.0 / (Ci * k * k) ** 0.5)
    x64, w64 = x.double().requires_grad_(True), w.double()
    y64 = F.conv2d(x64, w64, None, s_, p_, d)
    Ho, Wo = y64.shape[2:]
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    y64.backward(dy.double())
    pk = ops.PackedConv(Co, Ci, k, k, DEV)
    pk.pack(w.to(DEV))
    monkeypatch.setitem(ops.TILE_CHOICE, ops.tile_key("fwd", N, H, W, Ci, Co, k, k, s_, p_, d), code)
    monkeypatch.setitem(ops.TILE_CHOICE, ops.tile_key("dgrad", N, H, W, Ci, Co, k, k, s_, p_, d), code)
    xb = nhwc(x).to(DEV)
    yb = torch.full((N, Ho, Wo, Co), float("nan"), device=DEV)
    st = torch.zeros(ops.NSLOT * 2 * Co, dtype=torch.float64, device=DEV)
    scratch = torch.empty(8 * 1024 * 1024, device=DEV)
    ops.conv_fwd(xb, Ci, pk, yb, Co, N, H, W, s_, p_, d, stats=st, nslot=ops.NSLOT, scratch=scratch)
    e_f = relerr(nchw(yb), y64.detach())
    sref = torch.stack([y64.detach().sum((0, 2, 3)), (y64.detach() ** 2).sum((0, 2, 3))])
    e_s = float((st.view(ops.NSLOT, 2, Co).sum(0).cpu() - sref).abs().max() / sref.abs().max())
    dyb = nhwc(dy).to(DEV)
    dxb = torch.full((N, H, W, Ci), float("nan"), device=DEV)
    ops.conv_dgrad(dyb, Co, pk, dxb, Ci, N, H, W, s_, p_, d, scratch=scratch)
    e_d = relerr(nchw(dxb), x64.grad)
    report("conv tile code %d %s: fwd %.2e stats %.2e dgrad %.2e" % (code, case, e_f, e_s, e_d))
    assert e_f < 2e-5 and e_s < 1e-5 and e_d < 2e-5


WINO_CASES = [  # N, H, W, Ci, Co, dilation
    (2, 12, 12, 64, 64, 1),
    (1, 15, 13, 64, 96, 2),      # odd sizes: partial tiles in some phases
    (2, 9, 11, 64, 128, 1),
    (1, 17, 17, 128, 128, 4),    # dilation 4: 16 phases of 5 / 4 rows
    (2, 30, 30, 256, 256, 2),    # layer3's conv2 at a small batch
    (1, 8, 8, 512, 64, 1),
]


@pytest.mark.parametrize("arith", [0, 3])
@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_conv_fwd_dgrad_wgrad(case, arith, report):
    """Winograd F(2x2, 3x3) path (input / filter / output transforms + the batched matrix-core GEMMs) against fp64
    F.conv2d and its gradients: kernel 3, stride 1, padding = dilation; operands in wider buffers; statistics and the
    accumulate-into-dx form included.  Same bound as the direct kernels, in both arithmetics (0 = SEMSEG_ARITH_F32,
    3 = SEMSEG_ARITH_BF16X3 on the batched row GEMMs and the batched K-major GEMM of the weight gradient)."""
    from semseg_amd import ops
    N, H, W, Ci, Co, d = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * (1.0 / (Ci * 9) ** 0.5)
    dy = torch.randn(N, Co, H, W, generator=g)
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, 1, d, d)
    y64.backward(dy.double())
    wc = ops.WinoConv(Co, Ci, DEV)
    wc.transform(w.to(DEV))
    T = ops.wino_tiles(N, H, W, d)
    ldx, ldy = Ci + 32, ops.roundup(Co, 128) + 64
    xb = torch.randn(N, H, W, ldx, device=DEV)
    xb[..., 16:16 + Ci] = nhwc(x).to(DEV)
    yb = torch.full((N, H, W, ldy), float("nan"), device=DEV)
    V = torch.empty(16 * T * Ci, device=DEV)
    Mbuf = torch.empty(16 * T * max(Ci, Co), device=DEV)
    st = torch.zeros(ops.NSLOT * 2 * Co, dtype=torch.float64, device=DEV)
    ops.wino_conv_fwd(xb[..., 16:], ldx, wc, yb, ldy, N, H, W, d, V, Mbuf, stats=st, nslot=ops.NSLOT, arith=arith)
    y = yb[..., :Co].permute(0, 3, 1, 2)
    e_f = relerr(y, y64.detach())
    assert torch.isnan(yb[..., Co:]).all()                     # nothing written beyond the valid channels
    s = st.view(ops.NSLOT, 2, Co).sum(0).cpu()
    ref_s = torch.stack([y64.detach().sum((0, 2, 3)), (y64.detach() ** 2).sum((0, 2, 3))])
    e_s = float((s - ref_s).abs().max() / ref_s.abs().max())
    # eval-mode epilogue: y = relu(conv * scale + shift + residual)
    sc, sh = torch.rand(Co, device=DEV) + 0.5, torch.randn(Co, device=DEV)
    res = torch.randn(N, H, W, ldy, device=DEV)
    ye = torch.full((N, H, W, ldy), float("nan"), device=DEV)
    ops.wino_output_transform(Mbuf, Co, ye, ldy, N, H, W, Co, d, add=res, ldadd=ldy, scale=sc, shift=sh, relu=True)
    ref_e = torch.relu(nhwc(y64.detach()) * sc.double().cpu() + sh.double().cpu() + res[..., :Co].double().cpu())
    assert relerr(ye[..., :Co], ref_e) < 2e-5
    # data gradient, accumulated onto an existing gradient
    dyb = torch.zeros(N, H, W, ldy, device=DEV)
    dyb[..., :Co] = nhwc(dy).to(DEV)
    base = torch.randn(N, H, W, ldx, device=DEV)
    dxb = base.clone()
    Vdy = torch.empty(16 * T * wc.Kc, device=DEV)
    ops.wino_conv_dgrad(dyb, ldy, wc, dxb[..., 16:], ldx, N, H, W, d, Vdy, Mbuf, add=base[..., 16:], ldadd=ldx, arith=arith)
    dx = (dxb[..., 16:16 + Ci] - base[..., 16:16 + Ci]).permute(0, 3, 1, 2)
    e_d = relerr(dx, x64.grad)
    assert torch.equal(dxb[..., :16], base[..., :16]) and torch.equal(dxb[..., 16 + Ci:], base[..., 16 + Ci:])
    # the same data gradient with the BatchNorm-backward reduction of the layer that produced x folded into the output
    # transform: g = (dx + add) * (act > 0), sums = {sum g, sum g * xhat}
    act = torch.randn(N, H, W, ldx, device=DEV)
    ybn = torch.randn(N, H, W, Ci + 4, device=DEV)
    mean, invstd = torch.randn(Ci, device=DEV), torch.rand(Ci, device=DEV) + 0.5
    sums = torch.zeros(ops.NSLOT * 2 * Ci, dtype=torch.float64, device=DEV)
    gb = torch.full((N, H, W, ldx), float("nan"), device=DEV)
    ops.wino_input_transform(dyb, ldy, Vdy, N, H, W, wc.Kc, d)
    ops.gemm_rows_batched(Vdy, wc.Kc, T * wc.Kc, wc.U_dgrad, wc.Ci_pad * wc.Kc, Mbuf, Ci, T * Ci, T, wc.Kc, Ci, 16, arith=arith)
    ops.wino_output_transform_bnreduce(Mbuf, Ci, gb[..., 16:], ldx, N, H, W, Ci, d, act[..., 16:], ldx, ybn, Ci + 4, mean,
                                       invstd, sums, ops.NSLOT, add=base[..., 16:], ldadd=ldx)
    g_ref = (nhwc(x64.grad) + base[..., 16:16 + Ci].double().cpu()) * (act[..., 16:16 + Ci] > 0).cpu()
    xh = (ybn[..., :Ci].double().cpu() - mean.double().cpu()) * invstd.double().cpu()
    s_ref = torch.stack([g_ref.sum((0, 1, 2)), (g_ref * xh).sum((0, 1, 2))])
    e_g = relerr(gb[..., 16:16 + Ci], g_ref)
    e_gs = float((sums.view(ops.NSLOT, 2, Ci).sum(0).cpu() - s_ref).abs().max() / s_ref.abs().max())
    assert e_g < 2e-5 and e_gs < 1e-5, (e_g, e_gs)
    if Ci % 32 == 0:    # the same with the mask as bits (what bn_apply writes): identical values
        sums_b = torch.zeros_like(sums)
        gb2 = torch.full((N, H, W, ldx), float("nan"), device=DEV)
        ops.wino_output_transform_bnreduce(Mbuf, Ci, gb2[..., 16:], ldx, N, H, W, Ci, d, None, 0, ybn, Ci + 4, mean,
                                           invstd, sums_b, ops.NSLOT, add=base[..., 16:], ldadd=ldx,
                                           relu_bits=relu_bits(act[..., 16:16 + Ci]))
        assert torch.equal(gb2[..., 16:16 + Ci], gb[..., 16:16 + Ci])
        sa, sb = sums.view(ops.NSLOT, 2, Ci).sum(0), sums_b.view(ops.NSLOT, 2, Ci).sum(0)
        assert float((sa - sb).abs().max() / sa.abs().max()) < 1e-12
    # weight gradient from the kept transformed input
    Yh = torch.zeros(16 * T * ops.roundup(Co, 128), device=DEV)
    dU = torch.empty(16 * Co * Ci, device=DEV)
    scratch = torch.empty(16 * 1024 * 1024, device=DEV)
    dw = torch.full((Co, Ci, 3, 3), float("nan"), device=DEV)
    ops.wino_conv_wgrad(V, dyb, ldy, wc, dw, N, H, W, d, Yh, dU, scratch, arith=arith)
    e_w = relerr(dw, w64.grad)
    report("winograd %s arith %d: fwd %.2e stats %.2e dgrad %.2e wgrad %.2e" % (case, arith, e_f, e_s, e_d, e_w))
    assert e_f < 2e-5 and e_d < 2e-5 and e_w < 2e-5 and e_s < 1e-5


def test_gemm_kmajor_batched_chunks_when_scratch_is_small(report):
    """a batch whose partial slabs do not fit the scratch arena runs in chunks instead of failing."""
    from semseg_amd import ops
    B, K, Ci, Co = 5, 300, 128, 128
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, K, Ci, generator=g).to(DEV)
    y = torch.randn(B, K, Co, generator=g).to(DEV)
    ref = torch.einsum("bko,bkc->boc", y.double().cpu(), x.double().cpu())
    out = torch.full((B, Co, Ci), float("nan"), device=DEV)
    scratch = torch.empty(2 * Co * Ci + 7, device=DEV)       # room for two slabs only -> chunks of 2, 2, 1
    ops.gemm_kmajor_batched(x, Ci, K * Ci, y, Co, K * Co, out, Co * Ci, scratch, K, Ci, Co, B)
    e = relerr(out, ref)
    report("gemm_kmajor_batched in chunks (5 items, scratch for 2): %.2e" % e)
    assert e < 2e-5
    with pytest.raises(ops.HipError):
        ops.gemm_kmajor_batched(x, Ci, K * Ci, y, Co, K * Co, out, Co * Ci, torch.empty(Co * Ci - 1, device=DEV), K, Ci,
                                Co, B)


@pytest.mark.parametrize("nsplit,bk,B", [(2, 16, 3), (2, 32, 3), (3, 16, 3), (3, 16, 40)])
def test_gemm_rows_bf16split(nsplit, bk, B, report):
    """The split-bf16 row GEMM of the Winograd path (csrc/gemm_bf16split.hip; nsplit 3 = SEMSEG_ARITH_BF16X3, DESIGN.md
    section 8.4) against fp64 and next to the fp32 matrix-core kernel on the same operands — ragged last row tile, a
    column tile that is half padding, strided A and C, batch strides.  Bounds from the error model: three pieces / six
    products carry 24 mantissa bits (rms within 2x of the fp32 kernel's own rounding noise + 1e-7); two pieces / three
    products (a measurement only, nothing in the engine selects it) carry 16 (rms <= 1e-5).  Batch 3 = 12 tiles of 256 rows: the
    launcher takes the 128-row instance (four waves, round 5); batch 40 = 160 tiles: the 256-row instance."""
    from semseg_amd import ops
    M, K, Nout, lda, ldc = 300, 1024, 192, 1024 + 64, 192 + 64
    g = torch.Generator().manual_seed(17 + nsplit + bk)
    a = torch.randn(B, M, lda, generator=g)
    bt = torch.zeros(B, 256, K)                      # panel rows padded to 256, the padding stays zero
    bt[:, :Nout] = torch.randn(B, Nout, K, generator=g) / K ** 0.5
    ref = torch.einsum("bmk,bnk->bmn", a[:, :, :K].double(), bt[:, :Nout].double())
    ad, btd = a.to(DEV), bt.to(DEV)
    out = torch.full((B, M, ldc), float("nan"), device=DEV)
    ops.gemm_rows_batched_bf16split(ad, lda, M * lda, btd, 256 * K, out, ldc, M * ldc, M, K, Nout, B, nsplit=nsplit, bk=bk)
    base = torch.full((B, M, ldc), float("nan"), device=DEV)
    ops.gemm_rows_batched(ad, lda, M * lda, btd, 256 * K, base, ldc, M * ldc, M, K, Nout, B)
    torch.cuda.synchronize()
    assert torch.isnan(out[:, :, Nout:]).all()       # nothing written past Nout
    rms = lambda t: float((t[:, :, :Nout].cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    e, e32 = rms(out), rms(base)
    report("gemm_rows_batched_bf16split nsplit %d bk %d batch %d: rms %.2e (fp32 matrix-core kernel %.2e)" % (nsplit, bk, B, e, e32))
    assert e <= (2.0 * e32 + 1e-7 if nsplit == 3 else 1e-5)
    with pytest.raises(ops.HipError):
        ops.gemm_rows_batched_bf16split(ad, lda, M * lda, btd, 256 * K, out, ldc, M * ldc, M, K, Nout, B, nsplit=3, bk=32)


@pytest.mark.parametrize("code", [128, 64, 1128, 1064])
def test_conv_arith_bf16x3_1x1(code, report, monkeypatch):
    """SEMSEG_ARITH_BF16X3 per launch: the SP instances of the forward / data-gradient kernel (three-way split bf16
    pieces, six bf16 matrix-core products) on a 1x1 conv with every fused epilogue — forward with batch statistics, data
    gradient with residual add, data gradient with the fused BatchNorm-backward reduction — next to the fp32 instances
    on the same operands: rms error within 2x of the fp32 kernel's (+1e-7), statistics and reduction sums to 1e-6."""
    from semseg_amd import ops
    N, H, W, Ci, Co = 3, 17, 15, 1024, 256
    g = torch.Generator().manual_seed(23)
    x = torch.relu(torch.randn(N, Ci, H, W, generator=g))
    w = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    dy = torch.randn(N, Co, H, W, generator=g)
    add = torch.randn(N, Ci, H, W, generator=g)
    y64 = F.conv2d(x.double(), w.double())
    dx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double()) + add.double()
    pk = ops.PackedConv(Co, Ci, 1, 1, DEV)
    pk.pack(w.to(DEV))
    for sfx in ("", "|sp"):          # the split instances look their tile up under the suffixed key
        monkeypatch.setitem(ops.TILE_CHOICE, ops.tile_key("fwd", N, H, W, Ci, Co, 1, 1, 1, 0, 1) + sfx, code)
        monkeypatch.setitem(ops.TILE_CHOICE, ops.tile_key("dgrad", N, H, W, Ci, Co, 1, 1, 1, 0, 1) + sfx, code)
    xd, dyd, addd = nhwc(x).contiguous().to(DEV), nhwc(dy).contiguous().to(DEV), nhwc(add).contiguous().to(DEV)
    rms = lambda a, ref: float((a.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    res = {}
    for split in (False, True):
        ar = ops.ARITH_BF16X3 if split else ops.ARITH_F32
        yb = torch.empty(N, H, W, Co, device=DEV)
        stats = torch.zeros(ops.NSLOT * 2 * Co, dtype=torch.float64, device=DEV)
        ops.conv_fwd(xd, Ci, pk, yb, Co, N, H, W, 1, 0, 1, stats=stats, nslot=ops.NSLOT,
                     scratch=torch.empty(1 << 24, device=DEV), arith=ar)
        dxb = torch.empty(N, H, W, Ci, device=DEV)
        ops.conv_dgrad(dyd, Co, pk, dxb, Ci, N, H, W, 1, 0, 1, add=addd, ldadd=Ci, scratch=torch.empty(1 << 24, device=DEV),
                       arith=ar)
        torch.cuda.synchronize()
        st = stats.view(ops.NSLOT, 2 * Co).sum(0).cpu()
        res[split] = (rms(nchw(yb).cpu(), y64), rms(nchw(dxb).cpu(), dx64),
                      float((st[:Co] - y64.sum((0, 2, 3))).abs().max() / y64.sum((0, 2, 3)).abs().max()),
                      float((st[Co:] - (y64 ** 2).sum((0, 2, 3))).abs().max() / (y64 ** 2).sum((0, 2, 3)).abs().max()))
    report("conv split mode tile %d: forward rms %.2e (fp32 %.2e)  data gradient %.2e (fp32 %.2e)  stats %.1e / %.1e"
           % (code, res[True][0], res[False][0], res[True][1], res[False][1], res[True][2], res[True][3]))
    assert res[True][0] <= 2 * res[False][0] + 1e-7 and res[True][1] <= 2 * res[False][1] + 1e-7
    assert res[True][2] < 1e-6 and res[True][3] < 1e-6
    with pytest.raises(ops.HipError):                             # an unknown arithmetic code is rejected, not ignored
        ops.conv_fwd(xd, Ci, pk, yb, Co, N, H, W, 1, 0, 1, arith=6)


@pytest.mark.parametrize("case", [(2, 23, 21, 256, 128, 1, 1, 0, 1), (2, 19, 17, 128, 256, 3, 1, 2, 2), (2, 21, 21, 256, 256, 1, 2, 0, 1),
                                  (1, 33, 33, 128, 128, 3, 1, 1, 1)])
@pytest.mark.parametrize("sp_variant", [8, 9, 0, 10])
def test_conv_wgrad_arith_bf16x3(case, sp_variant, report, monkeypatch):
    """SEMSEG_ARITH_BF16X3 per launch: the SP instance of the 128 x 128 weight-gradient kernel (pixel-contiguous bf16
    piece planes, six bf16 matrix-core products) in its three gather modes (1x1, "same" 3x3 with dilation, strided)
    next to the fp32 kernels on the same operands, against fp64: rms within 2x of the fp32 path's (+1e-7)."""
    from semseg_amd import ops
    # wgrad_small=0: keep the 128 x 128 path on these small grids
    # wgrad_sp 8 / 9: the direct-to-LDS ring with the split at fragment time (4 stages / 3 stages); 0: the register-staged SP kernel;
    # 10: the 128 x 256 kernel with 64 x 128 wave tiles (layers with Ci % 256 == 0, else variant 8)
    monkeypatch.setenv("SEMSEG_DEBUG", "wgrad_small=0,wgrad_sp=%s" % sp_variant)
    N, H, W, Ci, Co, k, s_, p_, d = case
    g = torch.Generator().manual_seed(31)
    x = torch.relu(torch.randn(N, Ci, H, W, generator=g))
    Ho, Wo = ops.conv_out(H, k, s_, p_, d), ops.conv_out(W, k, s_, p_, d)
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, k, k), dy.double(), stride=s_, padding=p_, dilation=d)
    xd, dyd = nhwc(x).contiguous().to(DEV), nhwc(dy).contiguous().to(DEV)
    scratch = torch.empty(1 << 25, device=DEV)
    rms = lambda a: float((a.cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    errs = []
    for split in (False, True):
        dw = torch.full((Co, Ci, k, k), float("nan"), device=DEV)
        ops.conv_wgrad(xd, Ci, dyd, Co, dw, scratch, N, H, W, Ci, Co, k, k, s_, p_, d,
                       arith=ops.ARITH_BF16X3 if split else ops.ARITH_F32)
        torch.cuda.synchronize()
        errs.append(rms(dw))
    report("conv_wgrad bf16x3 variant %d %s: rms %.2e (fp32 path %.2e)" % (sp_variant, case, errs[1], errs[0]))
    assert errs[1] <= 2 * errs[0] + 1e-7


@pytest.mark.parametrize("case", [(2, 96, 96, 256, 128), (5, 59, 59, 512, 256), (3, 80, 77, 1024, 128)])
def test_conv_wgrad_bf16x3_long_1x1_reductions_run_two_accumulator_sets(case, report):
    """1x1 weight gradients with more than 16 384 pixels to reduce (WGRAD_ACC2_MIN_M; Ci % 256 == 0: the 128 x 256 kernel) accumulate the
    five small cross products of bf16x3 apart from the leading one (csrc/conv_wgrad.hip, ACC2): against fp64 the
    bf16x3 result stays within 1.3 x the exact-fp32-product path's rms (measured on the first run of this test: 0.6-1.2 x — 1.21 x at
    17 405 pixels, where its first bound of 1.1 x failed, 0.64 x at 55 696; one accumulator set measured 1.8-2.5 x on these lengths; the
    bound of the short-reduction tests above is 2 x)."""
    from semseg_amd import ops
    N, H, W, Ci, Co = case
    M = N * H * W
    assert M > 16384 and Ci % 256 == 0
    g = torch.Generator().manual_seed(Ci + H)
    x = torch.relu(torch.randn(M, Ci, generator=g)).to(DEV)
    dy = (torch.randn(M, Co, generator=g) * 1e-3).to(DEV)
    ref = dy.double().t() @ x.double()
    rr = float(ref.pow(2).mean().sqrt())
    ldy = ops.roundup(Co, 128)
    dyp = torch.zeros(M, ldy, device=DEV)
    dyp[:, :Co] = dy
    scratch = torch.empty(1 << 26, device=DEV)
    errs = {}
    for name, ar in (("f32", ops.ARITH_F32), ("bf16x3", ops.ARITH_BF16X3)):
        dw = torch.full((Co, Ci, 1, 1), float("nan"), device=DEV)
        ops.conv_wgrad(x, Ci, dyp, ldy, dw, scratch, N, H, W, Ci, Co, 1, 1, 1, 0, 1, arith=ar)
        torch.cuda.synchronize()
        errs[name] = float((dw.view(Co, Ci).double() - ref).pow(2).mean().sqrt()) / rr
    report("1x1 weight gradient over %d pixels, %d -> %d channels: rms bf16x3 (two accumulator sets) %.2e, exact fp32 products %.2e"
           % (M, Ci, Co, errs["bf16x3"], errs["f32"]))
    assert errs["bf16x3"] <= 1.3 * errs["f32"] + 2e-8 and errs["f32"] < 2e-6


@pytest.mark.parametrize("case", [(2, 23, 21, 64, 64, 3, 1, 1, 1),       # "same" 3x3, 64 -> 64 (layer0 / layer1 conv2): linear gather
                                  (2, 23, 21, 64, 256, 1, 1, 0, 1),      # 1x1, 64 input channels
                                  (2, 21, 21, 256, 64, 1, 2, 0, 1),      # strided 1x1, 64 output channels: generic gather
                                  (1, 33, 33, 64, 128, 3, 1, 1, 1),      # layer0.6: 64 -> 128, M = 1089 (K tail)
                                  (2, 15, 15, 256, 256, 1, 1, 0, 1)])    # a 128-multiple layer on a grid small enough for the 64 x 64 rule
def test_conv_wgrad_arith_bf16x3_64_tiles(case, report, monkeypatch):
    """The SP instance of the 64 x 64 register-staged weight-gradient kernel (waves 0-1 stage dy, waves 2-3 stage x) under
    SEMSEG_ARITH_BF16X3, next to the exact-fp32 64 x 64 kernel (SEMSEG_DEBUG wgrad_sp64=0) on the same operands, against fp64: rms
    within 2x of the fp32 kernel's (+1e-7), the criterion of the 128 x 128 instances."""
    from semseg_amd import ops
    N, H, W, Ci, Co, k, s_, p_, d = case
    g = torch.Generator().manual_seed(37)
    x = torch.relu(torch.randn(N, Ci, H, W, generator=g))
    Ho, Wo = ops.conv_out(H, k, s_, p_, d), ops.conv_out(W, k, s_, p_, d)
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, k, k), dy.double(), stride=s_, padding=p_, dilation=d)
    ldx, ldy = Ci + 32, ops.roundup(Co, 128)
    xd = torch.randn(N, H, W, ldx, device=DEV)
    xd[..., :Ci] = nhwc(x).to(DEV)
    dyd = torch.zeros(N, Ho, Wo, ldy, device=DEV)
    dyd[..., :Co] = nhwc(dy).to(DEV)
    scratch = torch.empty(1 << 24, device=DEV)
    rms = lambda a: float((a.cpu().double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    errs = []
    for sp64 in ("0", "1"):
        monkeypatch.setenv("SEMSEG_DEBUG", "wgrad_sp64=" + sp64)
        dw = torch.full((Co, Ci, k, k), float("nan"), device=DEV)
        ops.conv_wgrad(xd, ldx, dyd, ldy, dw, scratch, N, H, W, Ci, Co, k, k, s_, p_, d, arith=ops.ARITH_BF16X3)
        torch.cuda.synchronize()
        errs.append(rms(dw))
    report("conv_wgrad bf16x3 on 64x64 tiles %s: rms %.2e (exact-fp32 64x64 kernel %.2e)" % (case, errs[1], errs[0]))
    assert errs[1] <= 2 * errs[0] + 1e-7 and errs[1] < 2e-5


@pytest.mark.parametrize("case", [(3, 13, 11, 128, 128, 3, 1, 2, 2), (2, 21, 21, 128, 256, 3, 2, 1, 1)])
def test_conv_arith_bf16x3_3x3(case, report):
    """SEMSEG_ARITH_BF16X3 per launch: the SP instances of the 3x3 (unrolled-tap) forward / data-gradient kernel, dilated
    and strided, next to the fp32 instances against fp64: rms within 2x (+1e-7)."""
    from semseg_amd import ops
    N, H, W, Ci, Co, k, s_, p_, d = case
    g = torch.Generator().manual_seed(41)
    x = torch.relu(torch.randn(N, Ci, H, W, generator=g))
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    y64 = F.conv2d(x.double(), w.double(), None, s_, p_, d)
    Ho, Wo = y64.shape[2:]
    dy = torch.randn(N, Co, Ho, Wo, generator=g)
    dx64 = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), stride=s_, padding=p_, dilation=d)
    pk = ops.PackedConv(Co, Ci, k, k, DEV)
    pk.pack(w.to(DEV))
    xd, dyd = nhwc(x).contiguous().to(DEV), nhwc(dy).contiguous().to(DEV)
    rms = lambda a, ref: float((a.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    res = {}
    for split in (False, True):
        ar = ops.ARITH_BF16X3 if split else ops.ARITH_F32
        yb = torch.empty(N, Ho, Wo, Co, device=DEV)
        ops.conv_fwd(xd, Ci, pk, yb, Co, N, H, W, s_, p_, d, scratch=torch.empty(1 << 24, device=DEV), arith=ar)
        dxb = torch.empty(N, H, W, Ci, device=DEV)
        ops.conv_dgrad(dyd, Co, pk, dxb, Ci, N, H, W, s_, p_, d, scratch=torch.empty(1 << 24, device=DEV), arith=ar)
        torch.cuda.synchronize()
        res[split] = (rms(nchw(yb).cpu(), y64), rms(nchw(dxb).cpu(), dx64))
    report("conv 3x3 split mode %s: forward rms %.2e (fp32 %.2e)  data gradient %.2e (fp32 %.2e)"
           % (case, res[True][0], res[False][0], res[True][1], res[False][1]))
    assert res[True][0] <= 2 * res[False][0] + 1e-7 and res[True][1] <= 2 * res[False][1] + 1e-7


@pytest.mark.parametrize("case", [
    # N, H, W, Ci, Co, k, stride, pad, dil
    (2, 60, 60, 1024, 256, 1, 1, 0, 1),      # layer3 conv1 at per-GPU batch 2: 114 tiles of 128 x 128, four K slices
    (2, 60, 60, 2048, 512, 1, 1, 0, 1),      # layer4 conv1
    (2, 30, 30, 128, 128, 3, 1, 1, 1),       # 3x3, eight slices
    (7, 60, 60, 64, 256, 3, 1, 2, 2),        # full tiles and a split stream-K tail in one launch
    (2, 21, 21, 256, 150, 1, 1, 0, 1),       # a partial column tile
])
@pytest.mark.parametrize("arith", ["f32", "bf16x3"])
def test_split_k_reduced_inside_the_launch_equals_the_separate_reduction(case, arith, report, monkeypatch):
    """Split-K tiles of semseg_conv_fwd / semseg_conv_dgrad[_bnreduce] reduced by the last slice's workgroup (tile_counters given) vs by
    splitk_epilogue_kernel (rounds 1-5): the slabs are summed in slice order in both, so outputs are BIT-identical; statistics / fused
    sums differ only by the arrival order of their fp64 atomics.  Repeated: the arrival order of the slices varies run to run, the
    result must not.  The counters must be left zero."""
    from semseg_amd import ops
    N, H, W, Ci, Co, k, stride, pad, dil = case
    ar = ops.ARITH_F32 if arith == "f32" else ops.ARITH_BF16X3
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(N, H, W, Ci, generator=g).to(DEV)
    w = (torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5).to(DEV)
    pk = ops.PackedConv(Co, Ci, k, k, DEV)
    pk.pack(w)
    Ho, Wo = ops.conv_out(H, k, stride, pad, dil), ops.conv_out(W, k, stride, pad, dil)
    ldy = ops.roundup(Co, 4)
    scratch = torch.empty(32 * 1024 * 1024, device=DEV)
    NS = ops.NSLOT
    bias = torch.randn(Co, generator=g).to(DEV)

    def fwd(fused, with_stats):
        monkeypatch.setattr(ops, "FUSED_SPLIT", fused)
        y = torch.full((N, Ho, Wo, ldy), float("nan"), device=DEV)
        st = torch.zeros(NS * 2 * Co, dtype=torch.float64, device=DEV) if with_stats else None
        ops.conv_fwd(x, Ci, pk, y, ldy, N, H, W, stride, pad, dil, stats=st, nslot=NS, scratch=scratch, arith=ar,
                     bias=None if with_stats else bias)
        return y, st
    for with_stats in (True, False):
        y0, st0 = fwd(False, with_stats)
        for _ in range(6):
            y1, st1 = fwd(True, with_stats)
            assert torch.equal(y0[..., :Co], y1[..., :Co]), "forward %s: in-kernel reduction differs" % (case,)
            if with_stats:
                a, b = st0.view(NS, -1).sum(0), st1.view(NS, -1).sum(0)
                assert float(((a - b).abs() / (b.abs() + 1e-30)).max()) < 1e-12
    # data gradient with the fused BatchNorm-backward reduction (input channels of the dgrad = Ci must be % 32 for the bit mask)
    if Ci % 32 == 0:
        M = N * H * W
        ldd = ops.roundup(Co, 128)
        dy = torch.zeros(N, Ho, Wo, ldd, device=DEV)
        dy[..., :Co] = torch.randn(N, Ho, Wo, Co, generator=g).to(DEV)
        act = torch.relu(torch.randn(M, Ci, generator=g)).to(DEV)
        bits = relu_bits(act.view(N, H, W, Ci))
        ybn = (torch.randn(M, Ci, generator=g) * 2 + 0.5).to(DEV)
        mean, inv = torch.randn(Ci, generator=g).to(DEV), (torch.rand(Ci, generator=g) + 0.5).to(DEV)
        add0 = torch.randn(M, Ci, generator=g).to(DEV)

        def dgrad(fused):
            monkeypatch.setattr(ops, "FUSED_SPLIT", fused)
            dx = add0.clone()
            sums = torch.zeros(NS * 2 * Ci, dtype=torch.float64, device=DEV)
            ops.conv_dgrad_bnreduce(dy, ldd, pk, dx, Ci, N, H, W, stride, pad, dil, None, 0, [(ybn, Ci, mean, inv, sums)], NS,
                                    add=dx, ldadd=Ci, scratch=scratch, arith=ar, relu_bits=bits)
            return dx, sums
        d0, s0 = dgrad(False)
        for _ in range(6):
            d1, s1 = dgrad(True)
            assert torch.equal(d0, d1), "data gradient %s: in-kernel reduction differs" % (case,)
            a, b = s0.view(NS, -1).sum(0), s1.view(NS, -1).sum(0)
            assert float(((a - b).abs() / (b.abs() + 1e-30)).max()) < 2e-5     # (the kernel epilogue sums 8 rows per lane in fp32 before fp64, the separate launch sums in fp64)
    cnt = ops._CNT[(scratch.device.index, scratch.data_ptr())]
    torch.cuda.synchronize()
    assert int(cnt.abs().sum().item()) == 0, "tile counters not left at zero"
    report("split-K reduced inside the launch %s [%s]: outputs bit-identical to the separate reduction launch, statistics equal to 1e-12, fused sums to 2e-5" % (case, arith))
