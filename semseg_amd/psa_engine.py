"""PSA head (model/psanet.py:53-98) on the HIP engine.

Per branch (collect = psa_type 0, distribute = 1; psa_type 2 runs both):
  reduce 1x1+BN+ReLU -> bilinear shrink -> attention (1x1+BN+ReLU, 1x1 -> mask taps)
  -> psamask (pixel-major affinity rows A[n,q,p]) -> softmax over p (x 1/normalization_factor)
  -> point-affinity contraction z[n,q,:] = sum_p A[n,q,p] x[n,p,:]   (torch.bmm, psanet.py:90-91)
then concat -> proj 1x1+BN+ReLU -> bilinear expand -> concat with the trunk feature.

The contraction and both of its gradients run on the matrix-core kernels: forward and dA through the
1x1 implicit-GEMM kernel (A rows are K-contiguous by construction; x is transposed once per step so
the B operand is K-contiguous too), dx through the K-major (weight-gradient) kernel.
"""
import torch

from . import ops
from .engine import Act

PADROWS = 128  # GEMM B panels are read in 128-row tiles: keep one tile of slack behind the last image


def _padded_act(eng, N, h, w, C, tag):
    t = eng.buf((N * h * w + PADROWS, C), zero=True, tag=tag)
    return Act(t[:N * h * w].view(N, h, w, C), N, h, w, C, C, tag)


def _branch(eng, x4, red, att, typ, psa, zcat, zoff):
    N = x4.N
    sf = psa.shrink_factor
    H, W = x4.H, x4.W
    if sf != 1:
        xr = eng.conv_bn(x4, red[0], red[1])
        h, w = (H - 1) // sf + 1, (W - 1) // sf + 1
        xs = _padded_act(eng, N, h, w, xr.C, "psa_xs")
        ops.bilinear_fwd(xr.data, xr.ld, xs.data, xs.ld, N, H, W, h, w, xr.C)
        if eng.training:
            def bwd_shrink():
                g = eng.grad_of(xr)
                ops.bilinear_bwd(xs.grad, xs.ld, g, xr.ld, N, H, W, h, w, xr.C)
                xr.ginit = True
            eng.push("upsample", bwd_shrink, x=xr, dy=lambda: xs.grad, lddy=xs.ld, Ho=h, Wo=w)
    else:
        h, w = H, W
        xs = eng.conv_bn(x4, red[0], red[1], out=_padded_act(eng, N, h, w, red[0].weight.shape[0], "psa_xs"))
    hw = h * w
    C = xs.C
    a1 = eng.conv_bn(xs, att[0], att[1])
    ym = eng.conv(a1, att[3])                       # [N,h,w,taps] (ld padded, pad = 0)
    P = ops.roundup(hw, 128)                        # affinity row stride (zero padded)
    aff = eng.buf((N * hw + PADROWS, P), zero=True, tag="psa_aff")
    mH, mW = psa.mask_h, psa.mask_w
    alpha = 1.0 / psa.normalization_factor
    if psa.compact:
        assert ym.C == hw, "compact PSA needs mask_h*mask_w == h*w"
        if typ == 1:
            raw = eng.buf((N * hw + PADROWS, P), zero=True, tag="psa_aff_raw")
            ops.transpose_batched(ym.data, ym.ld, hw * ym.ld, raw, P, hw * P, N, hw, hw)
            ops.softmax_rows_fwd(raw, P, aff, P, N * hw, hw, alpha, psa.psa_softmax)
        else:
            ops.softmax_rows_fwd(ym.data, ym.ld, aff, P, N * hw, hw, alpha, psa.psa_softmax)
    else:
        raw = eng.buf((N * hw + PADROWS, P), zero=True, tag="psa_aff_raw")
        ops.psamask_nhwc_forward(typ, ym.data, ym.ld, raw, P, N, h, w, mH, mW)
        ops.softmax_rows_fwd(raw, P, aff, P, N * hw, hw, alpha, psa.psa_softmax)
    # B operand of the contraction: xT[n][c][p], K(p)-contiguous, zero padded to P
    xT = eng.buf((N * C + PADROWS, P), zero=True, tag="psa_xT")
    ops.transpose_batched(xs.data, xs.ld, hw * xs.ld, xT, P, C * P, N, hw, C)
    zdst = zcat.slice(zoff, C)
    # z[n] = A[n] x[n] for every image in ONE launch (blockIdx.y = image)
    gflops = 2.0 * N * hw * hw * C      # algorithmic FLOPs of one contraction (the zero padding of P is not counted)
    ev = eng._t0("conv_igemm_kernel<128,128,false,1>(+splitk_epilogue)", gflops)
    ops.gemm_rows_batched(aff, P, hw * P, xT, C * P, zdst.data, zdst.ld, hw * zdst.ld, hw, P, C, N)
    eng._t1(ev)
    if eng.training:
        def bwd_contract():
            gz = zcat.grad[..., zoff:]
            daff = eng.buf((N * hw + PADROWS, P), zero=True, tag="psa_daff")
            gxs = eng.grad_of(xs)
            scr = eng.scratch()
            # dA[n][q,p] = sum_c dz[q,c] x[p,c]   (B^T rows = p, K = c: x in its native layout), all images at once
            ev = eng._t0("conv_igemm_kernel<128,128,false,1>(+splitk_epilogue)", gflops)
            ops.gemm_rows_batched(gz, zcat.ld, hw * zcat.ld, xs.data, hw * xs.ld, daff, P, hw * P, hw, C, hw, N)
            eng._t1(ev)
            ev = eng._t0("conv_wgrad_dma_kernel<128x128>+reduce", gflops)
            # dx[n][p,c] = sum_q A[q,p] dz[q,c]   (K-major GEMM, all images at once)
            ops.gemm_kmajor_batched(gz, zcat.ld, hw * zcat.ld, aff, P, hw * P, gxs, hw * xs.ld, scr,
                                    hw, C, hw, N, accumulate=xs.ginit)
            eng._t1(ev)
            xs.ginit = True
            if ym.grad is None and not psa.compact:
                # zeroed ONCE (Engine.buf keeps it for the life of the engine) and written by the psamask backward only, which touches
                # the in-window taps: the out-of-window taps of a fixed geometry are the same elements every step
                ym.grad = eng.buf((ym.N, ym.H, ym.W, ym.ld), zero=True, tag="g:psa_ym")
            gym = eng.grad_of(ym)
            if psa.compact and typ == 0:
                ops.softmax_rows_bwd(aff, P, daff, P, gym, ym.ld, N * hw, hw, alpha, psa.psa_softmax)
            else:
                ops.softmax_rows_bwd(aff, P, daff, P, daff, P, N * hw, hw, alpha, psa.psa_softmax)
                if psa.compact:
                    ops.transpose_batched(daff, P, hw * P, gym, ym.ld, hw * ym.ld, N, hw, hw)
                else:
                    ops.psamask_nhwc_backward(typ, daff, P, gym, ym.ld, N, h, w, mH, mW, prezeroed=True)
            ym.ginit = True
        # appended last => runs first in backward, before the attention convs' backward
        eng.push("psa_contract", bwd_contract, xs=xs, ym=ym, aff=aff, zcat=zcat, zoff=zoff, typ=typ, psa=psa, P=P,
                 h=h, w=w, alpha=alpha)
    return h, w


def psa_forward(eng, x4, cat):
    """x4 = cat[..., :2048] (layer4 output, written in place by the trunk); fills cat[..., 2048:]."""
    m = eng.model.psa
    N = x4.N
    nb = 2 if m.psa_type == 2 else 1
    sf = m.shrink_factor
    h, w = ((x4.H - 1) // sf + 1, (x4.W - 1) // sf + 1) if sf != 1 else (x4.H, x4.W)
    zcat = eng.act(N, h, w, 512 * nb, tag="psa_z")
    if m.psa_type == 2:
        _branch(eng, x4, m.reduce, m.attention, 0, m, zcat, 0)
        _branch(eng, x4, m.reduce_p, m.attention_p, 1, m, zcat, 512)
    else:
        _branch(eng, x4, m.reduce, m.attention, m.psa_type, m, zcat, 0)
    dst = cat.slice(2048, 2048)
    if sf != 1:
        ap = eng.conv_bn(zcat, m.proj[0], m.proj[1])
        Ho, Wo = (h - 1) * sf + 1, (w - 1) * sf + 1
        assert (Ho, Wo) == (x4.H, x4.W), "PSA expand size must match the trunk feature map"
        ops.bilinear_fwd(ap.data, ap.ld, dst.data, dst.ld, N, h, w, Ho, Wo, ap.C)
        if eng.training:
            def bwd_expand():
                g = eng.grad_of(ap)
                ops.bilinear_bwd(cat.grad[..., 2048:], cat.ld, g, ap.ld, N, h, w, Ho, Wo, ap.C)
                ap.ginit = True
            eng.push("upsample", bwd_expand, x=ap, dy=lambda: cat.grad[..., 2048:], lddy=cat.ld, Ho=Ho, Wo=Wo)
    else:
        eng.conv_bn(zcat, m.proj[0], m.proj[1], out=dst)
        if eng.training:
            def link2():
                dst.grad = cat.grad[..., 2048:]
                dst.ginit = True
            eng.push("link", link2)
    if eng.training:
        def link():
            # cls' data-gradient wrote cat.grad; its first 2048 channels are x4's gradient so far:
            # the reduce convs' data-gradients accumulate into it
            x4.grad = cat.grad
            x4.ginit = True
        eng.push("link", link)  # appended last => runs first among the PSA backward closures
    return cat
