"""The hand-derived fp64 formulas of tests/insitu.py (BatchNorm backward, PSA contraction / softmax / psamask adjoint)
agree with torch autograd through the oracle's own forward ops — so the in-situ GPU test compares the HIP kernels
against the right thing.  CPU only."""
import torch
import torch.nn.functional as F

from insitu import InsituChecker, conv_bwd_taps


def test_bn_backward_formula_matches_autograd():
    torch.manual_seed(0)
    y = torch.randn(3, 8, 5, 7, dtype=torch.float64, requires_grad=True)
    gamma = torch.randn(8, dtype=torch.float64, requires_grad=True)
    beta = torch.randn(8, dtype=torch.float64, requires_grad=True)
    res = torch.randn(3, 8, 5, 7, dtype=torch.float64)
    out = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5) + res)
    dout = torch.randn_like(out)
    gy, gg, gb = torch.autograd.grad(out, (y, gamma, beta), dout)
    mean = y.detach().mean((0, 2, 3))
    invstd = 1.0 / torch.sqrt(y.detach().var((0, 2, 3), unbiased=False) + 1e-5)
    g = dout * (out.detach() > 0)
    dy, dg, db = InsituChecker._bn_bwd(g, y.detach(), mean, invstd, gamma.detach(), 3 * 5 * 7, torch.float64)
    assert (dy - gy).abs().max() < 1e-12 and (dg - gg).abs().max() < 1e-12 and (db - gb).abs().max() < 1e-12


def _psa_ref(gz, xv, A, alpha, softmax, typ, mh, mw, h, w):
    """Same algebra as InsituChecker._chk_psa_contract.ref (non-compact)."""
    from oracle import segnet
    from oracle import psamask as pm
    N, hw, C = xv.shape
    dx = torch.einsum("nqp,nqc->npc", A, gz)
    dA = torch.einsum("nqc,npc->nqp", gz, xv)
    if softmax:
        sm = A / alpha
        t = dA * alpha
        draw = sm * (t - (t * sm).sum(-1, keepdim=True))
    else:
        draw = dA * alpha
    dref = draw.transpose(1, 2).reshape(N, hw, h, w).contiguous()
    dmask = segnet._perm(pm.psa_mask_backward, dref, typ, mh, mw)
    return dx, dmask


def test_psa_adjoint_formula_matches_autograd_of_the_oracle():
    """Forward as model/psanet.py:80-91 (psa_mask -> softmax(dim=1) -> bmm with 1/normalization_factor) built from
    the oracle's ops; its autograd gradients must equal the closed-form adjoint used in-situ."""
    from oracle import segnet
    torch.manual_seed(1)
    N, C, h, w, mh, mw = 2, 6, 5, 4, 5, 7
    hw = h * w
    for typ in (0, 1):
        for softmax, nf in ((True, 1.0), (False, 3.0)):
            mask = torch.randn(N, mh * mw, h, w, dtype=torch.float64, requires_grad=True)
            x = torch.randn(N, C, h, w, dtype=torch.float64, requires_grad=True)
            y = segnet._PsaMask.apply(mask, typ, mh, mw)                      # [N, HW, h, w]
            if softmax:
                y = F.softmax(y, dim=1)
            out = torch.bmm(x.view(N, C, hw), y.view(N, hw, hw)).view(N, C, h, w) * (1.0 / nf)
            gout = torch.randn_like(out)
            gm, gx = torch.autograd.grad(out, (mask, x), gout)
            # engine layout: A[n,q,p] = alpha * y[n,p,q];  z[n,q,c] = sum_p A[n,q,p] x[n,p,c]
            alpha = 1.0 / nf
            A = (y.detach().view(N, hw, hw).transpose(1, 2) * alpha).contiguous()
            gz = gout.view(N, C, hw).transpose(1, 2).contiguous()
            xv = x.detach().view(N, C, hw).transpose(1, 2).contiguous()
            dx, dmask = _psa_ref(gz, xv, A, alpha, softmax, typ, mh, mw, h, w)
            assert (dx.transpose(1, 2).reshape(N, C, h, w) - gx).abs().max() < 1e-12
            assert (dmask - gm).abs().max() < 1e-12


def test_per_tap_conv_backward_matches_torch():
    torch.manual_seed(2)
    for (N, Ci, Co, H, k, s, p, d) in [(2, 5, 7, 11, 3, 1, 2, 2), (2, 4, 6, 13, 3, 2, 1, 1), (3, 6, 4, 9, 1, 1, 0, 1),
                                       (2, 3, 5, 12, 1, 2, 0, 1), (1, 4, 4, 10, 3, 1, 4, 4)]:
        x = torch.randn(N, Ci, H, H, dtype=torch.float64)
        w = torch.randn(Co, Ci, k, k, dtype=torch.float64)
        Ho = (H + 2 * p - d * (k - 1) - 1) // s + 1
        dy = torch.randn(N, Co, Ho, Ho, dtype=torch.float64)
        gw, gx = conv_bwd_taps(x.float().double(), w.shape, w, dy, s, p, d)
        kw = dict(stride=s, padding=p, dilation=d)
        assert (gw - torch.nn.grad.conv2d_weight(x.float().double(), w.shape, dy, **kw)).abs().max() < 1e-10
        assert (gx - torch.nn.grad.conv2d_input(x.shape, w, dy, **kw)).abs().max() < 1e-10
