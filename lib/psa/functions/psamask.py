"""`PSAMask` autograd Function — same contract as the reference's lib/psa/functions/psamask.py:6-39
(argument checks, zero-filled [N, H*W, H, W] output, adjoint backward).  The native call is the reference's own:
`src.gpu.psamask_forward(psa_type, input, output, num_, feature_H_, feature_W_, mask_H_, mask_W_, half_mask_H_,
half_mask_W_)` (psamask.py:20,35) on the pybind module `psamask_gpu` (lib/psa/src/gpu/operator.cpp), which forwards
to the gfx950 kernels through the C ABI (semseg_psamask_forward/backward, include/semseg_hip.h) on the *current*
stream.  fp32 CUDA tensors only; anything else raises (the reference would read garbage through `.data<float>()`;
there is deliberately no CPU path here).
"""
import torch
from torch.autograd import Function

from .. import src


def _check(t, what):
    if not t.is_cuda:
        raise RuntimeError("psa_mask: %s must live on the MI355X (cuda); semseg_amd has no CPU path" % what)
    if t.dtype != torch.float32:
        raise TypeError("psa_mask: %s must be float32 (lib/psa/src/cpu/psamask.cpp:117)" % what)


def _geometry(input, psa_type, mask_H_, mask_W_):
    """Argument checks of psamask.py:9-17 -> (N, mask channels, H, W, mask_H, mask_W, half_h, half_w)."""
    assert psa_type in [0, 1]  # 0 collect, 1 distribute
    assert (mask_H_ is None) == (mask_W_ is None)
    n, chans, fh, fw = input.size()
    if mask_H_ is None:
        mask_H_, mask_W_ = 2 * fh - 1, 2 * fw - 1
    assert mask_H_ % 2 == 1 and mask_W_ % 2 == 1
    assert chans == mask_H_ * mask_W_
    return n, chans, fh, fw, mask_H_, mask_W_, (mask_H_ - 1) // 2, (mask_W_ - 1) // 2


class PSAMask(Function):
    @staticmethod
    def forward(ctx, input, psa_type=0, mask_H_=None, mask_W_=None):
        geo = _geometry(input, psa_type, mask_H_, mask_W_)
        n, chans, fh, fw, mh, mw, hh, hw = geo
        _check(input, "input")
        # psamask.py:17 zero-fills the output because the kernel only writes in-window elements.  With a full-size mask
        # (mask >= 2 * feature - 1, the default) EVERY element is in the window and is written exactly once, so the
        # 4 * N * (H*W)^2-byte memset is skipped; smaller masks keep the zero fill.
        full = mh >= 2 * fh - 1 and mw >= 2 * fw - 1
        output = (torch.empty if full else torch.zeros)((n, fh * fw, fh, fw), dtype=input.dtype, device=input.device)
        src.gpu.psamask_forward(psa_type, input.contiguous(), output, n, fh, fw, mh, mw, hh, hw)
        ctx.cfg = (psa_type,) + geo
        return output

    @staticmethod
    def backward(ctx, grad_output):
        psa_type, n, chans, fh, fw, mh, mw, hh, hw = ctx.cfg
        _check(grad_output, "grad_output")
        # the reference assumes a dense gradient (a stride-0 grad makes it read garbage)
        grad_output = grad_output.contiguous()
        grad_input = torch.zeros((n, chans, fh, fw), dtype=grad_output.dtype, device=grad_output.device)
        src.gpu.psamask_backward(psa_type, grad_output, grad_input, n, fh, fw, mh, mw, hh, hw)
        return grad_input, None, None, None


psa_mask = PSAMask.apply
