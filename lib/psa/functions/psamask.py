"""`PSAMask` autograd Function — same contract as the reference's lib/psa/functions/psamask.py:6-39
(argument checks, zero-filled [N, H*W, H, W] output, adjoint backward), with the native call routed to
the gfx950 kernels through the C ABI (semseg_psamask_forward/backward, include/semseg_hip.h) on the
*current* stream.  fp32 CUDA tensors only; anything else raises (the reference would read garbage
through `.data<float>()`; there is deliberately no CPU path here).
"""
import torch
from torch.autograd import Function

from semseg_amd import ops


def _check(t, what):
    if not t.is_cuda:
        raise RuntimeError("psa_mask: %s must live on the MI355X (cuda); semseg_amd has no CPU path" % what)
    if t.dtype != torch.float32:
        raise TypeError("psa_mask: %s must be float32 (lib/psa/src/cpu/psamask.cpp:117)" % what)


class PSAMask(Function):
    @staticmethod
    def forward(ctx, input, psa_type=0, mask_H_=None, mask_W_=None):
        assert psa_type in [0, 1]  # 0-col, 1-dis
        assert (mask_H_ is None and mask_W_ is None) or (mask_H_ is not None and mask_W_ is not None)
        num_, channels_, feature_H_, feature_W_ = input.size()
        if mask_H_ is None and mask_W_ is None:
            mask_H_, mask_W_ = 2 * feature_H_ - 1, 2 * feature_W_ - 1
        assert (mask_H_ % 2 == 1) and (mask_W_ % 2 == 1)
        assert channels_ == mask_H_ * mask_W_
        _check(input, "input")
        half_h, half_w = (mask_H_ - 1) // 2, (mask_W_ - 1) // 2
        output = torch.zeros([num_, feature_H_ * feature_W_, feature_H_, feature_W_], dtype=input.dtype,
                             device=input.device)
        ops.psamask_forward(psa_type, input.contiguous(), output, num_, feature_H_, feature_W_, mask_H_,
                            mask_W_, half_h, half_w)
        ctx.cfg = (psa_type, num_, channels_, feature_H_, feature_W_, mask_H_, mask_W_, half_h, half_w)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        psa_type, num_, channels_, fH, fW, mH, mW, half_h, half_w = ctx.cfg
        _check(grad_output, "grad_output")
        # the reference assumes a dense gradient (a stride-0 grad makes it read garbage)
        grad_output = grad_output.contiguous()
        grad_input = torch.zeros([num_, channels_, fH, fW], dtype=grad_output.dtype, device=grad_output.device)
        ops.psamask_backward(psa_type, grad_output, grad_input, num_, fH, fW, mH, mW, half_h, half_w)
        return grad_input, None, None, None


psa_mask = PSAMask.apply
