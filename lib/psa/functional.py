"""Functional entry point of the PSA op, `lib.psa.functional.psa_mask` — the name and signature
model/psanet.py:4,75,94 import (reference lib/psa/functional.py:4-5).  The work is done by the autograd
Function in `lib/psa/functions/psamask.py`, which calls the gfx950 kernels through the C ABI."""
from .functions.psamask import PSAMask as _PSAMaskFunction

__all__ = ["psa_mask"]


def psa_mask(input, psa_type=0, mask_H_=None, mask_W_=None):
    """[N, mask_H_*mask_W_, H, W] attention logits -> [N, H*W, H, W] point-wise affinity
    (psa_type 0 = collect, 1 = distribute; mask size defaults to (2H-1) x (2W-1))."""
    return _PSAMaskFunction.apply(input, psa_type, mask_H_, mask_W_)
