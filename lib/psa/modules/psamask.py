"""`lib.psa.modules.PSAMask`: the nn.Module face of the op (reference lib/psa/modules/psamask.py:5-15; the
models call the functional form, this class exists for API completeness).  The reference constructor's
`mask_H_ in None` (modules/psamask.py:9) raises TypeError for every argument; the check it meant - both mask
sizes given or both omitted - is what is implemented."""
from torch import nn

from ..functional import psa_mask


class PSAMask(nn.Module):
    def __init__(self, psa_type=0, mask_H_=None, mask_W_=None):
        super().__init__()
        if psa_type not in (0, 1):
            raise AssertionError("psa_type is 0 (collect) or 1 (distribute), got %r" % (psa_type,))
        if (mask_H_ is None) != (mask_W_ is None):
            raise AssertionError("give both mask_H_ and mask_W_, or neither")
        self.psa_type = psa_type
        self.mask_H_, self.mask_W_ = mask_H_, mask_W_

    def extra_repr(self):
        return "psa_type=%d, mask=%sx%s" % (self.psa_type, self.mask_H_, self.mask_W_)

    def forward(self, input):
        return psa_mask(input, self.psa_type, self.mask_H_, self.mask_W_)
