"""nn.Module wrapper, as lib/psa/modules/psamask.py:5-15 (unused by the models).  The reference's
constructor check `mask_H_ in None` (modules/psamask.py:9) raises TypeError; the intended check is
implemented here."""
from torch import nn

from .. import functional as F


class PSAMask(nn.Module):
    def __init__(self, psa_type=0, mask_H_=None, mask_W_=None):
        super().__init__()
        assert psa_type in [0, 1]
        assert (mask_H_ is None and mask_W_ is None) or (mask_H_ is not None and mask_W_ is not None)
        self.psa_type, self.mask_H_, self.mask_W_ = psa_type, mask_H_, mask_W_

    def forward(self, input):
        return F.psa_mask(input, self.psa_type, self.mask_H_, self.mask_W_)
