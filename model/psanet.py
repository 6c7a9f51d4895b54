"""PSANet — drop-in for the reference's `model.psanet.PSANet` (constructor, forward contract,
sub-module attributes and state-dict keys: reference model/psanet.py:9-179), executed by hand-written
gfx950 kernels through semseg_amd.engine / semseg_amd.psa_engine instead of torch ops.
"""
from torch import nn

import model.resnet as models
from model.pspnet import seg_head
from semseg_amd.module_base import HipSegModule, _holder_forward


def _cbr(cin, cout):
    return [nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True)]


class PSA(nn.Module):
    """Point-wise spatial attention parameters: reduce / attention (collect branch), reduce_p /
    attention_p (distribute branch, psa_type 2 only) and proj."""
    forward = _holder_forward

    def __init__(self, in_channels=2048, mid_channels=512, psa_type=2, compact=False, shrink_factor=2,
                 mask_h=59, mask_w=59, normalization_factor=1.0, psa_softmax=True):
        super().__init__()
        assert psa_type in [0, 1, 2]
        self.psa_type = psa_type
        self.compact = compact
        self.shrink_factor = shrink_factor
        self.mask_h = mask_h
        self.mask_w = mask_w
        self.psa_softmax = psa_softmax
        if normalization_factor is None:
            normalization_factor = mask_h * mask_w
        self.normalization_factor = normalization_factor
        taps = mask_h * mask_w
        self.reduce = nn.Sequential(*_cbr(in_channels, mid_channels))
        self.attention = nn.Sequential(*_cbr(mid_channels, mid_channels),
                                       nn.Conv2d(mid_channels, taps, kernel_size=1, bias=False))
        if psa_type == 2:
            self.reduce_p = nn.Sequential(*_cbr(in_channels, mid_channels))
            self.attention_p = nn.Sequential(*_cbr(mid_channels, mid_channels),
                                             nn.Conv2d(mid_channels, taps, kernel_size=1, bias=False))
        self.proj = nn.Sequential(*_cbr(mid_channels * (2 if psa_type == 2 else 1), in_channels))


class PSANet(HipSegModule):
    kind = "psa"

    def __init__(self, layers=50, dropout=0.1, classes=2, zoom_factor=8, use_psa=True, psa_type=2,
                 compact=False, shrink_factor=2, mask_h=59, mask_w=59, normalization_factor=1.0,
                 psa_softmax=True, criterion=nn.CrossEntropyLoss(ignore_index=255), pretrained=True):
        super().__init__()
        assert layers in [50, 101, 152]
        assert classes > 1
        assert zoom_factor in [1, 2, 4, 8]
        assert psa_type in [0, 1, 2]
        self.zoom_factor = zoom_factor
        self.use_psa = use_psa
        self.criterion = criterion
        trunk = models.build_trunk(layers, pretrained)
        self.layer0 = trunk.stem()
        self.layer1, self.layer2, self.layer3, self.layer4 = (trunk.layer1, trunk.layer2,
                                                              trunk.layer3, trunk.layer4)
        fea_dim = 2048
        if use_psa:
            self.psa = PSA(fea_dim, 512, psa_type, compact, shrink_factor, mask_h, mask_w,
                           normalization_factor, psa_softmax)
            fea_dim *= 2
        self.cls = seg_head(fea_dim, 512, classes, dropout)
        if self.training:
            self.aux = seg_head(1024, 256, classes, dropout)
