"""PSPNet — drop-in for the reference's `model.pspnet.PSPNet` (constructor, forward contract,
sub-module attributes and state-dict keys: reference model/pspnet.py:8-105), executed by hand-written
gfx950 kernels through semseg_amd.engine instead of torch ops.
"""
from torch import nn

import model.resnet as models
from semseg_amd.module_base import HipSegModule, _holder_forward


class PPM(nn.Module):
    """Pyramid pooling parameters: features[i] = (AdaptiveAvgPool2d(bin), conv1x1, bn, relu)."""
    forward = _holder_forward

    def __init__(self, in_dim, reduction_dim, bins):
        super().__init__()
        self.features = nn.ModuleList([
            nn.Sequential(nn.AdaptiveAvgPool2d(b), nn.Conv2d(in_dim, reduction_dim, 1, bias=False),
                          nn.BatchNorm2d(reduction_dim), nn.ReLU(inplace=True)) for b in bins])


def seg_head(cin, mid, classes, dropout):
    return nn.Sequential(nn.Conv2d(cin, mid, kernel_size=3, padding=1, bias=False),
                         nn.BatchNorm2d(mid), nn.ReLU(inplace=True), nn.Dropout2d(p=dropout),
                         nn.Conv2d(mid, classes, kernel_size=1))


class PSPNet(HipSegModule):
    kind = "psp"

    def __init__(self, layers=50, bins=(1, 2, 3, 6), dropout=0.1, classes=2, zoom_factor=8,
                 use_ppm=True, criterion=nn.CrossEntropyLoss(ignore_index=255), pretrained=True):
        super().__init__()
        assert layers in [50, 101, 152]
        assert 2048 % len(bins) == 0
        assert classes > 1
        assert zoom_factor in [1, 2, 4, 8]
        self.zoom_factor = zoom_factor
        self.use_ppm = use_ppm
        self.criterion = criterion
        trunk = models.build_trunk(layers, pretrained)
        self.layer0 = trunk.stem()
        self.layer1, self.layer2, self.layer3, self.layer4 = (trunk.layer1, trunk.layer2,
                                                              trunk.layer3, trunk.layer4)
        fea_dim = 2048
        if use_ppm:
            assert len(bins) <= 4
            self.ppm = PPM(fea_dim, int(fea_dim / len(bins)), bins)
            fea_dim *= 2
        self.cls = seg_head(fea_dim, 512, classes, dropout)
        if self.training:
            self.aux = seg_head(1024, 256, classes, dropout)
