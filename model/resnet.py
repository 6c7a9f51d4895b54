"""Parameter containers for the dilated deep-stem ResNet-50/101/152 trunk.

Drop-in for the reference's `model/resnet.py` as far as PSPNet/PSANet use it (state-dict key names,
shapes, initialisation, `./initmodel/resnet{L}_v2.pth` loading — reference model/resnet.py:58-72,
97-145,192-229).  The modules here hold parameters only: the arithmetic of the trunk runs in the HIP
engine (semseg_amd/engine.py), so their forward() refuses to run.
"""
import torch
from torch import nn

_DEPTHS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


from semseg_amd.module_base import _holder_forward


class Bottleneck(nn.Module):
    """conv1(1x1)-bn1-relu, conv2(3x3, stride/dilation)-bn2-relu, conv3(1x1)-bn3, (+downsample), relu."""
    expansion = 4
    forward = _holder_forward

    def __init__(self, cin, width, stride=1, dilation=1, project=False):
        super().__init__()
        cout = width * self.expansion
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=dilation, dilation=dilation,
                               bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(cout))
        self.stride = stride


class Trunk(nn.Module):
    """Namespace with the reference ResNet's attribute names (conv1..bn3, layer1..layer4) so that a
    `resnet{L}_v2.pth` checkpoint loads with strict=False exactly as in model/resnet.py:199-200."""
    forward = _holder_forward

    def __init__(self, depth):
        super().__init__()
        blocks = _DEPTHS[depth]
        self.conv1 = nn.Conv2d(3, 64, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.conv2 = nn.Conv2d(64, 64, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(64)
        self.conv3 = nn.Conv2d(64, 128, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(128)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        cin = 128
        # (width, stride of the first block, dilation): strides of layer3/4 are traded for dilation
        # 2/4 — the surgery the reference applies after construction (model/pspnet.py:49-58).
        spec = [(64, 1, 1), (128, 2, 1), (256, 1, 2), (512, 1, 4)]
        for i, ((width, stride, dil), n) in enumerate(zip(spec, blocks)):
            seq = []
            for b in range(n):
                seq.append(Bottleneck(cin, width, stride if b == 0 else 1, dil, project=(b == 0)))
                cin = width * Bottleneck.expansion
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*seq))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def stem(self):
        return nn.Sequential(self.conv1, self.bn1, self.relu, self.conv2, self.bn2, self.relu,
                             self.conv3, self.bn3, self.relu, self.maxpool)


def build_trunk(layers, pretrained):
    assert layers in _DEPTHS
    t = Trunk(layers)
    if pretrained:
        t.load_state_dict(torch.load("./initmodel/resnet%d_v2.pth" % layers), strict=False)
    return t


def resnet50(pretrained=False, **kw):
    return build_trunk(50, pretrained)


def resnet101(pretrained=False, **kw):
    return build_trunk(101, pretrained)


def resnet152(pretrained=False, **kw):
    return build_trunk(152, pretrained)
