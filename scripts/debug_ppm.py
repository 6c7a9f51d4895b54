import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from semseg_amd import ops
from tests.test_ops_gpu import relerr, nhwc, nchw
DEV="cuda"
for hi, ho in [(3,60),(3,10),(3,90),(2,60),(6,60)]:
    N, C = 2, 512
    g = torch.Generator().manual_seed(hi*100+ho)
    x = torch.randn(N, C, hi, hi, generator=g).double().requires_grad_(True)
    y = F.interpolate(x, (ho, ho), mode="bilinear", align_corners=True)
    dy = torch.randn(y.shape, generator=g).double()
    y.backward(dy)
    ld = 4096
    dyb = torch.zeros(N, ho, ho, ld, device=DEV)
    dyb[..., 3072:3072+C] = nhwc(dy.float()).to(DEV)
    dx = torch.empty(N, hi, hi, C, device=DEV)
    ops.bilinear_bwd(dyb[..., 3072:], ld, dx, C, N, hi, hi, ho, ho, C)
    print("bilinear bwd %d->%d: %.2e" % (hi, ho, relerr(nchw(dx), x.grad)))
    # per low-res pixel error
    d = (nchw(dx).double().cpu() - x.grad).abs().amax(dim=(0,1))
    print(d)
