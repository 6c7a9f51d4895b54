"""Rounding noise of the weight-gradient kernel (reduction over M = N*Ho*Wo pixels, split over workgroups) vs
fp64, next to torch CPU fp32, on identical fp32 operands."""
import sys, torch
sys.path.insert(0, ".")
from semseg_amd import ops
DEV = torch.device("cuda")
rms = lambda a, b: float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
for N, H, Ci, Co, k, dil in [(2, 8, 512, 512, 3, 1), (16, 60, 256, 256, 3, 2), (16, 60, 512, 512, 3, 4), (16, 60, 1024, 256, 1, 1), (16, 119, 64, 64, 3, 1)]:
    g = torch.Generator().manual_seed(N + Ci)
    x = torch.relu(torch.randn(N, Ci, H, H, generator=g))
    dy = torch.randn(N, Co, H, H, generator=g)
    p = dil * (k // 2)
    wshape = (Co, Ci, k, k)
    ref = torch.nn.grad.conv2d_weight(x.double(), wshape, dy.double(), 1, p, dil)
    cpu = torch.nn.grad.conv2d_weight(x, wshape, dy, 1, p, dil).double()
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    dw = torch.empty(wshape, device=DEV)
    scratch = torch.empty(64 * 1024 * 1024, device=DEV)  # the engine's arena size (engine.py:353): lets the launcher pick its split
    ops.conv_wgrad(xd, Ci, dyd, Co, dw, scratch, N, H, H, Ci, Co, k, k, 1, p, dil)
    torch.cuda.synchronize()
    print("wgrad M %6d Ci %4d Co %4d %dx%d | hip rms %.2e | torch-cpu rms %.2e | ratio %.1f"
          % (N * H * H, Ci, Co, k, k, rms(dw.cpu().double(), ref), rms(cpu, ref), rms(dw.cpu().double(), ref) / rms(cpu, ref)), flush=True)
