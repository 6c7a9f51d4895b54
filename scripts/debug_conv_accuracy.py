"""Rounding noise of the conv forward vs reduction length K = Ci*R*S: HIP (fp32 MFMA, sequential along K within a
split) and torch CPU fp32, both against fp64, on identical fp32 operands.  With and without split-K scratch."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from semseg_amd import ops
DEV = torch.device("cuda")
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
rms = lambda a, b: float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
for Ci, k, Co, N, H in [(64, 1, 256, 2, 15), (512, 1, 128, 2, 8), (2048, 1, 512, 2, 8), (512, 3, 512, 2, 8),
                        (4096, 3, 512, 2, 8), (4096, 3, 512, 8, 8)]:
    g = torch.Generator().manual_seed(Ci + k)
    x = torch.relu(torch.randn(N, Ci, H, H, generator=g))          # post-ReLU activations: positive mean
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    p = k // 2
    ref = F.conv2d(x.double(), w.double(), None, 1, p)
    cpu = F.conv2d(x, w, None, 1, p).double()
    pk = ops.PackedConv(Co, Ci, k, k, DEV); pk.pack(w.to(DEV))
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    res = []
    for scratch in (None, torch.empty(64 * 1024 * 1024, device=DEV)):
        y = torch.empty(N, H, H, Co, device=DEV)
        ops.conv_fwd(xd, Ci, pk, y, Co, N, H, H, 1, p, 1, scratch=scratch)
        torch.cuda.synchronize()
        yy = y.permute(0, 3, 1, 2).cpu().double()
        res.append((rel(yy, ref), rms(yy, ref)))
    print("K %6d rows %4d | hip no-split max %.2e rms %.2e | hip split-K max %.2e rms %.2e | torch-cpu max %.2e rms %.2e"
          % (Ci * k * k, N * H * H, res[0][0], res[0][1], res[1][0], res[1][1], rel(cpu, ref), rms(cpu, ref)), flush=True)
