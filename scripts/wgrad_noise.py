"""Rounding noise of the weight-gradient kernels on one shape, per arithmetic and kernel variant, against fp64 — next to a plain
fp32 CPU-style blocked sum of the same products (torch fp32 matmul of 1024-pixel blocks summed in fp32).
python scripts/wgrad_noise.py [N] [H] [Ci] [Co]     (1x1 conv; layer3.0.conv1 of PSANet-101 465^2 batch 16: 16 59 512 256)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

N, H, Ci, Co = (int(a) for a in (sys.argv[1:5] + ["16", "59", "512", "256"][len(sys.argv) - 1:]))
W = H
M = N * H * W
dev = "cuda"
for dist in ("relu(randn) x randn", "heavy-tailed: relu(randn)^3 x randn * exp(2 randn)"):
    g = torch.Generator().manual_seed(7)
    x = torch.relu(torch.randn(M, Ci, generator=g))
    dy = torch.randn(M, Co, generator=g) * 1e-3
    if dist.startswith("heavy"):
        x = x ** 3
        dy = dy * torch.exp(2 * torch.randn(M, 1, generator=g))
    xd, dyd = x.to(dev), dy.to(dev)
    ref = (dyd.double().t() @ xd.double())                       # [Co, Ci]
    rr = float(ref.pow(2).mean().sqrt())
    rms = lambda a: float((a.double() - ref).pow(2).mean().sqrt()) / rr
    blk = sum((dyd[i:i + 1024].t() @ xd[i:i + 1024]) for i in range(0, M, 1024))
    print("%s  M=%d Ci=%d Co=%d: fp32 blocked matmul (1024-pixel blocks, rocBLAS) rms %.2e" % (dist, M, Ci, Co, rms(blk)))
    scratch = torch.empty(64 * 1024 * 1024, device=dev)
    ldy = ops.roundup(Co, 128)
    dyp = torch.zeros(M, ldy, device=dev)
    dyp[:, :Co] = dyd
    for name, arith, dbg in (("exact fp32", ops.ARITH_F32, ""), ("bf16x3 128x256 (policy 10)", ops.ARITH_BF16X3, "wgrad_sp=10"),
                             ("bf16x3 128x128 ring (8)", ops.ARITH_BF16X3, "wgrad_sp=8"), ("bf16x3 register-staged (0)", ops.ARITH_BF16X3, "wgrad_sp=0")):
        os.environ["SEMSEG_DEBUG"] = dbg
        dw = torch.empty(Co, Ci, 1, 1, device=dev)
        ops.conv_wgrad(xd, Ci, dyp, ldy, dw, scratch, N, H, W, Ci, Co, 1, 1, 1, 0, 1, arith=arith)
        torch.cuda.synchronize()
        print("   %-30s rms %.2e" % (name, rms(dw.view(Co, Ci))))
