"""Accuracy of the train-mode BN forward (channel_stats -> bn_finalize -> bn_apply) against an fp64 evaluation,
next to torch's CPU fp32 F.batch_norm, for the ill-conditioned shapes of the PPM branches (few rows per channel,
|mean| >> std).  Errors are max|out - out64| over all elements (outputs are O(1))."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from semseg_amd import ops

dev = torch.device("cuda")
C = 512
for M, B, sig in [(2, 1.0, 1.0), (2, 1.0, 0.01), (2, 5.0, 0.01), (8, 1.0, 0.01), (128, 3.0, 0.1), (128, 0.0, 1.0)]:
    errs_h, errs_c = [], []
    for seed in range(6):
        g = torch.Generator().manual_seed(seed)
        x = (B * torch.randn(1, C, generator=g) + sig * torch.randn(M, C, generator=g)).float()
        gamma = 1 + 0.1 * torch.randn(C, generator=g)
        beta = 0.1 * torch.randn(C, generator=g)
        x64 = x.double()
        mu = x64.mean(0); var = x64.var(0, unbiased=False)
        ref = (x64 - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double()
        cpu = F.batch_norm(x.t().reshape(1, C, M).permute(2, 1, 0).contiguous(), None, None, gamma, beta, True, 0.1, 1e-5)
        cpu = cpu.reshape(M, C)
        xd = x.to(dev)
        nslot = ops.NSLOT
        stats = torch.zeros(nslot * 2 * C, dtype=torch.float64, device=dev)
        ops.channel_stats(xd, C, stats, M, C, nslot)
        mean = torch.empty(C, device=dev); invstd = torch.empty(C, device=dev)
        scale = torch.empty(C, device=dev); shift = torch.empty(C, device=dev)
        rm = torch.zeros(C, device=dev); rv = torch.ones(C, device=dev)
        nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        ops.bn_finalize(stats, M, gamma.to(dev), beta.to(dev), rm, rv, nbt, 0.1, 1e-5, mean, invstd, scale, shift, C, nslot)
        out = torch.empty(M, C, device=dev)
        ops.bn_apply(xd, C, scale, shift, out, C, M, C, M, False)
        torch.cuda.synchronize()
        errs_h.append(float((out.cpu().double() - ref).abs().max()))
        errs_c.append(float((cpu.double() - ref).abs().max()))
    med = lambda v: sorted(v)[len(v) // 2]
    print("M %4d |mean|~%.1f std~%.2f | hip max-abs-err median %.2e (worst %.2e) | torch-cpu-fp32 %.2e (worst %.2e)"
          % (M, B, sig, med(errs_h), max(errs_h), med(errs_c), max(errs_c)), flush=True)
