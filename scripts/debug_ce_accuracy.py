"""Loss accuracy of the fused upsample+CE forward on identical fp32 scores: HIP vs fp64, next to torch CPU fp32
vs fp64 (signed errors, to expose a systematic bias of the fast exp/log intrinsics)."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from semseg_amd import ops
DEV = torch.device("cuda")
for C, h, H, amp in [(11, 8, 8, 1.0), (11, 8, 57, 1.0), (21, 10, 73, 3.0), (150, 10, 73, 3.0), (19, 10, 73, 10.0), (11, 8, 8, 0.1)]:
    sh, sc = [], []
    for seed in range(6):
        N, w, W = 2, h, H
        ld = ops.roundup(C, 64) if C > 64 else 64
        g = torch.Generator().manual_seed(seed)
        z = (torch.randn(N, C, h, w, generator=g) * amp).float()
        lab = torch.randint(0, C, (N, H, W), generator=g)
        up64 = F.interpolate(z.double(), (H, W), mode="bilinear", align_corners=True)
        l64 = float(F.cross_entropy(up64, lab, ignore_index=255))
        l32 = float(F.cross_entropy(F.interpolate(z, (H, W), mode="bilinear", align_corners=True), lab, ignore_index=255))
        zb = torch.zeros(N, h, w, ld, device=DEV)
        zb[..., :C] = z.permute(0, 2, 3, 1).to(DEV)
        lse = torch.empty(N, H, W, device=DEV)
        pred = torch.empty(N, H, W, dtype=torch.int64, device=DEV)
        acc = torch.zeros(3, dtype=torch.float64, device=DEV)
        lossd = torch.empty(1, device=DEV)
        ops.ce_head_fwd(zb, ld, lab.to(DEV), lse, pred, acc, lossd, N, h, w, H, W, C, 255)
        # the fp64 accumulator before the final fp32 rounding of the loss
        lh = float(acc[0].item() / acc[1].item())
        sh.append((lh - l64) / l64); sc.append((l32 - l64) / l64)
    f = lambda v: " ".join("%+.1e" % q for q in v)
    print("C %3d %2d->%2d amp %4.1f | hip signed rel err: %s\n%29s| cpu32 signed rel err: %s" % (C, h, H, amp, f(sh), "", f(sc)), flush=True)
