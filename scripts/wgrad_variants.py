"""Interleaved A/B of the weight-gradient kernel variants (SEMSEG_WGRAD_DMA = 0 register-staged, 1..5 direct-to-LDS
rings) on the PSPNet-101 bs16 473^2 shapes that run the 128 x 128 tile, with a numerical cross-check of every variant
against variant 0 and against an fp64 reference on one small shape.  python scripts/wgrad_variants.py [bs] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
VARS = [int(v) for v in os.environ.get("VARIANTS", "0,1,2,3,4,5").split(",")]
SHAPES = [  # name, H, Ci, Co, k, stride, pad, dil, count(R101)
    ("stem3 64->128 3x3 @237", 237, 64, 128, 3, 1, 1, 1, 0),   # Ci % 128 != 0 -> 64-tile kernel, listed for reference
    ("l2 conv2 128->128 3x3 @60", 60, 128, 128, 3, 1, 1, 1, 3),
    ("l3 conv1 1024->256 1x1", 60, 1024, 256, 1, 1, 0, 1, 22),
    ("l3 conv2 256->256 3x3 d2", 60, 256, 256, 3, 1, 2, 2, 23),
    ("l3 conv3 256->1024 1x1", 60, 256, 1024, 1, 1, 0, 1, 23),
    ("l4 conv1 2048->512 1x1", 60, 2048, 512, 1, 1, 0, 1, 2),
    ("l4 conv2 512->512 3x3 d4", 60, 512, 512, 3, 1, 4, 4, 3),
    ("l4 conv3 512->2048 1x1", 60, 512, 2048, 1, 1, 0, 1, 3),
    ("l4 ds 1024->2048 1x1", 60, 1024, 2048, 1, 1, 0, 1, 1),
    ("cls.0 4096->512 3x3", 60, 4096, 512, 3, 1, 1, 1, 1),
    ("aux.0 1024->256 3x3", 60, 1024, 256, 3, 1, 1, 1, 1),
    ("l2 ds 256->512 1x1 s2 @119", 119, 256, 512, 1, 2, 0, 1, 1),
    ("l2.0 conv2 128->128 3x3 s2", 119, 128, 128, 3, 2, 1, 1, 1),
]
dev = "cuda"
scratch = torch.empty(64 * 1024 * 1024, device=dev)


def setv(v):
    os.environ["SEMSEG_WGRAD_DMA"] = str(v)


def check_fp64():
    os.environ["SEMSEG_WGRAD_SMALL"] = "0"      # tiny shapes would otherwise take the 64 x 64 kernel
    _check_fp64()
    os.environ.pop("SEMSEG_WGRAD_SMALL")


def _check_fp64():
    """One small 3x3 dilated + one strided case against fp64 (all variants): catches a wrong OOB / tail behaviour."""
    import torch.nn.functional as F
    for (H, Ci, Co, k, s, p, d, n) in [(13, 128, 128, 3, 1, 2, 2, 3), (17, 128, 256, 3, 2, 1, 1, 2), (9, 256, 128, 1, 1, 0, 1, 5)]:
        g = torch.Generator().manual_seed(H)
        x = torch.randn(n, Ci, H, H, generator=g)
        Ho = ops.conv_out(H, k, s, p, d)
        dy = torch.randn(n, Co, Ho, Ho, generator=g)
        ref = torch.nn.grad.conv2d_weight(x.double(), (Co, Ci, k, k), dy.double(), stride=s, padding=p, dilation=d)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        dyd = dy.permute(0, 2, 3, 1).contiguous().to(dev)
        for v in VARS:
            setv(v)
            dw = torch.full((Co, Ci, k, k), float("nan"), device=dev)
            ops.conv_wgrad(xd, Ci, dyd, Co, dw, scratch, n, H, H, Ci, Co, k, k, s, p, d)
            e = float((dw.cpu().double() - ref).abs().max() / ref.abs().max())
            print("fp64 check H=%d %d->%d k%d s%d d%d variant %d: rel err %.2e %s" % (H, Ci, Co, k, s, d, v, e, "OK" if e < 2e-5 else "FAIL"))


check_fp64()
tot = {v: 0.0 for v in VARS}
print("%-30s %8s |" % ("shape", "GF") + "".join("   v%d us    TF |" % v for v in VARS))
for name, H, Ci, Co, k, s, p, d, cnt in SHAPES:
    Ho = ops.conv_out(H, k, s, p, d)
    x = torch.randn(N, H, H, Ci, device=dev)
    ldy = Co
    dy = torch.randn(N, Ho, Ho, ldy, device=dev)
    fl = 2.0 * N * Ho * Ho * Co * Ci * k * k
    outs, times = {}, {v: [] for v in VARS}
    for v in VARS:
        setv(v)
        dw = torch.empty(Co, Ci, k, k, device=dev)
        ops.conv_wgrad(x, Ci, dy, ldy, dw, scratch, N, H, H, Ci, Co, k, k, s, p, d)
        outs[v] = dw
    torch.cuda.synchronize()
    base = outs[VARS[0]]
    errs = {v: float((outs[v] - base).abs().max() / base.abs().max()) for v in VARS}
    for r in range(ROUNDS):
        for v in VARS:
            setv(v)
            dw = outs[v]
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record()
            for _ in range(4):
                ops.conv_wgrad(x, Ci, dy, ldy, dw, scratch, N, H, H, Ci, Co, k, k, s, p, d)
            e_.record()
            torch.cuda.synchronize()
            times[v].append(s_.elapsed_time(e_) / 4 * 1e3)
    med = {v: sorted(times[v])[len(times[v]) // 2] for v in VARS}
    print("%-30s %8.1f |" % (name, fl / 1e9) + "".join(" %7.1f %5.1f |" % (med[v], fl / med[v] / 1e6) for v in VARS) +
          "  maxdiff vs v%d: %s" % (VARS[0], " ".join("%.1e" % errs[v] for v in VARS[1:])))
    for v in VARS:
        tot[v] += med[v] * cnt
print("weighted totals per step (ms, incl. reduce):", {v: round(t / 1e3, 2) for v, t in tot.items()})
