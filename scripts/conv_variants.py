"""Interleaved A/B of a per-launch environment switch of the forward / data-gradient kernel (VAR, default
SEMSEG_CONV_TL: two-level accumulation rule) on the PSPNet-101 bs16 473^2 shapes.  python scripts/conv_variants.py [bs] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
VARS = [int(v) for v in os.environ.get("VARIANTS", "0,1").split(",")]
VAR = os.environ.get("VAR", "SEMSEG_CONV_TL")    # e.g. VAR=SEMSEG_CONV_TL to A/B the two-level accumulation rule
SHAPES = [  # name, H, Ci, Co, k, stride, pad, dil, count(R101)
    ("stem3 64->128 3x3 @237", 237, 64, 128, 3, 1, 1, 1, 1),
    ("l1 conv3 64->256 1x1 @119", 119, 64, 256, 1, 1, 0, 1, 3),
    ("l2 conv2 128->128 3x3 @60", 60, 128, 128, 3, 1, 1, 1, 3),
    ("l3 conv1 1024->256 1x1", 60, 1024, 256, 1, 1, 0, 1, 22),
    ("l3 conv2 256->256 3x3 d2", 60, 256, 256, 3, 1, 2, 2, 23),
    ("l3 conv3 256->1024 1x1", 60, 256, 1024, 1, 1, 0, 1, 23),
    ("l4 conv1 2048->512 1x1", 60, 2048, 512, 1, 1, 0, 1, 2),
    ("l4 conv2 512->512 3x3 d4", 60, 512, 512, 3, 1, 4, 4, 3),
    ("l4 conv3 512->2048 1x1", 60, 512, 2048, 1, 1, 0, 1, 3),
    ("cls.0 4096->512 3x3", 60, 4096, 512, 3, 1, 1, 1, 1),
    ("aux.0 1024->256 3x3", 60, 1024, 256, 3, 1, 1, 1, 1),
]
dev = "cuda"
scratch = torch.empty(64 * 1024 * 1024, device=dev)
tot = {(v, d): 0.0 for v in VARS for d in ("fwd", "dgrad")}
print("%-28s %8s |" % ("shape", "GF") + "".join("  v%d fwd us   TF | v%d dgrad us  TF |" % (v, v) for v in VARS))
for name, H, Ci, Co, k, s, p, d, cnt in SHAPES:
    Ho = ops.conv_out(H, k, s, p, d)
    pk = ops.PackedConv(Co, Ci, k, k, dev)
    pk.pack(torch.randn(Co, Ci, k, k, device=dev) * 0.05)
    x = torch.randn(N, H, H, Ci, device=dev)
    ldy = Co if Co % 64 == 0 else ops.roundup(Co, 128)
    y = torch.zeros(N, Ho, Ho, ldy, device=dev)
    dy = torch.zeros(N, Ho, Ho, ldy, device=dev); dy[..., :Co].normal_()
    dx = torch.empty(N, H, H, Ci, device=dev)
    st = torch.zeros(2 * Co * ops.NSLOT, dtype=torch.float64, device=dev)
    fl = 2.0 * N * Ho * Ho * Co * Ci * k * k
    fns = {"fwd": lambda: ops.conv_fwd(x, Ci, pk, y, ldy, N, H, H, s, p, d, stats=st, nslot=ops.NSLOT, scratch=scratch),
           "dgrad": lambda: ops.conv_dgrad(dy, ldy, pk, dx, Ci, N, H, H, s, p, d, scratch=scratch)}
    outs = {}
    for v in VARS:
        os.environ[VAR] = str(v)
        fns["fwd"](); fns["dgrad"]()
        outs[v] = (y.clone(), dx.clone())
    torch.cuda.synchronize()
    same = all(torch.equal(outs[v][0], outs[VARS[0]][0]) and torch.equal(outs[v][1], outs[VARS[0]][1]) for v in VARS)
    times = {(v, dd): [] for v in VARS for dd in fns}
    for r in range(ROUNDS):
        for v in VARS:
            os.environ[VAR] = str(v)
            for dd, fn in fns.items():
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record()
                for _ in range(4):
                    fn()
                e_.record(); torch.cuda.synchronize()
                times[(v, dd)].append(s_.elapsed_time(e_) / 4 * 1e3)
    med = {kk: sorted(vv)[len(vv) // 2] for kk, vv in times.items()}
    print("%-28s %8.1f |" % (name, fl / 1e9) + "".join(" %9.1f %5.1f | %9.1f %5.1f |" % (
        med[(v, "fwd")], fl / med[(v, "fwd")] / 1e6, med[(v, "dgrad")], fl / med[(v, "dgrad")] / 1e6) for v in VARS) +
        ("  bit-identical" if same else "  DIFFERENT RESULTS"))
    for kk in med:
        tot[kk] += med[kk] * cnt
print("weighted totals per step (ms):", {"v%d %s" % kk: round(t / 1e3, 2) for kk, t in tot.items()})
