"""Per-shape timing of the implicit-GEMM conv kernels (fwd / dgrad / wgrad) on the PSPNet101 bs16
473^2 layer shapes.  python scripts/conv_bench.py [bs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SHAPES = [  # name, H, Ci, Co, k, stride, pad, dil, count(R101)
    ("stem2 64->64 3x3 @237", 237, 64, 64, 3, 1, 1, 1, 1),
    ("stem3 64->128 3x3 @237", 237, 64, 128, 3, 1, 1, 1, 1),
    ("l1 conv2 64->64 3x3 @119", 119, 64, 64, 3, 1, 1, 1, 3),
    ("l1 conv3 64->256 1x1 @119", 119, 64, 256, 1, 1, 0, 1, 3),
    ("l1 conv1 256->64 1x1 @119", 119, 256, 64, 1, 1, 0, 1, 2),
    ("l2 conv2 128->128 3x3 @60", 60, 128, 128, 3, 1, 1, 1, 3),
    ("l3 conv1 1024->256 1x1", 60, 1024, 256, 1, 1, 0, 1, 22),
    ("l3 conv2 256->256 3x3 d2", 60, 256, 256, 3, 1, 2, 2, 23),
    ("l3 conv3 256->1024 1x1", 60, 256, 1024, 1, 1, 0, 1, 23),
    ("l4 conv1 2048->512 1x1", 60, 2048, 512, 1, 1, 0, 1, 2),
    ("l4 conv2 512->512 3x3 d4", 60, 512, 512, 3, 1, 4, 4, 3),
    ("l4 conv3 512->2048 1x1", 60, 512, 2048, 1, 1, 0, 1, 3),
    ("cls.0 4096->512 3x3", 60, 4096, 512, 3, 1, 1, 1, 1),
    ("aux.0 1024->256 3x3", 60, 1024, 256, 3, 1, 1, 1, 1),
    ("cls.4 512->150 1x1", 60, 512, 150, 1, 1, 0, 1, 1),
]
dev = "cuda"
scratch = torch.empty(64 * 1024 * 1024, device=dev)
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
print("%-28s %8s | %8s %6s | %8s %6s | %8s %6s" % ("shape", "GF", "fwd us", "TF", "dgrad us", "TF", "wgrad us", "TF"))
for name, H, Ci, Co, k, s, p, d, cnt in SHAPES:
    W = H
    Ho = ops.conv_out(H, k, s, p, d)
    pk = ops.PackedConv(Co, Ci, k, k, dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    pk.pack(w)
    LDXP, LDYP = int(os.environ.get("LDX_PAD", "0")), int(os.environ.get("LDY_PAD", "0"))
    ldx = Ci + LDXP
    x = torch.randn(N, H, W, ldx, device=dev)
    ldy = (Co if Co % 64 == 0 else ops.roundup(Co, 128)) + LDYP
    y = torch.zeros(N, Ho, Ho, ldy, device=dev)
    dy = torch.zeros(N, Ho, Ho, ldy, device=dev); dy[..., :Co].normal_()
    dx = torch.empty(N, H, W, ldx, device=dev)
    dw = torch.empty(Co, Ci, k, k, device=dev)
    stats8 = torch.zeros(2 * Co * ops.NSLOT, dtype=torch.float64, device=dev)
    fl = 2.0 * N * Ho * Ho * Co * Ci * k * k
    def timeit(fn, it=5):
        fn(); torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(it): fn()
        e_.record(); torch.cuda.synchronize()
        return s_.elapsed_time(e_) / it * 1e3
    st_arg = None if os.environ.get('NOSTATS') else stats8
    tf = timeit(lambda: ops.conv_fwd(x, ldx, pk, y, ldy, N, H, W, s, p, d, stats=st_arg, nslot=ops.NSLOT, scratch=scratch))
    td = timeit(lambda: ops.conv_dgrad(dy, ldy, pk, dx, ldx, N, H, W, s, p, d, scratch=scratch))
    tw = timeit(lambda: ops.conv_wgrad(x, ldx, dy, ldy, dw, scratch, N, H, W, Ci, Co, k, k, s, p, d))
    print("%-28s %8.1f | %8.1f %6.1f | %8.1f %6.1f | %8.1f %6.1f" % (name, fl / 1e9, tf, fl / tf / 1e6, td, fl / td / 1e6, tw, fl / tw / 1e6))
    tot["fwd"] += tf * cnt; tot["dgrad"] += td * cnt; tot["wgrad"] += tw * cnt
print("weighted totals (ms):", {k: round(v / 1e3, 2) for k, v in tot.items()})
