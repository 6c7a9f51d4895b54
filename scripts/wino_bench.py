"""Winograd F(2x2,3x3) path vs the direct implicit-GEMM kernels on the stride-1 3x3 shapes of PSPNet-101 473^2:
per direction, the three steps of the Winograd path timed separately.  python scripts/wino_bench.py [bs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
SHAPES = [("l3 conv2 256->256 d2", 60, 256, 256, 2, 23), ("l4 conv2 512->512 d4", 60, 512, 512, 4, 3),
          ("aux.0 1024->256", 60, 1024, 256, 1, 1), ("cls.0 4096->512", 60, 4096, 512, 1, 1),
          ("l2 conv2 128->128", 60, 128, 128, 1, 3), ("l1 conv2 64->64 @119", 119, 64, 64, 1, 3),
          ("stem3 64->128 @237", 237, 64, 128, 1, 1)]
dev = "cuda"
scratch = torch.empty(64 * 1024 * 1024, device=dev)


def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


tot = {"direct": 0.0, "wino": 0.0}
for name, H, Ci, Co, d, cnt in SHAPES:
    W = H
    pk = ops.PackedConv(Co, Ci, 3, 3, dev)
    w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.02
    pk.pack(w)
    wc = ops.WinoConv(Co, Ci, dev)
    wc.transform(w)
    x = torch.randn(N, H, W, Ci, device=dev)
    ldy = ops.roundup(Co, 128) if Co % 64 else Co
    y = torch.zeros(N, H, W, ldy, device=dev)
    dy = torch.randn(N, H, W, ldy, device=dev)
    dx = torch.empty(N, H, W, Ci, device=dev)
    dw = torch.empty(Co, Ci, 3, 3, device=dev)
    st = torch.zeros(2 * Co * ops.NSLOT, dtype=torch.float64, device=dev)
    T = ops.wino_tiles(N, H, W, d)
    V = torch.empty(16 * T * Ci, device=dev)
    Vdy = torch.empty(16 * T * wc.Kc, device=dev)
    Mbuf = torch.empty(16 * T * max(Ci, Co), device=dev)
    Yh = torch.zeros(16 * T * ops.roundup(Co, 128), device=dev)
    dU = torch.empty(16 * Co * Ci, device=dev)
    fl = 2.0 * N * H * W * Co * Ci * 9
    f_d = timeit(lambda: ops.conv_fwd(x, Ci, pk, y, ldy, N, H, W, 1, d, d, stats=st, nslot=ops.NSLOT, scratch=scratch))
    d_d = timeit(lambda: ops.conv_dgrad(dy, ldy, pk, dx, Ci, N, H, W, 1, d, d, scratch=scratch))
    w_d = timeit(lambda: ops.conv_wgrad(x, Ci, dy, ldy, dw, scratch, N, H, W, Ci, Co, 3, 3, 1, d, d))
    f_w = timeit(lambda: ops.wino_conv_fwd(x, Ci, wc, y, ldy, N, H, W, d, V, Mbuf, stats=st, nslot=ops.NSLOT))
    d_w = timeit(lambda: ops.wino_conv_dgrad(dy, ldy, wc, dx, Ci, N, H, W, d, Vdy, Mbuf))
    w_w = timeit(lambda: ops.wino_conv_wgrad(V, dy, ldy, wc, dw, N, H, W, d, Yh, dU, scratch))
    # the steps of the forward path
    t_in = timeit(lambda: ops.wino_input_transform(x, Ci, V, N, H, W, Ci, d))
    t_g = timeit(lambda: ops.gemm_rows_batched(V, Ci, T * Ci, wc.U_fwd, wc.Co_pad * Ci, Mbuf, Co, T * Co, T, Ci, Co, 16))
    t_out = timeit(lambda: ops.wino_output_transform(Mbuf, Co, y, ldy, N, H, W, Co, d, stats=st, nslot=ops.NSLOT))
    t_f = timeit(lambda: wc.transform(w))
    print("%-22s direct fwd/dgrad/wgrad %8.1f %8.1f %8.1f us (%.0f TF) | winograd %8.1f %8.1f %8.1f us | fwd steps: in %.1f gemm %.1f "
          "(%.0f TF) out %.1f | filter transform %.1f" % (name, f_d, d_d, w_d, fl / f_d / 1e6, f_w, d_w, w_w, t_in, t_g,
                                                           fl / 2.25 / t_g / 1e6, t_out, t_f), flush=True)
    tot["direct"] += (f_d + d_d + w_d) * cnt
    tot["wino"] += (f_w + d_w + w_w + t_f) * cnt
print("per step (ms, weighted by layer count):", {k: round(v / 1e3, 2) for k, v in tot.items()})
