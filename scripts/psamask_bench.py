"""Timing + effective HBM rate of the psamask kernels:
  * the C-ABI NCHW operator semseg_psamask_forward/backward (what lib.psa / psamask_gpu call), and
  * the engine's pixel-major form semseg_psamask_nhwc_forward/backward,
at the BASELINE.json configs[3] shape (N = 16, 30 x 30 feature map, 59 x 59 mask) and a 45 x 45 / 89 x 89 case
(705^2 Cityscapes PSANet).  Algorithmic bytes = SURVEY.md section 8(d): 2 * 4 * N * (HW)^2 (read the in-window taps +
write them); the backward additionally has to define the N * HW * mH * mW output, which the C-ABI contract leaves to the
caller's zero fill and the pixel-major form writes itself (counted as 'written').  python scripts/psamask_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd import ops

dev = "cuda"


def timeit(fn, it=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


rows = []
for (N, H, mH) in [(16, 30, 59), (16, 45, 89), (2, 30, 59)]:
    W, mW = H, mH
    HW, T = H * W, mH * mW
    alg = 2 * 4 * N * HW * HW
    for typ in (0, 1):
        x = torch.randn(N, T, H, W, device=dev)
        out = torch.zeros(N, HW, H, W, device=dev)
        gy = torch.randn(N, HW, H, W, device=dev)
        gin = torch.zeros(N, T, H, W, device=dev)
        hh = (mH - 1) // 2
        tf = timeit(lambda: ops.psamask_forward(typ, x, out, N, H, W, mH, mW, hh, hh))
        tb = timeit(lambda: ops.psamask_backward(typ, gy, gin, N, H, W, mH, mW, hh, hh))
        rows.append(("C-ABI NCHW", N, H, mH, typ, "fwd", tf, alg, alg))
        rows.append(("C-ABI NCHW", N, H, mH, typ, "bwd", tb, alg, alg))
        ldm = ops.roundup(T, 128)
        P = ops.roundup(HW, 128)
        m = torch.randn(N, HW, ldm, device=dev)
        aff = torch.zeros(N * HW + 128, P, device=dev)
        daff = torch.randn(N * HW + 128, P, device=dev)
        dm = torch.zeros(N, HW, ldm, device=dev)
        tf = timeit(lambda: ops.psamask_nhwc_forward(typ, m, ldm, aff, P, N, H, W, mH, mW))
        tb = timeit(lambda: ops.psamask_nhwc_backward(typ, daff, P, dm, ldm, N, H, W, mH, mW))
        rows.append(("pixel-major", N, H, mH, typ, "fwd", tf, alg, alg))
        # backward writes the whole tap row (zeros included)
        rows.append(("pixel-major", N, H, mH, typ, "bwd", tb, alg, 4 * N * HW * HW + 4 * N * HW * ldm))
        # the engine's form (round 6): gradient buffer zeroed once, only in-window taps written
        dm.zero_()
        tz = timeit(lambda: ops.psamask_nhwc_backward(typ, daff, P, dm, ldm, N, H, W, mH, mW, prezeroed=True))
        rows.append(("px-major 0'd", N, H, mH, typ, "bwd", tz, alg, alg))
print("%-12s %3s %3s %3s %4s %4s %9s %12s %10s %12s %10s" % ("form", "N", "H", "mH", "type", "dir", "us", "alg MB", "alg TB/s",
                                                           "moved MB", "moved TB/s"))
for form, N, H, mH, typ, d, t, alg, moved in rows:
    print("%-12s %3d %3d %3d %4d %4s %9.1f %12.1f %10.2f %12.1f %10.2f" % (form, N, H, mH, typ, d, t, alg / 1e6, alg / t / 1e6,
                                                                     moved / 1e6, moved / t / 1e6))
