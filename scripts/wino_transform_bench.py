"""Times the Winograd transform kernels (semseg_amd/csrc/winograd.hip) alone at the shapes of the PSPNet-101 473x473 batch-16 step.

    python scripts/wino_transform_bench.py [--batch 16] [--reps 20]
    SEMSEG_HIP_LIB=gpurun_variants/lib_X.so python scripts/wino_transform_bench.py     # a variant library

Every kernel is timed with HIP events over `reps` launches that rotate through enough buffer sets to exceed the 256 MB
infinity cache, so a launch never finds its operands cached by the previous one.  Prints one line per (shape, kernel):
us per launch, algorithmic GB moved, TB/s.  Count per step: layer2 x3, layer3 x23, layer4 x3, cls.0, aux.0; per conv one
input + one output(stats) transform forward, one input + one output(bnreduce) transform in the data gradient, one dy
transform in the weight gradient.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semseg_amd import ops  # noqa: E402

SHAPES = [  # name, Ci, Co, dilation, convs per step
    ("layer2", 128, 128, 1, 3),
    ("layer3", 256, 256, 2, 23),
    ("layer4", 512, 512, 4, 3),
    ("cls.0", 4096, 512, 1, 1),
    ("aux.0", 1024, 256, 1, 1),
]


def timed(fn, nset, reps):
    for i in range(nset):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % nset)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--hw", type=int, default=60)
    ap.add_argument("--reps", type=int, default=24)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    N, H, W = a.batch, a.hw, a.hw
    total = {}
    print("lib:", os.environ.get("SEMSEG_HIP_LIB", "default"))
    for name, Ci, Co, d, cnt in SHAPES:
        T = ops.wino_tiles(N, H, W, d)
        px = N * H * W

        def sets(nbytes):
            return max(2, min(8, int(600e6 // nbytes) + 1))

        # forward / data-gradient input transform: C channels -> V[16][T][C]
        for tag, C in (("input(Ci)", Ci), ("input(Co)", Co)):
            ns = sets(4 * (px * C + 16 * T * C))
            xs = [torch.randn(px, C, device=dev) for _ in range(ns)]
            Vs = [torch.empty(16, T, C, device=dev) for _ in range(ns)]
            us = timed(lambda i: ops.wino_input_transform(xs[i], C, Vs[i], N, H, W, C, d), ns, a.reps)
            gb = 4e-9 * (px * C + 16 * T * C)
            print("%-7s %-22s %8.1f us  %6.3f GB  %5.2f TB/s" % (name, tag, us, gb, gb / us * 1e3))
            total[tag] = total.get(tag, 0.0) + us * cnt
            del xs, Vs
        # forward output transform with statistics: M[16][T][Co] -> y
        C = Co
        ns = sets(4 * (px * C + 16 * T * C))
        Ms = [torch.randn(16, T, C, device=dev) for _ in range(ns)]
        ys = [torch.empty(px, C, device=dev) for _ in range(ns)]
        st = torch.zeros(ops.NSLOT * 2 * C, dtype=torch.float64, device=dev)
        us = timed(lambda i: ops.wino_output_transform(Ms[i], C, ys[i], C, N, H, W, C, d, stats=st, nslot=ops.NSLOT), ns,
                   a.reps)
        gb = 4e-9 * (px * C + 16 * T * C)
        print("%-7s %-22s %8.1f us  %6.3f GB  %5.2f TB/s" % (name, "output+stats(Co)", us, gb, gb / us * 1e3))
        total["output+stats"] = total.get("output+stats", 0.0) + us * cnt
        us = timed(lambda i: ops.wino_output_transform(Ms[i], C, ys[i], C, N, H, W, C, d), ns, a.reps)
        print("%-7s %-22s %8.1f us  %6.3f GB  %5.2f TB/s" % (name, "output plain(Co)", us, gb, gb / us * 1e3))
        del Ms, ys
        # data-gradient output transform with the fused BatchNorm-backward reduction: M[16][T][Ci] -> dx, + act, ybn
        C = Ci
        ns = sets(4 * (3 * px * C + 16 * T * C))
        Ms = [torch.randn(16, T, C, device=dev) for _ in range(ns)]
        ys = [torch.empty(px, C, device=dev) for _ in range(ns)]
        acts = [torch.randn(px, C, device=dev).clamp_(min=0) for _ in range(ns)]
        ybn = [torch.randn(px, C, device=dev) for _ in range(ns)]
        mean = torch.zeros(C, device=dev)
        invstd = torch.ones(C, device=dev)
        st = torch.zeros(ops.NSLOT * 2 * C, dtype=torch.float64, device=dev)
        us = timed(lambda i: ops.wino_output_transform_bnreduce(Ms[i], C, ys[i], C, N, H, W, C, d, acts[i], C, ybn[i], C,
                                                                mean, invstd, st, ops.NSLOT), ns, a.reps)
        gb = 4e-9 * (3 * px * C + 16 * T * C)
        print("%-7s %-22s %8.1f us  %6.3f GB  %5.2f TB/s" % (name, "output+bnreduce(Ci)", us, gb, gb / us * 1e3))
        total["output+bnreduce"] = total.get("output+bnreduce", 0.0) + us * cnt
        del Ms, ys, acts, ybn
        # weight-gradient dy transform
        C = Co
        ns = sets(4 * (px * C + 16 * T * C))
        dys = [torch.randn(px, C, device=dev) for _ in range(ns)]
        Yh = [torch.zeros(16, T, C, device=dev) for _ in range(ns)]
        us = timed(lambda i: ops.wino_dy_transform_wgrad(dys[i], C, Yh[i], C, N, H, W, C, d), ns, a.reps)
        gb = 4e-9 * (px * C + 16 * T * C)
        print("%-7s %-22s %8.1f us  %6.3f GB  %5.2f TB/s" % (name, "dy(wgrad)(Co)", us, gb, gb / us * 1e3))
        total["dy(wgrad)"] = total.get("dy(wgrad)", 0.0) + us * cnt
        del dys, Yh
        torch.cuda.empty_cache()
    print("per step (ms):", {k: round(v * 1e-3, 3) for k, v in total.items()},
          "sum %.3f" % (sum(total.values()) * 1e-3))


if __name__ == "__main__":
    main()
