"""Per-shape times of the matrix-core launches inside a train step (one stream, HIP events around each launch).

    python scripts/per_shape.py [batch] [steps]          # PSPNet-101 473x473, 150 classes

Groups the KernelTimer records of `steps` steps by (family, FLOPs of the launch): launches per step, us per launch,
TFLOP/s (fp32-equivalent), ms per step.  The FLOP count identifies the layer shape inside a family.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    from model.pspnet import PSPNet
    from semseg_amd import engine as E
    from semseg_amd.trainer import Trainer
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = PSPNet(layers=101, classes=150, zoom_factor=8, pretrained=False).to(dev).train()
    tr = Trainer(m, base_lr=0.01, momentum=0.9, weight_decay=1e-4, aux_weight=0.4, sync_bn=True)
    g = torch.Generator().manual_seed(1000)
    x = torch.randn(B, 3, 473, 473, generator=g).to(dev)
    y = torch.randint(0, 150, (B, 473, 473), generator=g).to(dev)
    for _ in range(3):
        tr.step(x, y, 0.01)
    kt = E.KernelTimer()
    # data-gradient launches are tagged with their shape and epilogue (the FLOP count alone does not tell conv1 from conv3)
    from semseg_amd import ops
    tags = []

    def wrap(name, fused):
        inner = getattr(ops, name)

        def f(dy, lddy, pk, *a, **k):
            tags.append("%d->%d%s%s" % (pk.Co, pk.Ci, " +bnr" if fused else "", " +add" if k.get("add") is not None else ""))
            return inner(dy, lddy, pk, *a, **k)
        setattr(ops, name, f)
    wrap("conv_dgrad_bnreduce", True)
    wrap("conv_dgrad", False)
    for e in tr.engines.values():
        e.side_wgrad, e.hipri_main, e.ktimer = False, False, kt
    for _ in range(steps):
        tr.step(x, y, 0.01)
    torch.cuda.synchronize()
    grp = {}
    ti = 0
    for family, flops, s, e in kt.rec:
        if (family.startswith("conv_igemm_kernel") and ",true," in family) or "<3,16,2>" in family:
            family = family.split("(")[0] + " " + tags[ti]
            ti += 1
        d = grp.setdefault((family, flops), [0, 0.0])
        d[0] += 1
        d[1] += s.elapsed_time(e) * 1e-3
    rows = sorted(grp.items(), key=lambda kv: -kv[1][1])
    print("%-62s %8s %5s %9s %8s %8s" % ("family", "GFLOP", "n", "us", "TF|TB/s", "ms/step"))
    for (family, flops), (n, t) in rows:
        rate = abs(flops) * n / t / 1e12
        print("%-62s %8.2f %5.1f %9.1f %8.1f %8.3f" % (family[:62], abs(flops) / 1e9, n / steps, t / n * 1e6, rate,
                                                       t / steps * 1e3))


if __name__ == "__main__":
    main()
