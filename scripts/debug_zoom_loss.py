"""Where does the train-mode main-loss deviation of the PPM path (HIP 0.5-2e-5 vs CPU-fp32 3e-7, both against
the fp64 oracle) come from?  (1) per-bin / per-batch split on the GPU; (2) sensitivity of the fp64 oracle's loss
to relative noise injected at the PPM 1x1-conv outputs (ahead of the train-mode BN)."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from oracle import segnet
from model.pspnet import PSPNet

r = lambda a, b: abs(float(a) - float(b)) / abs(float(b))


def case(zoom, bins, batch, gpu=True, classes=11, size=57):
    m = PSPNet(layers=50, classes=classes, zoom_factor=zoom, bins=bins, dropout=0.0, pretrained=False)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = segnet.recipe_state_dict(shapes, seed=77)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(batch, 3, size, size, generator=g)
    hh = int((size - 1) / 8 * zoom + 1)
    y = torch.randint(0, classes, (batch, hh, hh), generator=g)
    y[torch.rand(batch, hh, hh, generator=g) < 0.1] = 255
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    return m, sd, sd64, x, y


def oracle_losses(sd, x, y, zoom, bins):
    with torch.no_grad():
        _, ml, al = segnet.forward({k: v.clone() for k, v in sd.items()}, x, 50, "psp", bins=bins,
                                   zoom_factor=zoom, training=True, y=y)
    return ml, al


if torch.cuda.is_available():
    for bins, batch in [((1,), 2), ((2,), 2), ((1, 2, 3, 6), 2), ((1,), 4), ((1, 2, 3, 6), 4), ((1, 2, 3, 6), 8)]:
        m, sd, sd64, x, y = case(1, bins, batch)
        ml32, _ = oracle_losses(sd, x, y, 1, bins)
        ml64, _ = oracle_losses(sd64, x.double(), y, 1, bins)
        m = m.cuda().train()
        with torch.no_grad():
            _, ml, _ = m(x.cuda(), y.cuda())
        print("bins %-12s batch %d | main: hip-vs-f64 %.2e cpu32-vs-f64 %.2e" % (bins, batch, r(ml, ml64), r(ml32, ml64)),
              flush=True)

# (2) noise sensitivity, CPU only
orig_conv = F.conv2d
for bins, batch in [((1,), 2), ((1, 2, 3, 6), 2), ((1, 2, 3, 6), 8)]:
    m, sd, sd64, x, y = case(1, bins, batch)
    base, _ = oracle_losses(sd64, x.double(), y, 1, bins)
    for eps in (3e-7, 2e-6):
        devs = []
        for trial in range(5):
            gen = torch.Generator().manual_seed(100 + trial)
            ppm_w = {id(sd64_k) for sd64_k in ()}

            def noisy_ppm(sdd, xx, bb, training, gen=gen, eps=eps):
                outs = [xx]
                for i, b in enumerate(bb):
                    f = F.adaptive_avg_pool2d(xx, b)
                    f = F.conv2d(f, sdd["ppm.features.%d.1.weight" % i])
                    f = f * (1 + eps * torch.randn(f.shape, generator=gen, dtype=f.dtype))
                    f = F.relu(segnet._bn(f, sdd, "ppm.features.%d.2" % i, training))
                    outs.append(F.interpolate(f, xx.shape[2:], mode="bilinear", align_corners=True))
                return torch.cat(outs, 1)
            keep = segnet.ppm
            segnet.ppm = noisy_ppm
            try:
                ml, _ = oracle_losses(sd64, x.double(), y, 1, bins)
            finally:
                segnet.ppm = keep
            devs.append(r(ml, base))
        print("f64 oracle, bins %-12s batch %d, rel noise %.0e at PPM conv out -> main loss moves %.2e (max of 5: %.2e)"
              % (bins, batch, eps, sorted(devs)[2], max(devs)), flush=True)
