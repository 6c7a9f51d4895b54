"""Seed sweep of the batch-2, 57x57, zoom-1 PSPNet50 case: main/aux loss and eval logits of the HIP path and of
the CPU fp32 oracle, both against the fp64 oracle."""
import sys, torch
sys.path.insert(0, ".")
from oracle import segnet
from model.pspnet import PSPNet
r = lambda a, b: abs(float(a) - float(b)) / abs(float(b))
rows = []
for seed in range(8):
    classes, size, batch, zoom = 11, 57, 2, 1
    m = PSPNet(layers=50, classes=classes, zoom_factor=zoom, dropout=0.0, pretrained=False)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = segnet.recipe_state_dict(shapes, seed=77 + seed)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(3 + seed)
    x = torch.randn(batch, 3, size, size, generator=g)
    hh = int((size - 1) / 8 * zoom + 1)
    y = torch.randint(0, classes, (batch, hh, hh), generator=g)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    cp = lambda d: {k: v.clone() for k, v in d.items()}
    with torch.no_grad():
        _, ml32, al32 = segnet.forward(cp(sd), x, 50, "psp", zoom_factor=zoom, training=True, y=y)
        _, ml64, al64 = segnet.forward(cp(sd64), x.double(), 50, "psp", zoom_factor=zoom, training=True, y=y)
        lg32 = segnet.forward(cp(sd), x, 50, "psp", zoom_factor=zoom, training=False)
        lg64 = segnet.forward(cp(sd64), x.double(), 50, "psp", zoom_factor=zoom, training=False)
        m = m.cuda().train()
        _, ml, al = m(x.cuda(), y.cuda())
        lg = m.eval()(x.cuda()).cpu().double()
    rl = lambda a: float((a.double() - lg64).abs().max() / lg64.abs().max())
    rows.append((r(ml, ml64), r(ml32, ml64), r(al, al64), r(al32, al64), rl(lg), rl(lg32)))
    print("seed %d | main hip %.2e cpu32 %.2e | aux hip %.2e cpu32 %.2e | eval logits hip %.2e cpu32 %.2e" % ((seed,) + rows[-1]),
          flush=True)
med = lambda i: sorted(q[i] for q in rows)[len(rows) // 2]
print("median | main hip %.2e cpu32 %.2e | aux hip %.2e cpu32 %.2e | eval logits hip %.2e cpu32 %.2e" % tuple(med(i) for i in range(6)))
