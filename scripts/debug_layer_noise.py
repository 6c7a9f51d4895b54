"""Layer-by-layer forward noise profile, train mode: relative max error of every pre-BN conv output of the HIP
engine and of the CPU fp32 oracle, both against the fp64 oracle (same weights, same input).  Shows whether the
HIP path's error grows uniformly (accumulation-order noise) or jumps at one kernel."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from oracle import segnet
from model.pspnet import PSPNet
from semseg_amd import engine as E

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
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

caps = {}
orig_bn = segnet._bn


def run_oracle(tag, s, xx):
    def spy(f, sdd, p, training):
        o = orig_bn(f, sdd, p, training)
        caps.setdefault(p, {})[tag] = (f.detach().double(), o.detach().double())
        return o
    segnet._bn = spy
    try:
        with torch.no_grad():
            return segnet.forward({k: v.clone() for k, v in s.items()}, xx, 50, "psp", zoom_factor=zoom, training=True, y=y)
    finally:
        segnet._bn = orig_bn


_, ml64, al64 = run_oracle("f64", sd64, x.double())
_, ml32, al32 = run_oracle("c32", sd, x)

m = m.cuda().train()
names = {mod: n for n, mod in m.named_modules()}
order = []
orig_bnact = E.Engine.bn_act


def bn_spy(self, y_, bm, **kw):
    o = orig_bnact(self, y_, bm, **kw)
    order.append((names[bm], y_))
    if kw.get("y2") is not None:
        order.append((names[kw["bm2"]], kw["y2"]))
    return o


E.Engine.bn_act = bn_spy
with torch.no_grad():
    _, ml, al = m(x.cuda(), y.cuda())
torch.cuda.synchronize()
r = lambda a, b: abs(float(a) - float(b)) / abs(float(b))
print("losses: main hip %.2e cpu32 %.2e | aux hip %.2e cpu32 %.2e" % (r(ml, ml64), r(ml32, ml64), r(al, al64), r(al32, al64)))
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
for name, act in order:
    if name not in caps:
        print("%-28s (no oracle capture)" % name); continue
    ref = caps[name]["f64"][0]
    c32 = caps[name]["c32"][0]
    h = act.data[..., :act.C].permute(0, 3, 1, 2).cpu().double()
    eh, ec = rel(h, ref), rel(c32, ref)
    print("%-28s rows %5d  conv-out err hip %.2e cpu32 %.2e ratio %5.1f" % (name, act.M, eh, ec, eh / max(ec, 1e-12)))
