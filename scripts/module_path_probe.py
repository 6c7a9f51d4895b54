"""Where the drop-in nn.Module loop (tool/train.py:269-276) spends its step compared with semseg_amd.Trainer:
HIP-event spans around forward / loss / zero_grad / backward / optimizer.step.  python scripts/module_path_probe.py [bs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from model.pspnet import PSPNet

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
torch.manual_seed(0)
model = PSPNet(layers=101, classes=150, zoom_factor=8, pretrained=False)
groups = [dict(params=m.parameters(), lr=0.01) for m in (model.layer0, model.layer1, model.layer2, model.layer3, model.layer4)]
groups += [dict(params=m.parameters(), lr=0.1) for m in (model.ppm, model.cls, model.aux)]
opt = torch.optim.SGD(groups, lr=0.01, momentum=0.9, weight_decay=1e-4)
model = model.to(dev).train()
x = torch.randn(bs, 3, 473, 473, device=dev)
y = torch.randint(0, 150, (bs, 473, 473), device=dev)


def step(spans=None):
    def ev():
        e = torch.cuda.Event(enable_timing=True); e.record(); return e
    t = [time.time()]
    e0 = ev(); out, ml, al = model(x, y); t.append(time.time())
    e1 = ev(); loss = ml + 0.4 * al; opt.zero_grad(); t.append(time.time())
    e2 = ev(); loss.backward(); t.append(time.time())
    e3 = ev(); opt.step(); t.append(time.time())
    e4 = ev()
    if spans is not None:
        torch.cuda.synchronize()
        spans.append(([a.elapsed_time(b) for a, b in ((e0, e1), (e1, e2), (e2, e3), (e3, e4))],
                      [1e3 * (b - a) for a, b in zip(t, t[1:])]))


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    step()
torch.cuda.synchronize()
print("module path, free running: %.2f ms/step" % ((time.time() - t0) / 5 * 1e3))
sp = []
for _ in range(4):
    step(sp)
for g, h in sp:
    print("gpu ms fwd %.2f | loss+zero_grad %.2f | backward %.2f | optimizer %.2f   host ms fwd %.1f zero %.1f bwd %.1f opt %.1f"
          % (g[0], g[1], g[2], g[3], h[0], h[1], h[2], h[3]))
from semseg_amd.trainer import Trainer
del model, opt
torch.cuda.empty_cache()
torch.manual_seed(0)
m2 = PSPNet(layers=101, classes=150, zoom_factor=8, pretrained=False).to(dev).train()
tr = Trainer(m2, sync_bn=True)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    tr.step(x, y)
torch.cuda.synchronize()
print("Trainer, free running: %.2f ms/step" % ((time.time() - t0) / 5 * 1e3))
