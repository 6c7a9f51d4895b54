"""Host enqueue time vs GPU time per train step at several per-GPU batch sizes (what N-GPU scaling sees)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from model.pspnet import PSPNet
from semseg_amd.trainer import Trainer
dev = torch.device("cuda", 0)
m = PSPNet(layers=101, classes=150, pretrained=False).to(dev).train()
tr = Trainer(m)
for B in (16, 8, 4, 2):
    x = torch.randn(B, 3, 473, 473, device=dev); y = torch.randint(0, 150, (B, 473, 473), device=dev)
    for _ in range(2): tr.step(x, y)
    torch.cuda.synchronize()
    host, tot = [], []
    for _ in range(3):
        t0 = time.time(); tr.step(x, y); t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
        host.append(t1 - t0); tot.append(t2 - t0)
    print("B=%2d host enqueue %.1f ms  total %.1f ms  -> %.1f img/s per GPU" % (B, min(host) * 1e3, min(tot) * 1e3, B / min(tot)))
