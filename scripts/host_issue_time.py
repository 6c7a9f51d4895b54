"""Host-side issue time of a train step against its GPU time: python scripts/host_issue_time.py [batch]
After a device synchronisation the step is issued (t_issue = until tr.step returns) and drained (t_total)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd.trainer import Trainer
from model.pspnet import PSPNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.manual_seed(0)
m = PSPNet(layers=101, classes=150, zoom_factor=8, pretrained=False).cuda().train()
tr = Trainer(m, base_lr=0.01, sync_bn=True)
x = torch.randn(B, 3, 473, 473).cuda()
y = torch.randint(0, 150, (B, 473, 473)).cuda()
for _ in range(3):
    tr.step(x, y, 0.01)
iss, tot = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(x, y, 0.01)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    iss.append((t1 - t0) * 1e3)
    tot.append((t2 - t0) * 1e3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.step(x, y, 0.01)
torch.cuda.synchronize()
back = (time.perf_counter() - t0) * 100
print("batch %d: issue %.2f ms (min %.2f), issue+drain %.2f ms, back-to-back %.2f ms/step"
      % (B, sum(iss) / len(iss), min(iss), sum(tot) / len(tot), back))
