import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_model_gpu import build, inputs, rel
from oracle import segnet
arch, layers, classes, size, batch = "psp", 50, 21, 73, 2
m, sd = build(arch, layers, classes)
x, y = inputs(batch, size, classes)
sd_t = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
        for k, v in sd.items()}
p_ref, ml_ref, al_ref = segnet.forward(sd_t, x, layers, arch, training=True, y=y)
(ml_ref + 0.4 * al_ref).backward()
m = m.cuda().train()
pred, ml, al = m(x.cuda(), y.cuda())
(ml + 0.4 * al).backward()
for k, p in m.named_parameters():
    print("%-40s %.2e  |ref| %.3e" % (k, rel(p.grad, sd_t[k].grad), float(sd_t[k].grad.abs().max())))
