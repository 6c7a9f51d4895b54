"""Throughput of the device-side multi-scale test pipeline (BASELINE.json configs[4]): PSPNet101, one
512x512 image, scales [0.5..1.75], base_size 512, crop 473 -> 23 crops x 2 flips = 46 forwards."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from model.pspnet import PSPNet
from semseg_amd.infer import MultiScaleTester
m = PSPNet(layers=101, classes=150, pretrained=False).cuda().eval()
t = MultiScaleTester(m, 150, 512, 473, 473, (0.5, 0.75, 1.0, 1.25, 1.5, 1.75), max_batch_crops=int(sys.argv[1]) if len(sys.argv) > 1 else 9)
img = torch.rand(512, 512, 3, device="cuda") * 255
for _ in range(2): t.predict(img)
torch.cuda.synchronize(); t0 = time.time(); n = 3
for _ in range(n): t.predict(img)
torch.cuda.synchronize(); dt = (time.time() - t0) / n
fl = 46 * 460.1e9
print("multi-scale test: %.1f ms / image (%d forwards of 473^2) -> %.2f images/s, %.1f TFLOP/s" % (dt * 1e3, t.num_forwards(512, 512), 1 / dt, fl / dt / 1e12))
