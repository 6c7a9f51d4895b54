"""One Trainer configuration, one JSON line: ms per train step (timed like bench.py, no barrier needed on one GPU) and — with
FAMILIES=1 — the serialized per-family HIP-event table of two more steps.  Everything else comes from the environment
(SEMSEG_HIP_LIB, SEMSEG_ARITH, SEMSEG_DEBUG=side_wgrad=0,..., ...), so that scripts/ab_libs.py can A/B libraries and switches process
by process.   python scripts/step_time.py [batch] [steps] [arch] [size] [classes]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd.trainer import Trainer
from semseg_amd import engine as E

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ARCH = sys.argv[3] if len(sys.argv) > 3 else "psp"
SIZE = int(sys.argv[4]) if len(sys.argv) > 4 else (473 if ARCH == "psp" else 465)
CLASSES = int(sys.argv[5]) if len(sys.argv) > 5 else 150
torch.manual_seed(0)
if ARCH == "psp":
    from model.pspnet import PSPNet
    model = PSPNet(layers=101, classes=CLASSES, zoom_factor=8, pretrained=False)
else:
    from model.psanet import PSANet
    model = PSANet(layers=101, classes=CLASSES, zoom_factor=8, pretrained=False)
model = model.cuda().train()
tr = Trainer(model, base_lr=0.01, sync_bn=True)
x = torch.randn(B, 3, SIZE, SIZE).cuda()
y = torch.randint(0, CLASSES, (B, SIZE, SIZE)).cuda()
for _ in range(5):          # 2 launch-by-launch steps + the 2 recorded steps of the step plan + its first replay
    tr.step(x, y, 0.01)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(STEPS):
    _, ml, _ = tr.step(x, y, 0.01)
torch.cuda.synchronize()
out = {"ms": round((time.time() - t0) / STEPS * 1e3, 3), "batch": B, "arch": ARCH, "size": SIZE, "arith": E.arith_name(),
       "loss": round(float(ml.item()), 5), "lib": os.environ.get("SEMSEG_HIP_LIB", "default")}
if os.environ.get("FAMILIES") == "1":
    kt = E.KernelTimer()
    for e in tr.engines.values():
        e.side_wgrad, e.hipri_main, e.ktimer = False, False, kt
    for _ in range(2):
        tr.step(x, y, 0.01)
    torch.cuda.synchronize()
    out["families"] = {k: {"ms_per_step": round(v["total_ms"] / 2, 3), "avg_us": v["avg_us"],
                           "rate": v.get("tflops", v.get("hbm_tb_per_s_algorithmic"))} for k, v in kt.summary().items()}
    out["serial_sum_ms"] = round(sum(v["total_ms"] for v in kt.summary().values()) / 2, 2)
print(json.dumps(out), flush=True)
