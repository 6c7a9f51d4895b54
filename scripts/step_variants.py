"""Whole-step A/B inside ONE process: the PSPNet-101 473^2 train step under combinations of
  dma   = SEMSEG_WGRAD_DMA (weight-gradient kernel variant, read per launch),
  side  = every weight gradient on the side stream (Engine.side_all),
  hipri = dependent backward chain on a high-priority stream (Engine.hipri_main),
interleaved over ROUNDS rounds.  python scripts/step_variants.py [batch] [rounds] [arch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semseg_amd.trainer import Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ARCH = sys.argv[3] if len(sys.argv) > 3 else "psp"
SIZE = 473 if ARCH == "psp" else 465
# "policy/side/hipri[/extra]": policy = SEMSEG_WGRAD_DMA value (a variant number or "small:big:tile-threshold");
# extra = "KEY=VAL+KEY=VAL": NSIDE (Engine.n_side), FUSE_BNR (Engine.fuse_bnr), anything else goes to the environment
CONFIGS = [(c.split("/")[0], int(c.split("/")[1]), int(c.split("/")[2]), 0,
            c.split("/")[3] if len(c.split("/")) > 3 else "") for c in
           os.environ.get("CONFIGS", "0/0/0,6/1/0,6/1/1,6/0/0,7/1/1").split(",")]
torch.manual_seed(0)
if ARCH == "psp":
    from model.pspnet import PSPNet
    model = PSPNet(layers=101, classes=150, zoom_factor=8, pretrained=False)
else:
    from model.psanet import PSANet
    model = PSANet(layers=101, classes=150, zoom_factor=8, pretrained=False)
model = model.cuda().train()
tr = Trainer(model, base_lr=0.01, sync_bn=True)
x = torch.randn(B, 3, SIZE, SIZE).cuda()
y = torch.randint(0, 150, (B, SIZE, SIZE)).cuda()
for _ in range(2):
    tr.step(x, y, 0.01)
eng = next(iter(tr.engines.values()))
res = {c: [] for c in CONFIGS}
for r in range(ROUNDS):
    for c in CONFIGS:
        dma, side, hipri, cdma, extra = c
        for c2 in CONFIGS:                              # a key one config sets must not leak into the next
            for kv in [e for e in c2[4].split("+") if e]:
                os.environ.pop(kv.split("=")[0], None)
        for kv in [e for e in extra.split("+") if e]:
            os.environ[kv.split("=")[0]] = kv.split("=")[1]
            if kv.split("=")[0] == "NSIDE":          # number of weight-gradient streams (Engine.n_side)
                eng.n_side = int(kv.split("=")[1])
            if kv.split("=")[0] == "FUSE_BNR":       # BatchNorm-backward reduction in the data-gradient epilogue
                eng.fuse_bnr = kv.split("=")[1] == "1"
        os.environ["SEMSEG_WGRAD_DMA"] = str(dma)
        eng.side_all, eng.hipri_main = bool(side), bool(hipri)
        tr.step(x, y, 0.01)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            _, ml, _ = tr.step(x, y, 0.01)
        torch.cuda.synchronize()
        res[c].append((time.time() - t0) / 5 * 1e3)
print("%s batch %d: ms per step (min over %d rounds / all)" % (ARCH, B, ROUNDS))
for c in CONFIGS:
    print("  wgrad %-8s side_all %d hipri %d %-18s: %8.2f   %s" % (c[0], c[1], c[2], c[4], min(res[c]), " ".join("%.2f" % v for v in res[c])))
print("final loss", float(ml.item()))
