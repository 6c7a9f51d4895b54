"""Worker for tests/test_dist_gpu.py: one rank of a world_size-2 data-parallel run of the fused Trainer
(SyncBN statistics + bucketed gradient all-reduce).  Backend gloo so that both ranks can share the single
GPU of the test box; the code path in semseg_amd/trainer.py is the one RCCL takes at N>1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make_model(arch="psp"):
    from oracle import segnet
    from model.pspnet import PSPNet
    m = PSPNet(layers=50, classes=7, zoom_factor=8, dropout=0.0, pretrained=False)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(segnet.recipe_state_dict(shapes, seed=99))
    return m


def data(batch, size=57, classes=7):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(batch, 3, size, size, generator=g)
    y = torch.randint(0, classes, (batch, size, size), generator=g)  # no ignore pixels: equal counts
    return x, y


def run(world, rank, steps=2, bucket_mb=8):
    from semseg_amd.trainer import Trainer
    dev = torch.device("cuda", 0)
    m = make_model().to(dev).train()
    tr = Trainer(m, base_lr=0.01, momentum=0.9, weight_decay=1e-4, bucket_mb=bucket_mb, sync_bn=True)
    lr = float(os.environ.get("LR", "0.01"))
    run.w0 = tr.flat_w.cpu().numpy().copy()
    gb = int(os.environ.get("GLOBAL_BATCH", "4"))
    x, y = data(gb)
    per = gb // world
    xs, ys = x[rank * per:(rank + 1) * per].to(dev), y[rank * per:(rank + 1) * per].to(dev)
    losses = []
    w_first = None
    for it in range(steps):
        _, ml, al = tr.step(xs, ys, lr=lr)
        losses.append((float(ml.item()), float(al.item())))
        if it == 0:
            w_first = tr.flat_w.cpu().numpy()
    torch.cuda.synchronize()
    from semseg_amd import syncbn_xchg
    xc = syncbn_xchg.DECISION.get(dev.index, (None, ""))
    if xc[0] is not None:
        xc[0].check()                         # raises if an exchange gave up waiting for a peer
    print("SyncBN exchange of rank %d: %s" % (rank, xc[1]), flush=True)
    sd = m.state_dict()
    ncoll = max(e.syncbn_collectives_per_step for e in tr.engines.values())
    print("step plan of rank %d: %s" % (rank, tr.plan_log), flush=True)
    run.plan_log = list(tr.plan_log)
    return np.array(losses), tr.flat_w.cpu().numpy(), sd["layer0.1.running_var"].cpu().numpy(), \
        sd["cls.1.running_mean"].cpu().numpy(), w_first, ncoll


if __name__ == "__main__":
    out = sys.argv[1]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    losses, w, rv, rm, w1, ncoll = run(world, rank, steps=int(os.environ.get("STEPS", "2")))
    np.savez(os.path.join(out, "rank%d_of%d.npz" % (rank, world)), losses=losses, w=w, rv=rv, rm=rm, w1=w1,
             ncoll=ncoll, plan_log=np.array(" | ".join(run.plan_log)), w0=run.w0)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
