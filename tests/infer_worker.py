"""Worker for tests/test_infer_gpu.py::test_sharded_multi_scale_two_ranks: one rank of the crop-sharded multi-scale
test pipeline (BASELINE.json configs[4]).  Backend gloo so that both ranks can share the single GPU of the test box;
the code path in semseg_amd/infer.py is the one RCCL takes at N>1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main(out):
    from model.pspnet import PSPNet
    from oracle import segnet
    from semseg_amd.infer import MultiScaleTester
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    classes, crop, base = 7, 73, 96
    m = PSPNet(layers=50, classes=classes, zoom_factor=8, pretrained=False)
    sd = segnet.recipe_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5)
    sd["cls.4.weight"] *= 1e-3
    sd["cls.4.bias"] *= 1e-3
    m.load_state_dict(sd)
    img = (np.random.default_rng(1).random((97, 130, 3)) * 255).astype(np.float32)
    t = MultiScaleTester(m.cuda(), classes, base, crop, crop, (0.5, 1.0, 1.75), shard=True)
    pred, prob = t.predict(img, return_prob=True)
    torch.cuda.synchronize()
    units = t.shard_units(t.plan(97, 130), rank, world)
    np.savez(os.path.join(out, "infer_rank%d_of%d.npz" % (rank, world)),
             pred=np.zeros(0) if pred is None else pred.cpu().numpy(),
             prob=np.zeros(0) if prob is None else prob.cpu().numpy(),
             ncrops=sum(len(v) for v in units.values()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
