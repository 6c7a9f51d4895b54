"""Worker for tests/test_dist_gpu.py::test_syncbn_peer_memory_exchange: one rank of W processes that all share the test box's
single GPU.  Part 1: the raw exchange (semseg_amd/syncbn_xchg.py, csrc/xchg.hip) on random fp64 vectors of the sizes and
slot counts the engine uses, checked against a gloo all-reduce of the same data; every rank must hold bit-identical
results.  Exit code 3 = an exchange timed out waiting for a peer (kernels of two processes not co-resident on this box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

if __name__ == "__main__":
    out = sys.argv[1]
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from semseg_amd import syncbn_xchg
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    x = syncbn_xchg.get(dev)
    worst, nex = 0.0, 0
    sizes = [(2 * 64, 8), (2 * 2048, 8), (8192, 1), (1536, 1), (2, 1), (16384, 1), (4096, 8)]
    for rep in range(12):
        for n, nslot in sizes:
            g = torch.Generator().manual_seed(1000 * rep + 17 * n + rank)
            t = (torch.randn(nslot * n, generator=g, dtype=torch.float64) * (1 + rep)).to(dev)
            local = t.view(nslot, n).cpu()
            folded = local[0].clone()
            for s in range(1, nslot):
                folded += local[s]                      # the kernel's order
            parts = [torch.zeros(n, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(parts, folded)
            ref = parts[0].clone()
            for q in range(1, world):
                ref += parts[q]                         # rank order, as the kernel sums
            res = torch.empty(n, dtype=torch.float64, device=dev)
            x.all_reduce(t, nslot=nslot, n=n, out=res)
            if rep % 3 == 0:                            # also the in-place form the engine uses
                x.all_reduce(t, nslot=nslot, n=n)
                torch.cuda.synchronize()
                assert torch.equal(t[:n].cpu(), res.cpu())
                nex += 1
            torch.cuda.synchronize()
            nex += 1
            if int(x.err.item()):
                print("TIMEOUT in exchange %d" % x.seq, flush=True)
                dist.destroy_process_group()
                sys.exit(3)
            worst = max(worst, float((res.cpu() - ref).abs().max()))
            assert torch.equal(res.cpu(), ref), "rank %d exchange %d (n %d, nslot %d): max diff %.3e" % (
                rank, x.seq, n, nslot, float((res.cpu() - ref).abs().max()))
    # back-to-back exchanges without any host synchronisation in between (the way a training step issues them)
    vec = torch.full((512,), float(rank + 1), dtype=torch.float64, device=dev)
    for _ in range(200):
        x.all_reduce(vec)                               # v <- sum_r v_r: grows by a factor `world` every time, exactly
        vec /= world
    torch.cuda.synchronize()
    x.check()
    expect = sum(range(1, world + 1)) / world
    assert torch.all(vec == expect), (float(vec[0]), expect)
    np.savez(os.path.join(out, "xchg_rank%d_of%d.npz" % (rank, world)), nex=nex + 200, worst=worst)
    dist.barrier()
    x.close()
    dist.destroy_process_group()
