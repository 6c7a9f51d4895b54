"""CPU, gloo, world_size 2: the protocol of the N>1 path without kernels —
 (1) SyncBN: all-reducing the per-rank [sum, sum of squares] fp64 vectors and finalising with the global
     count reproduces full-batch BatchNorm statistics (what semseg_bn_finalize consumes);
 (2) SyncBN backward: all-reduced [sum g, sum g*xhat] reproduce the full-batch input gradient;
 (3) gradient buckets: contiguous, disjoint, cover every parameter, fire exactly once, heads first.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(11)
        C, N, HW = 8, 4, 5
        x = (torch.randn(N, C, HW, HW, generator=g) * 2 + 1).double()
        gamma, beta = (torch.rand(C, generator=g) + 0.5).double(), torch.randn(C, generator=g).double()
        dout = torch.randn(N, C, HW, HW, generator=g).double()
        per = N // world
        xl, dl = x[rank * per:(rank + 1) * per], dout[rank * per:(rank + 1) * per]
        # forward protocol
        stats = torch.cat([xl.sum((0, 2, 3)), (xl * xl).sum((0, 2, 3))])
        dist.all_reduce(stats)
        cnt = per * HW * HW * world
        mean = stats[:C] / cnt
        var = stats[C:] / cnt - mean * mean
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        xf = x.clone().requires_grad_(True)
        ref = F.batch_norm(xf, None, None, gamma, beta, True, 0.1, 1e-5)
        yl = (xl - mean.view(1, C, 1, 1)) * (invstd * gamma).view(1, C, 1, 1) + beta.view(1, C, 1, 1)
        ok_fwd = torch.allclose(yl, ref[rank * per:(rank + 1) * per].detach(), atol=1e-10)
        # backward protocol: parameter gradients from LOCAL sums, input gradient from GLOBAL sums
        ref.backward(dout)
        xh = (xl - mean.view(1, C, 1, 1)) * invstd.view(1, C, 1, 1)
        sums = torch.cat([dl.sum((0, 2, 3)), (dl * xh).sum((0, 2, 3))])
        dist.all_reduce(sums)
        dxl = (gamma * invstd).view(1, C, 1, 1) * (dl - (sums[:C] / cnt).view(1, C, 1, 1) -
                                                  xh * (sums[C:] / cnt).view(1, C, 1, 1))
        ok_bwd = torch.allclose(dxl, xf.grad[rank * per:(rank + 1) * per], atol=1e-10)
        q.put((rank, bool(ok_fwd), bool(ok_bwd)))
    finally:
        dist.destroy_process_group()


def test_syncbn_protocol_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(0, True, True), (1, True, True)], res


def _group_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from semseg_amd.engine import SyncGroup

        class _BL:
            def __init__(self, C):
                self.C = C
        bls = [_BL(8), _BL(16), _BL(4)]
        g = SyncGroup(bls, torch.device("cpu"))
        gen = torch.Generator().manual_seed(100 + rank)
        local = [torch.randn(2 * b.C, generator=gen, dtype=torch.float64) for b in bls]
        for b, v in zip(bls, local):
            g.view(b).copy_(v)
        # views are adjacent pieces of ONE vector in member order
        ok_layout = g.buf.numel() == 56 and all(g.view(b).data_ptr() == g.buf.data_ptr() + 8 * o
                                                for b, o in zip(bls, (0, 16, 48)))
        dist.all_reduce(g.buf)                      # one exchange for the group ...
        sep = [v.clone() for v in local]
        for v in sep:
            dist.all_reduce(v)                      # ... equals one exchange per layer
        ok_sum = all(torch.equal(g.view(b), v) for b, v in zip(bls, sep))
        q.put((rank, bool(ok_layout), bool(ok_sum)))
    finally:
        dist.destroy_process_group()


def test_syncbn_group_staging_world2():
    """One all-reduce of a SyncGroup's staging vector == one all-reduce per BatchNorm layer (bitwise: the same two
    addends per element)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_group_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert res == [(0, True, True), (1, True, True)], res


def test_gradient_buckets_cover_parameters_once():
    from model.pspnet import PSPNet
    from semseg_amd.trainer import Trainer

    class _E:
        pass
    m = PSPNet(layers=50, classes=5, pretrained=False)
    tr = Trainer.__new__(Trainer)
    tr.model = m
    tr.params = list(m.parameters())
    tr.offsets, off = {}, 0
    for p in tr.params:
        tr.offsets[p] = (off, p.numel())
        off += ((p.numel() + 3) // 4) * 4
    tr.bucket_elems = 4 * 1024 * 1024 // 4
    e = _E()
    tr._make_buckets(e)
    spans = sorted((lo, hi) for lo, hi, _ in e._buckets)
    assert spans[0][0] == 0 and spans[-1][1] == off
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))           # contiguous, disjoint
    members = [p for _, _, ps in e._buckets for p in ps]
    assert len(members) == len(tr.params) and len({id(p) for p in members}) == len(tr.params)
    for lo, hi, ps in e._buckets:
        for p in ps:
            o, n = tr.offsets[p]
            assert lo <= o and o + n <= hi
    # the first bucket holds the heads (their gradients are produced first in backward)
    assert any(p is m.aux[4].bias for p in e._buckets[0][2])
    assert len(e._buckets) > 4


def test_poly_lr_matches_reference_formula():
    from semseg_amd.trainer import poly_learning_rate
    # util/util.py:34-37: base_lr * (1 - curr_iter / max_iter) ** power
    assert poly_learning_rate(0.01, 0, 100) == pytest.approx(0.01)
    assert poly_learning_rate(0.01, 50, 100, 0.9) == pytest.approx(0.01 * 0.5 ** 0.9)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[4]: multi-scale test path sharded over crops (semseg_amd/infer.py) — work plan + reduce
# ---------------------------------------------------------------------------------------------------------------
def _tester(cls=None):
    from semseg_amd.infer import MultiScaleTester
    cls = cls or MultiScaleTester
    t = cls.__new__(cls)
    t.classes, t.base_size, t.crop_h, t.crop_w = 5, 512, 473, 473
    t.scales, t.stride_rate, t.max_batch_crops = (0.5, 0.75, 1.0, 1.25, 1.5, 1.75), 2.0 / 3.0, 16
    t.shard, t.group, t.all_ranks = True, None, False
    t.device = torch.device("cpu")
    return t


def test_crop_shards_partition_the_units():
    """512x512 image, six ADE scales -> 1+1+4+4+4+9 = 23 crops (46 forwards); every world size
    gets each (scale, crop) exactly once, shard sizes differ by at most one."""
    t = _tester()
    plan = t.plan(512, 512)
    assert [len(s["pos"]) for s in plan] == [1, 1, 4, 4, 4, 9] and t.num_forwards(512, 512) == 46
    for world in (1, 2, 3, 4, 8, 23, 32):
        seen, sizes = [], []
        for r in range(world):
            mine = t.shard_units(plan, r, world)
            sizes.append(sum(len(v) for v in mine.values()))
            seen += [(si, ci) for si, cs in mine.items() for ci in cs]
        assert sorted(seen) == [(si, ci) for si, s in enumerate(plan) for ci in range(len(s["pos"]))]
        assert max(sizes) - min(sizes) <= 1


def _infer_worker(rank, world, port, q):
    """The real predict() control flow (plan -> shard -> per-scale accumulation -> ONE reduce -> argmax on rank 0)
    with the device kernels replaced by a deterministic CPU stand-in per (scale, crop)."""
    from semseg_amd.infer import MultiScaleTester
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)

    class Stub(MultiScaleTester):
        calls = 0

        def _accumulate_scale(self, img, h, w, sc, crops, total, sharded):
            for ci in crops:
                g = torch.Generator().manual_seed(1000 * int(sc["sh"]) + ci)
                total += torch.rand(total.shape, generator=g) / len(self.scales)
                Stub.calls += 1

        def _argmax(self, total, C, h, w):
            return total.argmax(0)
    try:
        t = _tester(Stub)
        out = t.predict(torch.zeros(40, 48, 3).numpy(), return_prob=True)
        q.put((rank, Stub.calls, None if out[0] is None else (out[0].numpy().copy(), out[1].numpy().copy())))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_sharded_multi_scale_predict_world2_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_infer_worker, args=(0, 1, 0, q))
    p.start()
    _, calls1, one = q.get(timeout=120)
    p.join(60)
    port = _free_port()
    ps = [ctx.Process(target=_infer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda r: r[0])
    for p in ps:
        p.join(60)
    assert res[0][1] + res[1][1] == calls1 and abs(res[0][1] - res[1][1]) <= 1     # every crop once, balanced
    assert res[1][2] is None                                                       # only rank 0 gets the result
    pred2, prob2 = res[0][2]
    import numpy as np
    assert np.allclose(prob2, one[1], atol=1e-5) and float((pred2 == one[0]).mean()) > 0.999
