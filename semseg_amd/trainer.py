"""Fused train step — the loop body of the reference's tool/train.py:269-276
(`model(input, target)`; `loss = main + aux_weight*aux`; `zero_grad`; `backward`; `optimizer.step`)
driven directly on the HIP engine: no autograd tape, parameters / gradients / momentum live in three
flat fp32 buffers with identical offsets, SGD is two fused launches (backbone lr, head lr*10 —
tool/train.py:134-140), and under torch.distributed the gradient is all-reduced over RCCL in
reverse-order buckets on a side communicator while backward is still running (the role
DistributedDataParallel plays at tool/train.py:157), with SyncBN statistics on the default group.

`poly_learning_rate` restates util/util.py:34-37.
"""
import os

import torch
import torch.distributed as dist


from . import ops
from .engine import Engine


def poly_learning_rate(base_lr, curr_iter, max_iter, power=0.9):
    return base_lr * (1 - float(curr_iter) / max_iter) ** power


class Trainer:
    def __init__(self, model, base_lr=0.01, momentum=0.9, weight_decay=1e-4, aux_weight=0.4,
                 ignore_index=255, bucket_mb=32, sync_bn=True):
        self.model = model
        self.base_lr, self.momentum, self.wd = base_lr, momentum, weight_decay
        self.aux_weight = aux_weight
        self.ignore_index = ignore_index
        self.device = next(model.parameters()).device
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.dist_on = self.world > 1 or (os.environ.get("SEMSEG_FORCE_DIST") == "1" and dist.is_initialized())
        self.sync_bn = sync_bn
        self.engines = {}
        self.steps = 0
        self._flatten()
        self.g_main = torch.ones(1, device=self.device)
        self.g_aux = torch.full((1,), float(aux_weight), device=self.device)
        self.bucket_elems = bucket_mb * 1024 * 1024 // 4
        self.grad_group = dist.new_group() if self.dist_on else None
        self.timers = None

    # parameters -> one flat buffer (offsets 16-byte aligned, same layout as Engine.flat_grad)
    def _flatten(self):
        params = list(self.model.parameters())
        total = sum(((p.numel() + 3) // 4) * 4 for p in params)
        self.flat_w = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=self.device)
        off = 0
        self.offsets = {}
        for p in params:
            n = p.numel()
            v = self.flat_w[off:off + n].view(p.shape)
            v.copy_(p.data)
            p.data = v
            self.offsets[p] = (off, n)
            off += ((n + 3) // 4) * 4
        self.total = total
        # lr groups: the five backbone stages at base lr, everything after them at 10x
        backbone = [self.model.layer0, self.model.layer1, self.model.layer2, self.model.layer3,
                    self.model.layer4]
        nb = sum(1 for m in backbone for _ in m.parameters())
        last = params[nb - 1]
        self.split = self.offsets[last][0] + ((self.offsets[last][1] + 3) // 4) * 4
        assert all(self.offsets[p][0] < self.split for m in backbone for p in m.parameters())
        self.params = params

    def engine(self, x):
        key = tuple(x.shape)
        e = self.engines.get(key)
        if e is None:
            e = Engine(self.model, x.shape[0], x.shape[2], x.shape[3], True, self.model.kind)
            e.force_sync_bn = self.sync_bn and self.dist_on
            assert e.flat_grad.numel() == self.total
            if self.dist_on:
                self._make_buckets(e)
                e.grads_ready_hook = lambda plist, e=e: self._on_ready(e, plist)
            self.engines[key] = e
        return e

    # ------------------------------------------------------------------ gradient buckets
    def _make_buckets(self, e):
        """Contiguous flat-buffer ranges built from the END (backward produces the heads first)."""
        order = list(reversed(self.params))
        buckets, cur, hi = [], [], None
        for p in order:
            off, n = self.offsets[p]
            if hi is None:
                hi = off + ((n + 3) // 4) * 4
            cur.append(p)
            if hi - off >= self.bucket_elems:
                buckets.append((off, hi, cur))
                cur, hi = [], None
        if cur:
            buckets.append((self.offsets[cur[-1]][0], hi, cur))
        e._buckets = buckets
        e._bucket_of = {}
        for bi, (_, _, ps) in enumerate(buckets):
            for p in ps:
                e._bucket_of[p] = bi

    def _on_ready(self, e, plist):
        for p in plist:
            bi = e._bucket_of[p]
            e._pending[bi] -= 1
            if e._pending[bi] == 0:
                lo, hi, _ = e._buckets[bi]
                # the bucket's gradients come from BOTH the main and the side stream, whichever issues this
                e.order_after_all_producers()
                e._works.append(dist.all_reduce(e.flat_grad[lo:hi], group=self.grad_group, async_op=True))

    # ------------------------------------------------------------------ one optimisation step
    def step(self, x, y, lr=None):
        """x [N,3,H,W] fp32 cuda, y [N,h,w] int64 cuda (per-rank shard).  Returns (pred, main, aux)."""
        lr = self.base_lr if lr is None else lr
        # the parameters must still be views of flat_w (model.to()/half()/load_state_dict(assign=True) break that
        # silently: SGD would then update a buffer nobody reads)
        for p in (self.params[0], self.params[-1]):
            if p.data_ptr() != self.flat_w.data_ptr() + 4 * self.offsets[p][0]:
                raise RuntimeError("model parameters no longer alias the Trainer's flat buffer (the model was moved / "
                                   "re-assigned after Trainer(model)); build a new Trainer")
        e = self.engine(x)
        if e.params_stale():
            raise RuntimeError("the module tree changed after the Trainer built its engine; build a new Trainer")
        pred, main_loss, aux_loss = e.forward_train(x, y, self.ignore_index)
        if self.dist_on:
            e._pending = [len(ps) for _, _, ps in e._buckets]
            e._works = []
        e.backward(self.g_main, self.g_aux)
        if self.dist_on:
            assert all(c == 0 for c in e._pending), "a gradient bucket never completed"
            for w in e._works:
                w.wait()
        first = self.steps == 0
        gs = 1.0 / self.world
        ops.sgd_step(self.flat_w, e.flat_grad, self.flat_m, self.split, lr, self.momentum, self.wd, gs, first)
        n2 = self.total - self.split
        ops.sgd_step(self.flat_w[self.split:], e.flat_grad[self.split:], self.flat_m[self.split:], n2,
                     lr * 10.0, self.momentum, self.wd, gs, first)
        self.steps += 1
        # label counts of earlier steps that have reached the host by now (non-blocking; see Engine.LabelWatch)
        e._label_watch().poll(self.model.cls[4].weight.shape[0])
        return pred, main_loss, aux_loss

    def check_labels(self):
        """Blocks until the out-of-range-label counts of every step so far are on the host; raises IndexError if a step
        saw a label that is neither ignore_index nor a class id (torch's CrossEntropyLoss, tool/train.py:121, raises on
        such a batch immediately; here the fused head counts them and the error surfaces at most RING steps later or at
        this call — call it at epoch end and before validation)."""
        for e in self.engines.values():
            e.check_labels()
