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


from . import engine as _engine
from . import ops
from ._lib import lib
from .engine import Engine
from ._lib import debug
from .plan import PlanError, StepPlan

# Step plan (semseg_amd/plan.py, csrc/plan.hip): after EAGER_STEPS eager steps of an engine the next TWO are recorded (they run
# launch by launch, as before) and, when both records hold the same calls with the same arguments, every later step is
# replayed from C.  SEMSEG_STEP_PLAN=0: every step is sequenced by Python, launch by launch (rounds 1-4).
STEP_PLAN = os.environ.get("SEMSEG_STEP_PLAN", "1") != "0"
EAGER_STEPS = 2


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
        self.use_plan = STEP_PLAN
        self._step_state = torch.zeros(2, dtype=torch.float32, device=self.device)    # {lr, lr of the heads}: semseg_sgd_step's lr_dev
        self.plan_log = []        # what happened to every recording attempt (tests, bench)

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
                view = e.flat_grad[lo:hi]
                e.host_op(lambda view=view: e._works.append(dist.all_reduce(view, group=self.grad_group, async_op=True)))

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
        if not (self.use_plan and e.ktimer is None and e.tape_hook is None and not e._plan_off):
            if e._plan_replays:
                self._set_step_state(e, lr, 0)      # the device part of the dropout counter belongs to replayed steps only
            return self._fresh(e, self._step_eager(e, x, y, lr))
        # planned steps (the eager ones before the recording included) run on one stream of their own: a recorded stream handle
        # must mean the same stream at every replay
        cur = torch.cuda.current_stream()
        if debug("plan_own_stream", "1") == "0":      # A/B only
            return self._fresh(e, self._step_planned(e, x, y, lr, cur))
        st = _engine._shared(self.device, "plan_stream", lambda: torch.cuda.Stream(device=self.device))
        ops.stream_wait(st, cur)
        with torch.cuda.stream(st):
            out = self._step_planned(e, x, y, lr, st)
        ops.stream_wait(cur, st)
        return self._fresh(e, out)

    @staticmethod
    def _fresh(e, out):
        """(pred, main_loss, aux_loss) as the caller gets them: the two losses are slices of a CLONE of the engine's [2] loss buffer
        (one 8-byte copy on the caller's stream), so a loss kept across steps keeps its value like the fresh tensors the
        reference returns (model/pspnet.py:101-103, `losses.append(main_loss)`); `pred` IS the engine's buffer — valid until
        the next step of this engine, `.clone()` to keep ("Output lifetime")."""
        l = e._losses.clone()
        return out[0], l[0:1], l[1:2]

    def _step_eager(self, e, x, y, lr, lr_dev=None):
        pred, main_loss, aux_loss = e.forward_train(x, y, self.ignore_index)
        if self.dist_on:
            e._pending = [len(ps) for _, _, ps in e._buckets]
            e._works = []
        e.backward(self.g_main, self.g_aux)
        if self.dist_on:
            assert all(c == 0 for c in e._pending), "a gradient bucket never completed"
            e.host_op(lambda: [w.wait() for w in e._works])
        first = self.steps == 0
        gs = 1.0 / self.world
        # a step whose SyncBN statistics came from a timed-out peer-memory exchange must not reach the weights: both SGD launches read
        # the exchange's error flag on the device and leave w / momentum untouched when it is set (the host raises at most RING steps
        # later, _watch_exchange; every exchange after a time-out gives up at once, so the flag stays set until reset())
        skip = None
        if self.dist_on:
            from . import syncbn_xchg
            xc = syncbn_xchg.DECISION.get(self.device.index, (None,))[0]
            skip = None if xc is None else xc.err
        # a recorded step reads both learning rates from device memory (semseg_step_state_set); the by-value argument is then
        # unused and kept at 0 so that two records of consecutive steps of a schedule hold the same calls
        ops.sgd_step(self.flat_w, e.flat_grad, self.flat_m, self.split, lr if lr_dev is None else 0.0, self.momentum, self.wd,
                     gs, first, lr_dev=None if lr_dev is None else lr_dev[0:1], skip_dev=skip)
        n2 = self.total - self.split
        ops.sgd_step(self.flat_w[self.split:], e.flat_grad[self.split:], self.flat_m[self.split:], n2,
                     lr * 10.0 if lr_dev is None else 0.0, self.momentum, self.wd, gs, first,
                     lr_dev=None if lr_dev is None else lr_dev[1:2], skip_dev=skip)
        self.steps += 1
        # label counts of earlier steps that have reached the host by now (non-blocking; see Engine.LabelWatch)
        e._label_watch().poll(self.model.cls[4].weight.shape[0])
        self._watch_exchange()
        return pred, main_loss, aux_loss

    def _watch_exchange(self, wait=False):
        """A SyncBN peer-memory exchange that timed out raises at most RING steps later (non-blocking poll of its error flag
        through a pinned ring, like the label counts), or here with wait=True."""
        if not self.dist_on:
            return
        from . import syncbn_xchg
        xc = syncbn_xchg.DECISION.get(self.device.index, (None,))[0]
        if xc is not None:
            xc.poll(wait)
            if not wait:
                xc.watch()

    # ------------------------------------------------------------------ step plan: record once, replay from C
    def _set_step_state(self, e, lr, drop_offset):
        ops._ck(lib.raw("semseg_step_state_set")(self._step_state.data_ptr(), float(lr), float(lr) * 10.0,
                                                 e.drop_dev.data_ptr(), int(drop_offset), ops._stream()), "step_state_set")

    def _host_signature(self, e):
        """Host state that a recorded step bakes into its launch arguments and the launch-by-launch step re-reads every step
        (ADVICE r5): BatchNorm / Dropout2d training flags and p, momentum, weight decay, ignore_index, the loss weights' buffers,
        the 1 / world gradient scale.  A replay is only valid while it is unchanged."""
        from torch import nn
        if getattr(e, "_sig_mods", None) is None:     # the module tree is fixed while the engine lives (Engine.params_stale)
            e._sig_mods = [m for m in self.model.modules() if isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.Dropout2d))]
        mods = tuple((m.training, getattr(m, "p", None), getattr(m, "momentum", None), getattr(m, "eps", None))
                     for m in e._sig_mods)
        return (mods, self.momentum, self.wd, self.ignore_index, self.aux_weight, self.world, self.g_main.data_ptr(),
                self.g_aux.data_ptr())

    def _step_planned(self, e, x, y, lr, st):
        plan = e._plan
        if plan is not None and e._plan_gen != _engine.ARENA_GEN[0]:
            plan = e._plan = None             # a process-wide arena the plan points into was replaced: record again
            self.plan_log.append("discarded: arena generation moved")
        if plan is not None and e._plan_sig != self._host_signature(e):
            plan = e._plan = None
            e._plan_candidate = None
            self.plan_log.append("discarded: host state baked into the record changed (training flags / dropout p / momentum / "
                                 "weight decay / ignore_index / world)")
        if plan is not None:
            k = e._plan_replays + 1
            # the recorded dropout launches carry the host counter of the RECORDED step by value; the device part makes up the
            # difference to where the host counter stands now (eager steps interleaved with replays advance it too: ADVICE r5)
            drop_off = e._drop_calls - e._plan_drop_base
            # one launch: the caller's batch into the recorded input buffers + this step's learning rates / dropout counter
            xs, ys = x.contiguous(), y.contiguous()
            assert xs.dtype == e._plan_x.dtype and ys.dtype == e._plan_y.dtype and xs.shape == e._plan_x.shape and ys.shape == e._plan_y.shape
            ops._ck(lib.raw("semseg_step_begin")(e._plan_x.data_ptr(), xs.data_ptr(), xs.numel() * xs.element_size(),
                                                 e._plan_y.data_ptr(), ys.data_ptr(), ys.numel() * ys.element_size(),
                                                 self._step_state.data_ptr(), float(lr), float(lr) * 10.0,
                                                 e.drop_dev.data_ptr(), int(drop_off), ops._stream()), "step_begin")
            e._drop_calls += e._plan_drops
            if self.dist_on:
                e._works = []
            e.model.__dict__["_hip_bn_epoch"] = e.model.__dict__.get("_hip_bn_epoch", 0) + 1
            ncls = self.model.cls[4].weight.shape[0]
            e._label_watch().poll(ncls)
            plan.replay()
            e._label_watch().watch(e._rec_main["acc"])
            self._watch_exchange()
            e._plan_replays = k
            self.steps += 1
            return e._plan_out
        n = e._plan_eager
        if n < EAGER_STEPS or self.steps == 0:
            e._plan_eager = n + 1
            self._set_step_state(e, lr, 0)
            return self._step_eager(e, x, y, lr)
        return self._record(e, x, y, lr, st)

    def _record(self, e, x, y, lr, st):
        if lib.recorder is not None:        # another Trainer of this process is recording right now: this step runs as it is
            self._set_step_state(e, lr, 0)
            return self._step_eager(e, x, y, lr)
        if e._plan_x is None:
            e._plan_x = torch.empty_like(x, memory_format=torch.contiguous_format)
            e._plan_y = torch.empty_like(y, memory_format=torch.contiguous_format)
        e._plan_x.copy_(x)
        e._plan_y.copy_(y)
        self._set_step_state(e, lr, 0)
        plan = StepPlan()
        sig = self._host_signature(e)
        drops0 = e._drop_calls
        allocs0 = torch.cuda.memory_stats(self.device).get("allocation.all.allocated", 0)
        e.recorder = plan
        plan.begin()
        try:
            out = self._step_eager(e, e._plan_x, e._plan_y, lr, lr_dev=self._step_state)
        finally:
            why = plan.end()          # a call that cannot be replayed does not stop the step: it only invalidates the record
            e.recorder = None
        if why is None and torch.cuda.memory_stats(self.device).get("allocation.all.allocated", 0) != allocs0:
            why = "device memory was allocated while the step was recorded (a buffer address in the plan may be temporary)"
        if why is not None:
            e._plan_tries += 1
            self.plan_log.append("recording failed: " + why)
            if e._plan_tries >= 3:
                e._plan_off = True
                import warnings
                warnings.warn("semseg_amd.Trainer: the step could not be recorded (%s); it stays on the launch-by-launch path" % why)
            return out
        e._plan_drops = e._drop_calls - drops0
        prev = e._plan_candidate
        if prev is None or prev[1] != _engine.ARENA_GEN[0]:
            e._plan_candidate = (plan, _engine.ARENA_GEN[0])     # accepted when the next step records the same calls
            self.plan_log.append("candidate: %d launches, %d host operations" % (plan.launches(), plan.host_ops()))
            return out
        e._plan_candidate = None
        diff = prev[0].same_as(plan, ignore=("semseg_dropout2d_mask", 4))
        if diff is not None:
            e._plan_tries += 1
            self.plan_log.append("recording failed: two consecutive steps issued different launch sequences: " + diff)
            if e._plan_tries >= 3:
                e._plan_off = True
                import warnings
                warnings.warn("semseg_amd.Trainer: the step is not replayable (%s); it stays on the launch-by-launch path" % diff)
            return out
        e._plan, e._plan_out, e._plan_replays = plan, out, 0
        e._plan_gen = _engine.ARENA_GEN[0]
        e._plan_drop_base = drops0          # host dropout counter in front of the accepted record
        e._plan_sig = sig
        msg = "recorded: %d launches in %d segments, %d host operations; verified against the previous step's record" % (
            plan.launches(), len(plan.segments) - plan.host_ops(), plan.host_ops())
        self.plan_log.append(msg)
        return out

    def check_labels(self):
        """Blocks until the out-of-range-label counts of every step so far are on the host; raises IndexError if a step
        saw a label that is neither ignore_index nor a class id (torch's CrossEntropyLoss, tool/train.py:121, raises on
        such a batch immediately; here the fused head counts them and the error surfaces at most RING steps later or at
        this call — call it at epoch end and before validation)."""
        for e in self.engines.values():
            e.check_labels()
        self._watch_exchange(wait=True)
