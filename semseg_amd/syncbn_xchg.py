"""Peer-memory exchange for the SyncBN statistics (csrc/xchg.hip): the [2C] fp64 vectors that
nn.SyncBatchNorm all-reduces once per layer and pass (tool/train.py:142) travel through IPC-mapped fine-grained device
memory — every rank writes its vector into a slot of every peer's buffer and sums the slots of its own — in ONE kernel
per rank, with no c10d call and no host involvement.  torch.distributed is used once, to hand the IPC handles round.

SEMSEG_SYNCBN_XCHG = 0 (default) | auto | 1.
  0     RCCL (`dist.all_reduce`).  The default since round 6: the exchange has never run across xGMI, so a job only
        takes it when asked to.
  auto  on a multi-rank one-node job the exchange is built and SELF-TESTED among the real peers at the first SyncBN
        collective (`active()`): 64 exchanges of known vectors of the sizes the engine uses, 2 s bound per exchange, every
        rank checks every result, and the ranks agree on the verdict (MIN all-reduce).  Passed: the training step uses the
        exchange.  Anything else — a handle that does not map, a flag that does not arrive, a wrong sum, ranks on different
        hosts, world > 8 or world == 1 — and every rank uses RCCL (`dist.all_reduce`), with the reason kept in `DECISION`.
        The path has only ever run with several processes on ONE GPU (tests/test_dist_gpu.py); the self-test is what
        stands between it and the first multi-GPU run.
        In training the wait for a peer's flag is then bounded by TRAIN_TIMEOUT_MS (10 minutes, the order of RCCL's own
        watchdog: a rank that saves a checkpoint or logs validation arrives late, not never), not by the kernel's 20 s default.
  1     forced (tests, the forced one-rank bench line); no self-test.
A timed-out exchange is not silent, and it does not reach the weights: both SGD launches of the step read the error flag on the
device and do nothing when it is set (semseg_sgd_step's skip_dev; running statistics of that step ARE garbage). Trainer.step polls the error flag through a pinned ring after every step
(`watch` / `poll`), Trainer.check_labels() and `check()` block and raise."""
import atexit
import ctypes
import os
import socket

import torch
import torch.distributed as dist

from ._lib import lib

MAX_DOUBLES = 16384      # SEMSEG_XCHG_MAX_DOUBLES (include/semseg_hip.h)
_INSTANCES = {}
DECISION = {}            # device index -> (exchange or None, reason)


TRAIN_TIMEOUT_MS = 600000


def mode():
    v = os.environ.get("SEMSEG_SYNCBN_XCHG", "0")
    return v if v in ("0", "1", "auto") else "0"


def enabled():
    """Forced on (SEMSEG_SYNCBN_XCHG=1)."""
    return mode() == "1"


def get(device, group=None):
    """The process-wide exchange of this (device, group); built collectively on first use (every rank must get here)."""
    key = (device.index, id(group))
    if key not in _INSTANCES:
        _INSTANCES[key] = SyncExchange(device, group)
    return _INSTANCES[key]


def active(device):
    """The exchange the SyncBN collectives of this process use, or None for RCCL.  COLLECTIVE on first use in auto mode."""
    d = DECISION.get(device.index)
    if d is not None:
        return d[0]
    m = mode()
    if m == "0" or not (dist.is_available() and dist.is_initialized()):
        d = (None, "SEMSEG_SYNCBN_XCHG=0" if m == "0" else "no process group")
    elif m == "1":
        d = (get(device), "forced (SEMSEG_SYNCBN_XCHG=1)")
    else:
        d = _auto(device)
    DECISION[device.index] = d
    return d[0]


def _agree(ok, device):
    """MIN over ranks of a 0/1 verdict, through the job's own backend."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def _auto(device):
    world = dist.get_world_size()
    if world == 1:
        return None, "auto: one rank"
    if world > 8:
        return None, "auto: world %d > 8" % world
    x, why = None, None
    try:
        x = SyncExchange(device, None, strict=False)
        if x.failed:
            why = x.failed
    except Exception as e:      # noqa: BLE001 — any failure to build the mappings means RCCL, on every rank
        why = "build failed: %r" % (e,)
    if not _agree(why is None, device):
        if x is not None:
            x.close()
        return None, "auto: " + (why or "a peer could not map the exchange buffers")
    why = x.selftest()
    if not _agree(why is None, device):
        x.close()
        return None, "auto: self-test failed" + (": " + why if why else " on a peer")
    _INSTANCES[(device.index, id(None))] = x
    x.timeout_ms = TRAIN_TIMEOUT_MS
    return x, "auto: self-test passed on %d ranks" % world


class SyncExchange:
    RING = 8

    def __init__(self, device, group=None, strict=True):
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if not 1 <= self.world <= 8:
            raise RuntimeError("the peer-memory SyncBN exchange serves one node (world <= 8), got world %d" % self.world)
        self.device = device
        self.group = group
        self.failed = None
        self._own, self._mapped = None, []
        with torch.cuda.device(device):
            own = ctypes.c_void_p()
            rc = lib.semseg_xchg_alloc(self.world, ctypes.byref(own))
            handle = (ctypes.c_ubyte * 64)()
            if rc == 0:
                self._own = own
                rc = lib.semseg_xchg_ipc_export(own, handle)
            if rc != 0:
                self.failed = "allocating / exporting the exchange buffer failed (%d)" % rc
            # every rank takes part in the handle exchange even when its own part failed: the others must not hang
            mine = (socket.gethostname(), os.getpid(), bytes(handle), self.failed is None)
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine, group=group)
            if any(h[0] != mine[0] for h in everyone):
                self.failed = "ranks on different hosts"
            elif not all(h[3] for h in everyone):
                self.failed = self.failed or "a peer could not allocate its exchange buffer"
            ptrs = []
            if self.failed is None:
                for r, (_, pid, hbytes, _) in enumerate(everyone):
                    if r == self.rank:
                        ptrs.append(own.value)
                        continue
                    p = ctypes.c_void_p()
                    buf = (ctypes.c_ubyte * 64).from_buffer_copy(hbytes)
                    rc = lib.semseg_xchg_ipc_import(buf, ctypes.byref(p))
                    if rc != 0:
                        self.failed = "hipIpcOpenMemHandle of rank %d's buffer failed (%d)" % (r, rc)
                        break
                    self._mapped.append(p)
                    ptrs.append(p.value)
            if self.failed is None:
                self._peers = (ctypes.c_void_p * self.world)(*ptrs)
            self.err = torch.zeros(1, dtype=torch.int32, device=device)
            # the exchange number lives in device memory and is advanced by the kernel: no per-call host argument, so a recorded
            # step plan replays the exchanges inside its C segments (no host operation per SyncBN layer)
            self.seq_dev = torch.zeros(1, dtype=torch.int64, device=device)
            self._host = torch.zeros(self.RING, dtype=torch.int32).pin_memory()
        self._events = [None] * self.RING
        self._nwatch = 0
        self.seq = 0
        self.timeout_ms = 0           # 0: the kernel's default (20 s)
        if strict and self.failed:
            raise RuntimeError("peer-memory SyncBN exchange: " + self.failed)
        if strict:
            dist.barrier(group=group)      # nobody starts exchanging before every mapping exists
        atexit.register(self.close)

    @staticmethod
    def _ck(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with code %d" % (what, rc))

    def all_reduce(self, t, nslot=1, n=None, out=None):
        """out[0:n] = sum over ranks of (sum over the nslot replicas t[s*n : (s+1)*n]); in place by default.  fp64 CUDA tensor."""
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        n = t.numel() // nslot if n is None else n
        out = t if out is None else out
        assert n <= MAX_DOUBLES and out.numel() >= n
        self.seq += 1          # host mirror of the device counter (diagnostics, reset)
        self._ck(lib.semseg_xchg_allreduce_f64(t.data_ptr(), nslot, n, out.data_ptr(), ctypes.addressof(self._peers), self.world,
                                               self.rank, 0, self.seq_dev.data_ptr(), self.err.data_ptr(), self.timeout_ms,
                                               torch.cuda.current_stream().cuda_stream), "xchg_allreduce_f64")

    def selftest(self, rounds=64, timeout_ms=2000):
        """`rounds` exchanges of known vectors among the real peers (sizes the engine uses, in place and out of place, with
        and without slot replicas), each bounded by timeout_ms; every element of every result is checked on this rank.
        Returns None or the reason it failed.  Collective; the caller agrees on the verdict across ranks."""
        W, r = self.world, self.rank
        sizes = [(2, 1), (128, 8), (1536, 1), (4096, 8), (8192, 1), (16384, 1), (512, 1), (2 * 2048, 8)]
        old = self.timeout_ms
        self.timeout_ms = timeout_ms
        why = None
        try:
            with torch.cuda.device(self.device):
                checks = []
                for i in range(rounds):
                    n, nslot = sizes[i % len(sizes)]
                    base = torch.arange(n, dtype=torch.float64, device=self.device) * 0.5 + (i + 1)
                    # replica s of rank q holds (q + 1) * base + s: the sum over ranks and replicas is known in closed form
                    t = torch.stack([(r + 1) * base + s for s in range(nslot)]).contiguous().view(-1)
                    res = torch.empty(n, dtype=torch.float64, device=self.device)
                    self.all_reduce(t, nslot=nslot, n=n, out=res)
                    expect = base * (W * (W + 1) / 2) + W * (nslot * (nslot - 1) / 2)
                    checks.append((res, expect))
                torch.cuda.synchronize(self.device)
                if int(self.err.item()):
                    why = "rank %d: an exchange timed out after %d ms waiting for a peer's flag" % (r, timeout_ms)
                else:
                    for i, (res, expect) in enumerate(checks):
                        if not torch.equal(res, expect):
                            why = "rank %d: exchange %d returned a wrong sum (max diff %.3e)" % (
                                r, i, float((res - expect).abs().max()))
                            break
        except Exception as e:      # noqa: BLE001
            why = "rank %d: %r" % (r, e)
        self.timeout_ms = old
        return why

    # ------------------------------------------------------------------ error flag: non-blocking watch, blocking check
    def watch(self):
        """Copies the error flag into a pinned ring slot behind the work enqueued so far (no synchronisation)."""
        i = self._nwatch % self.RING
        self._nwatch += 1
        if self._events[i] is not None:
            self._events[i].synchronize()
            self._take(i)
        self._host[i:i + 1].copy_(self.err, non_blocking=True)
        self._events[i] = torch.cuda.Event()
        self._events[i].record()

    def _take(self, i):
        self._events[i] = None
        if int(self._host[i].item()):
            self._raise()

    def poll(self, wait=False):
        """Raises if a watched copy that has reached the host shows a timed-out exchange."""
        for i, ev in enumerate(self._events):
            if ev is None:
                continue
            if wait:
                ev.synchronize()
            elif not ev.query():
                continue
            self._take(i)

    def _raise(self):
        raise RuntimeError("SyncBN peer-memory exchange timed out waiting for a peer's flag (ranks out of step, a peer "
                           "died, or their kernels were not co-resident): the BatchNorm statistics of this rank are "
                           "garbage from that exchange on and every later exchange gives up at once; restart the job "
                           "(or call reset() on EVERY rank after re-synchronising them)")

    def check(self):
        """Synchronises and raises if an exchange gave up waiting for a peer (its result is then garbage)."""
        if int(self.err.item()):
            self._raise()

    def reset(self):
        """Explicit re-handshake after a time-out: COLLECTIVE — every rank drains its stream, the ranks meet at a barrier,
        the flag is cleared and the sequence numbers restart from a common value."""
        torch.cuda.synchronize(self.device)
        seq = torch.tensor([int(self.seq_dev.item())], dtype=torch.int64,
                           device=self.device if dist.get_backend(self.group) == "nccl" else "cpu")
        dist.all_reduce(seq, op=dist.ReduceOp.MAX, group=self.group)
        self.seq = int(seq.item()) + 2          # both parities' stale flags are below it on every rank
        self.seq_dev.fill_(self.seq)
        self.err.zero_()
        self._events = [None] * self.RING
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)

    def close(self):
        if self._own is None and not self._mapped:
            return
        try:
            torch.cuda.synchronize(self.device)
            for p in self._mapped:
                lib.semseg_xchg_ipc_close(p)
            self._mapped = []
            if self._own is not None:
                lib.semseg_xchg_free(self._own)
                self._own = None
        except Exception:      # noqa: BLE001 — interpreter shutdown: the driver is going away with the process
            pass
