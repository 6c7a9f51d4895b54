"""Peer-memory exchange for the SyncBN statistics (csrc/xchg.hip): the [2C] fp64 vectors that
nn.SyncBatchNorm all-reduces once per layer and pass (tool/train.py:142) travel through IPC-mapped fine-grained device
memory — every rank writes its vector into a slot of every peer's buffer and sums the slots of its own — in ONE kernel
per rank, with no c10d call and no host involvement.  torch.distributed is used once, to hand the IPC handles round.

OPT-IN (SEMSEG_SYNCBN_XCHG=1): exercised with two and four processes on one GPU (tests/test_dist_gpu.py); it has not run
across xGMI, so RCCL (`dist.all_reduce`) stays the default exchange.  One node, world <= 8."""
import ctypes
import os
import socket

import torch
import torch.distributed as dist

from ._lib import lib

MAX_DOUBLES = 16384      # SEMSEG_XCHG_MAX_DOUBLES (include/semseg_hip.h)
_INSTANCES = {}


def enabled():
    return os.environ.get("SEMSEG_SYNCBN_XCHG", "0") == "1"


def get(device, group=None):
    """The process-wide exchange of this (device, group); built collectively on first use (every rank must get here)."""
    key = (device.index, id(group))
    if key not in _INSTANCES:
        _INSTANCES[key] = SyncExchange(device, group)
    return _INSTANCES[key]


class SyncExchange:
    def __init__(self, device, group=None):
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if not 1 <= self.world <= 8:
            raise RuntimeError("the peer-memory SyncBN exchange serves one node (world <= 8), got world %d" % self.world)
        self.device = device
        with torch.cuda.device(device):
            own = ctypes.c_void_p()
            self._ck(lib.semseg_xchg_alloc(self.world, ctypes.byref(own)), "xchg_alloc")
            handle = (ctypes.c_ubyte * 64)()
            self._ck(lib.semseg_xchg_ipc_export(own, handle), "xchg_ipc_export")
            mine = (socket.gethostname(), os.getpid(), bytes(handle))
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine, group=group)
            if any(h[0] != mine[0] for h in everyone):
                raise RuntimeError("the peer-memory SyncBN exchange needs all ranks on one node")
            self._own, self._mapped = own, []
            ptrs = []
            for r, (_, pid, hbytes) in enumerate(everyone):
                if r == self.rank:
                    ptrs.append(own.value)
                    continue
                p = ctypes.c_void_p()
                buf = (ctypes.c_ubyte * 64).from_buffer_copy(hbytes)
                self._ck(lib.semseg_xchg_ipc_import(buf, ctypes.byref(p)), "xchg_ipc_import (rank %d)" % r)
                self._mapped.append(p)
                ptrs.append(p.value)
            self._peers = (ctypes.c_void_p * self.world)(*ptrs)
            self.err = torch.zeros(1, dtype=torch.int32, device=device)
        self.seq = 0
        dist.barrier(group=group)      # nobody starts exchanging before every mapping exists

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
        self.seq += 1
        self._ck(lib.semseg_xchg_allreduce_f64(t.data_ptr(), nslot, n, out.data_ptr(), self._peers, self.world, self.rank,
                                               self.seq, self.err.data_ptr(), torch.cuda.current_stream().cuda_stream),
                 "xchg_allreduce_f64")

    def check(self):
        """Synchronises and raises if an exchange gave up waiting for a peer (its result is then garbage)."""
        if int(self.err.item()):
            raise RuntimeError("SyncBN peer-memory exchange timed out waiting for a peer's flag (ranks out of step, a peer "
                               "died, or their kernels were not co-resident)")

    def close(self):
        torch.cuda.synchronize(self.device)
        for p in self._mapped:
            lib.semseg_xchg_ipc_close(p)
        self._mapped = []
        if self._own is not None:
            lib.semseg_xchg_free(self._own)
            self._own = None
