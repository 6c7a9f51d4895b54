"""Host side of the step plan (include/semseg_hip.h "Step plan", csrc/plan.hip): records the launches of one train step —
the loop body of the reference's tool/train.py:269-276 — as they are made through the C ABI, and replays them from C.

Recording is transparent to the engine: while a StepPlan is `lib.recorder`, every successful call of an entry point is
appended to the C-side plan (entry id + one 64-bit slot per argument).  What is not a C-ABI launch — a collective of
torch.distributed (SyncBN exchange, gradient buckets) — is registered by the engine as a host operation
(`StepPlan.py_op`); it ends the current C segment, so a replay is  segment, host op, segment, ...  in the recorded order.

A recorded argument must still mean the same thing at replay time.  Guaranteed by construction: the engine's buffers live as
long as the engine (Engine.buf), streams are cached per device, per-step scalars are read from device memory
(semseg_step_state_set).  Checked here: a pointer argument must be an integer address or None (a ctypes temporary would
dangle), and the caller verifies that torch's allocator handed out no new block while recording (Trainer._record).
"""
import ctypes
import struct
import threading

import torch

from ._lib import lib

_M64 = (1 << 64) - 1


class PlanError(RuntimeError):
    pass


def _slot(ct, a, name, i):
    if ct is ctypes.c_void_p:
        if a is None:
            return 0
        if isinstance(a, int):
            return a & _M64
        raise PlanError("%s: argument %d is a host object (%r), not a device address: not replayable" % (name, i, type(a)))
    if ct is ctypes.c_float:
        return struct.unpack("<I", struct.pack("<f", float(a)))[0]
    if ct is ctypes.c_double:
        return struct.unpack("<Q", struct.pack("<d", float(a)))[0]
    return int(a) & _M64          # int / long long / size_t / unsigned long long: sign-extended two's complement


# entry points that synchronise with the host or whose arguments are host memory: a step that calls one while recording
# cannot be replayed
_NOT_REPLAYABLE = ("semseg_label_check", "semseg_xchg_alloc", "semseg_xchg_free", "semseg_xchg_ipc", "semseg_aug", "semseg_plan_")


class StepPlan:
    def __init__(self):
        h = ctypes.c_void_p()
        if lib.raw("semseg_plan_create")(ctypes.byref(h)) != 0:
            raise PlanError("semseg_plan_create failed")
        self.handle = h
        self.segments = []          # ("c", first, last) | ("py", fn, stream)
        self._seg_start = 0
        self._ids = {}
        self.error = None
        self.recording = False
        self._append = lib.raw("semseg_plan_append")
        self._replay = lib.raw("semseg_plan_replay")

    def __del__(self):
        try:
            if self.handle:
                lib.raw("semseg_plan_destroy")(self.handle)
                self.handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ recording
    def begin(self):
        assert lib.recorder is None, "another plan is recording"
        lib.load()
        self._thread = threading.get_ident()      # only this thread's calls belong to the step (a loader thread's do not)
        lib.recorder = self
        self.recording = True

    def end(self):
        """Stops recording; returns None or the reason the recorded step cannot be replayed (the step itself ran normally)."""
        if lib.recorder is self:
            lib.recorder = None
        if self.recording:
            self.recording = False
            self._close_segment()
        return self.error

    def size(self):
        return lib.raw("semseg_plan_size")(self.handle)

    def _close_segment(self):
        n = self.size()
        if n > self._seg_start:
            self.segments.append(("c", self._seg_start, n))
        self._seg_start = n

    def wrap(self, name, f, argtypes):
        if name.startswith("semseg_plan_") or f.restype is not ctypes.c_int:      # plan calls and size queries are not launches
            return f

        def call(*args):
            rc = f(*args)
            if rc == 0 and self.error is None and threading.get_ident() == self._thread:
                try:
                    self._record(name, argtypes, args)
                except PlanError as e:
                    self.error = str(e)
            return rc
        return call

    def _record(self, name, argtypes, args):
        if name.startswith(_NOT_REPLAYABLE):
            raise PlanError("%s was called while a step was being recorded: not replayable" % name)
        fid = self._ids.get(name)
        if fid is None:
            fid = self._ids[name] = lib.raw("semseg_plan_fn_id")(name.encode())
            if fid < 0:
                raise PlanError("no thunk for %s (csrc/plan_thunks.inc is stale?)" % name)
        n = len(argtypes)
        if len(args) != n:
            raise PlanError("%s: %d arguments for %d parameters" % (name, len(args), n))
        slots = (ctypes.c_ulonglong * max(n, 1))(*[_slot(ct, a, name, i) for i, (ct, a) in enumerate(zip(argtypes, args))])
        if self._append(self.handle, fid, n, slots) < 0:
            raise PlanError("semseg_plan_append(%s) failed" % name)

    def py_op(self, fn):
        """A host-side operation inside the step (a torch.distributed collective): run it now and, when recording, keep it
        as the boundary between two C segments, together with the stream that is current."""
        if not self.recording:
            fn()
            return
        self._close_segment()
        self.segments.append(("py", fn, torch.cuda.current_stream() if torch.cuda.is_available() else None))
        lib.recorder = None           # launches the host operation makes itself (peer-memory exchange) are part of IT
        try:
            fn()
        finally:
            lib.recorder = self

    def entries_of(self, name):
        """Entry indices whose entry point is `name` (for patching a slot: semseg_plan_set_slot)."""
        fid = lib.raw("semseg_plan_fn_id")(name.encode())
        ef = lib.raw("semseg_plan_entry_fn")
        return [i for i in range(self.size()) if ef(self.handle, i) == fid]

    def set_slot(self, entry, arg, bits):
        if lib.raw("semseg_plan_set_slot")(self.handle, entry, arg, bits & _M64) != 0:
            raise PlanError("semseg_plan_set_slot(%d, %d) failed" % (entry, arg))

    def get_slot(self, entry, arg):
        v = ctypes.c_ulonglong()
        if lib.raw("semseg_plan_get_slot")(self.handle, entry, arg, ctypes.byref(v)) != 0:
            raise PlanError("semseg_plan_get_slot(%d, %d) failed" % (entry, arg))
        return v.value

    def same_as(self, other, ignore=None):
        """None when `other` recorded the same step — the same calls with the same arguments, host operations at the same
        places — else what differs.  ignore = (entry point name, argument index) of the one argument that legitimately
        advances from step to step on the host side (the dropout call counter)."""
        fn, arg = (-1, -1) if ignore is None else (lib.raw("semseg_plan_fn_id")(ignore[0].encode()), ignore[1])
        where = (ctypes.c_int * 2)(-1, -1)
        rc = lib.raw("semseg_plan_compare")(self.handle, other.handle, fn, arg, where)
        if rc != 0:
            if rc < 0:
                return "semseg_plan_compare failed (%d)" % rc
            e = where[0]
            ef = lib.raw("semseg_plan_entry_fn")
            names = {v: k for k, v in {**self._ids, **other._ids}.items()}
            na = names.get(ef(self.handle, e), "?") if e < self.size() else "(end)"
            nb = names.get(ef(other.handle, e), "?") if e < other.size() else "(end)"
            return "entry %d differs (%s vs %s, argument %d)" % (e, na, nb, where[1])
        sa = [(s[0], s[1], s[2]) if s[0] == "c" else (s[0], s[2]) for s in self.segments]
        sb = [(s[0], s[1], s[2]) if s[0] == "c" else (s[0], s[2]) for s in other.segments]
        if sa != sb:
            return "host operations at different places (%d vs %d segments)" % (len(sa), len(sb))
        return None

    # ------------------------------------------------------------------ replay
    def launches(self):
        return sum(s[2] - s[1] for s in self.segments if s[0] == "c")

    def host_ops(self):
        return sum(1 for s in self.segments if s[0] == "py")

    def replay(self):
        for seg in self.segments:
            if seg[0] == "c":
                rc = self._replay(self.handle, seg[1], seg[2])
                if rc != 0:
                    raise PlanError("replay failed with code %d at entry %d" %
                                    (rc, lib.raw("semseg_plan_failed_entry")(self.handle)))
            elif seg[2] is None:
                seg[1]()
            else:
                with torch.cuda.stream(seg[2]):
                    seg[1]()
