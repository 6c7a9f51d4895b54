"""Step plan (include/semseg_hip.h "Step plan", csrc/plan.hip, semseg_amd/plan.py) without a GPU: the generated call thunks
are current, the 64-bit slot encoding survives every scalar type of the C ABI, recording through the ctypes layer appends
what was called, replay walks segments and host operations in the recorded order, a patched slot is what the next replay
passes, a failing entry is reported.  Uses semseg_host_probe (host-only) as the recorded entry point; no kernel is launched."""
import ctypes
import os
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def test_thunks_are_current_and_cover_every_entry_point():
    from semseg_amd import build
    assert open(build.THUNKS).read() == build.thunks_text(), "run python -m semseg_amd.build (plan_thunks.inc is stale)"
    from semseg_amd._lib import lib, parse_header
    lib.load()
    for name, (ret, args) in parse_header().items():
        fid = lib.semseg_plan_fn_id(name.encode())
        if ret is ctypes.c_int and not name.startswith("semseg_plan_"):
            assert fid >= 0, name
            assert lib.semseg_plan_fn_nargs(fid) == len(args), name
        else:
            assert fid < 0, name        # size queries and the plan's own entry points cannot be recorded


def test_record_replay_patch_and_failure():
    from semseg_amd._lib import lib
    from semseg_amd.plan import PlanError, StepPlan
    out = (ctypes.c_ulonglong * 7)()
    addr = ctypes.addressof(out)
    plan = StepPlan()
    order = []
    plan.begin()
    assert lib.semseg_host_probe(addr, -7, -(1 << 40), 123456789012, 0.1, -2.5e-3, 0xdeadbeef) == 0
    plan.py_op(lambda: order.append(("host op", out[6])))
    assert lib.semseg_host_probe(addr, 5, 6, 7, 1.5, 2.5, None) == 0
    assert lib.semseg_adaptive_avgpool_scratch_floats is not None       # size queries pass through unrecorded
    assert plan.end() is None
    assert lib.recorder is None
    assert plan.launches() == 2 and plan.host_ops() == 1 and [s[0] for s in plan.segments] == ["c", "py", "c"]
    assert order == [("host op", 1)] and out[6] == 2                    # the recording pass itself executed everything once
    plan.replay()
    assert order[-1] == ("host op", 3) and out[6] == 4                  # probe, host op, probe
    assert ctypes.c_longlong(out[0]).value == 5 and out[1] == 6 and out[2] == 7 and out[5] == 0
    # first entry's values as the callee saw them
    assert lib.raw("semseg_plan_replay")(plan.handle, 0, 1) == 0
    assert ctypes.c_longlong(out[0]).value == -7 and ctypes.c_longlong(out[1]).value == -(1 << 40)
    assert out[2] == 123456789012 and out[5] == 0xdeadbeef
    assert struct.unpack("<f", struct.pack("<I", out[3]))[0] == _f32(0.1)
    assert struct.unpack("<d", struct.pack("<Q", out[4]))[0] == -2.5e-3
    # patch a slot: the float argument of entry 0
    (e0, e1) = plan.entries_of("semseg_host_probe")
    plan.set_slot(e0, 4, struct.unpack("<I", struct.pack("<f", 0.75))[0])
    assert lib.raw("semseg_plan_replay")(plan.handle, 0, 1) == 0
    assert struct.unpack("<f", struct.pack("<I", out[3]))[0] == 0.75
    assert plan.get_slot(e1, 1) == 5
    # a failing entry stops the replay and is named
    plan.set_slot(e1, 1, -12345)
    with pytest.raises(PlanError, match="entry 1"):
        plan.replay()
    # out-of-range requests are refused
    assert lib.raw("semseg_plan_replay")(plan.handle, 0, 3) != 0
    assert lib.raw("semseg_plan_set_slot")(plan.handle, 0, 7, 0) != 0


def test_unreplayable_calls_invalidate_the_record_not_the_step():
    from semseg_amd._lib import lib
    from semseg_amd.plan import StepPlan
    out = (ctypes.c_ulonglong * 7)()
    plan = StepPlan()
    plan.begin()
    # a host OBJECT where an address is expected (a ctypes temporary would dangle at replay time): the call itself still runs
    assert lib.semseg_host_probe(out, 1, 2, 3, 0.0, 0.0, None) == 0
    why = plan.end()
    assert out[6] == 1 and why is not None and "not replayable" in why
    plan2 = StepPlan()
    plan2.begin()
    assert lib.semseg_host_probe(ctypes.addressof(out), -12345, 2, 3, 0.0, 0.0, None) != 0     # failed calls are not recorded
    assert plan2.end() is None and plan2.launches() == 0


def test_two_records_are_compared_call_by_call():
    """The acceptance rule of the Trainer: a record is replayed only when the next step recorded the same calls."""
    from semseg_amd._lib import lib
    from semseg_amd.plan import StepPlan
    out = (ctypes.c_ulonglong * 7)()
    addr = ctypes.addressof(out)

    def rec(a, d, host_op=True, extra=False):
        p = StepPlan()
        p.begin()
        lib.semseg_host_probe(addr, a, 2, 3, d, 0.5, None)
        if host_op:
            p.py_op(lambda: None)
        lib.semseg_host_probe(addr, 7, 8, 9, 1.0, 2.0, 0x10)
        if extra:
            lib.semseg_host_probe(addr, 7, 8, 9, 1.0, 2.0, 0x10)
        assert p.end() is None
        return p
    base = rec(1, 0.25)
    assert base.same_as(rec(1, 0.25)) is None
    assert "entry 0" in base.same_as(rec(2, 0.25)) and "argument 1" in base.same_as(rec(2, 0.25))
    assert base.same_as(rec(2, 0.25), ignore=("semseg_host_probe", 1)) is None          # the one argument allowed to advance
    assert "argument 4" in base.same_as(rec(1, 0.5), ignore=("semseg_host_probe", 1))
    assert "host operations" in base.same_as(rec(1, 0.25, host_op=False))
    assert "entry 2" in base.same_as(rec(1, 0.25, extra=True))


def test_syncbn_exchange_mode_and_single_process_decision(monkeypatch):
    """SEMSEG_SYNCBN_XCHG = 0 (default since round 6: opt-in) | auto | 1; without a process group the decision is RCCL / nothing, with its reason kept."""
    import torch
    from semseg_amd import syncbn_xchg as sx
    monkeypatch.delenv("SEMSEG_SYNCBN_XCHG", raising=False)
    assert sx.mode() == "0" and not sx.enabled()
    monkeypatch.setenv("SEMSEG_SYNCBN_XCHG", "1")
    assert sx.mode() == "1" and sx.enabled()
    monkeypatch.setenv("SEMSEG_SYNCBN_XCHG", "bogus")
    assert sx.mode() == "0"
    monkeypatch.setenv("SEMSEG_SYNCBN_XCHG", "0")
    sx.DECISION.clear()
    dev = torch.device("cuda", 0)
    assert sx.active(dev) is None and sx.DECISION[0][1] == "SEMSEG_SYNCBN_XCHG=0"
    monkeypatch.setenv("SEMSEG_SYNCBN_XCHG", "auto")
    sx.DECISION.clear()
    assert sx.active(dev) is None and sx.DECISION[0][1] == "no process group"
    sx.DECISION.clear()


def test_only_the_recording_thread_is_recorded():
    """Calls another thread makes through the same library while a step is being recorded (a loader thread, an eval engine) are
    executed and NOT appended."""
    import threading
    from semseg_amd._lib import lib
    from semseg_amd.plan import StepPlan
    out = (ctypes.c_ulonglong * 7)()
    addr = ctypes.addressof(out)
    plan = StepPlan()
    plan.begin()
    lib.semseg_host_probe(addr, 1, 2, 3, 0.0, 0.0, None)
    t = threading.Thread(target=lambda: lib.semseg_host_probe(addr, 9, 9, 9, 0.0, 0.0, None))
    t.start()
    t.join()
    lib.semseg_host_probe(addr, 4, 5, 6, 0.0, 0.0, None)
    assert plan.end() is None
    assert out[6] == 3 and plan.launches() == 2


def test_debug_switch_list(monkeypatch):
    """SEMSEG_DEBUG = "name=value,name=value": the one variable every A/B / test switch lives in (semseg_amd._lib.debug; the C side
    parses the same string, csrc/common.h semseg_debug)."""
    from semseg_amd._lib import debug
    monkeypatch.delenv("SEMSEG_DEBUG", raising=False)
    assert debug("side_wgrad") is None and debug("side_wgrad", "1") == "1"
    monkeypatch.setenv("SEMSEG_DEBUG", "side_wgrad=0, wgrad_dma=1:3:32 ,fused_split_max=4")
    assert debug("side_wgrad", "1") == "0" and debug("wgrad_dma") == "1:3:32" and debug("fused_split_max") == "4"
    assert debug("wgrad") is None and debug("hipri_main", "1") == "1"
