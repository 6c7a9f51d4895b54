"""CPU: host side of the split-bf16 experiment — the process-wide kernel switch is off by default,
ops.conv_split() sets and restores it (also on an exception), the split instances use their own tile-table keys, and the
committed tables are well formed.  No kernel is launched."""
import json
import os

import pytest


def test_switch_default_off_and_context_manager_restores():
    from semseg_amd import ops
    lib = ops.lib
    assert int(lib.semseg_experiment_conv_split(0)) == 0          # returns the previous value: off by default
    assert int(lib.semseg_experiment_conv_split(7)) == 0          # an unknown piece count is ignored ...
    assert int(lib.semseg_experiment_conv_split(0)) == 0          # ... and leaves the switch where it was
    with ops.conv_split(False):
        assert not ops._SPLIT_ON and int(lib.semseg_experiment_conv_split(0)) == 0
    with ops.conv_split(True):
        assert ops._SPLIT_ON
        assert int(lib.semseg_experiment_conv_split(3)) == 3
        with ops.conv_split(True):                                # nesting keeps it on
            assert int(lib.semseg_experiment_conv_split(3)) == 3
        assert ops._SPLIT_ON and int(lib.semseg_experiment_conv_split(3)) == 3
    assert not ops._SPLIT_ON and int(lib.semseg_experiment_conv_split(0)) == 0
    with pytest.raises(RuntimeError):
        with ops.conv_split(True):
            raise RuntimeError("boom")
    assert not ops._SPLIT_ON and int(lib.semseg_experiment_conv_split(0)) == 0


def test_split_instances_use_their_own_tile_keys(monkeypatch):
    from semseg_amd import ops
    key = ops.tile_key("fwd", 16, 60, 60, 1024, 256, 1, 1, 1, 0, 1)
    monkeypatch.setitem(ops.TILE_CHOICE, key, 128)
    monkeypatch.setitem(ops.TILE_CHOICE, key + "|sp", 64)
    never = lambda *a: (_ for _ in ()).throw(AssertionError("no launch expected"))
    assert ops._tuned_tile(key, 128, None, 0, never) == 128
    with ops.conv_split(True):
        assert ops._tuned_tile(key, 128, None, 0, never) == 64
    monkeypatch.delitem(ops.TILE_CHOICE, key + "|sp")
    with ops.conv_split(True):
        assert ops._tuned_tile(key, 128, None, 0, never) == 128   # unknown shape: the default, no timing at run time


def test_committed_tile_tables_are_well_formed():
    from semseg_amd import ops
    for path, sp in ((ops.TILE_TABLE_PATH, False), (ops.TILE_TABLE_SP_PATH, True)):
        assert os.path.exists(path), path
        tiles = json.load(open(path))["tiles"]
        assert tiles and all(k.endswith("|sp") == sp for k in tiles)
        assert all(int(v) in ops.TILE_CODES for v in tiles.values())
        assert all(k.split("|")[0] in ("fwd", "dgrad") for k in tiles)


def test_engine_flag_is_opt_in():
    from semseg_amd import engine
    if "SEMSEG_SPLIT_BF16" not in os.environ:
        assert engine.SPLIT_BF16 == 0 and engine.SPLIT_LAYERS == ["cls.0"]
