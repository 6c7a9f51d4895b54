"""CPU: host side of the per-launch arithmetic argument (include/semseg_hip.h: SEMSEG_ARITH_F32 / SEMSEG_ARITH_BF16X3;
DESIGN.md section 8.4) — the library has NO process-wide switch any more, an unknown code is rejected, the engine default
is bf16x3 and can be forced to exact fp32, the split instances use their own tile-table keys, and the committed tables
are well formed and consistent with the packed-panel padding.  No kernel is launched."""
import ctypes
import json
import os

import pytest


def test_library_has_no_process_wide_switch_and_rejects_unknown_codes():
    from semseg_amd import ops
    lib = ops.lib
    lib.load()
    assert not hasattr(lib._dll, "semseg_experiment_conv_split")
    with pytest.raises(AttributeError):
        lib.semseg_experiment_conv_split
    # argument validation happens before anything touches a device: a bad arithmetic code is SEMSEG_EINVAL (-1) even
    # with otherwise plausible (fake, never dereferenced) pointers, a good one gets past that check (null pointers: -1 too,
    # so use distinct failure reasons: Ci % 32)
    fake = ctypes.c_void_p(4096)
    args = lambda arith, ci=32: (fake, 32, fake, fake, 32, 1, 1, 1, ci, 1, 1, 32, 1, 1, 1, 0, 1, None, None, 0, None, 0,
                                 None, 1, 64, arith, None, 0, None, None)
    for bad in (1, 2, 6, -1, 7):
        assert lib.semseg_conv_fwd(*args(bad)) == -1
    assert lib.semseg_gemm_rows_batched(fake, 32, 0, fake, 0, fake, 32, 0, 1, 32, 32, 1, 5, None) == -1
    assert lib.semseg_gemm_kmajor_batched(fake, 64, 0, fake, 64, 0, fake, 0, fake, 1 << 20, 32, 64, 64, 0, 1, 9, None) == -1
    assert ops.ARITH_F32 == 0 and ops.ARITH_BF16X3 == 3


def test_engine_default_and_forcing_exact_fp32():
    from semseg_amd import engine, ops
    if "SEMSEG_ARITH" not in os.environ:
        assert engine.ARITH == ops.ARITH_BF16X3 and engine.arith_name() == "bf16x3"
    old = engine.set_arith("f32")
    try:
        assert engine.ARITH == ops.ARITH_F32 and engine.arith_name() == "f32"
        assert engine.set_arith("bf16x3") == "f32"
        with pytest.raises(KeyError):
            engine.set_arith("bf16")
    finally:
        engine.set_arith(old)


def test_split_instances_use_their_own_tile_keys(monkeypatch):
    from semseg_amd import ops
    key = ops.tile_key("fwd", 16, 60, 60, 1024, 256, 1, 1, 1, 0, 1)
    monkeypatch.setitem(ops.TILE_CHOICE, key, 128)
    monkeypatch.setitem(ops.TILE_CHOICE, key + "|sp", 64)
    never = lambda *a: (_ for _ in ()).throw(AssertionError("no launch expected"))
    assert ops._tuned_tile(key, 128, None, 0, never) == 128
    assert ops._tuned_tile(key, 128, None, 0, never, ops.ARITH_BF16X3) == 64
    monkeypatch.delitem(ops.TILE_CHOICE, key + "|sp")
    assert ops._tuned_tile(key, 128, None, 0, never, ops.ARITH_BF16X3) == 128   # unknown shape: the default, no timing


def test_table_entry_wider_than_the_packed_panel_is_not_used(monkeypatch):
    """A layer with fewer than 128 output columns has panels padded to 64 rows (ops.PackedConv): a regenerated or
    hand-edited table that names a 128-column tile for it must not reach the kernel (it would read past the panel)."""
    from semseg_amd import ops
    key = ops.tile_key("fwd", 2, 119, 119, 256, 64, 1, 1, 1, 0, 1)
    never = lambda *a: (_ for _ in ()).throw(AssertionError("no launch expected"))
    for bad in (128, 1128):
        monkeypatch.setitem(ops.TILE_CHOICE, key, bad)
        assert ops._tuned_tile(key, 64, None, 0, never) == 64
    monkeypatch.setitem(ops.TILE_CHOICE, key, 1064)
    assert ops._tuned_tile(key, 64, None, 0, never) == 1064


def test_split_gemm_tile_code_is_only_taken_for_eligible_shapes(monkeypatch):
    """Tile code 2128: 1x1, stride 1, no padding, whole 128-column panels, bf16x3 table; data gradients only with a reduction of
    at most 1024.  A table entry naming it for anything else is dropped at load time; the exact-fp32 lookup never returns it."""
    from semseg_amd import ops
    ok = ops.tile_key("fwd", 16, 60, 60, 1024, 256, 1, 1, 1, 0, 1) + "|sp"
    assert ops._split_gemm_eligible(ok)
    okd = ops.tile_key("dgrad", 16, 60, 60, 256, 1024, 1, 1, 1, 0, 1) + "|sp"     # 1024 -> 256 data gradient: K = 1024
    assert ops._split_gemm_eligible(okd)
    for bad in (ops.tile_key("dgrad", 16, 60, 60, 512, 2048, 1, 1, 1, 0, 1) + "|sp",      # data gradient over K = 2048 > 1024
                ops.tile_key("dgrad", 16, 60, 60, 64, 256, 1, 1, 1, 0, 1) + "|sp",        # 64 input channels: no 128-column panel
                ops.tile_key("fwd", 16, 60, 60, 1024, 256, 3, 3, 1, 1, 1) + "|sp",        # 3x3
                ops.tile_key("fwd", 16, 119, 119, 256, 512, 1, 1, 2, 0, 1) + "|sp",       # strided
                ops.tile_key("fwd", 16, 60, 60, 512, 150, 1, 1, 1, 0, 1) + "|sp"):        # 150 columns
        assert not ops._split_gemm_eligible(bad)
    import json as _json, tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        badd = ops.tile_key("dgrad", 16, 60, 60, 512, 2048, 1, 1, 1, 0, 1) + "|sp"
        _json.dump({"tiles": {ok: 2128, okd: 2128, badd: 2128}}, f)
    monkeypatch.setattr(ops, "TILE_TABLE_SP_PATH", f.name)
    t = ops._load_tables()
    assert t.get(ok) == 2128 and t.get(okd) == 2128 and badd not in t
    never = lambda *a: (_ for _ in ()).throw(AssertionError("no launch expected"))
    monkeypatch.setitem(ops.TILE_CHOICE, ok, 2128)
    assert ops._tuned_tile(ok[:-3], 128, None, 0, never, ops.ARITH_BF16X3) == 2128
    assert ops._tuned_tile(ok[:-3], 128, None, 0, never, ops.ARITH_F32) in ops.TILE_CODES


def test_committed_tile_tables_are_well_formed():
    from semseg_amd import ops
    for path, sp in ((ops.TILE_TABLE_PATH, False), (ops.TILE_TABLE_SP_PATH, True)):
        assert os.path.exists(path), path
        tiles = json.load(open(path))["tiles"]
        assert tiles and all(k.endswith("|sp") == sp for k in tiles)
        # code 2128 (the 256 x 128 bf16x3 GEMM kernel for 1x1 forward convs) only in the bf16x3 table, only for eligible shapes
        assert all(int(v) in ops.TILE_CODES or (sp and int(v) in ops.SPLIT_GEMM_CODES and ops._split_gemm_eligible(k))
                   for k, v in tiles.items())
        assert all(k.split("|")[0] in ("fwd", "dgrad") for k in tiles)
        for k, v in tiles.items():          # column width never exceeds the padding of the layer's packed panels
            f = k.split("|")
            ncols = int(f[5]) if f[0] == "fwd" else int(f[4])
            assert ncols >= 128 or int(v) % 1000 == 64, (k, v)
        assert all(k in ops.TILE_CHOICE for k in tiles)        # nothing was dropped by the loader's consistency check
