"""CPU side of the input-pipeline parity: the oracle against the golden outputs of the
reference's own transform classes, independent cross-checks of the restated cv2 primitives, the planner's draw
order and region propagation, and the API's error behaviour.  No GPU, no compute through the C ABI."""
import math
import os
import random
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import transform_cases as tc                    # noqa: E402
from oracle import cv2_restated as ocv          # noqa: E402
from oracle import transform as otf             # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "transform_ref.npz"))


def _np(v):
    return v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


@pytest.mark.parametrize("name", sorted(tc.CASES))
def test_oracle_equals_reference_classes(name):
    """golden = /root/reference/util/transform.py classes run with cv2 := oracle/cv2_restated (make_golden_transform.py)"""
    H, W, ops, seeds = tc.CASES[name]
    img, lab = tc.make_input(name, H, W)
    for seed in seeds:
        random.seed(seed)
        oi, ol = otf.run(ops, np.float32(img), lab.copy())
        assert np.array_equal(_np(oi), GOLD["%s/%d/image" % (name, seed)])
        assert np.array_equal(_np(ol), GOLD["%s/%d/label" % (name, seed)])


def test_golden_covers_every_branch():
    """the seeds must exercise rotate on/off, blur on/off, flip on/off, padding, both crop types"""
    from semseg_amd import transform as T
    seen = set()
    for name, (H, W, ops, seeds) in tc.CASES.items():
        for seed in seeds:
            random.seed(seed)
            plan = tc.build_chain(T, ops).plan(H, W)
            kinds = [it["k"] for it in plan.items]
            seen.add("rotate" if "rotate" in kinds else "no_rotate")
            seen.add("blur" if "blur" in kinds else "no_blur")
            for it in plan.items:
                if it["k"] == "map" and it["sx"] == -1:
                    seen.add("hflip")
                if it["k"] == "map" and it["sy"] == -1:
                    seen.add("vflip")
                if it["k"] == "map" and (it["oy"] < 0 or it["ox"] < 0):
                    seen.add("pad")
    assert {"rotate", "no_rotate", "blur", "no_blur", "hflip", "vflip", "pad"} <= seen


# ---------------- independent cross-checks of the restated cv2 primitives ----------------
@pytest.mark.parametrize("fx,fy", [(0.5, 0.5), (0.76, 0.76), (1.0, 1.0), (1.31, 0.9), (2.0, 2.0)])
def test_resize_linear_vs_torch_half_pixel(fx, fy):
    """cv2 samples with the REQUESTED factor (scale = 1/fx), like torch with recompute_scale_factor=False"""
    rng = np.random.default_rng(1)
    img = (rng.random((37, 53, 3)) * 255).astype(np.float32)
    out = ocv.resize(img, None, fx=fx, fy=fy, interpolation=ocv.INTER_LINEAR)
    t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
    ref = F.interpolate(t, scale_factor=(fy, fx), mode="bilinear", align_corners=False, recompute_scale_factor=False)
    assert ref.shape[2:] == out.shape[:2]          # factors chosen so that floor (torch) == round (cv2)
    ref = ref[0].permute(1, 2, 0).numpy()
    # cv2 rounds the source coordinate to float32 (<= 2^-18 px at these sizes) and accumulates in float32
    assert np.abs(out - ref).max() < 2e-3


def test_resize_identity_and_nearest():
    rng = np.random.default_rng(2)
    img = (rng.random((20, 31, 3)) * 255).astype(np.float32)
    lab = rng.integers(0, 255, size=(20, 31)).astype(np.uint8)
    assert np.array_equal(ocv.resize(img, None, fx=1.0, fy=1.0), img)
    assert np.array_equal(ocv.resize(lab, None, fx=1.0, fy=1.0, interpolation=ocv.INTER_NEAREST), lab)
    up = ocv.resize(lab, None, fx=2.0, fy=2.0, interpolation=ocv.INTER_NEAREST)
    assert np.array_equal(up, np.repeat(np.repeat(lab, 2, axis=0), 2, axis=1))
    dn = ocv.resize(lab, None, fx=0.5, fy=0.5, interpolation=ocv.INTER_NEAREST)
    assert np.array_equal(dn, lab[0::2, 0::2][:10, :16])


@pytest.mark.parametrize("k", [3, 5, 7])
def test_blur_vs_scipy_mirror(k):
    from scipy.ndimage import correlate1d
    rng = np.random.default_rng(3)
    img = (rng.random((33, 29, 3)) * 255).astype(np.float32)
    out = ocv.GaussianBlur(img, (k, k), 0)
    w = np.asarray(ocv.SMALL_GAUSSIAN_TAB[k], dtype=np.float64)
    assert abs(w.sum() - 1.0) < 1e-15
    ref = correlate1d(correlate1d(img.astype(np.float64), w, axis=1, mode="mirror"), w, axis=0, mode="mirror")
    assert np.abs(out - ref).max() < 1e-4
    # the tabulated 5-tap kernel is the sigma=1.1 Gaussian OpenCV derives for ksize 5, rounded to 1/16ths
    if k == 5:
        g = np.exp(-np.arange(-2, 3) ** 2 / (2 * 1.1 ** 2))
        assert np.abs(g / g.sum() - w).max() < 0.02


@pytest.mark.parametrize("angle", [-37.0, -10.0, 3.3, 25.0, 90.0])
def test_rotate_vs_scipy_affine(angle):
    from scipy.ndimage import affine_transform
    H, W = 48, 64
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.stack([100 + 50 * np.sin(yy / 6) + 40 * np.cos(xx / 9 + c) for c in range(3)], axis=2).astype(np.float32)
    M = ocv.getRotationMatrix2D((W / 2, H / 2), angle, 1)
    out = ocv.warpAffine(img, M, (W, H), flags=ocv.INTER_LINEAR, borderValue=[7.0, 8.0, 9.0])
    m = ocv.invert_affine(M)                       # dst (x, y) -> src (x, y)
    A = np.array([[m[4], m[3]], [m[1], m[0]]])     # scipy works in (row, col)
    off = np.array([m[5], m[2]])
    inside = np.ones((H, W), dtype=np.float64)
    for c in range(3):
        ref = affine_transform(img[:, :, c].astype(np.float64), A, offset=off, order=1, mode="constant", cval=7.0 + c)
        cover = affine_transform(inside, A, offset=off, order=1, mode="constant", cval=0.0)
        sel = cover > 0.999                        # away from the border blend
        # 1/32 px coordinate grid x gradient (<= ~10 per px here) bounds the difference
        assert np.abs(out[:, :, c][sel] - ref[sel]).max() < 0.5
    # 90 degrees about (w/2, h/2) — the reference's centre, half a pixel off the pixel-grid centre — maps the grid onto
    # itself shifted by one row: destination row 0 falls outside (border), the rest is rot90 exactly (labels: nearest)
    lab = (np.arange(32 * 32).reshape(32, 32) % 251).astype(np.uint8)
    M90 = ocv.getRotationMatrix2D((16, 16), 90.0, 1)
    r = ocv.warpAffine(lab, M90, (32, 32), flags=ocv.INTER_NEAREST, borderValue=255)
    assert (r[0] == 255).all() and np.array_equal(r[1:, :], np.rot90(lab, 1)[:-1, :])


def test_rotate_zero_is_identity():
    rng = np.random.default_rng(4)
    img = (rng.random((21, 34, 3)) * 255).astype(np.float32)
    lab = rng.integers(0, 255, size=(21, 34)).astype(np.uint8)
    M = ocv.getRotationMatrix2D((17, 10.5), 0.0, 1)
    assert np.array_equal(ocv.warpAffine(img, M, (34, 21), flags=ocv.INTER_LINEAR, borderValue=[1, 2, 3]), img)
    assert np.array_equal(ocv.warpAffine(lab, M, (34, 21), flags=ocv.INTER_NEAREST, borderValue=255), lab)


def test_reflect101():
    assert ocv.border_reflect_101(np.array([-2, -1, 0, 4, 5, 6]), 5).tolist() == [2, 1, 0, 4, 3, 2]
    assert ocv.border_reflect_101(np.array([-3, 3]), 2).tolist() == [1, 1]


# ---------------- planner (host side of the product; no kernels run) ----------------
def test_planner_draws_like_the_oracle():
    """same seed -> same number of draws: after planning, the next random number must match the oracle's"""
    from semseg_amd import transform as T
    for name, (H, W, ops, seeds) in tc.CASES.items():
        img, lab = tc.make_input(name, H, W)
        for seed in seeds:
            random.seed(seed)
            otf.run(ops, np.float32(img), lab.copy())
            after_oracle = random.random()
            random.seed(seed)
            plan = tc.build_chain(T, ops).plan(H, W)
            assert random.random() == after_oracle, (name, seed)
            gi = GOLD["%s/%d/image" % (name, seed)]
            hw = gi.shape[1:] if plan.tensor else gi.shape[:2]
            assert (plan.h, plan.w) == tuple(hw)


def test_region_propagation_covers_every_tap():
    from semseg_amd import transform as T
    rng = random.Random(5)
    for _ in range(200):
        n, scale = rng.randint(3, 300), rng.uniform(0.3, 3.0)
        nd = max(int(round(n / scale)), 1)
        a = rng.randint(0, nd - 1)
        b = rng.randint(a, nd - 1)
        lo, hi = T._resize_need(a, b, scale, n)
        s0, s1, _, _ = ocv._linear_coeffs(nd, n, scale)
        near = np.minimum(np.floor(np.arange(nd) * scale).astype(np.int64), n - 1)
        assert lo <= min(s0[a:b + 1].min(), near[a:b + 1].min()) and hi >= max(s1[a:b + 1].max(), near[a:b + 1].max())
        assert 0 <= lo <= hi <= n - 1
    for _ in range(60):
        H, W = rng.randint(8, 90), rng.randint(8, 90)
        m = T._rotation_inverse(W / 2, H / 2, rng.uniform(-60, 60))
        y0, x0 = rng.randint(0, H - 1), rng.randint(0, W - 1)
        h, w = rng.randint(1, H - y0), rng.randint(1, W - x0)
        stage = dict(k="rotate", in_h=H, in_w=W, m=m)
        ny, nx, nh, nw = T._need(stage, (y0, x0, h, w))
        for flags in (ocv.INTER_LINEAR, ocv.INTER_NEAREST):
            X, Y = ocv.affine_fixed_coords(m, W, H, flags)
            X, Y = X[y0:y0 + h, x0:x0 + w], Y[y0:y0 + h, x0:x0 + w]
            if flags == ocv.INTER_LINEAR:
                X, Y = X >> 5, Y >> 5
                taps = [(Y + dy, X + dx) for dy in (0, 1) for dx in (0, 1)]
            else:
                taps = [(Y, X)]
            for ty, tx in taps:
                ok = (ty >= 0) & (ty < H) & (tx >= 0) & (tx < W)
                if ok.any():
                    assert nh > 0 and ty[ok].min() >= ny and ty[ok].max() < ny + nh
                    assert tx[ok].min() >= nx and tx[ok].max() < nx + nw


def test_rotation_matrix_matches_oracle():
    from semseg_amd import transform as T
    for (W, H, ang) in [(131, 97, -7.25), (64, 48, 33.0), (713, 713, 10.0), (50, 51, -0.001)]:
        a = T._rotation_inverse(W / 2, H / 2, ang)
        b = ocv.invert_affine(ocv.getRotationMatrix2D((W / 2, H / 2), ang, 1))
        assert a == b


def test_api_errors_like_the_reference():
    """util/transform.py raises RuntimeError / AssertionError on these (lines 79-92, 113-141, 170-185, 151-152)"""
    from semseg_amd import transform as T
    with pytest.raises(RuntimeError):
        T.RandScale([2.0, 0.5])
    with pytest.raises(RuntimeError):
        T.RandScale([0.5, 2.0], aspect_ratio=[1.5, 0.5])
    with pytest.raises(AssertionError):
        T.RandScale(0.5)
    with pytest.raises(RuntimeError):
        T.Crop([0, 5])
    with pytest.raises(RuntimeError):
        T.Crop(5, crop_type="corner")
    with pytest.raises(RuntimeError):
        T.Crop(5, padding=[1, 2])
    with pytest.raises(RuntimeError):
        T.Crop(5, padding=(1, 2, 3))
    with pytest.raises(RuntimeError):
        T.Crop(5, ignore_label=2.5)
    with pytest.raises(RuntimeError):
        T.RandRotate([10, -10], padding=[0, 0, 0])
    with pytest.raises(AssertionError):
        T.RandRotate([-10, 10], padding=None)
    with pytest.raises(AssertionError):
        T.Normalize([1, 2, 3], [1, 2])
    with pytest.raises(RuntimeError):          # Crop must pad but has no padding value (transform.py:151-152)
        T.Compose([T.Crop(50)]).plan(20, 20)
    with pytest.raises(RuntimeError):
        T.Compose([T.Normalize([1, 2, 3])]).plan(20, 20)
    with pytest.raises(RuntimeError):
        T.Compose([T.ToTensor(), T.RandomHorizontalFlip()]).plan(20, 20)


def test_stage_grouping():
    from semseg_amd import transform as T
    random.seed(0)
    c = tc.build_chain(T, tc.CASES["crop_then_scale"][2])
    st = c.plan(80, 80)
    kinds = [s["k"] for s in T._stages(st, 80, 80)]
    assert kinds == ["gather", "resize", "gather"]          # maps before a resampling op are materialised first
    c = tc.build_chain(T, tc.CASES["test_only_tensor"][2])
    assert [s["k"] for s in T._stages(c.plan(40, 33), 40, 33)] == ["gather"]
    c = T.Compose([T.RGB2BGR()] * 8 + [T.ToTensor()])
    sg = T._stages(c.plan(10, 10), 10, 10)
    assert [len(s["maps"]) for s in sg] == [6, 2] and [s["final"] for s in sg] == [False, True]


def test_f1_and_f3_restatements_of_cv2_resize_agree():
    """oracle/test_pipeline.py (test-time pipeline, float64 torch bilinear) and oracle/cv2_restated.py (float32 two-pass,
    OpenCV's coefficient rounding) restate the same cv2.resize(INTER_LINEAR) call: they must agree to float32 noise."""
    from oracle import test_pipeline as tp
    rng = np.random.default_rng(7)
    img = (rng.random((61, 83, 3)) * 255).astype(np.float32)
    for (nh, nw) in [(61, 83), (92, 125), (30, 41), (123, 166)]:
        a = tp.cv2_resize_linear(img, nw, nh)
        b = ocv.resize(img, (nw, nh), interpolation=ocv.INTER_LINEAR)
        assert a.shape == b.shape and np.abs(a - b).max() < 2e-3


# ---------------- the host schedule executed on the CPU, region by region ----------------
def _emulate(ops, img, lab, seed):
    from semseg_amd import transform as T
    import transform_emulator as em
    chain_obj = tc.build_chain(T, ops)
    random.seed(seed)
    plans, chains = chain_obj.schedule([lab.shape])
    return em.run_chain(plans[0], chains[0], img, lab)


@pytest.mark.parametrize("name", sorted(tc.CASES))
def test_schedule_reproduces_golden_on_cpu(name):
    """stage grouping + index-map composition + needed regions: executing the schedule with every unmaterialised pixel
    poisoned gives the reference's output bit for bit"""
    H, W, ops, seeds = tc.CASES[name]
    img, lab = tc.make_input(name, H, W)
    for seed in seeds:
        gi, gl = _emulate(ops, img, lab, seed)
        assert np.array_equal(gl, GOLD["%s/%d/label" % (name, seed)]), (name, seed)
        assert np.array_equal(gi, GOLD["%s/%d/image" % (name, seed)]), (name, seed)


def test_schedule_random_chains_match_oracle():
    """random sizes, crops, orders and seeds beyond the golden cases (property test of the region propagation)"""
    rng = random.Random(11)
    for trial in range(60):
        H, W = rng.randint(24, 90), rng.randint(24, 90)
        ch, cw = rng.randint(9, 70), rng.randint(9, 70)
        geo = [("rand_scale", (0.5, 2.0), rng.choice([None, (0.7, 1.4)])),
               ("rand_rotate", (-rng.uniform(5, 40), rng.uniform(5, 40)), tc.MEAN, 255, rng.choice([0.5, 1.0])),
               ("rand_blur", rng.choice([3, 5, 7])), ("rand_hflip", 0.5), ("rand_vflip", 0.5),
               ("crop", (ch, cw), rng.choice(["rand", "center"]), tc.MEAN, 255), ("swap_rb",),
               ("resize", (rng.randint(16, 60), rng.randint(16, 60)))]
        rng.shuffle(geo)
        ops = geo[:rng.randint(1, len(geo))]
        if rng.random() < 0.7:
            ops = ops + [("to_tensor",)] + ([("normalize", tc.MEAN, rng.choice([tc.STD, None]))] if rng.random() < 0.7 else [])
        img, lab = tc.make_input("r%d" % trial, H, W)
        seed = rng.randint(0, 10 ** 6)
        random.seed(seed)
        try:
            oi, ol = otf.run(ops, np.float32(img), lab.copy())
        except AssertionError:
            continue                                    # the drawn scale collapsed the image to nothing
        gi, gl = _emulate(ops, img, lab, seed)
        assert np.array_equal(gl, _np(ol)), (trial, ops)
        assert np.array_equal(gi, _np(oi)), (trial, ops)
