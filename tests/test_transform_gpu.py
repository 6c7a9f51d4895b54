"""GPU parity of the device-side input pipeline (semseg_amd/transform.py + csrc/augment.hip through
the C ABI semseg_augment_round) against the golden outputs of the reference's own transform classes and, at the
reference's real image sizes, against the oracle on the same seeds.  Bit-exact: labels are integer work, and the
float stages evaluate the same IEEE expressions in the same order (no FMA contraction) as the oracle."""
import os
import random
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import transform_cases as tc                    # noqa: E402
from oracle import transform as otf             # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(HERE, "golden", "transform_ref.npz"))


def _np(v):
    return v.cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


def _T():
    from semseg_amd import transform as T
    return T


@pytest.mark.parametrize("name", sorted(tc.CASES))
def test_golden(name):
    T = _T()
    H, W, ops, seeds = tc.CASES[name]
    img, lab = tc.make_input(name, H, W)
    chain = tc.build_chain(T, ops)
    for seed in seeds:
        for src in (img, np.float32(img)):                   # decoded uint8, and float32 as SemData hands it over
            random.seed(seed)
            gi, gl = chain(src, lab)
            ref_i, ref_l = GOLD["%s/%d/image" % (name, seed)], GOLD["%s/%d/label" % (name, seed)]
            gi, gl = _np(gi), _np(gl)
            assert gi.dtype == np.float32 and gi.shape == ref_i.shape
            assert gl.dtype == (np.int64 if any(o[0] == "to_tensor" for o in ops) else np.uint8)
            assert np.array_equal(gl, ref_l), (name, seed)
            bad = np.flatnonzero(gi.reshape(-1) != ref_i.reshape(-1))
            assert bad.size == 0, "%s/%d: %d of %d values differ, max |d| %g" % (
                name, seed, bad.size, gi.size, np.abs(gi - ref_i).max())


def test_batch_is_the_sequence_of_singles():
    T = _T()
    H, W, ops, _ = tc.CASES["train_65"]
    imgs, labs = zip(*[tc.make_input("b%d" % i, H + 3 * i, W - 2 * i) for i in range(8)])
    chain = tc.build_chain(T, ops)
    random.seed(123)
    bi, bl = chain.batch(list(imgs), list(labs))
    assert tuple(bi.shape) == (8, 3, 65, 65) and bi.dtype == torch.float32
    assert tuple(bl.shape) == (8, 65, 65) and bl.dtype == torch.int64
    random.seed(123)
    for i in range(8):
        oi, ol = otf.run(ops, np.float32(imgs[i]), labs[i].copy())
        assert np.array_equal(_np(bi[i]), oi.numpy()) and np.array_equal(_np(bl[i]), ol.numpy()), i


def test_device_resident_sources_and_mixed_sizes():
    T = _T()
    ops = tc.CASES["no_tensor"][2]
    a = tc.make_input("x", 60, 70)
    b = tc.make_input("y", 90, 55)
    chain = tc.build_chain(T, ops)
    dev = [(torch.from_numpy(i).cuda(), torch.from_numpy(l).cuda()) for i, l in (a, b)]
    random.seed(9)
    ims, lbs = chain.batch([d[0] for d in dev], [d[1] for d in dev])
    random.seed(9)
    for k, (i, l) in enumerate((a, b)):
        oi, ol = otf.run(ops, np.float32(i), l.copy())
        assert np.array_equal(_np(ims[k]), oi) and np.array_equal(_np(lbs[k]), ol)


def _big_input(H, W, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    base = np.stack([127 + 90 * np.sin(yy / 37.0 + c) * np.cos(xx / 23.0 - c) for c in range(3)], axis=2)
    img = np.clip(base + rng.normal(0, 20, size=(H, W, 3)), 0, 255).astype(np.uint8)
    lab = ((yy // 41 + xx // 57) % 150).astype(np.uint8)
    lab[rng.random((H, W)) < 0.03] = 255
    return img, lab


@pytest.mark.parametrize("H,W,crop,seeds", [(512, 683, 473, [0, 1, 2, 3, 4, 5]),      # ADE20K-like, ade20k_pspnet50.yaml
                                             (1024, 2048, 713, [0, 4, 9, 10])])            # Cityscapes, cityscapes_pspnet50.yaml
def test_train_chain_at_dataset_sizes(H, W, crop, seeds):
    T = _T()
    ops = tc.train_chain((crop, crop))
    img, lab = _big_input(H, W, H)
    chain = tc.build_chain(T, ops)
    kinds = set()
    for seed in seeds:
        random.seed(seed)
        gi, gl, plans = chain.batch([img], [lab], return_plans=True)
        kinds |= {it["k"] for it in plans[0].items}
        random.seed(seed)
        oi, ol = otf.run(ops, np.float32(img), lab.copy())
        assert np.array_equal(_np(gl[0]), ol.numpy()), seed
        d = np.abs(_np(gi[0]) - oi.numpy())
        assert d.max() == 0, "seed %d: %d values differ, max %g" % (seed, int((d > 0).sum()), d.max())
    assert {"resize", "rotate", "blur", "map"} <= kinds


def test_properties_at_full_size():
    T = _T()
    img, lab = _big_input(1024, 2048, 7)
    mean, std = tc.MEAN, tc.STD
    # val chain = plain centre crop + normalise
    vi, vl = T.Compose([T.Crop([713, 713], crop_type="center", padding=mean, ignore_label=255), T.ToTensor(),
                        T.Normalize(mean, std)])(img, lab)
    y0, x0 = int((1024 - 713) / 2), int((2048 - 713) / 2)
    ref = torch.from_numpy(np.float32(img[y0:y0 + 713, x0:x0 + 713]).transpose(2, 0, 1).copy())
    for c in range(3):
        ref[c].sub_(mean[c]).div_(std[c])
    assert torch.equal(vi.cpu(), ref) and np.array_equal(_np(vl), lab[y0:y0 + 713, x0:x0 + 713].astype(np.int64))
    # two flips and a unit-factor resize are the identity
    ii, ll = T.Compose([T.RandomHorizontalFlip(1.0), T.RandomVerticalFlip(1.0), T.Resize((1024, 2048)),
                        T.RandomVerticalFlip(1.0), T.RandomHorizontalFlip(1.0)])(img, lab)
    assert np.array_equal(_np(ii), np.float32(img)) and np.array_equal(_np(ll), lab)
    # same seed -> same bits; labels only take source values or the ignore label
    chain = tc.build_chain(T, tc.train_chain((713, 713)))
    random.seed(77)
    a = chain(img, lab)
    random.seed(77)
    b = chain(img, lab)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert set(np.unique(_np(a[1]))) <= set(np.unique(lab)) | {255}
    # padding: an image smaller than the crop is centred in mean / ignore (normalised mean = 0)
    si, sl = T.Compose([T.Crop([473, 473], crop_type="center", padding=mean, ignore_label=255), T.ToTensor(),
                        T.Normalize(mean, std)])(img[:100, :200], lab[:100, :200])
    si, sl = _np(si), _np(sl)
    t, l = int(373 / 2), int(273 / 2)
    assert (sl[:t] == 255).all() and (sl[t + 100:] == 255).all() and (sl[:, :l] == 255).all()
    assert np.abs(si[:, :t]).max() < 1e-6 and np.array_equal(sl[t:t + 100, l:l + 200], lab[:100, :200].astype(np.int64))


def test_rejects_bad_input():
    T = _T()
    c = T.Compose([T.ToTensor()])
    with pytest.raises(RuntimeError):
        c(np.zeros((4, 5), dtype=np.uint8), np.zeros((4, 5), dtype=np.uint8))          # 2-dim image
    with pytest.raises(RuntimeError):
        c(np.zeros((4, 5, 3), dtype=np.uint8), np.zeros((4, 6), dtype=np.uint8))       # shape mismatch (dataset.py:65-66)
    with pytest.raises(RuntimeError):
        c(np.zeros((4, 5, 3), dtype=np.float64), np.zeros((4, 5), dtype=np.uint8))


@pytest.mark.parametrize("ops", [[("rand_scale", (0.7, 1.6), None)],
                                 [("rand_rotate", (-30, 30), tc.MEAN, 255, 1.0)],
                                 [("rand_hflip", 1.0), ("rand_blur", 5)],
                                 [("resize", (50, 41)), ("to_tensor",)],
                                 [("swap_rb",)] * 8 + [("rand_scale", (1.2, 1.3), None), ("swap_rb",), ("to_tensor",)]],
                         ids=["scale_only", "rotate_only", "flip_blur", "resize_tensor", "many_maps"])
def test_chains_ending_in_a_resampling_stage(ops):
    """no final gather (the last stage's region IS the result), > 6 index maps in a row, blur as the last stage"""
    T = _T()
    img, lab = tc.make_input("tail", 57, 66)
    chain = tc.build_chain(T, ops)
    for seed in (1, 2, 3):
        random.seed(seed)
        gi, gl = chain(img, lab)
        random.seed(seed)
        oi, ol = otf.run(ops, np.float32(img), lab.copy())
        assert np.array_equal(_np(gi), _np(oi)) and np.array_equal(_np(gl), _np(ol)), seed
