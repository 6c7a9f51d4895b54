"""util/dataset.py mirror: list parsing and error behaviour (CPU), decode + device collate (GPU)."""
import os
import random
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import transform_cases as tc          # noqa: E402


def _write_set(tmp, n=3):
    from PIL import Image
    lines = []
    samples = []
    for i in range(n):
        img, lab = tc.make_input("ds%d" % i, 70 + 5 * i, 90 - 3 * i)
        Image.fromarray(img, "RGB").save(os.path.join(tmp, "im%d.png" % i))
        Image.fromarray(lab, "L").save(os.path.join(tmp, "lb%d.png" % i))
        lines.append("im%d.png lb%d.png" % (i, i))
        samples.append((img, lab))
    with open(os.path.join(tmp, "train.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(tmp, "test.txt"), "w") as f:
        f.write("\n".join(l.split()[0] for l in lines) + "\n")
    return samples


def test_make_dataset_and_decode(tmp_path):
    from semseg_amd import dataset as D
    tmp = str(tmp_path)
    samples = _write_set(tmp)
    pairs = D.make_dataset("train", tmp, os.path.join(tmp, "train.txt"))
    assert pairs[1] == (os.path.join(tmp, "im1.png"), os.path.join(tmp, "lb1.png"))
    tpairs = D.make_dataset("test", tmp, os.path.join(tmp, "test.txt"))
    assert tpairs[2][0] == tpairs[2][1] == os.path.join(tmp, "im2.png")     # placeholder label (dataset.py:33)
    with pytest.raises(RuntimeError):
        D.make_dataset("train", tmp, os.path.join(tmp, "missing.txt"))
    with pytest.raises(RuntimeError):
        D.make_dataset("train", tmp, os.path.join(tmp, "test.txt"))          # one column where two are needed
    with pytest.raises(RuntimeError):
        D.make_dataset("test", tmp, os.path.join(tmp, "train.txt"))
    with pytest.raises(AssertionError):
        D.make_dataset("trainval", tmp, os.path.join(tmp, "train.txt"))
    ds = D.SemData("train", tmp, os.path.join(tmp, "train.txt"))
    assert len(ds) == 3
    for i, (img, lab) in enumerate(samples):
        gi, gl = ds[i]
        assert gi.dtype == np.uint8 and np.array_equal(gi, img) and np.array_equal(gl, lab)
    ts = D.SemData("test", tmp, os.path.join(tmp, "test.txt"))
    assert ts[0][1].shape == samples[0][1].shape
    # mismatched label size (dataset.py:65-66)
    from PIL import Image
    Image.fromarray(samples[0][1][:10], "L").save(os.path.join(tmp, "lb0.png"))
    with pytest.raises(RuntimeError):
        ds[0]


@pytest.mark.gpu
def test_loader_with_device_collate(tmp_path):
    import torch
    from semseg_amd import dataset as D, transform as T
    from oracle import transform as otf
    tmp = str(tmp_path)
    samples = _write_set(tmp, n=4)
    ops = tc.train_chain((49, 49))
    ds = D.SemData("train", tmp, os.path.join(tmp, "train.txt"))
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=0,
                                         collate_fn=D.DeviceCollate(tc.build_chain(T, ops)))
    random.seed(3)
    x, y = next(iter(loader))
    assert x.is_cuda and tuple(x.shape) == (4, 3, 49, 49) and y.dtype == torch.int64
    random.seed(3)
    for i, (img, lab) in enumerate(samples):
        oi, ol = otf.run(ops, np.float32(img), lab.copy())
        assert np.array_equal(x[i].cpu().numpy(), oi.numpy()) and np.array_equal(y[i].cpu().numpy(), ol.numpy())
    # the reference call shape: per-sample transform inside the dataset
    ds2 = D.SemData("val", tmp, os.path.join(tmp, "train.txt"), transform=tc.build_chain(T, tc.val_chain((65, 65))))
    vi, vl = ds2[1]
    oi, ol = otf.run(tc.val_chain((65, 65)), np.float32(samples[1][0]), samples[1][1].copy())
    assert np.array_equal(vi.cpu().numpy(), oi.numpy()) and np.array_equal(vl.cpu().numpy(), ol.numpy())
