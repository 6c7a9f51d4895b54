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


class _HostChain(object):
    """Stand-in for transform.Compose on a box without a GPU: same `batch` call shape, centre-crops on the host."""

    def __init__(self, size):
        self.size = size
        self.pids = []

    def batch(self, images, labels, stack=True, device=None):
        import torch
        self.pids.append(os.getpid())
        s = self.size
        x = torch.stack([torch.from_numpy(np.ascontiguousarray(im[:s, :s].transpose(2, 0, 1))).float() for im in images])
        y = torch.stack([torch.from_numpy(np.ascontiguousarray(lb[:s, :s])).long() for lb in labels])
        return x, y


def test_workers_decode_only_and_chain_runs_in_training_process(tmp_path):
    """num_workers = 2: the workers return raw uint8 pairs, the chain runs in THIS process when the batch is taken;
    DeviceCollate / SemData(transform=) refuse to run inside a worker (util/dataset.py:61-71 decodes in 16 workers)."""
    import torch
    from semseg_amd import dataset as D
    tmp = str(tmp_path)
    samples = _write_set(tmp, n=6)
    ds = D.SemData("train", tmp, os.path.join(tmp, "train.txt"))
    chain = _HostChain(40)
    inner = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, num_workers=2, collate_fn=D.raw_collate)
    loader = D.DeviceLoader(inner, chain)
    assert len(loader) == 2 and loader.batch_size == 3 and loader.dataset is ds
    seen = 0
    for x, y in loader:
        assert tuple(x.shape) == (3, 3, 40, 40) and y.dtype == torch.int64
        for j in range(3):
            img, lab = samples[seen + j]
            assert np.array_equal(x[j].numpy(), img[:40, :40].transpose(2, 0, 1).astype(np.float32))
            assert np.array_equal(y[j].numpy(), lab[:40, :40].astype(np.int64))
        seen += 3
    assert seen == 6 and set(chain.pids) == {os.getpid()}
    # the old recipe (chain as the workers' collate_fn) fails loudly instead of initialising HIP in a forked child
    bad = torch.utils.data.DataLoader(ds, batch_size=3, num_workers=2, collate_fn=D.DeviceCollate(chain))
    with pytest.raises(RuntimeError, match="worker"):
        next(iter(bad))
    ds_t = D.SemData("val", tmp, os.path.join(tmp, "train.txt"), transform=lambda a, b: (a, b))
    bad2 = torch.utils.data.DataLoader(ds_t, batch_size=1, num_workers=1, collate_fn=D.raw_collate)
    with pytest.raises(RuntimeError, match="worker"):
        next(iter(bad2))
    with pytest.raises(RuntimeError, match="raw_collate"):
        D.DeviceLoader(torch.utils.data.DataLoader(ds, batch_size=3), chain)


@pytest.mark.gpu
def test_device_loader_with_worker_processes(tmp_path):
    """The documented recipe on the GPU: 2 decode workers + the device chain in the
    training process, bit-exact against the oracle."""
    import torch
    from semseg_amd import dataset as D, transform as T
    from oracle import transform as otf
    tmp = str(tmp_path)
    samples = _write_set(tmp, n=8)
    ops = tc.train_chain((49, 49))
    torch.zeros(1, device="cuda")       # HIP is initialised in the parent BEFORE the workers fork
    ds = D.SemData("train", tmp, os.path.join(tmp, "train.txt"))
    inner = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=2, collate_fn=D.raw_collate)
    loader = D.DeviceLoader(inner, tc.build_chain(T, ops))
    random.seed(5)
    got = [(x, y) for x, y in loader]
    assert len(got) == 2
    random.seed(5)
    for b, (x, y) in enumerate(got):
        assert x.is_cuda and tuple(x.shape) == (4, 3, 49, 49) and y.dtype == torch.int64
        for i in range(4):
            img, lab = samples[4 * b + i]
            oi, ol = otf.run(ops, np.float32(img), lab.copy())
            assert np.array_equal(x[i].cpu().numpy(), oi.numpy()) and np.array_equal(y[i].cpu().numpy(), ol.numpy())


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
