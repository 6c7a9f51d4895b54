"""The pieces either side of the train step composed the way tool/train.py:194-304 composes them: list file ->
SemData (decode in 2 worker processes) -> DeviceLoader (device-side transform chain in the training process) -> fused train step (forward, two
cross-entropy losses, backward, SGD with poly LR) -> intersectionAndUnionGPU on the returned prediction -> state-dict
save / load.  Checks plumbing (shapes, dtypes, devices), that the loss goes down on a learnable toy set, and that a
reloaded checkpoint predicts identically."""
import io
import os
import random
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import transform_cases as tc          # noqa: E402

pytestmark = pytest.mark.gpu


def _toy_set(tmp, n, classes):
    """images whose class is readable from colour: vertical bands of `classes` hues + noise; label = band index"""
    from PIL import Image
    rng = np.random.default_rng(0)
    lines = []
    for i in range(n):
        H, W = 80 + 4 * (i % 3), 100 + 6 * (i % 4)
        band = (np.arange(W) * classes // W + i) % classes
        lab = np.repeat(band[None, :], H, axis=0).astype(np.uint8)
        palette = np.array([[220, 40, 40], [40, 220, 40], [40, 40, 220], [220, 220, 40]], dtype=np.float64)[:classes]
        img = np.clip(palette[lab] + rng.normal(0, 10, size=(H, W, 3)), 0, 255).astype(np.uint8)
        lab[rng.random((H, W)) < 0.02] = 255
        Image.fromarray(img, "RGB").save(os.path.join(tmp, "im%d.png" % i))
        Image.fromarray(lab, "L").save(os.path.join(tmp, "lb%d.png" % i))
        lines.append("im%d.png lb%d.png" % (i, i))
    with open(os.path.join(tmp, "train.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


def test_train_loop_end_to_end(tmp_path):
    from model.pspnet import PSPNet
    from semseg_amd import dataset as D, transform as T
    from semseg_amd.metrics import intersectionAndUnionGPU
    from semseg_amd.trainer import Trainer, poly_learning_rate
    tmp = str(tmp_path)
    classes, crop, bs = 4, 65, 4
    _toy_set(tmp, 8, classes)
    chain = tc.build_chain(T, tc.train_chain((crop, crop), scale=(0.8, 1.25), rotate=(-10, 10)))
    torch.zeros(1, device="cuda")
    loader = D.DeviceLoader(torch.utils.data.DataLoader(D.SemData("train", tmp, os.path.join(tmp, "train.txt")),
                                                        batch_size=bs, shuffle=True, num_workers=2, drop_last=True,
                                                        collate_fn=D.raw_collate), chain)
    torch.manual_seed(0)
    random.seed(0)
    model = PSPNet(layers=50, classes=classes, zoom_factor=8, dropout=0.1, pretrained=False).cuda().train()
    tr = Trainer(model, base_lr=0.01, sync_bn=True)
    epochs, it, max_iter = 15, 0, 15 * len(loader)
    losses, accs = [], []
    for _ in range(epochs):
        for x, y in loader:
            assert x.is_cuda and x.dtype == torch.float32 and tuple(x.shape) == (bs, 3, crop, crop)
            assert y.is_cuda and y.dtype == torch.int64 and tuple(y.shape) == (bs, crop, crop)
            pred, main_loss, aux_loss = tr.step(x, y, poly_learning_rate(0.01, it, max_iter))
            inter, union, target = intersectionAndUnionGPU(pred, y, classes, 255)
            losses.append(float(main_loss))
            accs.append(float(inter.sum() / (target.sum() + 1e-10)))
            it += 1
    assert all(np.isfinite(losses))
    first, last = np.mean(losses[:4]), np.mean(losses[-4:])
    assert last < 0.8 * first, (first, last, losses)              # colour -> class: the loss must come down in 30 steps
    assert np.mean(accs[-6:]) > np.mean(accs[:6]), (accs[:6], accs[-6:])

    # checkpoint round trip in the reference's format (train.py:231-237 saves state_dict; test.py:110-113 loads it)
    buf = io.BytesIO()
    torch.save({"epoch": epochs, "state_dict": model.state_dict()}, buf)
    buf.seek(0)
    model2 = PSPNet(layers=50, classes=classes, zoom_factor=8, pretrained=False).cuda()
    model2.load_state_dict(torch.load(buf)["state_dict"], strict=True)
    model.eval()
    model2.eval()
    xv, _ = next(iter(loader))
    with torch.no_grad():
        a, b = model(xv), model2(xv)
    assert tuple(a.shape) == (bs, classes, crop, crop) and torch.equal(a, b)
