"""The two BASELINE.json configurations that carry the headline numbers, at their REAL sizes, against fixtures produced by the
imported reference itself (tests/golden/make_golden_headline.py, run in the build container):

  * metric configuration: PSPNet-101, 473x473, 150 classes, BATCH 16 train step (model/pspnet.py:80-105) — through
    `Trainer.step` (what bench.py times) and through the drop-in nn.Module path (`loss.backward()`), in the default
    arithmetic (bf16x3) and with exact fp32 forced;
  * configs[4]: the multi-scale test path (tool/test.py:149-204) on a 512x512 image, base_size 512, crop 473, the six ADE
    scales = 23 crops = 46 forwards through `MultiScaleTester`.

Bounds (fixed before the first run): losses 1e-5 relative (the bound of every other train-loss check here); argmax sample
agreement >= 0.999; running statistics 1e-4 of their maximum; gradients of the last conv of each head 5e-4 of their maximum (the
tests against the fp64 oracle use 2e-4 for each of the two fp32 implementations compared here); norms of all 340 gradients:
median deviation <= 2e-3, 90 % quantile <= 1e-2, maximum <= 1e-1 (ReLU-mask flips make deeper gradients of two fp32
implementations differ element-wise, their norms far less); multi-scale probabilities 2e-4 absolute, argmax agreement >= 0.998
(the bounds of tests/test_infer_gpu.py)."""
import os

import numpy as np
import pytest
import torch

from test_model_gpu import build, inputs

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))


def _check_train(report, name, gold, pred, ml, al, grads, bufs):
    e_ml = abs(ml - float(gold["main_loss"])) / abs(float(gold["main_loss"]))
    e_al = abs(al - float(gold["aux_loss"])) / abs(float(gold["aux_loss"]))
    agree = float((pred[:, ::5, ::5].cpu().numpy().astype(np.uint8) == gold["pred_sample"]).mean())
    e_buf = {k[4:]: _rel(bufs[k[4:]].cpu().numpy(), gold[k]) for k in gold.files if k.startswith("buf/")}
    e_grad = {k[5:]: _rel(grads[k[5:]].cpu().numpy(), gold[k]) for k in gold.files if k.startswith("grad/")}
    names = [str(n) for n in gold["gnorm_names"]]
    gn = np.array([float(grads[n].double().norm().item()) for n in names])
    dev = np.abs(gn - gold["gnorm"]) / np.maximum(gold["gnorm"], 1e-30)
    q = lambda f: float(np.sort(dev)[min(len(dev) - 1, int(f * len(dev)))])
    report("%s vs the reference's batch-16 fixture: losses %.2e / %.2e, argmax sample agreement %.5f, running statistics %s, "
           "head gradients %s, gradient norms (340 tensors) median %.1e q90 %.1e max %.1e (%s)"
           % (name, e_ml, e_al, agree, {k: "%.1e" % v for k, v in e_buf.items()}, {k: "%.1e" % v for k, v in e_grad.items()},
              q(.5), q(.9), dev.max(), names[int(dev.argmax())]))
    assert e_ml < 1e-5 and e_al < 1e-5
    assert agree >= 0.999
    assert all(v < 1e-4 for v in e_buf.values()), e_buf
    for k in ("cls.4.weight", "cls.4.bias", "aux.4.weight", "aux.4.bias"):
        assert e_grad[k] < 5e-4, (k, e_grad[k])
    assert q(.5) <= 2e-3 and q(.9) <= 1e-2 and dev.max() <= 1e-1


@pytest.mark.skipif(os.environ.get("SEMSEG_SKIP_BIG_INSITU") == "1", reason="big cases disabled")
@pytest.mark.parametrize("arith", ["bf16x3", "f32"])
@pytest.mark.parametrize("path", ["trainer", "module"])
def test_headline_batch16_train_step(path, arith, report):
    from semseg_amd import engine as E
    from semseg_amd.trainer import Trainer
    gold = np.load(os.path.join(GOLD, "pspnet101_c150_s473_b16.npz"))
    old = E.set_arith(arith)
    try:
        m, _ = build("psp", 101, 150)
        x, y = inputs(16, 473, 150)
        m = m.cuda().train()
        xd, yd = x.cuda(), y.cuda()
        if path == "trainer":
            tr = Trainer(m, base_lr=0.01, momentum=0.9, weight_decay=1e-4, aux_weight=0.4, sync_bn=False)
            pred, ml, al = tr.step(xd, yd, 0.01)
            eng = tr.engine(xd)
            assert eng.arith == E._ARITH_NAMES[arith]
            grads = {k: eng.grad_views[p] for k, p in m.named_parameters()}
            tr.check_labels()
        else:
            pred, ml, al = m(xd, yd)
            (ml + 0.4 * al).backward()
            grads = {k: p.grad for k, p in m.named_parameters()}
        torch.cuda.synchronize()
        bufs = {k: v for k, v in m.state_dict().items() if "running" in k}
        _check_train(report, "PSPNet-101 473^2 batch 16 [%s, %s]" % (path, arith), gold, pred, float(ml.item()),
                     float(al.item()), grads, bufs)
    finally:
        E.set_arith(old)
        torch.cuda.empty_cache()


@pytest.mark.skipif(os.environ.get("SEMSEG_SKIP_BIG_INSITU") == "1", reason="big cases disabled")
def test_config5_multi_scale_512_six_scales(report):
    """BASELINE configs[4] at its real size: 512x512 image, scales 0.5 ... 1.75, base_size 512, crop 473 -> 46 forwards."""
    from model.pspnet import PSPNet
    from oracle import segnet
    from semseg_amd.infer import MultiScaleTester
    gold = np.load(os.path.join(GOLD, "pspnet101_c150_ms512.npz"))
    classes, crop, base = 150, 473, 512
    scales = (0.5, 0.75, 1.0, 1.25, 1.5, 1.75)
    m = PSPNet(layers=101, classes=classes, zoom_factor=8, pretrained=False)
    sd = segnet.recipe_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed=5)
    sd["cls.4.weight"] *= 1e-3          # as in the fixture (and tests/test_infer_gpu.py): well-conditioned probabilities
    sd["cls.4.bias"] *= 1e-3
    m.load_state_dict(sd)
    img = (np.random.default_rng(1).random((512, 512, 3)) * 255).astype(np.float32)
    mean = [0.485 * 255, 0.456 * 255, 0.406 * 255]
    std = [0.229 * 255, 0.224 * 255, 0.225 * 255]
    t = MultiScaleTester(m.cuda(), classes, base, crop, crop, scales, mean, std)
    assert t.num_forwards(512, 512) == int(gold["forwards"]) == 46
    pred, prob = t.predict(img, return_prob=True)
    prob = prob.permute(1, 2, 0).cpu().numpy()
    e = float(np.abs(prob[::4, ::4, :] - gold["prob_sample"]).max())
    e_max = float(np.abs(prob.max(axis=2) - gold["prob_max"]).max())
    agree = float((pred.cpu().numpy().astype(np.uint8) == gold["argmax"]).mean())
    report("config 5 (512x512, six scales, 46 forwards of PSPNet-101 473^2) vs the reference-network fixture: prob sample "
           "max-abs err %.2e, max-prob err %.2e, argmax agreement %.5f" % (e, e_max, agree))
    assert e < 2e-4 and e_max < 2e-4 and agree > 0.998
