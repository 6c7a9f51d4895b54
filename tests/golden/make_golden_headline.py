"""Generates the fixtures of the two BASELINE.json configurations that carry the headline numbers, by running the REAL
reference (/root/reference, imported here on CPU):

  pspnet101_c150_s473_b16.npz   BASELINE metric configuration: PSPNet-101, 473x473, 150 classes, batch 16 train step
                                (model/pspnet.py:80-105, dropout 0, recipe weights, 5 % ignore pixels): losses, argmax
                                sample, four running-statistics buffers, cls.4 / aux.4 gradients, every gradient norm
  pspnet101_c150_ms512.npz      BASELINE configs[4]: the multi-scale test path of tool/test.py:149-204 on one synthetic
                                512x512 image, base_size 512, crop 473, the six ADE scales (23 crops = 46 forwards): the
                                network is the imported reference PSPNet-101 (eval), the loop around it is
                                oracle/test_pipeline.py (cv2 is not installable here: its resize / copyMakeBorder are
                                restated from their formula there, that part stays "parity unpinned")

Run in the build container only (needs ~50 GB of RAM for the batch-16 step):
    python tests/golden/make_golden_headline.py [train] [ms] [psa] [psp50]

  pspnet50_c150_s473_b16.npz    BASELINE configs[1] at its stated batch (round 6): PSPNet-50, 473x473, 150 classes, BATCH 16
                                train step (model/pspnet.py:30-105): same contents plus sampled gradients of the last
                                dilated conv of layer3 and layer4

  psanet101_c150_s465_b16.npz   BASELINE configs[3] at its stated size: PSANet-101, 465x465, 150 classes, psa_type 2,
                                shrink 2, full 59x59 mask, BATCH 16 train step (model/psanet.py:154-179): same contents
                                plus the gradients of the PSA module's last layers
The fixtures travel to the GPU box; /root/reference does not.
"""
import os
import resource
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402


def headline_inputs(batch, size, classes, zoom=8):
    """Same generator as tests/test_model_gpu.py::inputs (seed 7, 5 % ignore pixels)."""
    g = torch.Generator().manual_seed(7)
    x = torch.randn(batch, 3, size, size, generator=g)
    hh = int((size - 1) / 8 * zoom + 1)
    y = torch.randint(0, classes, (batch, hh, hh), generator=g)
    y[torch.rand(batch, hh, hh, generator=g) < 0.05] = 255
    return x, y


PSA_CFG = dict(psa_type=2, compact=False, shrink_factor=2, mask_h=59, mask_w=59, normalization_factor=1.0,
               psa_softmax=True)


def train_fixture(rp, segnet, arch="psp", layers=101):
    classes, batch = 150, 16
    size = 473 if arch == "psp" else 465
    psa_cfg = None if arch == "psp" else PSA_CFG
    out_name = "pspnet%d_c150_s473_b16.npz" % layers if arch == "psp" else "psanet101_c150_s465_b16.npz"
    if arch == "psp":
        m = rp.PSPNet(layers=layers, classes=classes, zoom_factor=8, dropout=0.0, pretrained=False)
    else:
        m = rp.PSANet(layers=layers, classes=classes, zoom_factor=8, dropout=0.0, pretrained=False, **PSA_CFG)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = segnet.recipe_state_dict(shapes, seed=1234)
    m.load_state_dict(sd)
    x, y = headline_inputs(batch, size, classes)
    m.train()
    t0 = time.time()
    grab = {}
    hook = m.cls.register_forward_hook(lambda mod, inp, out: grab.__setitem__("scores", out.detach()))
    pred, ml, al = m(x, y)
    hook.remove()
    (ml + 0.4 * al).backward()
    print("reference %s-%d %d^2 batch 16 train step on the CPU: %.1f s, main %.6f aux %.6f"
          % ("PSPNet" if arch == "psp" else "PSANet", layers, size, time.time() - t0, ml.item(), al.item()), flush=True)
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    new_sd = {k: v.clone() for k, v in m.state_dict().items()}
    # top-2 margin of the reference's own upsampled train-mode scores at the sampled pixels, relative to max |score|: an
    # argmax may legitimately differ between two fp32 implementations only where this margin is inside the logits tolerance
    import torch.nn.functional as F
    up = F.interpolate(grab["scores"], size=(size, size), mode="bilinear", align_corners=True)
    assert torch.equal(up.max(1)[1], pred)
    top2 = up[:, :, ::5, ::5].topk(2, dim=1)[0]
    margin = ((top2[:, 0] - top2[:, 1]) / up.abs().max()).numpy().astype(np.float32)
    del up, top2
    fx = {
        "main_loss": np.float64(ml.item()), "aux_loss": np.float64(al.item()),
        "pred_sample": pred[:, ::5, ::5].numpy().astype(np.uint8),
        "margin_sample": margin,
        "pred_hist": np.bincount(pred.reshape(-1).numpy(), minlength=classes).astype(np.int64),
    }
    names = list(grads)
    fx["gnorm_names"] = np.array(names)
    fx["gnorm"] = np.array([grads[k].double().norm().item() for k in names])
    head = ["cls.4.weight", "cls.4.bias", "aux.4.weight", "aux.4.bias", "layer0.1.weight", "layer0.1.bias"]
    bufs = ["layer0.1.running_mean", "layer0.1.running_var", "layer4.2.bn3.running_var", "cls.1.running_mean"]
    if layers == 50:   # configs[1]: also the stored gradient of one dilated conv of each of layer3 / layer4 (sampled)
        head += ["layer3.5.conv2.weight", "layer4.2.conv2.weight"]
    if arch == "psa":   # the PSA module's own last layers (model/psanet.py:29-51) and one of its BatchNorms
        head += ["psa.proj.0.weight", "psa.attention.3.weight", "psa.attention_p.3.weight"]
        bufs += ["psa.proj.1.running_var"]
    for k in head:
        if grads[k].numel() > (1 << 19):   # the big weights: a [::8, ::8] sample and the full maximum
            fx["gradsub/" + k] = grads[k].numpy()[::8, ::8].copy()
            fx["gradmax/" + k] = np.float64(grads[k].abs().max().item())
        else:
            fx["grad/" + k] = grads[k].numpy()
    for k in bufs:
        fx["buf/" + k] = new_sd[k].numpy()
    del m, pred, grads
    # the oracle on the same step: pins oracle/segnet.py to the reference at the headline configuration too
    sd_t = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
            for k, v in sd.items()}
    p2, ml2, al2 = segnet.forward(sd_t, x, layers, arch, zoom_factor=8, training=True, y=y, psa_cfg=psa_cfg)
    assert ml2.item() == fx["main_loss"] and al2.item() == fx["aux_loss"], "oracle != reference at batch 16"
    assert np.array_equal(p2[:, ::5, ::5].numpy().astype(np.uint8), fx["pred_sample"])
    for k in ["layer0.1.running_mean", "layer4.2.bn3.running_var", "cls.1.running_mean"]:
        assert torch.equal(sd_t[k], new_sd[k]), k
    np.savez_compressed(os.path.join(HERE, out_name), **fx)
    print(out_name + " written; oracle == reference (losses, argmax sample, running statistics)", flush=True)


def ms_fixture(rp, segnet):
    sys.path.insert(0, ROOT)
    from oracle import test_pipeline as tp
    layers, classes, crop, base = 101, 150, 473, 512
    scales = (0.5, 0.75, 1.0, 1.25, 1.5, 1.75)
    m = rp.PSPNet(layers=layers, classes=classes, zoom_factor=8, pretrained=False)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = segnet.recipe_state_dict(shapes, seed=5)
    m.load_state_dict(sd)
    m.eval()
    g = np.random.default_rng(1)
    img = (g.random((512, 512, 3)) * 255).astype(np.float32)
    mean = [0.485 * 255, 0.456 * 255, 0.406 * 255]
    std = [0.229 * 255, 0.224 * 255, 0.225 * 255]
    # The recipe's eval-mode logits of this 101-layer net are huge (random running statistics), and softmax turns the fp32
    # round-off of such logits (~1e-6 relative) into O(0.01) probability changes at near-tie pixels (first version of this
    # fixture, classifier scaled by 1e-3 as tests/test_infer_gpu.py does for PSPNet-50: the HIP path differed by 1.6e-2 in
    # probability at 0.998 argmax agreement).  The classifier is therefore scaled so that max |logit| = 10 on the centre
    # crop: probabilities are then well conditioned and the 2e-4 bound means what it says.  The factor travels in the file.
    with torch.no_grad():
        probe = torch.from_numpy(((img[19:492, 19:492] - np.array(mean, np.float32)) / np.array(std, np.float32))
                                 .transpose(2, 0, 1))[None]
        amax = float(m(probe).abs().max())
    cls_scale = np.float32(10.0 / amax)
    print("max |logit| on the centre crop %.4g -> classifier scale %.6g" % (amax, cls_scale), flush=True)
    with torch.no_grad():
        m.cls[4].weight.mul_(float(cls_scale))
        m.cls[4].bias.mul_(float(cls_scale))
    calls = [0]

    def net(x):
        calls[0] += x.shape[0]
        return m(x)
    t0 = time.time()
    arg, prob = tp.multi_scale_predict(net, img, classes, base, crop, crop, scales, mean, std)
    print("reference PSPNet-101 multi-scale test of one 512x512 image: %d forwards, %.1f s" % (calls[0], time.time() - t0),
          flush=True)
    assert calls[0] == 46
    np.savez_compressed(os.path.join(HERE, "pspnet101_c150_ms512.npz"),
                        argmax=arg.astype(np.uint8), prob_sample=prob[::16, ::16, :].astype(np.float32),
                        prob_max=prob.max(axis=2)[::2, ::2].astype(np.float32), forwards=np.int64(calls[0]),
                        cls_scale=cls_scale)
    print("pspnet101_c150_ms512.npz written", flush=True)


def main():
    what = sys.argv[1:] or ["train", "ms"]
    resource.setrlimit(resource.RLIMIT_AS, (58 << 30, 58 << 30))   # fail with MemoryError instead of waking the OOM killer
    rp, rpa, _ = import_reference()
    sys.path.insert(0, ROOT)
    from oracle import segnet
    torch.manual_seed(0)
    torch.set_num_threads(8)
    if "ms" in what:
        ms_fixture(rp, segnet)
    if "train" in what:
        train_fixture(rp, segnet)
    if "psa" in what:
        train_fixture(rpa, segnet, arch="psa")
    if "psp50" in what:
        train_fixture(rp, segnet, layers=50)


if __name__ == "__main__":
    main()
