"""Generates tests/golden/*.npz by running the REAL reference (/root/reference, imported here on CPU)
and checks the oracle (oracle/segnet.py, oracle/psamask.py) against it bit-for-bit on the way.
Run in the build container only:   python tests/golden/make_golden.py
The fixtures travel to the GPU box; /root/reference does not.
"""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def import_reference():
    """The reference JIT-builds lib/psa into its own (read-only) directory: work from a temp copy."""
    tmp = tempfile.mkdtemp(prefix="semseg_ref_")
    for d in ("model", "lib", "util"):
        shutil.copytree(os.path.join(REF, d), os.path.join(tmp, d))
    sys.path.insert(0, tmp)
    import model.pspnet as rp  # noqa
    import model.psanet as rpa  # noqa
    sys.path.remove(tmp)
    mods = {k: sys.modules.pop(k) for k in list(sys.modules) if k.split(".")[0] in ("model", "lib", "util")}
    return rp, rpa, mods


def main():
    rp, rpa, _ = import_reference()
    sys.path.insert(0, ROOT)
    from oracle import segnet, psamask as opm
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---------- psamask fixtures from the reference's own compiled CPU op ----------
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from build_ref import build, load_ref
    build()
    ref_op = load_ref()
    rng = np.random.default_rng(0)
    cases = {}
    for (H, W, mH, mW) in [(5, 5, 9, 9), (5, 7, 9, 13), (6, 6, 5, 5), (4, 4, 3, 3), (7, 6, 13, 11)]:
        x = rng.standard_normal((2, mH * mW, H, W)).astype(np.float32)
        gy = rng.standard_normal((2, H * W, H, W)).astype(np.float32)
        for t in (0, 1):
            out = torch.zeros(2, H * W, H, W)
            ref_op.psamask_forward(t, torch.from_numpy(x), out, 2, H, W, mH, mW, (mH - 1) // 2, (mW - 1) // 2)
            gin = torch.zeros(2, mH * mW, H, W)
            ref_op.psamask_backward(t, torch.from_numpy(gy), gin, 2, H, W, mH, mW, (mH - 1) // 2, (mW - 1) // 2)
            assert np.array_equal(out.numpy(), opm.psa_mask_forward(x, t, mH, mW)), "oracle != reference"
            assert np.array_equal(gin.numpy(), opm.psa_mask_backward(gy, t, mH, mW)), "oracle != reference"
            key = "H%d_W%d_m%dx%d_t%d" % (H, W, mH, mW, t)
            cases[key + "_x"] = x
            cases[key + "_gy"] = gy
            cases[key + "_out"] = out.numpy()
            cases[key + "_gin"] = gin.numpy()
    np.savez_compressed(os.path.join(HERE, "psamask_ref.npz"), **cases)
    print("psamask: oracle == compiled reference on %d cases" % (len(cases) // 4))

    # ---------- network fixtures: reference modules, recipe weights ----------
    def run(name, ctor, arch, layers, classes, size, batch, psa_cfg=None, zoom=8):
        m = ctor()
        shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        sd = segnet.recipe_state_dict(shapes, seed=1234)
        m.load_state_dict(sd)
        g = torch.Generator().manual_seed(7)
        x = torch.randn(batch, 3, size, size, generator=g)
        hh = int((size - 1) / 8 * zoom + 1)
        y = torch.randint(0, classes, (batch, hh, hh), generator=g)
        y[torch.rand(batch, hh, hh, generator=g) < 0.05] = 255
        # eval logits
        m.eval()
        with torch.no_grad():
            ref_logits = m(x)
            sd_e = {k: v.clone() for k, v in sd.items()}
            orc_logits = segnet.forward(sd_e, x, layers, arch, zoom_factor=zoom, training=False, psa_cfg=psa_cfg)
        assert torch.equal(ref_logits, orc_logits), "%s: eval oracle != reference" % name
        # train step (dropout p=0 so no RNG enters)
        m.train()
        pred, ml, al = m(x, y)
        (ml + 0.4 * al).backward()
        sd_t = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
                for k, v in sd.items()}
        p2, ml2, al2 = segnet.forward(sd_t, x, layers, arch, zoom_factor=zoom, training=True, y=y, psa_cfg=psa_cfg)
        (ml2 + 0.4 * al2).backward()
        assert torch.equal(pred, p2) and torch.equal(ml, ml2) and torch.equal(al, al2), "%s train fwd" % name
        grads = {k: p.grad for k, p in m.named_parameters()}
        for k, gref in grads.items():
            assert torch.allclose(gref, sd_t[k].grad, rtol=0, atol=0) or \
                (gref - sd_t[k].grad).abs().max() <= 1e-6 * gref.abs().max(), "%s grad %s" % (name, k)
        new_sd = m.state_dict()
        for k in new_sd:
            if "running" in k:
                assert torch.equal(new_sd[k], sd_t[k]), k
        # keep a compact fixture: strided logits sample + checksums, losses, gradient norms + samples
        fx = {
            "logits_sample": ref_logits[:, ::7, ::5, ::5].numpy(),
            "logits_absmax": np.float64(ref_logits.abs().max().item()),
            "logits_sum": np.float64(ref_logits.double().sum().item()),
            "main_loss": np.float64(ml.item()), "aux_loss": np.float64(al.item()),
            "pred_sample": pred[:, ::5, ::5].numpy(),
            "pred_full": pred.numpy().astype(np.uint8),
        }
        for k, gref in grads.items():
            fx["gnorm/" + k] = np.float64(gref.double().norm().item())
        for k in ["layer0.0.weight", "cls.4.bias", "cls.4.weight", "aux.4.bias", "layer0.1.weight", "layer0.1.bias",
                  "layer3.0.bn2.weight", "layer4.2.bn3.bias"]:
            fx["grad/" + k] = grads[k].numpy()
        for k in ["layer0.1.running_mean", "layer0.1.running_var", "layer4.2.bn3.running_var", "cls.1.running_mean"]:
            fx["buf/" + k] = new_sd[k].numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **fx)
        print("%s: oracle == reference (eval logits, train losses, argmax, grads, running stats); "
              "|logits|max=%.3f main=%.4f aux=%.4f" % (name, fx["logits_absmax"], ml.item(), al.item()))

    run("pspnet50_c21_s73_b2", lambda: rp.PSPNet(layers=50, classes=21, zoom_factor=8, dropout=0.0, pretrained=False),
        "psp", 50, 21, 73, 2)
    cfg = dict(psa_type=2, compact=False, shrink_factor=2, mask_h=9, mask_w=9, normalization_factor=1.0,
               psa_softmax=True)
    run("psanet50_c19_s65_b2", lambda: rpa.PSANet(layers=50, classes=19, zoom_factor=8, dropout=0.0, pretrained=False,
                                                  **cfg), "psa", 50, 19, 65, 2, psa_cfg=cfg)


if __name__ == "__main__":
    main()
