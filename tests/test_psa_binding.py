"""The reference-side native binding of lib/psa (SURVEY.md section 8b "Native operator ABI").

`lib/psa/src/__init__.py` builds the pybind module `psamask_gpu` with torch.utils.cpp_extension; it exports
psamask_forward / psamask_backward with the reference's signature (lib/psa/src/gpu/operator.h:3-4) on top of the C ABI
of libsemseg_hip.so.  CPU part: the module builds, loads and has that signature, and the reference's UNMODIFIED
lib/psa/functions/psamask.py binds to it (the call reaches the front-end's own device check instead of dying on a
pybind signature mismatch).  GPU part: golden vectors bit-exactly through that module, on a non-default stream too, and
through the reference's unmodified Python front-end on top of it.

The reference's front-end (functional.py, functions/__init__.py, functions/psamask.py) travels to the GPU box as the
byte-exact, sha256-checked fixture tests/golden/ref_psa_frontend.npz (tests/golden/make_golden_refpsa.py); where
/root/reference exists the fixture is also compared with the files there.
"""
import hashlib
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = "/root/reference/lib/psa"
GOLD = os.path.join(ROOT, "tests", "golden", "psamask_ref.npz")
FRONTEND = os.path.join(ROOT, "tests", "golden", "ref_psa_frontend.npz")
FRONTEND_SHA256 = {   # printed by tests/golden/make_golden_refpsa.py
    "functional": ("functional.py", "0ab753f2741e0eedcad9bca3449fd4475a46d91f3941eaa2736ff6660efa8a86"),
    "functions_init": ("functions/__init__.py", "a04f0f6335f78b8181c1754d0ef8c59f3b750d2f378923d0dc68a5fafe5ba2a4"),
    "functions_psamask": ("functions/psamask.py", "94ef86dc71875eda0cf3ac70fc6d2851052c3e8765e0feb96ac08d46a8445717"),
}


def _module():
    import lib.psa.src as src
    return src


def test_psamask_gpu_module_builds_and_has_the_reference_signature():
    src = _module()
    assert src.gpu.__name__ == "psamask_gpu" and src.gpu.__file__.endswith("psamask_gpu.so")
    for fn in ("psamask_forward", "psamask_backward"):
        doc = getattr(src.gpu, fn).__doc__
        sig = doc.split("->")[0]
        assert sig.count("torch.Tensor") == 2 and sig.count("Int") == 8 and "-> None" in doc, doc
    with pytest.raises(RuntimeError, match="no CPU"):
        src.cpu
    # CPU tensors convert fine (signature ok) and are refused by the front-end itself: no CPU path
    with pytest.raises(RuntimeError, match="MI355X"):
        src.gpu.psamask_forward(0, torch.zeros(1, 9, 3, 3), torch.zeros(1, 9, 3, 3), 1, 3, 3, 3, 3, 1, 1)
    with pytest.raises(TypeError):
        src.gpu.psamask_forward(0, torch.zeros(1, 9, 3, 3), 1, 3, 3, 3, 3, 1, 1)   # wrong arity / types


def _reference_package(tmp_path):
    """A package `refpsa` = the reference's own lib/psa Python files (unpacked from the sha256-checked fixture at test
    time) with its `src` sub-package replaced by this repo's lib/psa/src."""
    fx = np.load(FRONTEND)
    pkg = tmp_path / "refpsa"
    (pkg / "functions").mkdir(parents=True)
    (pkg / "__init__.py").write_text("")
    for key, (rel, digest) in FRONTEND_SHA256.items():
        raw = fx[key].tobytes()
        assert hashlib.sha256(raw).hexdigest() == digest, "fixture %s does not have the recorded digest" % key
        if os.path.isdir(REF_DIR):   # build container: the fixture IS the reference's file
            assert raw == open(os.path.join(REF_DIR, rel), "rb").read(), rel
        (pkg / rel).write_bytes(raw)
    os.symlink(os.path.join(ROOT, "lib", "psa", "src"), pkg / "src")
    sys.path.insert(0, str(tmp_path))
    for k in [k for k in sys.modules if k == "refpsa" or k.startswith("refpsa.")]:
        del sys.modules[k]
    try:
        return importlib.import_module("refpsa.functional")
    finally:
        sys.path.remove(str(tmp_path))


def test_unmodified_reference_function_binds_to_psamask_gpu(tmp_path, monkeypatch):
    PF = _reference_package(tmp_path)
    # no GPU here: make the reference take its `is_cuda` branch with host tensors
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    with pytest.raises(RuntimeError, match="MI355X"):     # reached psamask_gpu.psamask_forward's own check
        PF.psa_mask(torch.randn(1, 9, 3, 3), 0, 3, 3)


@pytest.mark.gpu
def test_psamask_gpu_module_golden_vectors(report):
    """Golden vectors (outputs of the reference's compiled CPU op) bit-exactly through the pybind module, called the
    way lib/psa/functions/psamask.py:17-22,31-35 calls it; with the reference's unmodified Python files when they
    are present."""
    src = _module()
    fx = np.load(GOLD)
    keys = sorted({k.rsplit("_", 1)[0] for k in fx.files})
    side = torch.cuda.Stream()
    for i, key in enumerate(keys):
        parts = key.split("_")
        mH, mW = (int(v) for v in parts[2][1:].split("x"))
        t = int(parts[3][1:])
        x = torch.from_numpy(fx[key + "_x"]).cuda()
        gy = torch.from_numpy(fx[key + "_gy"]).cuda()
        n, c, fh, fw = x.shape
        with torch.cuda.stream(side if i % 2 else torch.cuda.current_stream()):
            out = torch.zeros(n, fh * fw, fh, fw, device="cuda")
            src.gpu.psamask_forward(t, x, out, n, fh, fw, mH, mW, (mH - 1) // 2, (mW - 1) // 2)
            gin = torch.zeros(n, c, fh, fw, device="cuda")
            src.gpu.psamask_backward(t, gy, gin, n, fh, fw, mH, mW, (mH - 1) // 2, (mW - 1) // 2)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), fx[key + "_out"]), key
        assert np.array_equal(gin.cpu().numpy(), fx[key + "_gin"]), key
    with pytest.raises(RuntimeError):
        src.gpu.psamask_forward(0, torch.zeros(1, 9, 3, 3, device="cuda"), torch.zeros(1, 8, 3, 3, device="cuda"),
                                1, 3, 3, 3, 3, 1, 1)            # wrong destination shape
    with pytest.raises(RuntimeError):
        src.gpu.psamask_forward(0, torch.zeros(1, 9, 3, 3, device="cuda").double(),
                                torch.zeros(1, 9, 3, 3, device="cuda"), 1, 3, 3, 3, 3, 1, 1)
    report("psamask_gpu pybind module == reference golden vectors (%d cases, two streams)" % len(keys))


@pytest.mark.gpu
def test_unmodified_reference_function_on_gpu(tmp_path, report):
    PF = _reference_package(tmp_path)
    fx = np.load(GOLD)
    for key in sorted({k.rsplit("_", 1)[0] for k in fx.files}):
        parts = key.split("_")
        mH, mW = (int(v) for v in parts[2][1:].split("x"))
        x = torch.from_numpy(fx[key + "_x"]).cuda().requires_grad_(True)
        out = PF.psa_mask(x, int(parts[3][1:]), mH, mW)
        out.backward(torch.from_numpy(fx[key + "_gy"]).cuda())
        assert np.array_equal(out.detach().cpu().numpy(), fx[key + "_out"])
        assert np.array_equal(x.grad.cpu().numpy(), fx[key + "_gin"])
    report("reference's unmodified lib/psa/functions/psamask.py on psamask_gpu: golden vectors bit-exact")
