"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/semseg_hip.h
declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from semseg_amd import build as b
    from semseg_amd._lib import parse_header, LIB_PATH
    b.build()
    protos = parse_header()
    assert len(protos) >= 26
    dll = ctypes.CDLL(LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), "missing export %s" % name


def test_header_is_plain_c():
    """No torch / C++ types in the boundary: the header must compile as C."""
    src = "#include \"semseg_hip.h\"\nint main(void){return 0;}\n"
    p = subprocess.run(["gcc", "-x", "c", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                        "-I", "/opt/rocm/include", "-"], input=src.encode(), capture_output=True)
    assert p.returncode == 0, p.stderr.decode()


def test_invalid_arguments_are_rejected_without_a_gpu():
    from semseg_amd._lib import lib
    assert lib.semseg_psamask_forward(0, None, None, 1, 1, 1, 1, 1, 0, 0, None) == -1
    assert lib.semseg_conv_fwd(None, 0, None, None, 0, 1, 1, 1, 32, 1, 1, 32, 1, 1, 1, 0, 1, None, None, 0, None,
                               0, None, 1, 64, 0, None, 0, None, None) == -1
    assert lib.semseg_sgd_step(None, None, None, 4, 0.1, None, 0.9, 0.0, 1.0, 1, None, None) == -1


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under semseg_amd/, model/, lib/ may reference it."""
    bad = []
    for top in ("semseg_amd", "model", "lib"):
        for dp, _, fs in os.walk(os.path.join(ROOT, top)):
            for f in fs:
                if f.endswith(".py"):
                    s = open(os.path.join(dp, f)).read()
                    if "import oracle" in s or "from oracle" in s:
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_cpu_model_refuses_to_run():
    import pytest
    import torch
    from model.pspnet import PSPNet
    m = PSPNet(layers=50, classes=5, pretrained=False).eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 9, 9))


def test_pretrained_trunk_file_is_loaded(tmp_path, monkeypatch):
    """`pretrained=True` reads ./initmodel/resnet{L}_v2.pth (reference model/resnet.py:196-200: published
    ImageNet file, keys conv1/bn1/.../layerK.B.*/fc.*, loaded with strict=False) and the weights appear under
    the segmentation model's own keys (layer0.N.*, layerK.*) — model/pspnet.py:37-43."""
    import sys
    import torch
    from model.pspnet import PSPNet
    from model import resnet as own
    keys_from = None
    if os.path.isdir("/root/reference/model"):      # key set of the reference's own ResNet when it is present
        sys.path.insert(0, "/root/reference")
        try:
            import importlib.util
            spec = importlib.util.spec_from_file_location("ref_resnet", "/root/reference/model/resnet.py")
            ref = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(ref)
            keys_from = ref.resnet50(pretrained=False).state_dict()
        finally:
            sys.path.remove("/root/reference")
    if keys_from is None:
        keys_from = dict(own.build_trunk(50, False).state_dict())
        keys_from["fc.weight"] = torch.zeros(1000, 2048)
        keys_from["fc.bias"] = torch.zeros(1000)
    g = torch.Generator().manual_seed(5)
    published = {k: (torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone())
                 for k, v in keys_from.items()}
    assert "fc.weight" in published and "conv1.weight" in published and "layer4.2.conv3.weight" in published
    (tmp_path / "initmodel").mkdir()
    torch.save(published, str(tmp_path / "initmodel" / "resnet50_v2.pth"))
    monkeypatch.chdir(tmp_path)
    m = PSPNet(layers=50, classes=5, pretrained=True)
    sd = m.state_dict()
    stem = {"conv1": "layer0.0", "bn1": "layer0.1", "conv2": "layer0.3", "bn2": "layer0.4", "conv3": "layer0.6",
            "bn3": "layer0.7"}
    checked = 0
    for k, v in published.items():
        if k.startswith("fc."):
            continue
        head, _, rest = k.partition(".")
        mk = (stem[head] + "." + rest) if head in stem else k
        assert mk in sd, (k, mk)
        assert torch.equal(sd[mk], v), mk
        checked += 1
    assert checked == len(published) - 2
    # without the file the constructor fails like the reference does
    monkeypatch.chdir(tmp_path / "initmodel")
    import pytest
    with pytest.raises(FileNotFoundError):
        PSPNet(layers=50, classes=5, pretrained=True)
