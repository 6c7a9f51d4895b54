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
                               0, None, 1, 64, None, 0, None) == -1
    assert lib.semseg_sgd_step(None, None, None, 4, 0.1, None, 0.9, 0.0, 1.0, 1, None) == -1


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
