import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def report():
    """Append-only text report under gpurun_out/ so a single gpurun call returns every error figure."""
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    f = open(os.path.join(out, "parity_report.txt"), "a")

    def log(msg):
        f.write(msg + "\n")
        f.flush()
        print(msg)
    yield log
    f.close()


@pytest.fixture(params=["bf16x3", "f32"])
def arith(request):
    """Both arithmetics of the conv GEMMs (include/semseg_hip.h): the engine default (SEMSEG_ARITH_BF16X3) and the forced exact
    path (SEMSEG_ARITH_F32).  Tests that take this fixture run twice; engines built inside the test pick the value up."""
    from semseg_amd import engine
    old = engine.set_arith(request.param)
    yield request.param
    engine.set_arith(old)


@pytest.fixture(autouse=True)
def _release_device_memory():
    """Engines hold tens of GB at the headline sizes and sit in reference cycles (tape closures <-> engine): without a
    collection between tests eight batch-16 cases in a row reached 281 GB and the next test ran out of memory (round 5)."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    except Exception:
        pass
