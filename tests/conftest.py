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
