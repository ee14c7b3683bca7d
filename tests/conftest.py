import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

# torch first: a process must hold ONE HIP runtime, and torch brings its own copy; libazhip.so
# (loaded later through ctypes) then binds to the runtime that is already resident
try:
    import torch  # noqa: F401,E402
except Exception:  # noqa: BLE001
    pass
