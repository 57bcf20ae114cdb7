import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Some GPU tests use torch for device memory next to the engine's own library. PyTorch ships its own HIP / HSA
    # runtime; when libiteres_amd.so (linked against /opt/rocm) brings up the system runtime FIRST, torch's later
    # initialisation in the same process finds no device. The other order works (the engine then runs on the runtime
    # that is already loaded — bench.py has always done it that way), so on a GPU box torch goes first, once.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
