import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    """The product library.  GPU tests fail loudly (never skip) if it cannot run."""
    # torch bundles its own copy of the HIP runtime: when a test also needs torch (device
    # tensors for the full-size cases) it has to be initialised before libbfhip.so pulls in
    # /opt/rocm's copy, the order bench.py uses
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass
    import brutefir_amd as bf
    bf.lib()
    assert bf.device_count() >= 1, "no HIP device visible"
    return bf
