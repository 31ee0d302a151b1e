import gzip
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with gzip.open(os.path.join(HERE, "golden", name + ".json.gz"), "rb") as f:
        return json.loads(f.read().decode())


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session", autouse=True)
def _torch_device_first():
    """Two HIP runtimes live in a GPU test process: /opt/rocm's (libhite_gpu.so links it) and the one PyTorch bundles
    (test_gpu_scale builds its synthetic genome with torch on the device).  bench.py always brings torch's up first; a test
    session does the same, whatever the order of the test files, instead of leaving it to the first test that needs it."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda:0")
    except ImportError:
        pass
    yield
