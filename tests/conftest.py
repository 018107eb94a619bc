import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from safetensors import safe_open

    def load(name):
        path = os.path.join(GOLDEN, name + ".safetensors")
        with safe_open(path, framework="pt") as f:
            tensors = {k: f.get_tensor(k) for k in f.keys()}
            meta = f.metadata() or {}
        return tensors, meta

    return load
