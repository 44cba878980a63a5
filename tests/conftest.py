import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


# PyTorch brings its own HIP runtime; whichever libamdhip64 is mapped first serves the whole process.  If the
# product library (linked against /opt/rocm) initialises HIP before torch does, torch then finds "No HIP GPUs".
# Tests and bench use torch for tensors and torch.distributed, so it is loaded first here (bench.py imports it
# first as well); a C caller of the library never meets torch.
try:
    import torch  # noqa: F401
except ImportError:
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib


@pytest.fixture(scope="session")
def golden_vectors():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "message_vectors.json")) as f:
        return json.load(f)
