import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


@pytest.fixture
def fp32_oracle():
    """Oracle in idealised fp32 mode (the mode the golden fixtures were generated in)."""
    from oracle import tcnn_ref

    prev = tcnn_ref.get_precision()
    tcnn_ref.set_precision("fp32")
    yield tcnn_ref
    tcnn_ref.set_precision(prev)


@pytest.fixture
def tcnn_oracle():
    """Oracle with tiny-cuda-nn's fp16 rounding points (what the HIP path is compared with)."""
    from oracle import tcnn_ref

    prev = tcnn_ref.get_precision()
    tcnn_ref.set_precision("tcnn")
    yield tcnn_ref
    tcnn_ref.set_precision(prev)
