import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path: no skip-on-missing-GPU, fail loudly instead."""
    import torch
    assert torch.cuda.is_available(), "a -m gpu test was started without a visible GPU"
    import spblas_reference_amd as sp
    sp._capi.lib()  # raises if the HIP library is not built
    return torch.device("cuda:0")
