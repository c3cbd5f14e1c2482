import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """The product library must already be built in-tree (it travels to the GPU box)."""
    from nntoolkitcore_amd import _build, capi
    if not os.path.exists(capi.LIB_PATH):
        _build.build()
    return capi.load()


@pytest.fixture(scope="session")
def gpu(built_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible: the HIP path has no CPU fallback")
    torch.cuda.set_device(0)
    from nntoolkitcore_amd import layers
    layers.use_torch_stream()
    return torch.device("cuda:0")
