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


OPTION_NAMES = ("rec_persistent", "rec_xw", "rec_pingpong", "rec_groups", "rec_spin_us", "rec_stream", "rec_fused2", "rec_rr", "rec_xf", "rec_fk", "rec_hf", "dense_frag3", "dense_f16x2", "train_bptt", "train_outer_plain",
                "spec_ppw", "spec_variant", "bn_fast", "gemm_tm_batch", "gemm_split_bf16", "gemm_wide", "conv_store", "conv_frag3_out", "conv_flatk", "conv_a4", "weights_check")


@pytest.fixture(autouse=True)
def _restore_library_options():
    """Tests flip tuning knobs with capi.set_option(); every test starts from the defaults again."""
    yield
    from nntoolkitcore_amd import capi
    if capi._lib is not None:
        for name in OPTION_NAMES:
            capi._lib.nntk_hip_set_option(name.encode(), b"auto")
