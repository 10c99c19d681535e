import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hostemu():
    """Test-only host-emulation build of the kernel sources (g++, -DFV3_HOST_EMU)."""
    from pace_amd import build

    build.build(64, hostemu=True, verbose=False)
    return "hostemu"


@pytest.fixture(scope="session")
def gpu_backend():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pace_amd import build, lib

    if not os.path.exists(build.lib_path(64)):
        build.build(64)
    lib.load(64)  # fails loudly if the HIP library is missing
    return "hip:gfx950"


@pytest.fixture(params=["hostemu", pytest.param("hip:gfx950", marks=pytest.mark.gpu)])
def backend(request):
    """Both builds of the kernel sources: the host emulation (CPU suite) and the HIP library (-m gpu)."""
    if request.param == "hostemu":
        request.getfixturevalue("hostemu")
    else:
        request.getfixturevalue("gpu_backend")
    return request.param


@pytest.fixture(autouse=True)
def _release_device_memory(request):
    """After every GPU test: collect what the test dropped and hand PyTorch's cached blocks back, so that the full-size cases
    (C768 L79 fp64, C768 L127 fp32: tens of GB each) start from an empty device whatever ran before them in the same process."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc

        import torch

        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
