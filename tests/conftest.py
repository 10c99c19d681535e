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
