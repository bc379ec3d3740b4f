import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _probe_build() -> bool:
    from l4p_amd import _lib

    return int(_lib.load().l4p_get_knob(b"probe_kernels")) == 1


@pytest.fixture
def probe_kernels():
    """Skip unless the library is a PROBES=1 build (make -C l4p_amd/csrc VARIANT=probes PROBES=1; L4P_HIP_LIB=.../libl4p_hip_probes.so):
    the measured-and-not-adopted kernels (gemm4w.hpp, the up-sampling loader of conv3_halo.hpp) are not in the shipped library."""
    if not _probe_build():
        pytest.skip("kernel not in the shipped library (PROBES=1 build only)")


@pytest.fixture
def knob():
    """Set a dispatch knob of the native launchers for one test (l4p_set_knob) and restore it afterwards."""
    from l4p_amd import _lib

    saved = {}

    def set_(name: str, value: int) -> None:
        saved.setdefault(name, int(_lib.load().l4p_get_knob(name.encode())))
        _lib.set_knob(name, value)

    yield set_
    for name, value in saved.items():
        _lib.set_knob(name, value)
