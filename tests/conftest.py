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
    """The product library bound to cuda:0.  No fallback: missing library or device is an error."""
    from lastz_amd import lzgpu
    lib = lzgpu.Lib()
    if lib.probe() != 0 and not os.environ.get("LZGPU_REQUIRE_GPU"):
        pytest.skip("no gfx950 device here (the GPU tier runs on the MI355X box; LZGPU_REQUIRE_GPU=1 makes this an error)")
    lib.init(0)
    yield lib
    lib.shutdown()
