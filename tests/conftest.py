import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu")


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on a real device.  Fails (never skips) when the HIP
    extension is missing or no gfx950 device answers: GPU tests must not pass on a fallback."""
    from zopfli_amd import api
    lib = api.library()
    assert lib.zmx_device_count() > 0, "no HIP device visible"
    return lib


@pytest.fixture(scope="session")
def gpu_ctx(gpu_lib):
    from zopfli_amd import Context
    ctx = Context(0, gpu_lib)
    yield ctx
    ctx.close()
