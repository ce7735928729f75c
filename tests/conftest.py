import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/libroaring_oracle.so), built on demand with gcc."""
    from oracle import pyoracle

    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """One fbk context on cuda:0.  Fails loudly (no skip, no fallback) if the HIP library
    is missing or no gfx950 device is visible."""
    from featurebase_amd.roaring import Context

    ctx = Context(0)
    yield ctx
    ctx.close()
