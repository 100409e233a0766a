import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def ctx():
    """A libmsfm context on cuda:0.  Fails (does not skip) when the HIP library or the GPU is
    missing: -m gpu tests must never pass on a fallback."""
    from metricsfm_amd import capi
    c = capi.Context(0)
    yield c
    c.close()
