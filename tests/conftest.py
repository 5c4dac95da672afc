import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _bounds_audit_verdict():
    """When the suite runs against the bounds-audit build (ODEVIO_LIB=.../libodevio_audit.so),
    no kernel may have computed an address outside its buffers in ANY test of the session: the library counts
    violations at every check() and at every plan destruction."""
    yield
    if "audit" not in os.path.basename(os.environ.get("ODEVIO_LIB", "")):
        return
    import gc
    gc.collect()                       # destroy the plans that are still alive: their status words are read then
    from odevio_amd import _lib
    n = _lib.load().odevio_audit_violations()
    assert n == 0, f"bounds audit: {n} plan(s) saw a kernel compute an out-of-bounds address"
