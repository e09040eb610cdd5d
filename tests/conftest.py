import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True, scope="session")
def _library_options_from_env():
    """MAGMA_TEST_OPTIONS="name=value,..." -> magma_set_option before the tests (diagnostic runs
    of the GPU suite with a library switch flipped; unset in normal runs)."""
    opts = [kv for kv in os.environ.get("MAGMA_TEST_OPTIONS", "").split(",") if kv]
    if opts:
        from magma_amd import _lib
        for kv in opts:
            name, val = kv.split("=")
            _lib.call("magma_set_option", name.strip().encode(), int(val))
    yield
