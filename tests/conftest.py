import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gandiva: marker used by pyarrow's own test_gandiva.py")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The C-ABI library and the oracle must exist; build them if this is a fresh checkout."""
    from gandiva_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    from oracle import oracle
    oracle.build()
