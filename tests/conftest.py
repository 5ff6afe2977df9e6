import os
import sys

import pytest

# the library's two test hooks (KG_TEST_TINY_LISTS, KG_TEST_FAIL_ALLOC) are inert unless the process opted in before its
# first scan (include/kmerguts_hip.h)
os.environ.setdefault("KG_ENABLE_TEST_HOOKS", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (C restatement), built on demand.  Test infrastructure only."""
    from oracle import kgo
    kgo.build()
    kgo.load()
    return kgo


@pytest.fixture(scope="session")
def native():
    """The HIP library through ctypes; fails loudly when it is missing."""
    from kmergutsjava_amd import _native
    _native.load()
    return _native
