import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; built in-tree by __graft_entry__.build().  GPU tests fail loudly
    when it is missing -- there is no fallback."""
    import irbaboon_amd
    return irbaboon_amd.lib()
