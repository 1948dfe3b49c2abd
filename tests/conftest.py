import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _native_up_to_date():
    """Rebuild the native pieces when their sources are newer (a no-op otherwise): tests must never run a
    stale libirbfdl.so.  hipcc cross-compiles gfx950 without a GPU; on the GPU box the in-tree .so travels."""
    import subprocess
    for d in (os.path.join(ROOT, "irbaboon_amd", "csrc"), os.path.join(ROOT, "oracle")):
        subprocess.run(["make", "-s", "-j3", "-C", d], check=True)


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; built in-tree by __graft_entry__.build().  GPU tests fail loudly
    when it is missing -- there is no fallback."""
    import irbaboon_amd
    return irbaboon_amd.lib()
