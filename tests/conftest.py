import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the native library and the CPU checkers once per session (no-ops when fresh)."""
    from gcn_amd import build as gbuild
    gbuild.build()
    if not os.path.exists(os.path.join(ROOT, "oracle", "libspmm_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    yield
