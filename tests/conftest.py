import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """a default per-test limit (pytest-timeout, when it is there and no --timeout was given): a test that hangs — a kernel
    whose waves never finish would — ends the run with its name and the stacks instead of sitting until the box is killed"""
    if not config.pluginmanager.hasplugin("timeout") or config.getoption("timeout", None):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the native library and the CPU checkers once per session (no-ops when fresh)."""
    from gcn_amd import build as gbuild
    gbuild.build()
    if not os.path.exists(os.path.join(ROOT, "oracle", "libspmm_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    yield
