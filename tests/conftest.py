import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# torch first: its bundled HIP runtime must be the one the process loads, whichever test file is collected first
# (libzkhip.so then binds to the same runtime; two runtimes in one process do not both see the GPU)
import torch  # noqa: E402,F401

import zkhip_pkg  # noqa: E402

zkhip_pkg.load()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def co():
    """The C oracle (built on demand with gcc; test infrastructure only)."""
    import coracle
    coracle.lib()
    return coracle


@pytest.fixture(scope="session")
def ctx():
    """A libzkhip context on cuda:0.  No fallback: fails when the HIP library or the device is missing."""
    import zkevm_circuits_amd as zk
    c = zk.Context(0)
    yield c
    c.close()
