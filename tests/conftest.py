import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (built on demand with gcc)."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hiplib():
    """libmpcg_hip.so, built on demand if a compiler is present (the prebuilt .so travels to the GPU box)."""
    from mpcgpu_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH) or (build.is_stale() and os.path.exists(build.HIPCC)):
        build.build()
    return _lib.load()
