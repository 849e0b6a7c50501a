import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # libmicformer_hip.so is a build artefact (git-ignored): on a clean checkout build it before any test imports the package
    # (hipcc cross-compiles gfx950 without a GPU; ~40 s once, a no-op afterwards).
    lib = os.path.join(ROOT, "micformer_amd", "libmicformer_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture()
def hook():
    """hook(name, value): set a test hook of the library (micf_set_option: a slower equivalent kernel, a shrunk capacity) for THIS
    test; every hook is restored afterwards.  The product path never sets one."""
    from micformer_amd import _lib
    saved = []

    def set_(name, value):
        saved.append((name, _lib.set_option(name, value)))

    yield set_
    for name, prev in reversed(saved):
        _lib.set_option(name, prev)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
