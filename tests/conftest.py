import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import hns_amd  # noqa: E402,F401  registers the package alias for multi-uav-pursuit-evasion_amd/

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built extension (the .so files are git-ignored): build once, here or on the GPU box
    from hns_amd import abi
    if not os.path.exists(abi.library_path()):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load
