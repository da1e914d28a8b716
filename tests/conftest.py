import os
import sys

import pytest
import torch  # noqa: F401  -- before pyhvx loads libhelix_vec_gfx950.so: the process must bind ONE HIP runtime, torch's

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "helix-db_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu2: needs TWO devices on one node (RCCL between real ranks); skipped where only one is visible")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.lib()
    return _orc
