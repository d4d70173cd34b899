import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Frames whose images come through mrh_upload_* are fused serially by default (host- and link-bound: the kernel pipeline buys
# nothing there); the parity tests feed exactly that way, so the suite asks for the pipeline on those frames too — every map the
# tests compare with the oracle has then been through the second stream, the zombies and the reclaim.  MRH_PIPE=0 (set by
# individual tests) is the serial path.
os.environ.setdefault("MRH_PIPE_UPLOADS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "soak: long-running GPU gates outside the suite's time budget (MRH_SOAK=1 python -m pytest tests -m soak)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference (test infrastructure; built on demand with gcc)."""
    import parity_utils as pu

    return pu.oracle_lib()


@pytest.fixture(scope="session")
def hip():
    """The product library. No fallback: a missing build or a missing device is a hard failure."""
    from mrhash_amd import capi

    return capi.load_hip()
