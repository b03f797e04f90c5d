import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _restore_constants():
    """modules read NFEATURES etc. from the package's constants at construction
    time (as the reference does); tests patch it, so restore after each test."""
    from opensetgaitrecognition_pcaa_amd import constants
    saved = {k: getattr(constants, k) for k in ("NFEATURES", "NMAX", "BATCH_SIZE")}
    yield
    for k, v in saved.items():
        setattr(constants, k, v)


@pytest.fixture(autouse=True)
def _pin_global_numerics_state():
    """Every test starts (and leaves the process) in the documented defaults: fp32 parity mode, no SyncBN group.
    A test that wants bf16 says so itself (``set_precision`` or the trainer's ``precision=``), so no result depends
    on which test ran before it."""
    from opensetgaitrecognition_pcaa_amd import functional as F_hip
    F_hip.set_precision("fp32")
    F_hip.set_sync_bn_group(None)
    yield
    F_hip.set_precision("fp32")
    F_hip.set_sync_bn_group(None)
