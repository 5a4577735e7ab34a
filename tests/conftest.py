import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # On a GPU box torch must be the FIRST to initialise HIP in the test process: its wheel carries a HIP runtime of its own under the
    # soname libleon_dna.so also links (/opt/rocm's), and whichever is loaded first serves both.  With the library's first (a test file
    # that never touches torch running before one that does) torch then finds "no ROCm-capable device".  bench.py imports torch first too.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:                       # noqa: BLE001 -- no torch, no GPU: nothing to order
        pass


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib


@pytest.fixture(params=["host-chains", "device-chains"])
def rc_chains(request, monkeypatch):
    """Small launches (up to 400 blocks, when the library's estimate says so) have their range-coder chains coded on host cores from the device modelers' records (host_blocks.h);
    larger ones by the device's coder waves (k_rc_encode).  Nearly every test launches a handful of blocks, so the tests that use this
    fixture run twice: as they come, and with LEON_RC_HOST_BLOCKS=0 (the device coder for every launch).  Same bytes either way."""
    # (named in the environment, the choice is taken as given; left alone, the library estimates both ways' times per launch)
    monkeypatch.setenv("LEON_RC_HOST_BLOCKS", "0" if request.param == "device-chains" else "400")
    return request.param
