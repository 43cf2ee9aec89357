import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def emu_ctx():
    """CPU wave-emulation of the kernel bodies (tests only; see tests/host_harness/harness.cpp)."""
    import hostemu
    return hostemu.context()


@pytest.fixture(scope="session")
def gpu_ctx():
    """The product: libdigiham_amd.so on cuda:0.  No fallback: missing library or GPU is an error."""
    import torch
    from digiham_amd import api
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    return api.Context(device=0)


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def ctx(request):
    """Runs a test once against the CPU wave emulation (CPU tier) and once against the GPU library (-m gpu)."""
    return request.getfixturevalue(request.param + "_ctx")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import json
    d = os.path.join(ROOT, "tests", "golden")
    return {"fec": np.load(os.path.join(d, "fec_ref.npz")), "hashes": json.load(open(os.path.join(d, "fec_ref_hashes.json"))),
            "chain": np.load(os.path.join(d, "chain_oracle.npz"))}
