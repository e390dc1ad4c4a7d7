import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; oracle/dfx_oracle.cpp), built on demand with g++."""
    from oracle import dfx_oracle
    dfx_oracle.build()
    return dfx_oracle


@pytest.fixture(scope="session")
def dfx():
    """The product: deepfactors_amd over libdfx.so.  Fails loudly when the HIP library is not built / no GPU."""
    import torch
    import deepfactors_amd
    assert os.path.exists(deepfactors_amd.LIB_PATH), "libdfx.so not built: run __graft_entry__.build()"
    assert torch.cuda.is_available(), "gpu-marked test without a HIP device"
    return deepfactors_amd
