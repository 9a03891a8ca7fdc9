import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pqn_oracle
    pqn_oracle.lib()
    return pqn_oracle


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from purejaxql_amd import _lib
    _lib.load()  # fail loudly if the HIP extension is missing on a GPU box
    return torch.device("cuda:0")
