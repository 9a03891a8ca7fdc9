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


# np.testing.assert_allclose treats NaN == NaN as equal by default: a reference array that went NaN would let NaN results pass
# (VERDICT r5 weak point 3).  Every use in this suite compares against a reference that must be finite; a test that really wants
# NaN-tolerant comparison says equal_nan=True itself.
import numpy as _np

_plain_allclose = _np.testing.assert_allclose


def _finite_allclose(actual, desired, *args, **kwargs):
    if "equal_nan" not in kwargs:
        assert _np.isfinite(_np.asarray(desired, dtype=_np.float64)).all(), "reference array of assert_allclose is not finite"
        kwargs["equal_nan"] = False
    return _plain_allclose(actual, desired, *args, **kwargs)


_np.testing.assert_allclose = _finite_allclose
