"""PLUMBING ONLY -- NOT PARITY.  The HIP-side consumers of the reference fixtures (tests/test_reference_fixtures_gpu.py) against
the oracle-written stand-in files of tests/plumbing_fixtures.py (see tests/test_fixture_plumbing_cpu.py): key layout, state
import / export maps and bit packing line up, so the consumers run the day the real files arrive.  The comparison itself
is HIP path vs oracle -- which the parity tests proper already make on larger inputs -- and is not reported as reference parity."""
import pytest

import tests.reference_fixture_maps as maps
import tests.test_reference_fixtures_gpu as consumers
from tests.plumbing_fixtures import write_all

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def plumbing_dir(tmp_path_factory, oracle):
    d = tmp_path_factory.mktemp("plumbing_not_reference")
    write_all(oracle, d)
    return d


def test_plumbing_only_hip_consumers_run(gpu, oracle, plumbing_dir, monkeypatch):
    monkeypatch.setattr(maps, "GOLDEN", str(plumbing_dir))
    consumers.test_hip_q_lambda_vs_reference(gpu)
    consumers.test_hip_radam_vs_reference(gpu)
    for mode in ("f32", "bf16x3"):
        consumers.test_hip_qnetwork_vs_reference(gpu, oracle, mode)
    for name, canon in (("Breakout-MinAtar", maps.breakout_canon), ("SpaceInvaders-MinAtar", maps.spaceinvaders_canon)):
        consumers.test_hip_step_env_vs_gymnax(gpu, name, canon)
