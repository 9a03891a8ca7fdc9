"""PLUMBING ONLY -- NOT PARITY.  Runs every consumer of the reference fixtures (tests/test_reference_fixtures_cpu.py,
through tests/reference_fixture_maps.py) against stand-in .npz files that tests/plumbing_fixtures.py writes FROM THE ORACLE
into a temporary directory, with the generator's file names, key layout, shapes and dtypes.  A pass says: the day
tests/golden/make_reference_fixtures.py has been run in the reference's jax environment, the consumers execute (key
names, canonical-state maps, shapes line up) instead of failing on their first line.  It says nothing about the oracle's
agreement with the reference -- the data is the oracle's own -- and DESIGN.md keeps "parity unpinned against reference
executions" until the real files exist."""
import inspect

import pytest

import tests.reference_fixture_maps as maps
import tests.test_reference_fixtures_cpu as consumers
from tests.plumbing_fixtures import write_all

CONSUMERS = sorted(n for n, f in vars(consumers).items() if n.startswith("test_") and inspect.isfunction(f))


@pytest.fixture(scope="module")
def plumbing_dir(tmp_path_factory, oracle):
    d = tmp_path_factory.mktemp("plumbing_not_reference")
    write_all(oracle, d)
    return d


def test_every_consumer_is_covered():
    assert len(CONSUMERS) == 9, CONSUMERS


@pytest.mark.parametrize("name", CONSUMERS)
def test_plumbing_only_consumer_runs(name, plumbing_dir, oracle, monkeypatch):
    monkeypatch.setattr(maps, "GOLDEN", str(plumbing_dir))
    getattr(consumers, name)(oracle)
