"""Regression for the round-5 GPU memory access fault: 16 seeds x 4096 envs of Breakout with the yaml default
TEST_DURING_TRAINING=True died after ~16 updates.  Root cause (round 6, tools/repro/update_replay.cpp): the HIP runtime bundled in
the PyTorch wheel faults when a long hipGraph is replayed on the legacy NULL stream with eager launches queued behind the replay;
make_train now runs everything on a created stream.  Each case runs in a child process: a fault aborts the process, and the suite
should report a failure instead of dying with it.  Reference behaviour matched: evaluations inside the training loop
(purejaxql/pqn_minatar.py:340-350, `TEST_DURING_TRAINING: True` in config/alg/pqn_minatar.yaml:24)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, torch
sys.path.insert(0, %(root)r)
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys
cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", "alg.NUM_ENVS=4096", "NUM_SEEDS=16"]))
n_upd = %(n_upd)d
cfg["MATMUL_DTYPE"] = "bf16x3"
cfg["TOTAL_TIMESTEPS"] = cfg["TOTAL_TIMESTEPS_DECAY"] = n_upd * 4096 * 32
cfg["TEST_DURING_TRAINING"] = %(test)s
cfg["TEST_INTERVAL"] = 2.0 / n_upd + 1e-9          # an evaluation every 2 updates
tr = make_train(cfg, device="cuda:0")
assert tr.stream is not None and tr.stream.cuda_stream != 0
update, finish = tr.make_batch_runner(seed_keys(0, 16))
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
lib = _lib.load()
for u in range(n_upd):
    update(u)
    if %(eager)d:      # the bisected trigger: ~100 tiny launches enqueued behind every replay, no host wait
        with torch.cuda.stream(tr.stream):
            for _ in range(%(eager)d):
                _lib.check(lib.pqn_fold_in_range(12345, 1, 8, _lib.ptr(buf), _lib.stream_ptr()), "fold")
outs = finish()
torch.cuda.synchronize()
rs = outs[0]["runner_state"]
assert rs["driver"] == "graph", rs["driver_graph_error"]
m = outs[0]["metrics"]
assert torch.isfinite(m["td_loss"]).all() and int(m["env_step"][-1]) == n_upd * 4096 * 32
if %(test)s:
    t = torch.stack([o["metrics"]["test/returned_episode_returns"] for o in outs])
    assert t.shape == (16, n_upd) and torch.isfinite(t).all()
print("CLEAN", n_upd)
"""


def _run(n_upd, test, eager):
    code = CHILD % {"root": ROOT, "n_upd": n_upd, "test": "True" if test else "False", "eager": eager}
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "CLEAN" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.gpu
def test_headline_shape_with_evaluations_every_two_updates_runs_clean(gpu):
    """16 x 4096, TEST_DURING_TRAINING=True, 44 updates, an evaluation every 2: round 5 died here after ~16 updates"""
    _run(44, True, 0)


@pytest.mark.gpu
def test_eager_launches_behind_every_replay_run_clean(gpu):
    """the bisected trigger of round 5 (100 launches of pqn_fold_in_range behind every replay, no host wait; it faulted at update 26)"""
    _run(44, False, 100)
