"""Regression pins for the oracle (and, through the parity tests, for the HIP kernels).

NOT reference-generated goldens: the reference cannot run here (no jax / gymnax).  These are SHA-256 digests
of the oracle's own outputs on seeded inputs, committed so that an accidental change of the env rules, the
counter-based RNG streams, the shuffle, Q(lambda) or eps-greedy is caught as a diff against history.
Regenerate (only after an intended change): python tests/golden/make_regression_pins.py > tests/golden/regression_pins.json
"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import pqn_oracle as O  # noqa: E402

GAMES = ("Breakout-MinAtar", "Asterix-MinAtar", "Freeway-MinAtar", "SpaceInvaders-MinAtar", "CartPole-v1", "Acrobot-v1")


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode() + str(a.shape).encode() + a.tobytes())
    return h.hexdigest()


def env_pin(name, n=37, steps=300, key=20260926):
    env = O.OracleEnv(name)
    rng = np.random.default_rng(7)
    obs, st = env.reset(key, n)
    h = hashlib.sha256(digest(obs).encode())
    tot_r, tot_d = 0.0, 0
    for t in range(steps):
        a = rng.integers(0, env.num_actions, n).astype(np.int32)
        obs, st, r, d, info = env.step(O.fold_in(key, 1 + t), st, a)
        h.update(digest(obs, r, d, info["returned_episode_returns"], info["returned_episode_lengths"], info["timestep"]).encode())
        tot_r += float(r.sum())
        tot_d += int(d.sum())
    return {"n": n, "steps": steps, "key": key, "sha256": h.hexdigest(), "sum_reward": tot_r, "num_done": tot_d}


def pins():
    out = {"_comment": "self-regression pins of the oracle (tests/golden/make_regression_pins.py); not reference goldens",
           "envs": {g: env_pin(g) for g in GAMES}}
    rng = np.random.default_rng(11)
    q = rng.standard_normal((1000, 5)).astype(np.float32)
    a, qm = O.eps_greedy(q, np.float32(0.3), 12345)
    out["eps_greedy"] = {"sha256": digest(a, qm), "num_greedy": int((a == q.argmax(-1)).sum())}
    out["permutation"] = {"key": 987654321, "n": 4096, "sha256": digest(O.permutation(987654321, 4096).astype(np.int64))}
    r = (rng.random((32, 64)) < 0.05).astype(np.float32)
    d = rng.random((32, 64)) < 0.02
    qmax = rng.standard_normal((32, 64)).astype(np.float32)
    lq = rng.standard_normal(64).astype(np.float32)
    out["q_lambda"] = {"sha256_quirk": digest(O.q_lambda(r, d, qmax, lq, 0.99, 0.65, quirk=True)),
                       "sha256_atari": digest(O.q_lambda(r, d, qmax, lq, 0.99, 0.65, quirk=False))}
    out["fold_in"] = {"key": 42, "values": [int(O.fold_in(42, i)) for i in range(4)]}
    return out


if __name__ == "__main__":
    print(json.dumps(pins(), indent=1))
