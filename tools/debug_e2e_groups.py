"""Whole-update comparison fused HIP / torch-op network / oracle at a given shape, parameter differences broken down by
parameter group (usage: debug_e2e_groups.py NUM_ENVS NUM_STEPS NUM_MINIBATCHES [ENV_NAME])."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pqn_oracle as oracle
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.networks import QNetwork
from purejaxql_amd.pqn import make_train, seed_keys

n_envs, steps, mbs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
env_name = sys.argv[4] if len(sys.argv) > 4 else "Breakout-MinAtar"
cfg = flatten(load_config(["+alg=pqn_minatar"]))
cfg.update({"NUM_ENVS": n_envs, "NUM_STEPS": steps, "NUM_MINIBATCHES": mbs, "NUM_EPOCHS": 2, "ENV_NAME": env_name,
            "TOTAL_TIMESTEPS": 1 * n_envs * steps, "TOTAL_TIMESTEPS_DECAY": 30 * n_envs * steps, "TEST_DURING_TRAINING": False})
key = seed_keys(0, 1)[0]
oe = oracle.OracleEnv(env_name)
net = QNetwork("cnn", oe.obs_shape, oe.num_actions, device="cuda:0")
theta0 = net.init(123)
outs = {}
for be in ("fused", "torch"):
    c = dict(cfg)
    c["_BACKEND"] = be
    c["_INIT_PARAMS"] = theta0
    rs = make_train(c, device="cuda:0")(key)["runner_state"]
    outs[be] = rs["theta"].cpu().numpy()
    if be == "torch":
        nu_gpu = rs["opt_nu"].cpu().numpy()
ores = oracle.make_train(dict(cfg))(key, theta0.cpu().numpy())
oo = ores["theta"]
t0 = theta0.cpu().numpy()
pairs = [("fused", "oracle", outs["fused"], oo), ("torch", "oracle", outs["torch"], oo), ("fused", "torch", outs["fused"], outs["torch"])]
for a, b, x, y in pairs:
    d = np.abs(x - y)
    mv = np.linalg.norm(y - t0)
    cos = float(np.dot(x - t0, y - t0) / (np.linalg.norm(x - t0) * mv))
    bad = d > (2e-5 + 2e-3 * np.abs(y))
    print(f"{a} vs {b}: max {d.max():.3e}  bad {bad.mean():.4%}  rel-L2 of the update {np.linalg.norm(x - y) / mv:.3e}  cos {cos:.6f}")
    for k, (off, n) in net.offsets.items():
        dd = d[off:off + n]
        bb = bad[off:off + n]
        mvk = np.abs(y[off:off + n] - t0[off:off + n])
        print(f"    {k:28s} n={n:7d} max {dd.max():.2e} bad {int(bb.sum()):5d}  median|move| {np.median(mvk):.2e} max|move| {mvk.max():.2e}")

# are the elements that differ the ones whose gradient is at rounding-noise level?  RAdam's step is scale-free
# (u = r * m_hat / sqrt(v_hat)): an element whose gradient is noise still moves by ~lr per step, in a direction that
# depends on the summation order of the implementation.
d = np.abs(outs["torch"] - oo)
bad = d > (2e-5 + 2e-3 * np.abs(oo))
rv = np.sqrt(ores["opt_nu"])
print("sqrt(v) of the oracle's second moment: all elements median %.3e | 'bad' elements median %.3e max %.3e" %
      (np.median(rv[rv > 0]), np.median(rv[bad]), rv[bad].max()))
print("quantile of the bad elements' sqrt(v) within all elements:", np.round([np.mean(rv <= q) for q in np.quantile(rv[bad], [0.1, 0.5, 0.9])], 4))
print("gpu (torch path) sqrt(v) at the bad elements: median %.3e" % np.median(np.sqrt(nu_gpu[bad])))
