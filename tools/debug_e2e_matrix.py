"""One update, torch-op network path vs oracle, for a matrix of shapes: which parameter (NUM_ENVS, minibatch size,
epochs) makes the two part?  Prints one summary line per shape + where in the fc1 kernel the differing elements sit."""
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

shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(4096, 32, 32, 2)]
for n_envs, steps, mbs, ep in shapes:
    cfg = flatten(load_config(["+alg=pqn_minatar"]))
    cfg.update({"NUM_ENVS": n_envs, "NUM_STEPS": steps, "NUM_MINIBATCHES": mbs, "NUM_EPOCHS": ep, "ENV_NAME": "Breakout-MinAtar",
                "TOTAL_TIMESTEPS": 1 * n_envs * steps, "TOTAL_TIMESTEPS_DECAY": 30 * n_envs * steps, "TEST_DURING_TRAINING": False})
    key = seed_keys(0, 1)[0]
    net = QNetwork("cnn", (10, 10, 4), 3, device="cuda:0")
    theta0 = net.init(123)
    c = dict(cfg)
    c["_BACKEND"] = "torch"
    c["_INIT_PARAMS"] = theta0
    x = make_train(c, device="cuda:0")(key)["runner_state"]["theta"].cpu().numpy()
    y = oracle.make_train(dict(cfg))(key, theta0.cpu().numpy())["theta"]
    t0 = theta0.cpu().numpy()
    d = np.abs(x - y)
    bad = d > (2e-5 + 2e-3 * np.abs(y))
    off, n = net.offsets["CNN_0/Dense_0/kernel"]
    bk = bad[off:off + n].reshape(1024, 128)
    rows, cols = bk.sum(1), bk.sum(0)
    print(f"N={n_envs} T={steps} MB={mbs} EP={ep} (B={n_envs * steps // mbs}): max {d.max():.2e} bad {bad.mean():.4%} "
          f"rel-L2 {np.linalg.norm(x - y) / np.linalg.norm(y - t0):.2e} | fc1 bad {int(bk.sum())}: rows with any {int((rows > 0).sum())} "
          f"(top {sorted(rows.tolist())[-3:]}), cols with any {int((cols > 0).sum())} (top {sorted(cols.tolist())[-3:]})", flush=True)
