import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pqn_oracle as oracle
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys
from purejaxql_amd.networks import QNetwork
n_envs, steps, mbs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = flatten(load_config(["+alg=pqn_minatar"]))
cfg.update({"NUM_ENVS": n_envs, "NUM_STEPS": steps, "NUM_MINIBATCHES": mbs, "NUM_EPOCHS": 2, "ENV_NAME": "Breakout-MinAtar",
            "TOTAL_TIMESTEPS": 3 * n_envs * steps, "TOTAL_TIMESTEPS_DECAY": 30 * n_envs * steps, "TEST_DURING_TRAINING": False})
key = seed_keys(0, 1)[0]
net = QNetwork("cnn", (10, 10, 4), 3, device="cuda:0")
theta0 = net.init(123)
outs = {}
for be in ("fused", "torch"):
    c = dict(cfg); c["_BACKEND"] = be; c["_INIT_PARAMS"] = theta0
    outs[be] = make_train(c, device="cuda:0")(key)
oo = oracle.make_train(dict(cfg))(key, theta0.cpu().numpy())
for u in range(3):
    print("update", u)
    for k in ("td_loss", "qvals", "returned_episode_returns", "returned_episode", "timestep"):
        print("  %-26s fused %.8f torch %.8f oracle %.8f" % (k, float(outs["fused"]["metrics"][k][u]), float(outs["torch"]["metrics"][k][u]), oo["metrics"][u][k]))
for be in ("fused", "torch"):
    d = np.abs(outs[be]["runner_state"]["theta"].cpu().numpy() - oo["theta"])
    print(be, "theta max abs diff", d.max(), "n>2e-5:", (d > 2e-5).sum(), "n>1e-4:", (d > 1e-4).sum())
