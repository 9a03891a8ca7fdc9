"""Fused (grid-barrier) fold+clip+RAdam vs the two-kernel version: 3 updates of every MinAtar game, bitwise."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys
_lib.load()
for game in ("Breakout-MinAtar", "Asterix-MinAtar", "Freeway-MinAtar", "SpaceInvaders-MinAtar"):
    outs = []
    for fused in (True, False):
        cfg = flatten(load_config(["+alg=pqn_minatar", f"alg.ENV_NAME={game}"]))
        cfg.update({"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2, "TOTAL_TIMESTEPS": 3 * 64 * 8,
                    "TOTAL_TIMESTEPS_DECAY": 40 * 64 * 8, "TEST_DURING_TRAINING": False, "_FUSED_OPT": fused})
        outs.append(make_train(cfg, device="cuda:0")(seed_keys(0, 1)[0]))
        torch.cuda.synchronize()
    a, b = outs
    print(game, "theta equal", torch.equal(a["runner_state"]["theta"], b["runner_state"]["theta"]),
          "maxdiff", (a["runner_state"]["theta"] - b["runner_state"]["theta"]).abs().max().item(),
          "td_loss equal", torch.equal(a["metrics"]["td_loss"], b["metrics"]["td_loss"]),
          "opt_mu equal", torch.equal(a["runner_state"]["opt_mu"], b["runner_state"]["opt_mu"]), flush=True)
