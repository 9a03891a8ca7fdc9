"""Yaml-default runs of the MinAtar suite (1 seed each): wall clock + final returns."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys
_lib.load()
torch.zeros(1, device="cuda").sum().item()
for game in sys.argv[1:] or ["Breakout-MinAtar", "Asterix-MinAtar", "Freeway-MinAtar", "SpaceInvaders-MinAtar"]:
    cfg = flatten(load_config(["+alg=pqn_minatar", f"alg.ENV_NAME={game}"]))
    t0 = time.time()
    train = make_train(cfg, device="cuda:0")
    out = train(seed_keys(0, 1)[0])
    torch.cuda.synchronize()
    m = out["metrics"]
    print(f"{game:24s} {time.time() - t0:6.2f} s  train ret {float(m['returned_episode_returns'][-1]):8.2f}  "
          f"test ret {float(m['test/returned_episode_returns'][-1]):8.2f}  backend {train.backend}/{out['runner_state']['driver']}", flush=True)
