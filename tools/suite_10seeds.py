"""10 seeds per game at the yaml defaults (seed-batched launches): wall clock and final return statistics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys, vmap_train
_lib.load()
torch.zeros(1, device="cuda").sum().item()
for game in sys.argv[1:] or ["Breakout-MinAtar", "Asterix-MinAtar", "Freeway-MinAtar", "SpaceInvaders-MinAtar"]:
    cfg = flatten(load_config(["+alg=pqn_minatar", f"alg.ENV_NAME={game}", "NUM_SEEDS=10"]))
    if os.environ.get("PQN_MATMUL"):
        cfg["MATMUL_DTYPE"] = os.environ["PQN_MATMUL"]
    t0 = time.time()
    outs = vmap_train(make_train(cfg, device="cuda:0"), seed_keys(0, 10))
    torch.cuda.synchronize()
    dt = time.time() - t0
    te = outs["metrics"]["test/returned_episode_returns"][:, -1].double()
    tr = outs["metrics"]["returned_episode_returns"][:, -1].double()
    print(f"{cfg.get('MATMUL_DTYPE', 'f32'):7s} {game:24s} {dt:6.2f} s  test ret mean {te.mean():7.2f} std {te.std():6.2f} min {te.min():7.2f} max {te.max():7.2f} | "
          f"train ret mean {tr.mean():7.2f} std {tr.std():6.2f}", flush=True)
