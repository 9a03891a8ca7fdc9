"""Learning sanity of MATMUL_DTYPE=f16 vs f32: 10 seeds of the yaml-default Breakout run each (seed-batched)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys, vmap_train
_lib.load()
torch.zeros(1, device="cuda").sum().item()
games = sys.argv[1:] or ["Breakout-MinAtar"]
for game in games:
    for dt in ("f32", "f16"):
        cfg = flatten(load_config(["+alg=pqn_minatar", f"alg.ENV_NAME={game}", "NUM_SEEDS=10"]))
        cfg["MATMUL_DTYPE"] = dt
        t0 = time.time()
        outs = vmap_train(make_train(cfg, device="cuda:0"), seed_keys(0, 10))
        torch.cuda.synchronize()
        te = outs["metrics"]["test/returned_episode_returns"][:, -1].double()
        tr = outs["metrics"]["returned_episode_returns"][:, -1].double()
        print(f"{game:22s} {dt}: {time.time() - t0:6.2f} s  test ret mean {te.mean():7.2f} std {te.std():6.2f} | "
              f"train ret mean {tr.mean():7.2f} std {tr.std():6.2f}", flush=True)
