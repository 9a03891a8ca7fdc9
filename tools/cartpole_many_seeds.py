""""Thousands of seeds in parallel" (reference README.md:27) on the gymnax MLP path: S CartPole-v1 seeds at the yaml
defaults (5e5 steps each, evaluation off), batched into the launches in groups of 128.  usage: ... [S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys, vmap_train
_lib.load()
torch.zeros(1, device="cuda").sum().item()
for S in [int(a) for a in sys.argv[1:]] or [128]:
    cfg = flatten(load_config(["+alg=pqn_cartpole", f"NUM_SEEDS={S}"]))
    cfg["TEST_DURING_TRAINING"] = False
    t0 = time.time()
    outs = vmap_train(make_train(cfg, device="cuda:0"), seed_keys(0, S))
    torch.cuda.synchronize()
    dt = time.time() - t0
    r = outs["metrics"]["returned_episode_returns"][:, -1].double()
    print(f"CartPole-v1 x {S} seeds: {dt:.2f} s wall ({S * cfg['TOTAL_TIMESTEPS'] / dt:.4g} env-steps/s aggregate), final train "
          f"returned_episode_returns mean {r.mean():.1f} std {r.std():.1f} min {r.min():.1f} (seeds >= 400: {(r >= 400).sum().item()}/{S})", flush=True)
