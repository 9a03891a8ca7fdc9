"""Wall-clock of the yaml-default CartPole-v1 run (pqn_gymnax, 5e5 timesteps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys, vmap_train
_lib.load()
cfg = flatten(load_config(["+alg=pqn_cartpole"]))
torch.zeros(1, device="cuda").sum().item()
t0 = time.time()
train = make_train(cfg, device="cuda:0")
out = train(seed_keys(0, 1)[0])
torch.cuda.synchronize()
dt = time.time() - t0
m = out["metrics"]
print(f"CartPole-v1 yaml defaults: {dt:.2f} s wall, backend {train.backend}, driver {out['runner_state']['driver']}, "
      f"{cfg['NUM_UPDATES']} updates, final train ret {float(m['returned_episode_returns'][-1]):.1f}, "
      f"test ret {float(m['test/returned_episode_returns'][-1]) if 'test/returned_episode_returns' in m else None}")
