import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys
_lib.load()
cfg = flatten(load_config(["+alg=pqn_cartpole"]))
print({k: cfg[k] for k in ("NUM_ENVS","NUM_STEPS","NUM_MINIBATCHES","NUM_EPOCHS","TEST_INTERVAL","TEST_NUM_ENVS","TOTAL_TIMESTEPS")})
torch.zeros(1, device="cuda").sum().item()
t0 = time.time(); train = make_train(cfg, device="cuda:0"); t1 = time.time()
update, finish = train.make_runner(seed_keys(0, 1)[0]); torch.cuda.synchronize(); t2 = time.time()
ts = []
for u in range(int(cfg["NUM_UPDATES"])):
    a = time.time(); update(u); torch.cuda.synchronize(); ts.append(time.time() - a)
out = finish(); torch.cuda.synchronize(); t3 = time.time()
import numpy as np
ts = np.array(ts)
print(f"make_train {t1-t0:.2f}s, make_runner(+first eval) {t2-t1:.2f}s, updates total {ts.sum():.2f}s: median {np.median(ts)*1e3:.2f} ms, "
      f"top {np.sort(ts)[-25:].sum():.2f}s in 25 slowest (evals), first two {ts[:2]}")
