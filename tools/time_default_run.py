"""Wall-clock of the yaml-default MinAtar run (128 envs x 32 steps, 1e7 timesteps) for S seeds:
usage: python tools/time_default_run.py [S] [concurrent 0/1] [TEST_DURING_TRAINING 0/1] [KEY=VALUE config overrides ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys, vmap_train

_lib.load()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
conc = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
test = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", f"NUM_SEEDS={S}"]))
cfg["TEST_DURING_TRAINING"] = test
for kv in sys.argv[4:]:
    k_, v_ = kv.split("=", 1)
    cfg[k_] = {"True": True, "False": False}.get(v_, v_ if not v_.replace(".", "").isdigit() else (int(v_) if v_.isdigit() else float(v_)))
torch.zeros(1, device="cuda").sum().item()
t0 = time.time()
train = make_train(cfg, device="cuda:0")
outs = vmap_train(train, seed_keys(0, S), concurrent=conc)
torch.cuda.synchronize()
dt = time.time() - t0
m = outs["metrics"]
k = "test/returned_episode_returns" if test else "returned_episode_returns"
print(f"S={S} concurrent={conc} test={test}: {dt:.2f} s wall, {S * cfg['TOTAL_TIMESTEPS'] / dt:.4g} env-steps/s, "
      f"final {k} = {[round(float(x), 2) for x in m[k][:, -1]]}", flush=True)
