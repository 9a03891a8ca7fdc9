"""BASELINE.json configs[4]: Craftax-Classic PQN at the yaml shape (1024 envs, 1 step x 1 minibatch x 1 epoch, 4 x 1024 MLP
with BatchRenorm input, 1-step loss, optimistic resets): wall clock, env-steps/s and the learning signal of a short run.
python tools/craftax_c5_run.py [updates] [driver 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys
updates = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
cfg = flatten(load_config(["+alg=pqn_craftax", "alg.ENV_NAME=Craftax-Classic-Symbolic-v1"]))
cfg["TOTAL_TIMESTEPS"] = updates * cfg["NUM_ENVS"] * cfg["NUM_STEPS"]
cfg["TOTAL_TIMESTEPS_DECAY"] = cfg["TOTAL_TIMESTEPS"]
if len(sys.argv) > 2:
    cfg["_DRIVER"] = bool(int(sys.argv[2]))
train = make_train(cfg, device="cuda:0", script="craftax")
update, finish = train.make_runner(seed_keys(0, 1)[0])
for u in range(50):
    update(u)
torch.cuda.synchronize()
t0 = time.time()
for u in range(50, updates):
    update(u)
torch.cuda.synchronize()
dt = time.time() - t0
res = finish()
m = res["metrics"]
rets = m["returned_episode_returns"]
ok = ~torch.isnan(rets)
first, last = rets[:updates // 5][ok[:updates // 5]].mean(), rets[-updates // 5:][ok[-updates // 5:]].mean()
print(f"Craftax-Classic C5 shape ({train.backend}, driver {res['runner_state']['driver']}): {updates - 50} updates x {cfg['NUM_ENVS']} envs in {dt:.2f} s = "
      f"{(updates - 50) * cfg['NUM_ENVS'] / dt:.3e} env-steps/s ({dt / (updates - 50) * 1e3:.3f} ms/update); mean finished-episode return "
      f"first fifth {float(first):.3f} -> last fifth {float(last):.3f}; td_loss {float(m['td_loss'][50]):.4f} -> {float(m['td_loss'][-1]):.4f}")
