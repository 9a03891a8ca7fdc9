import sys, time, torch
sys.path.insert(0, "/root/repo")
from purejaxql_amd.config_loader import load_config, flatten
from purejaxql_amd.pqn import make_train, seed_keys
cfg = flatten(load_config(["+alg=pqn_cartpole", "alg.ENV_NAME=Acrobot-v1", "alg.TOTAL_TIMESTEPS=500000"]))
tr = make_train(cfg, device="cuda:0")
t0 = time.time(); out = tr(seed_keys(0, 1)[0]); torch.cuda.synchronize()
m = out["metrics"]
print("backend", tr.backend, "wall %.1fs" % (time.time() - t0), "returns first/last", float(m["returned_episode_returns"][:5].mean()), float(m["returned_episode_returns"][-5:].mean()), "test", [round(float(x), 1) for x in m.get("test/returned_episode_returns", m["returned_episode_returns"])[-3:]])
