import os, sys
sys.path.insert(0, os.getcwd())
import torch
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys
cfg = flatten(load_config(["+alg=pqn_craftax", "alg.ENV_NAME=Craftax-Classic-Symbolic-v1"]))
n_upd = 300
cfg.update({"NUM_ENVS": 128, "HIDDEN_SIZE": 256, "NUM_LAYERS": 2, "TOTAL_TIMESTEPS": n_upd * 128, "TOTAL_TIMESTEPS_DECAY": n_upd * 128,
            "LOG_ACHIEVEMENTS": True, "EPS_START": 1.0, "EPS_FINISH": 1.0})
tr = make_train(cfg, device="cuda:0", script="craftax")
update, finish = tr.make_runner(seed_keys(2, 1)[0])
for u in range(n_upd):
    update(u)
torch.cuda.synchronize()
d = update.driver
print(type(d).__name__, "ach_buf", None if d.ach_buf is None else (d.ach_buf.shape, int((d.ach_buf != 0).sum())),
      "args.achievements", d.args.achievements, "args.ach_metrics", d.args.ach_metrics)
print("ach_metrics nan rows", int(torch.isnan(d.ach_metrics[:, 0]).sum()), "nonzero", int((torch.nan_to_num(d.ach_metrics) != 0).sum()))
print("metrics nan", int(torch.isnan(d.metrics[:, 7]).sum()), d.metrics[:3])
out = finish()
print(out["metrics"]["Achievements/collect_wood"][:20])
