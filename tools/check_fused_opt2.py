import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys
_lib.load()
outs = []
for fused in (True, False):
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar"]))
    cfg.update({"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 1, "NUM_EPOCHS": 1, "TOTAL_TIMESTEPS": 1 * 64 * 8,
                "TOTAL_TIMESTEPS_DECAY": 40 * 64 * 8, "TEST_DURING_TRAINING": False, "_FUSED_OPT": fused, "_GRAPH": False})
    outs.append(make_train(cfg, device="cuda:0")(seed_keys(0, 1)[0]))
    torch.cuda.synchronize()
a, b = (o["runner_state"] for o in outs)
lay = a["kernel_layout"]
for name in ("theta", "opt_mu", "opt_nu"):
    x, y = a[name], b[name]
    if name != "theta":
        x, y = lay.to_flax(x), lay.to_flax(y)
    d = (x - y).abs()
    nz = (d > 0).nonzero().flatten()
    print(name, "n diff", int(nz.numel()), "of", x.numel(), "max", d.max().item(), "first idx", nz[:8].tolist(),
          "vals", x[nz[:3]].tolist(), y[nz[:3]].tolist())
print("td_loss", outs[0]["metrics"]["td_loss"].tolist(), outs[1]["metrics"]["td_loss"].tolist())
