"""Debug: batched seeds vs solo runs, first update, piece by piece."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys
_lib.load()
cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar"]))
cfg.update({"NUM_ENVS": 32, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2, "TOTAL_TIMESTEPS": 4 * 32 * 8,
            "TOTAL_TIMESTEPS_DECAY": 40 * 32 * 8, "TEST_DURING_TRAINING": False, "_GRAPH": False})
keys = seed_keys(0, 3)
tr = make_train(dict(cfg), device="cuda:0")
upd, fin = tr.make_batch_runner(keys)
drv = upd.driver
upd(0)
torch.cuda.synchronize()
N, T, S = 32, 8, 3
for s in range(S):
    tr1 = make_train(dict(cfg), device="cuda:0")
    u1, f1 = tr1.make_runner(keys[s])
    d1 = u1.driver
    u1(0)
    torch.cuda.synchronize()
    ro_b, ro_1 = drv._keep[0], d1._keep[1]
    sl = slice(s * N, (s + 1) * N)
    print("seed", s,
          "action", torch.equal(ro_b.action[:, sl], ro_1.action),
          "reward", torch.equal(ro_b.reward[:, sl], ro_1.reward),
          "qmax", torch.equal(ro_b.qmax[:, sl], ro_1.qmax),
          "target", torch.equal(ro_b.target[:, sl], ro_1.target),
          "bits", torch.equal(ro_b.bits[:, sl], ro_1.bits),
          "perm(last epoch)", torch.equal(drv.sk_out[s * T * N:(s + 1) * T * N] & ((1 << 25) - 1), d1.sk_out & 0xFFFFFFFF),
          "loss_buf", (drv.loss_buf[s] - d1.loss_buf).abs().max().item(), drv.loss_buf[s][:3].tolist(), d1.loss_buf[:3].tolist(),
          "theta maxdiff", (drv.theta[s, :drv.layout.total] - d1._keep[0].theta).abs().max().item())
