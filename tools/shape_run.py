"""One launch shape of the Breakout loop for a kernel trace: python tools/shape_run.py NUM_ENVS SEEDS UPDATES [MATMUL_DTYPE]
(run under rocprofv3 --kernel-trace; prints env-steps/s and the kernel forms taken)"""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys

n_envs, seeds, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
warm = 3
cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", f"alg.NUM_ENVS={n_envs}", "alg.TEST_DURING_TRAINING=False"]))
if len(sys.argv) > 4:
    cfg["MATMUL_DTYPE"] = sys.argv[4]
cfg["TOTAL_TIMESTEPS"] = (steps + warm + 2) * n_envs * cfg["NUM_STEPS"]
tr = make_train(cfg, device="cuda:0")
upd, _ = tr.make_batch_runner(seed_keys(0, seeds)) if seeds > 1 else tr.make_runner(seed_keys(0, 1)[0])
for u in range(warm):
    upd(u)
torch.cuda.synchronize()
t0 = time.perf_counter()
for u in range(warm, warm + steps):
    upd(u)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("NUM_ENVS=%d seeds=%d: %.4g env-steps/s, %.4f ms per update, forms %s" % (n_envs, seeds, n_envs * cfg["NUM_STEPS"] * seeds * steps / dt, 1e3 * dt / steps, _lib.last_kernel_form()), flush=True)
