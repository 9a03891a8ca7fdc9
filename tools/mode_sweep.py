"""Whole-loop env-steps/s of Breakout under the two f32-grade operand modes over launch sizes: the data behind MATMUL_DTYPE=auto
(purejaxql_amd/qnet.py: resolve_matmul_dtype).  python tools/mode_sweep.py  (on a GPU box)"""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys


def rate(n_envs, seeds, mode, steps=12, warm=4):
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", f"alg.NUM_ENVS={n_envs}", "alg.TEST_DURING_TRAINING=False"]))
    cfg["MATMUL_DTYPE"] = mode
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
    return n_envs * cfg["NUM_STEPS"] * seeds * steps / dt, _lib.last_kernel_form()


for n_envs, seeds, steps in ((128, 1, 200), (128, 16, 100), (256, 1, 100), (512, 1, 100), (1024, 1, 60), (1024, 4, 40), (1024, 16, 20), (2048, 1, 40), (4096, 1, 20), (4096, 4, 12), (4096, 16, 12)):
    row = []
    for mode in ("f32", "bf16x3"):
        v, forms = rate(n_envs, seeds, mode, steps=steps, warm=max(3, steps // 5))
        row.append("%s %.3g %s" % (mode, v, forms))
    print("NUM_ENVS=%d seeds=%d minibatch=%d:  " % (n_envs, seeds, n_envs) + "   ".join(row), flush=True)
