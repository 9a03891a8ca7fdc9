"""Whole-loop env-steps/s of Breakout (operand mode MD, default f16x2) with the kernel forms the library picks by default against the position-parallel forms forced
(options bwd_pos = rollout_pos = 2): where should the form threshold sit?  python tools/form_sweep.py  (on a GPU box)"""
import os
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys


def rate(n_envs, seeds, pos, steps, warm=3):
    _lib.set_option("bwd_pos", pos)
    _lib.set_option("rollout_pos", pos)
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", f"alg.NUM_ENVS={n_envs}", "alg.TEST_DURING_TRAINING=False"]))
    cfg["MATMUL_DTYPE"] = os.environ.get("MD", "f16x2")
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


for n_envs, seeds, steps in ((512, 16, 30), (1024, 2, 40), (1024, 4, 30), (1024, 8, 24), (1024, 16, 16), (2048, 2, 30), (2048, 4, 20), (2048, 8, 16),
                             (4096, 1, 20), (4096, 2, 16), (4096, 4, 12), (4096, 8, 12)):
    row = []
    for pos in (1, 2):
        v, forms = rate(n_envs, seeds, pos, steps)
        row.append("%s %.3g %s" % ("default" if pos == 1 else "forced-pos", v, forms))
    print("NUM_ENVS=%d seeds=%d (256-sample tiles x seeds = %d):  " % (n_envs, seeds, n_envs // 256 * seeds) + "   ".join(row), flush=True)
