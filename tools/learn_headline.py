"""Learning sanity at the bench shape: 16 seeds x 4096 envs of a MinAtar game, TOTAL_TIMESTEPS per seed from argv (default
2e7), evaluations on -- with the kernels the launch takes by default (position-parallel forms) in the f16x2 (package default at this
shape) and bf16x3 operand modes, and with PQN_BWD_POS=0 PQN_ROLLOUT_POS=0 (the pair kernels of round 4, bf16x3).
python tools/learn_headline.py [game] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from purejaxql_amd import _lib
from purejaxql_amd.config_loader import flatten, load_config
from purejaxql_amd.pqn import make_train, seed_keys, vmap_train
_lib.load()
torch.zeros(1, device="cuda").sum().item()
game = sys.argv[1] if len(sys.argv) > 1 else "Breakout-MinAtar"
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 2e7
for label, dtype, opts in (("position-parallel forms, f16x2 (default)", "auto", {}), ("position-parallel forms, bf16x3", "bf16x3", {}),
                           ("pair kernels (bwd_pos = rollout_pos = 0)", "bf16x3", {"bwd_pos": 0, "rollout_pos": 0})):
    cfg = flatten(load_config(["+alg=pqn_minatar", f"alg.ENV_NAME={game}", "alg.NUM_ENVS=4096", "NUM_SEEDS=16"]))
    cfg["MATMUL_DTYPE"] = dtype
    cfg["TOTAL_TIMESTEPS"] = cfg["TOTAL_TIMESTEPS_DECAY"] = steps
    with _lib.options(**opts):
        t0 = time.time()
        outs = vmap_train(make_train(cfg, device="cuda:0"), seed_keys(0, 16))
        torch.cuda.synchronize()
        dt = time.time() - t0
        forms = _lib.last_kernel_form()
    te = outs["metrics"]["test/returned_episode_returns"][:, -1].double()
    tr = outs["metrics"]["returned_episode_returns"][:, -1].double()
    print(f"{game:22s} {label:42s} forms {forms}  {dt:6.2f} s  test ret mean {te.mean():7.2f} std {te.std():6.2f} min {te.min():7.2f} max {te.max():7.2f} | "
          f"train ret mean {tr.mean():7.2f} std {tr.std():6.2f}", flush=True)
