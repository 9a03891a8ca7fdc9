"""Experiment: S independent seeds of the bench workload on S HIP streams of one GPU (hipGraph replay per seed).
usage: python tools/multi_seed_streams.py [S ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import workload_config
from purejaxql_amd import _lib
from purejaxql_amd.pqn import make_train, seed_keys

_lib.load()
K, W = 10, 3
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    cfg = workload_config(4096, "seeds")
    cfg["TOTAL_TIMESTEPS"] = (K + W + 3) * cfg["NUM_ENVS"] * cfg["NUM_STEPS"]
    cfg["_FUSED_OPT"] = False   # updates of different seeds are in flight at once: two-kernel fold + optimizer
    streams = [torch.cuda.Stream() for _ in range(S)]
    runners = []
    for s, key in zip(streams, seed_keys(0, S)):
        with torch.cuda.stream(s):
            train = make_train(dict(cfg), device="cuda:0")
            runners.append(train.make_runner(key)[0])
    for u in range(W):
        for s, upd in zip(streams, runners):
            with torch.cuda.stream(s):
                upd(u)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for u in range(W, W + K):
        for s, upd in zip(streams, runners):
            with torch.cuda.stream(s):
                upd(u)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"S={S}: {S * K * 4096 * 32 / dt:.4g} env-steps/s aggregate, {dt / K * 1e3:.3f} ms per round", flush=True)
    del runners
