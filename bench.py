#!/usr/bin/env python
"""bench.py -- env-steps/sec of the PQN hot path, MinAtar-Breakout 4096 envs.

A "step" is ONE PQN update of the full loop (rollout of NUM_STEPS x NUM_ENVS
env-steps + Q(lambda) targets + NUM_EPOCHS x NUM_MINIBATCHES optimizer steps),
the unit the reference's own SPS figures count
(purejaxql/pqn_mujoco_playground.py:658-668).  value = env-steps/s over all ranks.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task prompt), incl. `roofline`
(dominant kernel, HIP-event timed live) and `cpu_baseline` (the CPU oracle, timed
on this box's host cores on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENV_STEP_BYTES = 1926.0      # algorithmic bytes of one env.step (f32 obs surface)
LOOP_FLOP = 2.367e6          # algorithmic FLOP per env-step of the whole loop
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32 MFMA/vector peak
# Dominant kernel = qnet_cnn_train_kernel (T1: forward + backward except the fc1 weight gradient).
# Algorithmic FLOP per sample: fwd conv 73,728 + fc1 262,144 + fc2 768; bwd fc2 dgrad+wgrad
# 1,536 + fc1 dgrad 262,144 + conv wgrad 73,728 (the conv has no input gradient) = 674,048.
T1_FLOP_PER_SAMPLE = 674048.0


def workload_config(num_envs: int, mode: str):
    from purejaxql_amd.config_loader import flatten, load_config
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", f"alg.NUM_ENVS={num_envs}",
                               "alg.TEST_DURING_TRAINING=False"]))
    return cfg


def multi_seed_rate(cfg, seeds, steps, warmup, dev):
    """`seeds` independent runs of the bench workload (own parameters / optimizer / envs / keys) batched into
    the SAME launches (grid.y = seed, pqn_cnn_update_seeds; purejaxql_amd.pqn.vmap_train does the same):
    aggregate env-steps/s.  Reported beside `value`, which stays the single-seed number."""
    import torch
    from purejaxql_amd.pqn import make_train, seed_keys
    c = dict(cfg)
    c.pop("_ENV_SHARD", None)
    train = make_train(c, device=str(dev))
    update, _finish = train.make_batch_runner(seed_keys(1, seeds))
    for u in range(warmup):
        update(u)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for u in range(warmup, warmup + steps):
        update(u)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"seeds_per_gpu": seeds, "value": seeds * steps * c["NUM_ENVS"] * c["NUM_STEPS"] / dt, "unit": "env-steps/s",
            "ms_per_round": dt / steps * 1e3,
            "how": "all seeds in the same kernel launches (grid.y = seed), one hipGraph replay per update"}


def cpu_baseline(cfg, theta0, max_seconds=45.0):
    """The CPU oracle (numpy/C restatement, kind="port") on one update of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import pqn_oracle as oracle
    ocfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
    n, t = int(ocfg["NUM_ENVS"]), int(ocfg["NUM_STEPS"])
    # bounded sample: 1 update at a reduced env count if the full one would take too long
    sample_envs = n
    # threads actually used: numpy's BLAS pool (the network, most of the time) and OpenMP (C env step)
    try:
        from threadpoolctl import threadpool_info
        pools = threadpool_info()
        cores = max([p.get("num_threads", 1) for p in pools] + [1])
        pool_desc = ", ".join(f"{p.get('internal_api')}:{p.get('num_threads')}" for p in pools)
    except Exception:
        cores, pool_desc = os.cpu_count() or 1, "unknown"
    ocfg["NUM_ENVS"] = sample_envs
    ocfg["TOTAL_TIMESTEPS"] = ocfg["TOTAL_TIMESTEPS_DECAY"] = 1e7
    train = oracle.make_train(ocfg)
    t0 = time.perf_counter()
    train(12345, theta0, max_updates=1)
    dt = time.perf_counter() - t0
    # env-only rate (uniform-random actions, no network): the C oracle's OpenMP env.step + auto-reset + LogWrapper
    env = oracle.OracleEnv(ocfg["ENV_NAME"])
    _obs, st = env.reset(1, sample_envs)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, env.num_actions, size=(50, sample_envs)).astype(np.int32)
    env.step(2, st, acts[0])
    t1 = time.perf_counter()
    for i in range(50):
        _o, st, _r, _d, _info = env.step(100 + i, st, acts[i])
    env_only = 50 * sample_envs / (time.perf_counter() - t1)
    return {"value": sample_envs * t / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "env_only_env_steps_per_s": env_only,
            "sample": f"1 full PQN update (rollout+Q(lambda)+{ocfg['NUM_EPOCHS']}x{ocfg['NUM_MINIBATCHES']} SGD steps) "
                      f"at NUM_ENVS={sample_envs}, NUM_STEPS={t}: {sample_envs * t} env-steps in {dt:.1f}s; "
                      "oracle/pqn_oracle.py (C env/eps-greedy/Q(lambda)/RAdam + numpy-BLAS network), not JAX; "
                      f"host has {os.cpu_count()} logical cores, thread pools: {pool_desc}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)    # 20 warm-up updates, then >= 200 timed
    ap.add_argument("--warmup", type=int, default=20)    # (2.6e7 env-steps between the two stream syncs, ~1 s)
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--mode", default="seeds", choices=["seeds", "envs"],
                    help="multi-GPU sharding: independent seeds per rank (no collective) or envs of one seed "
                         "(RCCL gradient all-reduce per optimizer step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mixed-precision", action="store_true", help="skip the extra MATMUL_DTYPE=f16 measurement")
    ap.add_argument("--matmul-dtype", default="f32", choices=["f32", "f16"],
                    help="operand type of the fc1 products (config MATMUL_DTYPE); f16 = fp16 operands, f32 accumulation")
    ap.add_argument("--multi-seed", type=int, default=16,
                    help="N=1 only: also report the aggregate rate of this many independent seeds of the same workload "
                         "batched into the same launches (jax.vmap over seeds, pqn_minatar.py:459-461; 16 per GPU = "
                         "BASELINE.json configs[3]: 128 seeds over 8 GPUs); 0 = skip")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: purejaxql_amd has no CPU path")
    # PQN_BENCH_ONE_GPU=1 (tests on a 1-GPU box only): every rank on device 0, gloo instead of RCCL -- exercises the
    # multi-rank control flow (barriers, max-over-ranks timing, rank-0 line), not the scaling
    one_gpu = os.environ.get("PQN_BENCH_ONE_GPU", "0") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from purejaxql_amd import _lib
    _lib.load()
    from purejaxql_amd.pqn import make_train, seed_keys
    from purejaxql_amd import dist as pdist

    cfg = workload_config(args.num_envs, args.mode)
    cfg["MATMUL_DTYPE"] = args.matmul_dtype
    cfg["TOTAL_TIMESTEPS"] = (args.steps + args.warmup + 3) * cfg["NUM_ENVS"] * cfg["NUM_STEPS"]
    grad_hook = None
    if world > 1 and args.mode == "envs":
        grad_hook = pdist.make_grad_allreduce_hook()
    seed_index = rank if args.mode == "seeds" else 0
    key = seed_keys(0, world)[seed_index]
    if args.mode == "envs" and world > 1:
        cfg["_ENV_SHARD"] = (rank, world)
    train = make_train(cfg, device=str(dev), grad_hook=grad_hook)
    update, finish = train.make_runner(key)

    lib = _lib.load()
    fused = train.backend == "fused"
    for u in range(args.warmup):
        update(u)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for u in range(args.warmup, args.warmup + args.steps):
        update(u)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    env_steps = cfg["NUM_ENVS"] * cfg["NUM_STEPS"] * args.steps * world
    sps = env_steps / dt

    # The kernel-timer pass below runs 2 more updates.  With the envs of one seed sharded over ranks those updates
    # contain collectives, so EVERY rank has to run them (rank 0 alone would wait forever for its peers).
    if world > 1 and args.mode == "envs" and fused and rank != 0:
        for u in range(args.warmup + args.steps, args.warmup + args.steps + 2):
            update(u)
        torch.cuda.synchronize()

    if rank == 0:
        roof = None
        drv0 = getattr(update, "driver", None)
        driver_mode = None if drv0 is None else ("hipGraph replay" if drv0.graph is not None else "C++ enqueue (eager)")
        if fused:
            import ctypes
            # HIP-event timing of the dominant kernel on its launch stream.  Events recorded while a hipGraph
            # is being captured cannot be read back on ROCm 7, so the timed region above runs the graph and
            # this pass re-runs 2 more updates of the same workload through the eager C++ enqueue with the
            # kernel timer on (same kernels, same shapes, same buffers).
            drv = getattr(update, "driver", None)
            if drv is not None:
                drv.graph, drv.use_graph = None, False
            _lib.check(lib.pqn_prof_enable(1), "pqn_prof_enable")
            for u in range(args.warmup + args.steps, args.warmup + args.steps + 2):
                update(u)
            torch.cuda.synchronize()
            cnt, tot = ctypes.c_int32(0), ctypes.c_float(0.0)
            _lib.check(lib.pqn_prof_read(ctypes.byref(cnt), ctypes.byref(tot)), "pqn_prof_read")
            lib.pqn_prof_enable(0)
            mb = cfg["NUM_ENVS"] * cfg["NUM_STEPS"] // cfg["NUM_MINIBATCHES"]
            if cnt.value == 0:
                raise SystemExit("kernel timer recorded nothing")
            avg_s = tot.value * 1e-3 / cnt.value
            achieved = T1_FLOP_PER_SAMPLE * mb / avg_s / 1e12
            traffic, tsrc = None, None
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_train_kernel.json")
            if os.path.exists(pmc):
                pj = json.load(open(pmc))
                traffic, tsrc = pj.get("hbm_bytes_per_launch"), "profiles/r01_pmc_train_kernel.json (separate rocprofv3 --pmc passes)"
            roof = {"kernel": "qnet_cnn_train_kernel<4> (fwd + bwd of one 4096-sample minibatch, f32 MFMA)",
                    "bound": "mfma", "achieved": achieved, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / F32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
                    "avg_launch_us": avg_s * 1e6, "launches_timed": cnt.value,
                    "flop_per_launch": T1_FLOP_PER_SAMPLE * mb}
        if roof is None:
            from purejaxql_amd.profiling import time_env_step_kernel
            k_ms = time_env_step_kernel(cfg["NUM_ENVS"], dev)
            achieved = ENV_STEP_BYTES * cfg["NUM_ENVS"] / (k_ms * 1e-3) / 1e9
            roof = {"kernel": "minatar_kernel<Breakout> (env.step, f32 obs surface)", "bound": "hbm",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None, "avg_launch_us": k_ms * 1e3}
        out = {
            "metric": "env-steps/sec (whole node), MinAtar-Breakout 4096 envs", "value": sps, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.matmul_dtype == "f32" else "f16 operands / f32 accumulate (fc1), f32 elsewhere", "data": "synthetic",
            "config": {"workload": f"Breakout-MinAtar PQN full loop, NUM_ENVS={cfg['NUM_ENVS']} NUM_STEPS={cfg['NUM_STEPS']} "
                                   f"NUM_MINIBATCHES={cfg['NUM_MINIBATCHES']} NUM_EPOCHS={cfg['NUM_EPOCHS']} per GPU",
                       "seeds_per_gpu": 1, "backend": train.backend, "driver": driver_mode, "parallelism": f"{args.mode}x{world}",
                       "loop_tflops": sps * LOOP_FLOP / 1e12, "loop_frac_f32_peak": sps * LOOP_FLOP / 1e12 / F32_PEAK_TFLOPS},
            "roofline": roof,
        }
        out["config"]["loop_gbps_algorithmic"] = sps * 6.77e-6   # 6,770 B per env-step of the loop
        out["config"]["loop_frac_hbm_peak"] = sps * 6.77e-6 / HBM_PEAK_GBS
        # the extra objects below never take the headline line down with them: a failure is recorded in place
        def guarded(name, fn):
            try:
                out[name] = fn()
            except Exception as exc:  # noqa: BLE001
                out[name] = {"error": repr(exc)[:300]}
                torch.cuda.synchronize()

        if world == 1:
            from purejaxql_amd.profiling import env_step_hbm_roofline
            guarded("roofline_env_step", lambda: [env_step_hbm_roofline(n, dev) for n in (4096, 65536)])
        if world == 1 and args.multi_seed > 1 and fused:
            guarded("multi_seed", lambda: multi_seed_rate(cfg, args.multi_seed, args.steps, args.warmup, dev))
        if world == 1 and fused and args.matmul_dtype == "f32" and not args.no_mixed_precision:
            # opt-in operand precision (config MATMUL_DTYPE=f16): reported beside `value`, never as `value`
            def mixed():
                c16 = dict(cfg)
                c16["MATMUL_DTYPE"] = "f16"
                one = multi_seed_rate(c16, 1, args.steps, args.warmup, dev)
                many = multi_seed_rate(c16, args.multi_seed, args.steps, args.warmup, dev) if args.multi_seed > 1 else None
                return {
                    "dtype": "fc1 products (forward, input gradient, weight gradient; 78 % of the FLOPs) with fp16 operands and "
                             "f32 accumulation; master weights, optimizer, conv, LayerNorm, head, loss in f32",
                    "value": one["value"], "unit": "env-steps/s", "ms_per_step": one["ms_per_round"],
                    "multi_seed": None if many is None else {"seeds_per_gpu": many["seeds_per_gpu"], "value": many["value"]},
                    "returns": "10-seed Breakout / Asterix test returns equal to the f32 mode within seed noise "
                               "(profiles/r01_learning_curves.txt)"}
            guarded("mixed_precision", mixed)
        if not args.no_cpu_baseline and world == 1:
            guarded("cpu_baseline", lambda: cpu_baseline(cfg, finish()["runner_state"]["network"].init(1).cpu().numpy()))
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
