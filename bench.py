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
BF16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak
# Dominant kernel = qnet_cnn_train_kernel (T1: forward + backward except the fc1 weight gradient).
# Algorithmic FLOP per sample: fwd conv 73,728 + fc1 262,144 + fc2 768; bwd fc2 dgrad+wgrad
# 1,536 + fc1 dgrad 262,144 + conv wgrad 73,728 (the conv has no input gradient) = 674,048.
T1_FLOP_PER_SAMPLE = 674048.0


# position-parallel form (round 5): the backward kernel is the dominant one.  Algorithmic FLOP per sample of the backward
# (cnn_pos_bwd_kernel): fc1 dgrad 262,144 + fc1 wgrad 262,144 + conv wgrad 73,728 = 598,016 (its recomputed conv is not
# algorithmic work); of the forward (cnn_pos_fwd_kernel): conv 73,728 + fc1 262,144 + fc2 forward / dgrad / wgrad 2,304 = 338,176.
def pos_flop_per_sample(channels: int, actions: int):
    bwd = 2.0 * 262144.0 + 18432.0 * channels
    fwd = 18432.0 * channels + 262144.0 + 3.0 * 256.0 * actions
    return bwd, fwd


def t1_flop_per_sample(channels: int, actions: int) -> float:
    """The same count for any MinAtar game: conv 3x3xC -> 16 on 8x8 positions (forward + weight gradient), fc1 1024 -> 128
    (forward + input gradient), fc2 128 -> A (forward, input and weight gradient); Breakout (C = 4, A = 3): 674,048."""
    return 2.0 * 18432.0 * channels + 2.0 * 262144.0 + 3.0 * 256.0 * actions


def workload_config(num_envs: int, mode: str):
    from purejaxql_amd.config_loader import flatten, load_config
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", f"alg.NUM_ENVS={num_envs}",
                               "alg.TEST_DURING_TRAINING=False"]))
    return cfg


def cpu_baseline(cfg, theta0):
    """The CPU path beside the GPU number (kind = "port"): the oracle loop -- C/OpenMP env step + LogWrapper
    + eps-greedy + Q(lambda) + shuffle + optax-exact clip/RAdam -- with the Q-network on torch-CPU (oracle/pqn_cpu_torch.py;
    held to the numpy oracle loop by tests/test_oracle_cpu.py), at the bench's own shape (NUM_ENVS = 4096, one seed).
    Thread count: MORE torch threads are SLOWER on this workload (4096-sample minibatches of a 130k-parameter network:
    measured on the 256-core GPU box 1.6 s per update at 32 threads, 4.8 s at 64, 10 s at 128, > 120 s at 256), so the
    count is calibrated first -- one warm + one timed update at 16 / 32 / 64 threads -- and the best one then runs whole
    updates until >= 3 are timed (~10 s).  `cores` = the threads of that best setting.  The reference's JAX-CPU path
    cannot run here (no jax in the image): this is the same algorithm, not the same library."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import pqn_oracle as oracle
    import pqn_cpu_torch as cpu_loop
    ocfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
    t, n_envs = int(ocfg["NUM_STEPS"]), int(cfg["NUM_ENVS"])
    ocfg["TOTAL_TIMESTEPS"] = ocfg["TOTAL_TIMESTEPS_DECAY"] = 1e7
    ncpu = os.cpu_count() or 1
    try:   # every native pool of the process (torch's OpenMP, numpy's OpenBLAS, the C oracle's OpenMP) at the SAME count: in round 5
        from threadpoolctl import threadpool_info, threadpool_limits   # openblas sat at 64 threads beside torch's 16
    except Exception:  # noqa: BLE001
        threadpool_info = threadpool_limits = None

    def run(thr, **kw):
        if threadpool_limits is None:
            return cpu_loop.make_train(ocfg, threads=thr)(12345, theta0, **kw), "unknown"
        with threadpool_limits(limits=thr):
            out = cpu_loop.make_train(ocfg, threads=thr)(12345, theta0, **kw)
            desc = ", ".join(f"{p.get('internal_api')}:{p.get('num_threads')}" for p in threadpool_info())
        return out, desc

    tried = {}
    for thr in sorted({min(16, ncpu), min(32, ncpu), min(64, ncpu)}):
        out, _ = run(thr, max_updates=2, time_budget=8.0, min_updates=1)
        tried[thr] = out["seconds_per_update"][-1]              # the second (warm) update, or the only one if it was slow
    best = min(tried, key=tried.get)
    out, pool_desc = run(best, max_updates=7, time_budget=12.0, min_updates=4)
    secs = out["seconds_per_update"][1:]                    # the first update is the warm-up
    dt = float(sum(secs))
    # env-only rate (uniform-random actions, no network): the C oracle's OpenMP env.step + auto-reset + LogWrapper
    env = oracle.OracleEnv(ocfg["ENV_NAME"])
    _obs, st = env.reset(1, n_envs)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, env.num_actions, size=(50, n_envs)).astype(np.int32)
    env.step(2, st, acts[0])
    t1 = time.perf_counter()
    for i in range(50):
        _o, st, _r, _d, _info = env.step(100 + i, st, acts[i])
    env_only = 50 * n_envs / (time.perf_counter() - t1)
    return {"value": len(secs) * n_envs * t / dt, "unit": "env-steps/s", "cores": best, "kind": "port",
            "host_logical_cores": ncpu, "cores_used_of_present": f"{best} of {ncpu}", "thread_pools": pool_desc, "seconds_per_update_by_threads": {str(k): round(v, 3) for k, v in tried.items()},
            "env_only_env_steps_per_s": env_only, "updates_timed": len(secs), "seconds_per_update": [round(x, 3) for x in secs],
            "sample": f"{len(secs)} timed full PQN updates (rollout+Q(lambda)+{ocfg['NUM_EPOCHS']}x{ocfg['NUM_MINIBATCHES']} SGD steps) of ONE "
                      f"seed at NUM_ENVS={n_envs}, NUM_STEPS={t} (the bench shape): {len(secs) * n_envs * t} env-steps in {dt:.1f}s, after one "
                      "untimed update of the same shape; oracle/pqn_cpu_torch.py = C/OpenMP env + eps-greedy + Q(lambda) + RAdam of the "
                      f"oracle, Q-network forward/backward on torch-CPU at the best of 16/32/64 threads ({best}; more threads are "
                      f"slower: see seconds_per_update_by_threads) on a host with {ncpu} logical cores; not JAX; thread pools: {pool_desc}"}


def timed_updates(update, steps, warmup, first=0, barrier=None):
    """`warmup` untimed + `steps` timed updates between two stream syncs; returns seconds."""
    import torch
    for u in range(first, first + warmup):
        update(u)
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for u in range(first + warmup, first + warmup + steps):
        update(u)
    torch.cuda.synchronize()
    if barrier:
        barrier()
    return time.perf_counter() - t0


def kernel_timer_pass(lib, update, first, mb_samples, seeds, mode=1):
    """HIP-event timing of the dominant kernel on its launch stream.  Events recorded while a hipGraph is being
    captured cannot be read back on ROCm 7, so the timed region runs the graph and this pass re-runs 2 more
    updates of the same workload through the eager C++ enqueue with the kernel timer on (same kernels, shapes,
    buffers).  Returns (avg seconds per launch, launches)."""
    import ctypes
    import torch
    from purejaxql_amd import _lib
    drv = getattr(update, "driver", None)
    if drv is not None:
        drv.graph, drv.use_graph = None, False
        if hasattr(drv, "graphs"):
            drv.graphs = None
    _lib.check(lib.pqn_prof_enable(mode), "pqn_prof_enable")
    for u in range(first, first + 2):
        update(u)
    torch.cuda.synchronize()
    cnt, tot = ctypes.c_int32(0), ctypes.c_float(0.0)
    _lib.check(lib.pqn_prof_read(ctypes.byref(cnt), ctypes.byref(tot)), "pqn_prof_read")
    lib.pqn_prof_enable(0)
    if cnt.value == 0:
        raise SystemExit("kernel timer recorded nothing")
    return tot.value * 1e-3 / cnt.value, cnt.value


def t1_roofline(avg_s, launches, mb_samples, seeds, matmul, form, flop_per_sample=T1_FLOP_PER_SAMPLE, channels=4, actions=3):
    if matmul == "f16x2" and form != "pos":
        matmul = "bf16x3"     # an f16x2 layout runs bf16x3 in every kernel form but the position-parallel one: priced as what ran
    if form == "pos":     # the timed kernel is the backward of the position-parallel form
        flop_per_sample = pos_flop_per_sample(channels, actions)[0]
    achieved = flop_per_sample * mb_samples * seeds / avg_s / 1e12
    traffic, tsrc, l2cu = None, None, None
    for name in ((f"r06_pmc_pos_bwd_kernel_{matmul}_seeds{seeds}.json",) + ((f"r05_pmc_pos_bwd_kernel_{matmul}_seeds{seeds}.json",) if matmul != "f16x2" else ()) if form == "pos" else
                 (f"r04_pmc_train_kernel_{matmul}_seeds{seeds}.json", f"r03_pmc_train_kernel_{matmul}_seeds{seeds}.json",
                 f"r02_pmc_train_kernel_{matmul}_seeds{seeds}.json",
                 f"r02_pmc_train_kernel_{matmul}.json", "r01_pmc_train_kernel.json")):
        pmc = os.path.join(ROOT, "profiles", name)
        if os.path.exists(pmc):
            pj = json.load(open(pmc))
            if pj.get("matmul", "f32") == matmul and pj.get("hbm_bytes_per_launch"):
                ps = max(1, int(pj.get("seeds_per_launch", 1)))
                traffic = pj["hbm_bytes_per_launch"] * seeds / ps       # every seed of a launch moves the same bytes
                l2cu = pj.get("l2_to_cu_bytes_per_launch")
                l2cu = l2cu * seeds / ps if l2cu else None
                tsrc = (f"from file profiles/{name}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the "
                        f"{ps}-seed launch of this kernel, gfx950-corrected as MI355X_MICROARCH.md prescribes"
                        + (f", x{seeds // ps} for the {seeds} seeds of this launch" if seeds != ps else "")
                        + f"; NOT measured in this run (counter passes need rocprofv3 around the process): file written {pj.get('date', 'in the round its name says')}"
                        + (f" at commit {pj['commit']}" if pj.get('commit') else ""))
                break
    # which form ran is asked of the library (pqn_cnn_last_kernel_form), not re-derived here
    kname = {"pair": "qnet_cnn_train_pair_kernel<4>", "single": "qnet_cnn_train_kernel<4>"}.get(form, form)
    what = (f"cnn_pos_bwd_kernel (position-parallel form: fc1 input + weight gradient, LayerNorm_0 / conv backward of one "
            f"{mb_samples}-sample minibatch of each of {seeds} seed(s) per launch)") if form == "pos" else \
        f"{kname} (fwd + bwd of one {mb_samples}-sample minibatch of each of {seeds} seed(s) per launch)"
    out = {"kernel": what,
           "bound": "mfma", "achieved": achieved, "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": achieved / F32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": tsrc, "l2_to_cu_bytes": l2cu,
           "avg_launch_us": avg_s * 1e6, "launches_timed": launches,
           "flop_per_launch": flop_per_sample * mb_samples * seeds,
           "peak_note": "f32 MFMA / vector peak of MI355X (157.3 TFLOP/s); algorithmic f32 FLOPs of the kernel"}
    if matmul == "bf16x3":
        # The f32 products of this mode run on the bf16 matrix pipe as 6 bf16 MFMA products each (exact 3-way operand split, f32
        # accumulate).  Since round 5 the kernels are past the f32 MFMA peak (the fraction against it exceeds 1), so the roofline is
        # priced against the pipe the work actually runs on: peak = dense bf16 peak / 6 = the f32-accurate product rate that pipe
        # can deliver; `achieved` stays the ALGORITHMIC f32 FLOP rate.  frac_f32_mfma_peak keeps the basis of rounds 1-4.
        x3_peak = BF16_PEAK_TFLOPS / 6.0
        out.update({"peak": x3_peak, "frac": achieved / x3_peak, "frac_f32_mfma_peak": achieved / F32_PEAK_TFLOPS,
                    "peak_note": "dense bf16 MFMA peak of MI355X (2500 TFLOP/s) / 6 bf16 products per f32 product of the exact 3-way "
                                 "split = 416.7 TFLOP/s of f32-accurate products; `achieved` = algorithmic f32 FLOPs of the kernel per second.  "
                                 "frac_f32_mfma_peak = the same rate against the f32 MFMA peak (157.3 TFLOP/s), the basis of the round 1-4 "
                                 "lines (round 4: 0.63)"})
        out["bf16_pipe"] = {"issued_tflops": achieved * 6.0, "peak": BF16_PEAK_TFLOPS, "frac": achieved * 6.0 / BF16_PEAK_TFLOPS,
                            "note": "bf16 MFMA FLOPs issued (6 per algorithmic f32 FLOP; an upper bound -- the conv weight gradient, 12 % of "
                                    "the backward's FLOPs, needs 3: its bit operand is exact in bf16) against the dense bf16 peak.  The "
                                    "position-parallel kernels keep the matrix pipe 41-47 % busy; the rest is instruction issue on the "
                                    "SIMDs (4.1 VALU instructions per MFMA in the backward: operand split, LayerNorm_0 backward, bit "
                                    "expansion) and the latency of a two-wave-per-SIMD schedule (DESIGN.md section 3.6)"}
    if matmul == "f16x2":
        h2_peak = BF16_PEAK_TFLOPS / 3.0     # the fp16 MFMA peak equals the bf16 one
        out.update({"peak": h2_peak, "frac": achieved / h2_peak, "frac_f32_mfma_peak": achieved / F32_PEAK_TFLOPS,
                    "peak_note": "dense fp16 MFMA peak of MI355X (2500 TFLOP/s) / 3 fp16 products per f32 product of the two-way split "
                                 "= 833.3 TFLOP/s of 22-bit products; `achieved` = algorithmic f32 FLOPs of the kernel per second.  "
                                 "frac_f32_mfma_peak = the same rate against the f32 MFMA peak (157.3 TFLOP/s)"})
        out["fp16_pipe"] = {"issued_tflops": achieved * 3.0, "peak": BF16_PEAK_TFLOPS, "frac": achieved * 3.0 / BF16_PEAK_TFLOPS,
                            "note": "fp16 MFMA FLOPs issued (3 per algorithmic f32 FLOP; the conv weight gradient stays on 3 bf16 products) "
                                    "against the dense peak"}
    return out


DTYPE_LABEL = {"f32": "f32",
               "bf16x3": "f32 (bf16x3 split operands, 6 bf16 MFMA products per f32 product, f32 accumulate)",
               "f16x2": "f32 (f16x2 split operands: two range-scaled fp16 pieces = 22 significand bits, 3 fp16 MFMA products per f32 "
                        "product, f32 accumulate; bf16x3 outside the position-parallel kernels)",
               "f16": "f16 operands / f32 accumulate (fc1), f32 elsewhere"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)    # 16 seeds x 131,072 env-steps per update: 2.1e8 env-steps timed
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--seeds-per-gpu", type=int, default=16,
                    help="independent seeds of the workload per GPU, batched into the same launches (jax.vmap over seeds, "
                         "pqn_minatar.py:459-461).  16 = BASELINE.json configs[3]: 128 seeds x 4096 envs over 8 GPUs")
    ap.add_argument("--mode", default="seeds", choices=["seeds", "envs"],
                    help="multi-GPU sharding: independent seeds per rank (no collective) or envs of one seed "
                         "(RCCL gradient all-reduce per optimizer step)")
    ap.add_argument("--seed-groups", type=int, default=0,
                    help="seed groups per GPU (0 / 1 = all seeds in one chain of launches, the default).  With G > 1 the "
                         "HBM-bound optimizer tail of one group runs on a second stream under the training kernel of the next "
                         "(pqn_cnn_update_seed_groups): measured slower, profiles/r04_v0_seed_groups_ab.txt")
    ap.add_argument("--groups-tail", default="graph",
                    help="graph (default: one two-branch hipGraph per update) | eager (C++ enqueue, high-priority tail stream) "
                         "| masked:<lo>:<hi> (eager, tail stream on CUs [lo, hi), training kernels on the rest)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only")
    ap.add_argument("--matmul-dtype", default="f16x2", choices=["f32", "bf16x3", "f16x2", "f16"],
                    help="operand mode of the fc1 / conv products (config MATMUL_DTYPE).  f16x2 (default here; what the package default "
                         "`auto` resolves to from 512-sample minibatches on) carries every f32 operand of the position-parallel kernels as two "
                         "range-scaled fp16 pieces (22 significand bits, 3 matrix instructions per product, f32 accumulate) and is bf16x3 -- "
                         "three bf16 pieces, 6 instructions -- in every other kernel form; both are held to the same f32 tolerances as the "
                         "f32-MFMA mode by the parity tests and measured against float64 (tests/test_qnet_gpu.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher -- one rank per GPU under torch.distributed.run (the
        # same command the driver issues for N > 1), rendezvous on 127.0.0.1, rank 0's JSON line passed through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: purejaxql_amd has no CPU path")
    from purejaxql_amd import _lib
    from purejaxql_amd import dist as pdist
    # PQN_BENCH_ONE_GPU=1 (tests on a 1-GPU box only): every rank on device 0, gloo instead of RCCL -- exercises the
    # multi-rank control flow (barriers, max-over-ranks timing, rank-0 line), not the scaling
    if os.environ.get("PQN_BENCH_ONE_GPU", "0") == "1":
        os.environ.setdefault("PQN_DIST_BACKEND", "gloo")
    if int(os.environ.get("WORLD_SIZE", "1")) > torch.cuda.device_count() and os.environ.get("PQN_BENCH_ONE_GPU", "0") != "1":
        raise SystemExit(f"bench.py: {os.environ['WORLD_SIZE']} ranks but {torch.cuda.device_count()} visible GPU(s) "
                         "(PQN_BENCH_ONE_GPU=1 puts every rank on device 0 over gloo -- control-flow test only)")
    rank, world, local_rank = pdist.init_from_env()
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    dev = torch.device("cuda", torch.cuda.current_device())
    lib = _lib.load()
    from purejaxql_amd.pqn import make_train, seed_keys

    cfg = workload_config(args.num_envs, args.mode)
    # The CPU-oracle leg runs FIRST (rank 0, N = 1 only): ~30 s of host work, after which every GPU leg follows back to
    # back -- an external GPU-busy sampler then sees the GPU work at the END of the run instead of missing its first 3 s.
    cpu_base = None
    if world == 1 and not args.no_extras and not args.no_cpu_baseline:
        from purejaxql_amd.networks import QNetwork
        try:
            cpu_base = cpu_baseline(cfg, QNetwork("cnn", (10, 10, 4), 3, device=dev).init(1).cpu().numpy())
        except Exception as exc:  # noqa: BLE001
            cpu_base = {"error": repr(exc)[:300]}
    if args.matmul_dtype:
        cfg["MATMUL_DTYPE"] = args.matmul_dtype
    matmul = str(cfg.get("MATMUL_DTYPE", "f32")).lower()
    cfg["SEED_GROUPS"] = args.seed_groups
    cfg["_SEED_GROUPS_TAIL"] = args.groups_tail
    SUSTAIN_MAX = 400
    n_total = args.steps + args.warmup + 10 + SUSTAIN_MAX
    cfg["TOTAL_TIMESTEPS"] = n_total * cfg["NUM_ENVS"] * cfg["NUM_STEPS"]
    barrier = dist.barrier if world > 1 else None
    spg = max(1, args.seeds_per_gpu)
    if args.mode == "envs":
        # the envs of ONE seed split over the ranks, gradient all-reduce per optimizer step
        spg = 1
        scfg = pdist.shard_env_config(cfg, rank, world) if world > 1 else dict(cfg)
        ghook = pdist.make_grad_allreduce_hook() if world > 1 else None   # one-shot peer all-reduce when hipIpc maps the ranks
        train = make_train(scfg, device=str(dev), grad_hook=ghook,
                           metrics_hook=pdist.allreduce_mean_scalars if world > 1 else None)
        update, finish = train.make_runner(seed_keys(0, 1)[0])
        env_steps_per_update = cfg["NUM_ENVS"] * cfg["NUM_STEPS"]          # the global env count, all ranks together
        scaling = "strong"
    else:
        ghook = None
        # seeds are independent runs: rank r trains seeds [r*spg, (r+1)*spg) of seed_keys(0, world*spg), no collective
        train = make_train(dict(cfg), device=str(dev))
        keys = seed_keys(0, world * spg)[rank * spg:(rank + 1) * spg]
        if spg > 1:
            update, finish = train.make_batch_runner(keys)
        else:
            update, finish = train.make_runner(keys[0])
        env_steps_per_update = cfg["NUM_ENVS"] * cfg["NUM_STEPS"] * spg * world
        scaling = "weak"
    fused = train.backend == "fused"
    # one tiny collective on a DEVICE tensor before the timed region: proof that the process group (RCCL under the nccl
    # backend) really spans `world` ranks -- in seeds mode the data path itself has no collective
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.float32, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(round(float(one.item())))
    dt = timed_updates(update, args.steps, args.warmup, 0, barrier)
    per_rank_ms = None
    if world > 1:
        # every rank's own time for the same timed region (value uses the MAX): a slow GPU of the node shows up as spread
        cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")
        mine = torch.tensor([dt], dtype=torch.float64, device=cdev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in every]
        dt = max(float(x.item()) for x in every)
    sps = env_steps_per_update * args.steps / dt

    # A second, longer timed region of the SAME runner (>= 6 s of back-to-back graph replays, N = 1 only): the rate once
    # clocks and caches have settled, and long enough for an external 5-s GPU-busy sampler to see the run.
    sustained, n_done = None, args.warmup + args.steps
    if world == 1 and not args.no_extras and fused:
        n_s = min(SUSTAIN_MAX, max(args.steps, int(6.0 / (dt / args.steps))))
        ds = timed_updates(update, n_s, 0, n_done)
        n_done += n_s
        sustained = {"value": env_steps_per_update * n_s / ds, "unit": "env-steps/s", "steps": n_s, "seconds": ds,
                     "note": "same runner and graph, continued right behind the headline region"}

    # The kernel-timer pass runs 2 more updates.  With the envs of one seed sharded over ranks those updates
    # contain collectives, so EVERY rank has to run them (rank 0 alone would wait forever for its peers).
    roof = None
    mb = train.config["NUM_ENVS"] * cfg["NUM_STEPS"] // cfg["NUM_MINIBATCHES"]   # minibatch per rank and seed
    drv0 = getattr(update, "driver", None)
    driver_mode = None if drv0 is None else ("hipGraph replay" if drv0.graph is not None else "C++ enqueue (eager)")
    groups = len(getattr(drv0, "drivers", [None]))          # seed groups of the headline runner (SeedGroupsDriver)
    spl = spg // groups                                      # seeds per training-kernel launch
    if fused and (rank == 0 or (world > 1 and args.mode == "envs")):
        avg_s, launches = kernel_timer_pass(lib, update, n_done, mb, spl)
        form0 = _lib.last_kernel_form()[0]
        roof = t1_roofline(avg_s, launches, mb, spl, matmul, form0)
        if form0 == "pos":   # the other two launches of the training step, timed the same way (2 more updates each)
            f_s, _ = kernel_timer_pass(lib, update, n_done + 2, mb, spl, mode=3)
            a_s, _ = kernel_timer_pass(lib, update, n_done + 4, mb, spl, mode=4)
            bwd_f, fwd_f = pos_flop_per_sample(4, 3)
            # flat scalars (a line parser that keeps scalars only still shows the whole training step)
            # round 6: the gather runs once per epoch (one launch for all minibatches); its share per optimizer step is added to the
            # forward + backward time the per-step timer sees
            try:
                g_s, g_n = kernel_timer_pass(lib, update, n_done + 6, mb, spl, mode=5)
                a_s += g_s / cfg["NUM_MINIBATCHES"]
                roof["epoch_gather_us"] = g_s * 1e6
            except SystemExit:      # a caller-sized workspace without the epoch region: the gather ran inside the timed step
                roof["epoch_gather_us"] = None
            roof["forward_kernel_us"] = f_s * 1e6
            roof["gather_forward_backward_us"] = a_s * 1e6
            roof["value_and_grad_frac"] = (bwd_f + fwd_f) * mb * spl / a_s / 1e12 / roof["peak"]
            roof["training_step"] = {
                "forward_kernel_us": f_s * 1e6, "forward_frac": fwd_f * mb * spl / f_s / 1e12 / roof["peak"],
                "forward_frac_f32_peak": fwd_f * mb * spl / f_s / 1e12 / F32_PEAK_TFLOPS,
                "gather_forward_backward_us": a_s * 1e6,
                "value_and_grad_frac": (bwd_f + fwd_f) * mb * spl / a_s / 1e12 / roof["peak"],
                "value_and_grad_frac_f32_peak": (bwd_f + fwd_f) * mb * spl / a_s / 1e12 / F32_PEAK_TFLOPS,
                "note": "cnn_pos_fwd_kernel alone, and pos_gather (once per epoch: 1 / NUM_MINIBATCHES of its launch) + cnn_pos_fwd + cnn_pos_bwd together = the whole "
                        "value_and_grad(_loss_fn) of an optimizer step incl. the fc1 weight gradient (936,192 algorithmic "
                        "FLOP per sample for Breakout); HIP events on the launch stream, 2 eager updates each; *_frac against "
                        "roofline.peak, *_frac_f32_peak against the f32 MFMA peak (the basis of rounds 1-4)"}
        if groups > 1:
            roof["note"] = (f"{groups} seed groups of {spl} seeds: each timed launch covers one group and runs while the "
                            "previous group's fc1 weight gradient / fold / RAdam kernels share the GPU on a second stream")

    if rank == 0:
        if roof is None:
            from purejaxql_amd.profiling import time_env_step_kernel
            k_ms = time_env_step_kernel(cfg["NUM_ENVS"], dev)
            achieved = ENV_STEP_BYTES * cfg["NUM_ENVS"] / (k_ms * 1e-3) / 1e9
            roof = {"kernel": "minatar_kernel<Breakout> (env.step, f32 obs surface)", "bound": "hbm",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None, "avg_launch_us": k_ms * 1e3}
        out = {
            "metric": "env-steps/sec (whole node), MinAtar-Breakout 4096 envs"
                      + (f" [{spg} seeds/GPU batched, {matmul} operands]" if args.mode == "seeds" else f" [1 seed, envs sharded, {matmul} operands]"),
            "value": sps, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": DTYPE_LABEL.get(matmul, matmul), "data": "synthetic",
            "config": {"workload": f"Breakout-MinAtar PQN full loop, NUM_ENVS={cfg['NUM_ENVS']} NUM_STEPS={cfg['NUM_STEPS']} "
                                   f"NUM_MINIBATCHES={cfg['NUM_MINIBATCHES']} NUM_EPOCHS={cfg['NUM_EPOCHS']}, "
                                   + (f"{spg} independent seed(s) per GPU batched into the launches "
                                      f"(BASELINE.json configs[3] = 128 seeds x 4096 envs over 8 GPUs is 16 per GPU)"
                                      if args.mode == "seeds" else f"ONE seed, its envs sharded over {world} rank(s)"),
                       "seeds_per_gpu": spg, "seeds_total": spg * world if args.mode == "seeds" else 1,
                       "seed_groups": groups, "seed_groups_tail": args.groups_tail if groups > 1 else None,
                       "env_steps_per_step": env_steps_per_update, "matmul_dtype": matmul,
                       "backend": train.backend, "driver": driver_mode, "parallelism": f"{args.mode}x{world}",
                       "kernel_forms": dict(zip(("train", "rollout"), _lib.last_kernel_form())),
                       "rccl_ranks": (ranks_seen if dist.get_backend() == "nccl" else 0) if world > 1 else 1,
                       "ranks_in_allreduce": ranks_seen,
                       "dist_backend": dist.get_backend() if world > 1 else None,
                       "grad_allreduce": (getattr(ghook, "mode", None) if args.mode == "envs" and world > 1 else None),
                       "gpus_visible": torch.cuda.device_count(),
                       "per_rank_ms_per_step": per_rank_ms,
                       "per_rank_spread": (None if not per_rank_ms else (max(per_rank_ms) - min(per_rank_ms)) / max(per_rank_ms)),
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

        extras = world == 1 and not args.no_extras
        if extras and fused and spg > 1:
            def single_seed():
                c1 = dict(cfg)
                tr1 = make_train(c1, device=str(dev))
                upd1, _fin1 = tr1.make_runner(seed_keys(0, 1)[0])
                d1, n1 = timed_updates(upd1, args.steps, args.warmup), args.steps
                done1 = args.warmup + args.steps
                if d1 < 0.2:   # a short region: time ~0.3 s more and keep the faster of the two (a host hiccup only ever adds)
                    n2 = min(SUSTAIN_MAX // 2, max(n1, int(0.3 / (d1 / n1))))
                    d2 = timed_updates(upd1, n2, 0, done1)
                    done1 += n2
                    if d2 / n2 < d1 / n1:
                        d1, n1 = d2, n2
                a1, l1 = kernel_timer_pass(lib, upd1, done1, mb, 1)
                v1 = cfg["NUM_ENVS"] * cfg["NUM_STEPS"] * n1 / d1
                return {"seeds_per_gpu": 1, "value": v1, "unit": "env-steps/s", "ms_per_step": d1 / n1 * 1e3, "updates_timed": n1,
                        "loop_frac_f32_peak": v1 * LOOP_FLOP / 1e12 / F32_PEAK_TFLOPS,
                        "roofline": t1_roofline(a1, l1, mb, 1, matmul, _lib.last_kernel_form()[0]),
                        "note": "ONE seed of 4096 envs alone on the GPU (round 1's headline configuration)"}
            guarded("single_seed", single_seed)
        if extras and fused:
            def minatar_suite():
                """BASELINE.json configs[2] (the MinAtar suite at 4096 envs: gymnax 0.0.6 has four games) and configs[1]
                (Breakout at 1024 envs: with 16 seeds per GPU and as ONE seed alone), each as the headline workload: seeds batched into the launches, the
                yaml's 32 steps x 32 minibatches x 2 epochs, same operand mode.  Per line: whole-loop env-steps/s, the kernel
                forms the library took, the training kernel's HIP-event duration and its fraction of the f32 peak."""
                from purejaxql_amd.config_loader import flatten, load_config
                from purejaxql_amd.envs import make
                res = []
                for env_name, n_envs, spg_g in (("Asterix-MinAtar", 4096, spg), ("Freeway-MinAtar", 4096, spg), ("SpaceInvaders-MinAtar", 4096, spg),
                                                ("Breakout-MinAtar", 4096, spg), ("Breakout-MinAtar", 1024, spg), ("Breakout-MinAtar", 1024, 1)):
                    try:
                        cg = flatten(load_config(["+alg=pqn_minatar", f"alg.ENV_NAME={env_name}", f"alg.NUM_ENVS={n_envs}",
                                                  "alg.TEST_DURING_TRAINING=False"]))
                        cg["MATMUL_DTYPE"] = matmul
                        w_g, s_g, s_more = 3, 12, 120
                        cg["TOTAL_TIMESTEPS"] = (w_g + s_g + s_more + 3) * n_envs * cg["NUM_STEPS"]
                        trg = make_train(cg, device=str(dev))
                        updg, _fg = trg.make_batch_runner(seed_keys(0, spg_g)) if spg_g > 1 else trg.make_runner(seed_keys(0, 1)[0])
                        dg = timed_updates(updg, s_g, w_g)
                        n_done_g = w_g + s_g
                        if dg < 0.2:   # a short region (12 updates of a small launch shape are ~45 ms): one host hiccup would be a third of it --
                            n2 = min(s_more, max(s_g, int(0.3 / (dg / s_g))))   # time ~0.3 s more and keep the faster of the two regions
                            d2 = timed_updates(updg, n2, 0, n_done_g)
                            n_done_g += n2
                            if d2 / n2 < dg / s_g:
                                dg, s_g = d2, n2
                        forms = dict(zip(("train", "rollout"), _lib.last_kernel_form()))
                        mbg = n_envs * cg["NUM_STEPS"] // cg["NUM_MINIBATCHES"]
                        ag, lg = kernel_timer_pass(lib, updg, n_done_g, mbg, spg_g)
                        env_g, _pg = make(env_name, device=dev)
                        ch, na = int(env_g.obs_shape[-1]), int(env_g.num_actions)
                        rg = t1_roofline(ag, lg, mbg, spg_g, matmul, forms["train"], t1_flop_per_sample(ch, na), ch, na)
                        res.append({"env": env_name, "num_envs": n_envs, "seeds_per_gpu": spg_g, "channels": ch, "actions": na,
                                    "value": n_envs * cg["NUM_STEPS"] * spg_g * s_g / dg, "unit": "env-steps/s",
                                    "ms_per_update": dg / s_g * 1e3, "kernel_forms": forms,
                                    "t1_avg_launch_us": rg["avg_launch_us"], "t1_frac": rg["frac"],
                                    "t1_frac_f32_peak": rg.get("frac_f32_mfma_peak", rg["frac"]),
                                    "t1_flop_per_sample": rg["flop_per_launch"] / (mbg * spg_g), "t1_kernel": rg["kernel"].split(" ")[0],
                                    "minibatch": mbg})
                        del updg, trg
                    except Exception as exc:  # noqa: BLE001
                        res.append({"env": env_name, "num_envs": n_envs, "error": repr(exc)[:300]})
                        torch.cuda.synchronize()
                return res
            guarded("minatar_suite", minatar_suite)
        if extras:
            from purejaxql_amd.profiling import env_step_hbm_roofline
            guarded("roofline_env_step", lambda: [env_step_hbm_roofline(n, dev) for n in (4096, 65536, 262144)])
        if extras:
            def craftax_c5():
                """BASELINE.json configs[4]: Craftax-Classic at the yaml shape, whole loop through pqn_bigmlp_update."""
                import ctypes
                from purejaxql_amd.config_loader import flatten, load_config
                c5 = flatten(load_config(["+alg=pqn_craftax", "alg.ENV_NAME=Craftax-Classic-Symbolic-v1"]))
                n5, warm5, steps5 = c5["NUM_ENVS"], 40, 400
                c5["TOTAL_TIMESTEPS"] = (warm5 + steps5 + 8) * n5 * c5["NUM_STEPS"]
                c5["TOTAL_TIMESTEPS_DECAY"] = c5["TOTAL_TIMESTEPS"]
                c5["TEST_DURING_TRAINING"] = False
                tr5 = make_train(c5, device=str(dev), script="craftax")
                upd5, _fin5 = tr5.make_runner(seed_keys(0, 1)[0])
                d5 = timed_updates(upd5, steps5, warm5)
                drv5 = getattr(upd5, "driver", None)
                mode5 = None if drv5 is None else ("hipGraph replay" if drv5.graph is not None else "C++ enqueue (eager)")
                # the GEMM kernel under HIP events: 4 more updates through the eager enqueue (events cannot be captured)
                if drv5 is not None:
                    drv5.graph, drv5.use_graph = None, False
                _lib.check(lib.pqn_prof_enable(2), "pqn_prof_enable")
                for u in range(warm5 + steps5, warm5 + steps5 + 4):
                    upd5(u)
                torch.cuda.synchronize()
                cnt, tot = ctypes.c_int32(0), ctypes.c_float(0.0)
                _lib.check(lib.pqn_prof_read(ctypes.byref(cnt), ctypes.byref(tot)), "pqn_prof_read")
                lib.pqn_prof_enable(0)
                d_in, h5, l5, a5 = 1345, c5["HIDDEN_SIZE"], c5["NUM_LAYERS"], 17
                per_row = 2.0 * (d_in * h5 + (l5 - 1) * h5 * h5 + h5 * a5)            # forward = weight gradient FLOP per row
                # input gradients: below the first layer only when the input normalisation trains (its scale / bias gradient)
                dgrad_row = per_row if c5.get("NORM_INPUT") else 2.0 * ((l5 - 1) * h5 * h5 + h5 * a5)
                nb5 = n5 * c5["NUM_STEPS"] // c5["NUM_MINIBATCHES"]
                flop_upd = per_row * n5 * c5["NUM_STEPS"] + c5["NUM_MINIBATCHES"] * c5["NUM_EPOCHS"] * (
                    per_row * 2 * nb5 + dgrad_row * nb5 + per_row * nb5)
                gemm_s = tot.value * 1e-3 / 4
                ach = flop_upd / gemm_s / 1e12
                return {"workload": f"Craftax-Classic-Symbolic-v1 PQN at config/alg/pqn_craftax.yaml's shape: {n5} envs, "
                                    f"{c5['NUM_STEPS']} step x {c5['NUM_MINIBATCHES']} minibatch x {c5['NUM_EPOCHS']} epoch, 1345 -> "
                                    f"{l5} x {h5} -> 17 LayerNorm MLP with BatchRenorm input, 1-step loss, optimistic resets",
                        "value": n5 * c5["NUM_STEPS"] * steps5 / d5, "unit": "env-steps/s", "ms_per_update": d5 / steps5 * 1e3,
                        "backend": tr5.backend, "driver": mode5, "updates_timed": steps5,
                        "roofline": {"kernel": "bm_gemm_kernel<128|64, 64> (every Dense product of the update: forward of the "
                                               "rollout batch and of concat(obs, next_obs), input and weight gradients)",
                                     "bound": "mfma", "achieved": ach, "peak": BF16_PEAK_TFLOPS / 6.0, "unit": "TFLOP/s",
                                     "frac": ach / (BF16_PEAK_TFLOPS / 6.0), "frac_f32_mfma_peak": ach / F32_PEAK_TFLOPS, "traffic": None,
                                     "gemm_share_of_update": gemm_s / (d5 / steps5),
                                     "flop_per_update": flop_upd, "gemm_us_per_update": gemm_s * 1e6,
                                     "gemm_launches_per_update": cnt.value / 4,
                                     "bf16_pipe": {"issued_tflops": ach * 6.0, "peak": BF16_PEAK_TFLOPS,
                                                   "frac": ach * 6.0 / BF16_PEAK_TFLOPS},
                                     "peak_note": "the headline's basis: algorithmic f32 FLOPs (2 M N K on the unpadded shapes) against "
                                                  "dense bf16 peak / 6 products per f32 product = 416.7 TFLOP/s (round 5 printed this "
                                                  "line against the f32 MFMA peak: that fraction is kept as frac_f32_mfma_peak)"}}
            guarded("craftax_c5", craftax_c5)

            def yaml_default():
                """The reference's own default (config/alg/pqn_minatar.yaml: 128 envs x 32 steps, 32 x 2 minibatches of 128
                samples, f32 operand mode): the small-minibatch regime, K-split training kernels."""
                from purejaxql_amd.config_loader import flatten, load_config
                cd = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar"]))
                cd["TEST_DURING_TRAINING"] = False
                warm_d, steps_d = 30, 300
                cd["TOTAL_TIMESTEPS_DECAY"] = cd["TOTAL_TIMESTEPS"]
                trd = make_train(cd, device=str(dev))
                updd, _find = trd.make_runner(seed_keys(0, 1)[0])
                dd = timed_updates(updd, steps_d, warm_d)
                per_upd = cd["NUM_ENVS"] * cd["NUM_STEPS"]
                return {"workload": f"Breakout-MinAtar PQN at the yaml defaults: NUM_ENVS={cd['NUM_ENVS']} NUM_STEPS={cd['NUM_STEPS']} "
                                    f"NUM_MINIBATCHES={cd['NUM_MINIBATCHES']} NUM_EPOCHS={cd['NUM_EPOCHS']}, one seed, MATMUL_DTYPE=auto -> f32 operands",
                        "value": per_upd * steps_d / dd, "unit": "env-steps/s", "ms_per_update": dd / steps_d * 1e3,
                        "updates_timed": steps_d, "kernel_forms": dict(zip(("train", "rollout"), _lib.last_kernel_form())),
                        "seconds_for_1e7_steps": 1e7 / (per_upd * steps_d / dd)}
            guarded("yaml_default", yaml_default)
        if extras and fused:
            def other_modes():
                res = {}
                for md in ("f32", "bf16x3", "f16x2", "f16"):
                    if md == matmul:
                        continue
                    c2 = dict(cfg)
                    c2["MATMUL_DTYPE"] = md
                    tr2 = make_train(c2, device=str(dev))
                    k2 = seed_keys(0, spg)
                    upd2, _f2 = tr2.make_batch_runner(k2) if spg > 1 else tr2.make_runner(k2[0])
                    d2 = timed_updates(upd2, max(10, args.steps // 4), max(3, args.warmup // 2))
                    res[md] = {"dtype": DTYPE_LABEL[md], "seeds_per_gpu": spg,
                               "value": cfg["NUM_ENVS"] * cfg["NUM_STEPS"] * spg * max(10, args.steps // 4) / d2,
                               "unit": "env-steps/s"}
                res["note"] = ("the same workload under the other operand modes of the fc1 products; f16 = fp16 operands "
                               "(narrower than the reference's f32: reported, never the headline)")
                return res
            guarded("matmul_modes", other_modes)
            if matmul == "f16x2" and isinstance(out.get("matmul_modes", {}).get("bf16x3"), dict):
                # the round-5 operand mode beside the headline, at the top level: same workload, same run
                out["value_bf16x3_operands"] = out["matmul_modes"]["bf16x3"]["value"]
                out["operand_mode_note"] = ("value: MATMUL_DTYPE f16x2 = what the package default `auto` runs at this shape (f32 operands as two range-scaled "
                                            "fp16 pieces in the position-parallel kernels, f32 accumulate; measured NEARER to float64 than the f32 fma chain "
                                            "per product and than bf16x3 per gradient: profiles/r06_v7_f16x2_accuracy.txt); value_bf16x3_operands: the "
                                            "same workload with the three-piece bf16 operands of round 5.  The reference's own f32 conv / dot run as TF32 (10 significand "
                                            "bits per operand) under XLA's default precision on the NVIDIA GPUs its numbers come from (SURVEY A.8)")
        if sustained is not None:
            out["sustained"] = sustained
        if cpu_base is not None:
            out["cpu_baseline"] = cpu_base
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        if ghook is not None:
            ghook.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
