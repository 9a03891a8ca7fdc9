"""single_run(config): the launcher half of the reference scripts (purejaxql/pqn_minatar.py:435-483,
534-545) on top of make_train: seeds = independent runs (jax.vmap(make_train) at :459-461), wall-clock
line (:462), per-seed .safetensors + resolved config yaml (:464-483).  wandb is not available offline:
WANDB_MODE other than "disabled" logs metrics as JSON lines to stdout instead, and tune() (HYP_TUNE, :484-531) runs the
sweep's search space as a grid.

Multi-GPU (one process per GPU, `python -m torch.distributed.run --nproc-per-node G -m purejaxql_amd.pqn_minatar ...`):
the reference's seed axis is sharded over the ranks -- rank r trains partition_seeds(NUM_SEEDS, G, r), batched
into its launches, with NO data-path collective; the [S, NUM_UPDATES] metrics are gathered on every rank and
each rank writes the checkpoints of its own seeds under their GLOBAL vmap index.  `SHARD: envs` (this build only)
instead splits the NUM_ENVS of every seed over the ranks with a gradient all-reduce per optimizer step."""
from __future__ import annotations

import json
import os
import time
from typing import Any, Callable, Dict, List, Optional

import torch
import yaml

from . import dist as pdist
from .config_loader import flatten, load_config
from .save_load import save_params


def single_run(config: Dict[str, Any], device: Optional[str] = None, make_train_fn: Optional[Callable] = None,
               vmap_fn: Optional[Callable] = None, seed_keys_fn: Optional[Callable] = None, script: str = "gymnax") -> Dict[str, Any]:
    """make_train_fn / vmap_fn / seed_keys_fn default to the product's (purejaxql_amd.pqn); the CPU tests of the
    multi-rank control flow inject stand-ins (no GPU there)."""
    if make_train_fn is None or vmap_fn is None or seed_keys_fn is None:
        from .pqn import make_train, seed_keys, vmap_train
        if make_train_fn is None and script != "gymnax":   # pqn_craftax.py's make_train
            from functools import partial
            make_train_fn = partial(make_train, script=script)
        make_train_fn, vmap_fn, seed_keys_fn = make_train_fn or make_train, vmap_fn or vmap_train, seed_keys_fn or seed_keys
    config = flatten(config)                       # {**config, **config["alg"]} (:437)
    alg_name = config.get("ALG_NAME", "pqn")
    env_name = config["ENV_NAME"]
    rank, world, local_rank = pdist.init_from_env()
    if device is None:
        device = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cuda"
    if config.get("WANDB_MODE", "disabled") != "disabled":
        def cb(u, m):
            print(json.dumps({k: (float(v) if torch.is_tensor(v) else v) for k, v in m.items()}), flush=True)
        config["_CALLBACK"] = cb
    num_seeds = int(config["NUM_SEEDS"])
    keys = seed_keys_fn(config["SEED"], num_seeds)      # split(PRNGKey(SEED), NUM_SEEDS) (:456-459)
    shard_mode = str(config.get("SHARD", "seeds")).lower()
    t0 = time.time()
    if world > 1 and shard_mode == "envs":
        # every rank runs ALL seeds on its share of the envs; gradients are averaged over ranks per optimizer step
        cfg = pdist.shard_env_config(config, rank, world)
        ghook = pdist.make_grad_allreduce_hook()
        train = make_train_fn(cfg, device=device, grad_hook=ghook, metrics_hook=pdist.allreduce_mean_scalars)
        mine = list(range(num_seeds))
        try:
            outs = vmap_fn(train, keys, concurrent=False)
        finally:
            ghook.close()      # unmap the peers' staging regions of the in-graph all-reduce, free our own
        metrics = outs["metrics"]
        saver = rank == 0
    else:
        mine = pdist.partition_seeds(num_seeds, world, rank) if world > 1 else list(range(num_seeds))
        train = make_train_fn(dict(config), device=device)
        outs = vmap_fn(train, [keys[i] for i in mine]) if mine else {"runner_state": [], "metrics": {}}
        metrics = pdist.gather_seed_metrics(outs["metrics"]) if world > 1 else outs["metrics"]
        saver = True
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        print(f"Took {time.time() - t0} seconds to complete.")      # (:462)
        rs0 = outs["runner_state"][0] if outs["runner_state"] else None
        forms = rs0.get("kernel_forms") if rs0 is not None and hasattr(rs0, "get") else None
        if forms:
            # which form of the training / rollout kernels these launches took.  A seed batch normally takes the kernels its
            # seeds would take alone and is bit-identical to the solo runs; the one exception (f32 operands, minibatches <=
            # 256 samples, more than t1_ksplit_tiles tiles x seeds per launch: "single" here, "ksplit" alone) agrees with
            # the solo runs to f32 summation order only -- SEED_BATCH_BIT_IDENTICAL=True pins the solo form
            print(f"kernel forms: training={forms.get('train')} rollout={forms.get('rollout')} "
                  f"(seed batch of {rs0.get('seed_batch', 1)}, SEED_BATCH_BIT_IDENTICAL={bool(config.get('SEED_BATCH_BIT_IDENTICAL', False))})")
    if config.get("SAVE_PATH", None) is not None:
        save_dir = os.path.join(config["SAVE_PATH"], env_name)
        os.makedirs(save_dir, exist_ok=True)
        if rank == 0:
            clean = {k: v for k, v in config.items() if not k.startswith("_")}
            with open(os.path.join(save_dir, f'{alg_name}_{env_name}_seed{config["SEED"]}_config.yaml'), "w") as f:
                yaml.safe_dump(clean, f)
        if saver:
            for i, rs in zip(mine, outs["runner_state"]):      # i = GLOBAL seed index: the vmap axis of :464-483
                save_params(rs["params"], os.path.join(
                    save_dir, f'{alg_name}_{env_name}_seed{config["SEED"]}_vmap{i}.safetensors'))
    return {"runner_state": outs["runner_state"], "seed_indices": mine, "metrics": metrics,
            "rank": rank, "world_size": world}


# the sweep the reference hands to wandb (pqn_minatar.py:509-527; the other scripts carry the same block): one parameter, four values,
# metric returned_episode_returns, goal maximize
SWEEP_PARAMETERS = {"LR": [0.001, 0.0005, 0.0001, 0.00005]}
SWEEP_METRIC = "returned_episode_returns"


def tune(default_config: Dict[str, Any], script: str = "gymnax", parameters: Optional[Dict[str, List[Any]]] = None,
         run_fn: Optional[Callable] = None) -> Dict[str, Any]:
    """tune(default_config) of the reference scripts (pqn_minatar.py:484-531) without the wandb service: the reference
    registers a sweep (method "bayes", metric returned_episode_returns, goal maximize) over LR in {1e-3, 5e-4, 1e-4, 5e-5}
    and lets wandb.agent call wrapped_make_train -- a copy of the default config with the drawn parameters written over
    it, NUM_SEEDS vmapped seeds (:495-507) -- up to 1000 times.  Offline the sweep's search space is what can be honoured:
    every combination of `parameters` (default: the reference's) is run once through single_run (same launcher path, seeds
    batched into the launches, checkpoints off), the metric is the mean over seeds of its last-update value (NaN-safe), and
    the ranking is printed and returned.  With WORLD_SIZE > 1 every rank runs every configuration on its share of the seeds;
    rank 0 reports."""
    import copy
    import itertools
    import math
    space = dict(SWEEP_PARAMETERS if parameters is None else parameters)
    names = sorted(space)
    run_fn = single_run if run_fn is None else run_fn
    trials = []
    for values in itertools.product(*(space[n] for n in names)):
        config = copy.deepcopy(default_config)
        flat_alg = config.get("alg") if isinstance(config.get("alg"), dict) else None
        for n, v in zip(names, values):       # wandb.config entries are written over the flattened config (:497-499)
            config[n] = v
            if flat_alg is not None and n in flat_alg:
                flat_alg[n] = v
        config["SAVE_PATH"] = None
        outs = run_fn(config, script=script)
        m = outs["metrics"].get(SWEEP_METRIC)
        last = torch.as_tensor(m)[..., -1].to(torch.float64).reshape(-1) if m is not None else torch.empty(0, dtype=torch.float64)
        ok = last[~torch.isnan(last)]
        score = float(ok.mean()) if ok.numel() else float("nan")
        trials.append({"parameters": dict(zip(names, values)), SWEEP_METRIC: score, "rank": outs.get("rank", 0)})
        if outs.get("rank", 0) == 0:
            print(f"sweep trial {len(trials)}: {dict(zip(names, values))} -> {SWEEP_METRIC} = {score:.4f}", flush=True)
    ranked = sorted(trials, key=lambda t: (math.isnan(t[SWEEP_METRIC]), -t[SWEEP_METRIC] if not math.isnan(t[SWEEP_METRIC]) else 0.0))
    if trials and trials[0]["rank"] == 0:
        print("sweep ranking (metric: %s, goal: maximize):" % SWEEP_METRIC)
        for t in ranked:
            print("  ", json.dumps({**t["parameters"], SWEEP_METRIC: t[SWEEP_METRIC]}))
    return {"trials": trials, "best": ranked[0] if ranked else None}


def main(argv: List[str], default_alg: str, script: str = "gymnax") -> Dict[str, Any]:
    """`python -m purejaxql_amd.pqn_minatar +alg=pqn_minatar alg.KEY=V KEY=V` (README.md:170-187)."""
    overrides = list(argv)
    if not any(o.startswith("+alg=") or o.startswith("alg=") for o in overrides):
        overrides.insert(0, f"+alg={default_alg}")
    config = load_config(overrides)
    rank, _world, _lr = pdist.world_info()
    if rank == 0:
        print("Config:\n", yaml.safe_dump(config))
    if config.get("HYP_TUNE", False):      # (:537-540)
        res = tune(config, script=script)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return res
    outs = single_run(config, script=script)
    if outs["rank"] == 0:
        m = outs["metrics"]
        last = {k: [round(float(x), 4) for x in v[:, -1]] for k, v in m.items()}
        print("final metrics per seed:", json.dumps(last))
    if outs["world_size"] > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return outs
