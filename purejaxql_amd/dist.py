"""Multi-GPU sharding of the PQN hot path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-device code.  Its two vmap axes shard as:
  * seeds  (jax.vmap(make_train) over rngs, pqn_minatar.py:459-461): independent
    runs -> partition seeds over ranks, NO collective on the data path;
  * envs of one seed (vmap_step, :110-112): the only cross-env reductions are the
    minibatch-mean loss gradient (:285-292), so each optimizer step all-reduces ONE
    flat fp32 gradient bucket (132,475 floats = 530 KB for Breakout) before
    clip + RAdam, which must see the averaged gradient (:159-162).
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def world_info() -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) of this process: the initialised process group, else the torchrun
    environment (RANK / WORLD_SIZE / LOCAL_RANK), else a single process."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", "0"))
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """One process per GPU (launched by `python -m torch.distributed.run`): bind cuda:LOCAL_RANK and join the
    process group.  backend: "nccl" (= RCCL over xGMI on ROCm; default when a GPU is visible) or "gloo"
    (PQN_DIST_BACKEND=gloo: CPU tests, or several ranks sharing one GPU).  No-op for a single process."""
    rank, world, local_rank = world_info()
    if world <= 1:
        return rank, world, local_rank
    backend = backend or os.environ.get("PQN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        if os.environ.get("PQN_BENCH_ONE_GPU", "0") == "1":   # several ranks on one GPU (tests on a 1-GPU box)
            local_rank = 0
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def partition_seeds(num_seeds: int, world_size: int, rank: int) -> List[int]:
    """Contiguous block partition of seed indices; ranks differ by at most one seed."""
    base, rem = divmod(int(num_seeds), int(world_size))
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def make_grad_allreduce_hook(group: Optional[dist.ProcessGroup] = None) -> Callable[[torch.Tensor], None]:
    """grad_hook for make_train (env-sharded mode): sum the flat gradient bucket over
    ranks, divide by world size (mean over the global minibatch of B*G samples)."""
    world = dist.get_world_size(group)

    def hook(flat_grad: torch.Tensor) -> None:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
        flat_grad.div_(world)

    return hook


def shard_env_config(config: Dict[str, Any], rank: int, world: int) -> Dict[str, Any]:
    """Env-sharded mode: rank r owns NUM_ENVS/world envs of the ONE seed and an equal share of the timestep budget
    (so NUM_UPDATES, the schedules and the per-update global batch are those of the unsharded run); the minibatch
    count stays, each rank's minibatch is 1/world of the global one."""
    if int(config["NUM_ENVS"]) % world:
        raise ValueError(f"NUM_ENVS={config['NUM_ENVS']} is not divisible by the {world} ranks")
    c = dict(config)
    c["NUM_ENVS"] = int(config["NUM_ENVS"]) // world
    c["TOTAL_TIMESTEPS"] = config["TOTAL_TIMESTEPS"] / world
    c["TOTAL_TIMESTEPS_DECAY"] = config["TOTAL_TIMESTEPS_DECAY"] / world
    c["_ENV_SHARD"] = (int(rank), int(world))
    return c


def allreduce_mean_scalars(values: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Metric means over env shards (pqn_minatar.py:330-338): one small all-reduce per update."""
    dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
    return values.div_(dist.get_world_size(group))


def gather_seed_metrics(metrics: dict, group: Optional[dist.ProcessGroup] = None) -> dict:
    """Stack per-rank [S_local, NUM_UPDATES] metric tensors into [S, NUM_UPDATES] on every rank
    (the leading axis jax.vmap would have produced on one device), rank order = seed order of partition_seeds.
    Ranks may hold different numbers of seeds, including none (NUM_SEEDS < world): the per-rank row counts and the
    metric names / trailing shapes are exchanged first, every block is padded to the largest one for the
    all_gather (mismatched sizes hang or corrupt memory under RCCL) and sliced back out."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    desc = {k: (int(v.shape[0]), tuple(v.shape[1:]), str(v.dtype).replace("torch.", "")) for k, v in metrics.items()}
    descs: List[Any] = [None] * world
    dist.all_gather_object(descs, desc, group=group)
    names = next((list(dd) for dd in descs if dd), [])
    if not names:
        return {}
    ref = next(dd for dd in descs if dd)
    counts = [(next(iter(dd.values()))[0] if dd else 0) for dd in descs]
    smax = max(counts)
    dev = next(iter(metrics.values())).device if metrics else (
        torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu"))
    out = {}
    for k in names:
        _n, tail, dt = ref[k]
        dtype = getattr(torch, dt)
        pad = torch.zeros((smax, *tail), dtype=dtype, device=dev)
        if k in metrics:
            v = metrics[k].contiguous()
            if v.shape[0] != counts[rank] or tuple(v.shape[1:]) != tuple(tail):
                raise ValueError(f"gather_seed_metrics: '{k}' has shape {tuple(v.shape)}, expected ({counts[rank]}, *{tail})")
            pad[:v.shape[0]] = v
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        out[k] = torch.cat([bb[:c] for bb, c in zip(bufs, counts)], dim=0)
    return out


def gather_seed_list(local: List[Any], group: Optional[dist.ProcessGroup] = None) -> List[Any]:
    """Concatenate per-rank python lists in rank order (host objects: per-seed summaries)."""
    world = dist.get_world_size(group)
    out: List[Any] = [None] * world
    dist.all_gather_object(out, local, group=group)
    return [x for part in out for x in part]
