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

from typing import Callable, List, Optional

import torch
import torch.distributed as dist


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


def allreduce_mean_scalars(values: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Metric means over env shards (pqn_minatar.py:330-338): one small all-reduce per update."""
    dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
    return values.div_(dist.get_world_size(group))


def gather_seed_metrics(metrics: dict, group: Optional[dist.ProcessGroup] = None) -> dict:
    """Stack per-rank [S_local, NUM_UPDATES] metric tensors into [S, NUM_UPDATES] on every rank
    (the leading axis jax.vmap would have produced on one device)."""
    world = dist.get_world_size(group)
    out = {}
    for k, v in metrics.items():
        bufs = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(bufs, v.contiguous(), group=group)
        out[k] = torch.cat(bufs, dim=0)
    return out
