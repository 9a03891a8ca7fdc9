"""CPU tests of the N>1 path: world_size-2 gloo process groups (no GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from purejaxql_amd import dist as pdist
    # env-sharded mode: flat gradient bucket is averaged over ranks
    g = torch.full((132475,), float(rank + 1))
    pdist.make_grad_allreduce_hook()(g)
    ok1 = bool(torch.allclose(g, torch.full_like(g, (1 + world) / 2)))
    # seed-sharded mode: per-rank metrics are gathered into the [S, NUM_UPDATES] the vmap would give
    seeds = pdist.partition_seeds(5, world, rank)
    m = {"td_loss": torch.tensor([[float(s)] * 3 for s in seeds])}
    # ranks may hold different seed counts -> pad to the max for all_gather
    pad = max(len(pdist.partition_seeds(5, world, r)) for r in range(world))
    mm = {"td_loss": torch.cat([m["td_loss"], torch.full((pad - len(seeds), 3), -1.0)])}
    allm = pdist.gather_seed_metrics(mm)["td_loss"]
    got = sorted(x for x in allm[:, 0].tolist() if x >= 0)
    ok2 = got == [0.0, 1.0, 2.0, 3.0, 4.0]
    v = pdist.allreduce_mean_scalars(torch.tensor([float(rank), 2.0]))
    ok3 = bool(torch.allclose(v, torch.tensor([(world - 1) / 2, 2.0])))
    q.put((rank, ok1, ok2, ok3))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_seeds():
    from purejaxql_amd.dist import partition_seeds
    assert [partition_seeds(128, 8, r) for r in range(8)] == [list(range(16 * r, 16 * r + 16)) for r in range(8)]
    parts = [partition_seeds(5, 2, r) for r in range(2)]
    assert parts == [[0, 1, 2], [3, 4]]
    assert partition_seeds(1, 4, 3) == []


def test_gloo_world2_allreduce_and_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok1 and ok2 and ok3 for _, ok1, ok2, ok3 in res), res
