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
    # ranks hold different seed counts (3 and 2): gather_seed_metrics pads internally and restores seed order
    allm = pdist.gather_seed_metrics(m)["td_loss"]
    ok2 = allm.shape == (5, 3) and allm[:, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]
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


# ---------------------------------------------------------------------------------------------------------
# run.single_run under WORLD_SIZE = 2 (the multi-GPU product path), driven on CPU with stand-in train functions
# ---------------------------------------------------------------------------------------------------------
def _stub_seed_keys(seed, num_seeds):
    return [1000 * int(seed) + i for i in range(int(num_seeds))]


def _stub_make_train(config, device=None, grad_hook=None, metrics_hook=None):
    def train(key):
        raise AssertionError("the stub is driven through vmap_fn")
    train.config = config
    train.grad_hook, train.metrics_hook = grad_hook, metrics_hook
    return train


def _stub_vmap(train, keys, concurrent=True):
    """What vmap_train returns, as a function of the keys only: seed with key k has td_loss row [k, k+1, k+2]."""
    nu = 3
    rows = torch.tensor([[float(k + u) for u in range(nu)] for k in keys], dtype=torch.float32).reshape(len(keys), nu)
    if train.metrics_hook is not None:          # env-sharded mode: per-rank means are averaged over the shards
        rank = dist.get_rank()
        rows = torch.stack([train.metrics_hook(r + float(rank)) for r in rows]) if len(keys) else rows
    rs = [{"params": {"Dense_0/kernel": torch.full((2, 2), float(k))}} for k in keys]
    return {"runner_state": rs, "metrics": {"td_loss": rows, "qvals": 2 * rows}}


def _run_worker(rank, world, port, q, save_path, num_seeds, shard):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), PQN_DIST_BACKEND="gloo")
    from purejaxql_amd.config_loader import load_config
    from purejaxql_amd.run import single_run
    cfg = load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", f"NUM_SEEDS={num_seeds}", "SEED=7",
                       f"SAVE_PATH={save_path}", f"+SHARD={shard}", "alg.NUM_ENVS=128"])
    out = single_run(cfg, device="cpu", make_train_fn=_stub_make_train, vmap_fn=_stub_vmap, seed_keys_fn=_stub_seed_keys)
    q.put((rank, out["seed_indices"], out["metrics"]["td_loss"].tolist(), out["metrics"]["qvals"].shape[0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("num_seeds,shard", [(5, "seeds"), (1, "seeds"), (3, "envs")])
def test_single_run_world2_partitions_seeds_gathers_metrics_and_saves_global_indices(tmp_path, num_seeds, shard):
    """pqn_minatar.py:456-483 under two ranks: rank r trains partition_seeds(S, 2, r) only, every rank ends with the
    [S, NUM_UPDATES] metrics in seed order, checkpoints carry GLOBAL vmap indices, rank 0 writes the config.
    SHARD=envs: every rank runs all seeds on half the envs, metric means are averaged over the two shards."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_worker, args=(r, world, port, q, str(tmp_path), num_seeds, shard)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    keys = _stub_seed_keys(7, num_seeds)
    if shard == "seeds":
        from purejaxql_amd.dist import partition_seeds
        want = [[float(k + u) for u in range(3)] for k in keys]
        for rank, mine, td, nq in res:
            assert mine == partition_seeds(num_seeds, world, rank)
            assert td == want and nq == num_seeds        # full [S, NUM_UPDATES], seed order, on every rank
    else:
        want = [[float(k + u) + 0.5 for u in range(3)] for k in keys]   # mean of the two shards' rows (+0, +1)
        for rank, mine, td, nq in res:
            assert mine == list(range(num_seeds)) and td == want
    files = sorted(os.listdir(os.path.join(tmp_path, "Breakout-MinAtar")))
    assert files == ["pqn_Breakout-MinAtar_seed7_config.yaml"] + [f"pqn_Breakout-MinAtar_seed7_vmap{i}.safetensors"
                                                                 for i in range(num_seeds)]
    from purejaxql_amd.save_load import load_params
    for i, k in enumerate(keys):   # file vmap{i} holds the parameters of GLOBAL seed i
        p = load_params(os.path.join(tmp_path, "Breakout-MinAtar", f"pqn_Breakout-MinAtar_seed7_vmap{i}.safetensors"))
        assert float(p["Dense_0/kernel"][0, 0]) == float(k)


def test_shard_env_config_keeps_the_global_schedule():
    from purejaxql_amd.dist import shard_env_config
    from purejaxql_amd.pqn import derive_config
    base = {"NUM_ENVS": 4096, "NUM_STEPS": 32, "NUM_MINIBATCHES": 32, "TOTAL_TIMESTEPS": 1e7, "TOTAL_TIMESTEPS_DECAY": 1e7}
    full = derive_config(dict(base))
    for world in (2, 4, 8):
        c = derive_config(shard_env_config(base, 1, world))
        assert c["NUM_ENVS"] == 4096 // world and c["_ENV_SHARD"] == (1, world)
        assert c["NUM_UPDATES"] == full["NUM_UPDATES"] == 76 and c["NUM_UPDATES_DECAY"] == full["NUM_UPDATES_DECAY"]
    with pytest.raises(ValueError):
        shard_env_config({**base, "NUM_ENVS": 100}, 0, 8)
