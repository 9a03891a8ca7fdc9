"""GPU test of the env-sharded multi-rank training step (-m gpu): two ranks share the one GPU of the box over gloo
(one process per rank, as `torch.distributed.run` would start them), each owning half of the envs of ONE seed, with
the per-minibatch gradient all-reduce between pqn_cnn_update_phase(GRAD) and (APPLY).  RCCL itself needs >= 2 GPUs;
the control flow, the phase split, the segment hipGraphs and the arithmetic are what this covers."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg(num_envs_global, n_upd):
    from purejaxql_amd.config_loader import flatten, load_config
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar"]))
    cfg.update({"NUM_ENVS": num_envs_global, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2,
                "TOTAL_TIMESTEPS": n_upd * num_envs_global * 8, "TOTAL_TIMESTEPS_DECAY": 30 * num_envs_global * 8,
                "TEST_DURING_TRAINING": False})
    return cfg


def _rank_main(rank, world, port, q, num_envs_global, n_upd, theta0, use_driver, peer, fault=""):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), PQN_DIST_BACKEND="gloo", PQN_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                      PQN_PEER_FAULT=fault)
    import torch.distributed as dist
    torch.set_num_threads(2)      # the ranks share the host: one thread per core EACH (the default) thrashes
    from purejaxql_amd import dist as pdist
    from purejaxql_amd.pqn import make_train, seed_keys
    pdist.init_from_env()
    cfg = pdist.shard_env_config(_cfg(num_envs_global, n_upd), rank, world)
    cfg["_INIT_PARAMS"] = torch.from_numpy(theta0).to("cuda:0")
    cfg["_DRIVER"] = use_driver
    train = make_train(cfg, device="cuda:0", grad_hook=pdist.make_grad_allreduce_hook(peer=peer),
                       metrics_hook=pdist.allreduce_mean_scalars)
    out = train(seed_keys(0, 1)[0])
    torch.cuda.synchronize()
    rs = out["runner_state"]
    q.put((rank, rs["theta"].cpu().numpy(), {k: v.cpu().numpy() for k, v in out["metrics"].items()}, rs["driver"],
           (rs["driver_graph_error"], rs["allreduce"], rs["driver_graphs"])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_driver,peer", [(True, True), (True, False), (False, True), (False, False)])
def test_two_rank_env_sharded_training_matches_oracle_with_averaged_gradients(gpu, oracle, use_driver, peer):
    """3 updates of make_train(grad_hook = all-reduce mean) on 2 ranks x 32 envs: theta identical on both ranks and
    equal (bulk rtol 2e-3, worst element < lr: the criterion of test_make_train_end_to_end_vs_oracle) to the oracle loop
    that averages the two shards' gradients per optimizer step; metric means are means over both shards.
    use_driver: the phase-split C++ enqueue with hipGraphs (update 0 eager, 1 captured, 2 replayed) / the per-kernel
    Python loop.  peer: the one-shot all-reduce over hipIpc-mapped peer buffers (csrc/pqn_peer.hip; two processes mapping
    each other's staging regions on the one GPU) -- the whole update, its 8 collectives included, is then ONE graph per
    rank -- / the torch.distributed collective issued from the host between 9 per-segment graphs."""
    from purejaxql_amd.dist import shard_env_config
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.pqn import seed_keys
    world, n_glob, n_upd = 2, 64, 3
    theta0 = QNetwork("cnn", (10, 10, 4), 3, device=gpu).init(17).cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q, n_glob, n_upd, theta0, use_driver, peer)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, th0, m0, drv0, err0), (_, th1, m1, _drv1, _err1) = res
    gerr, mode, n_graphs = err0
    assert mode == ("peer" if peer else "host"), (mode, gerr)
    if use_driver:
        assert drv0 == "graph", gerr
        assert n_graphs == (1 if peer else 9), (n_graphs, gerr)
    else:
        assert drv0 is None
    np.testing.assert_array_equal(th0, th1)                   # both ranks applied the same averaged gradients
    for k in m0:
        np.testing.assert_array_equal(m0[k], m1[k], err_msg=k)
    ocfg = {k: v for k, v in shard_env_config(_cfg(n_glob, n_upd), 0, world).items() if not k.startswith("_")}
    oout = oracle.make_train(ocfg)(seed_keys(0, 1)[0], theta0, shard_world=world)
    assert len(oout["metrics"]) == n_upd
    for u in range(n_upd):
        om = oout["metrics"][u]
        assert float(m0["env_step"][u]) == om["env_step"] == (u + 1) * n_glob * 8
        assert float(m0["grad_steps"][u]) == om["grad_steps"]
        for k in ("td_loss", "qvals", "returned_episode_returns", "returned_episode_lengths", "timestep",
                  "returned_episode", "discount"):
            assert abs(float(m0[k][u]) - om[k]) <= 1e-3 * max(1.0, abs(om[k])), (u, k, float(m0[k][u]), om[k])
    d = np.abs(th0 - oout["theta"])
    bad = d > (2e-5 + 2e-3 * np.abs(oout["theta"]))
    assert bad.mean() < 1e-3 and d.max() < 5e-4, (int(bad.sum()), float(d.max()))


@pytest.mark.parametrize("fault", ["open:1", "selftest:0"])
def test_peer_path_setup_failure_on_one_rank_falls_back_to_the_host_collective_on_all(gpu, fault):
    """First-contact readiness for an 8-GPU node: when ONE rank cannot map a peer region (as if
    hipIpcOpenMemHandle failed across devices) or fails the path's self-test, EVERY rank must land on the torch.distributed
    collective -- no hang, no rank left on the peer kernels -- and the ranks' parameters stay identical.  2 ranks x 32 envs, 3
    updates, phase-split driver (9 segment graphs: the host collective sits between them)."""
    from purejaxql_amd.networks import QNetwork
    world, n_glob, n_upd = 2, 64, 3
    theta0 = QNetwork("cnn", (10, 10, 4), 3, device=gpu).init(17).cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q, n_glob, n_upd, theta0, True, True, fault)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, th0, m0, drv0, (gerr0, mode0, ng0)), (_, th1, m1, drv1, (gerr1, mode1, ng1)) = res
    assert mode0.startswith("host") and mode1.startswith("host") and "self-test" in mode0, (mode0, mode1)
    assert drv0 == drv1 == "graph" and ng0 == ng1 == 9, (gerr0, gerr1, ng0, ng1)
    assert np.isfinite(th0).all()
    np.testing.assert_array_equal(th0, th1)
    for k in m0:
        np.testing.assert_array_equal(m0[k], m1[k], err_msg=k)


def _recapture_main(rank, world, port, q, theta0, peer):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), PQN_DIST_BACKEND="gloo", PQN_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.set_num_threads(2)      # the ranks share the host: one thread per core EACH (the default) thrashes
    from purejaxql_amd import _lib
    from purejaxql_amd import dist as pdist
    from purejaxql_amd.pqn import make_train, seed_keys
    pdist.init_from_env()
    thetas, graphs = [], []
    for flip in (False, True):
        cfg = pdist.shard_env_config(_cfg(64, 5), rank, world)
        cfg["_INIT_PARAMS"] = torch.from_numpy(theta0).to("cuda:0")
        train = make_train(cfg, device="cuda:0", grad_hook=pdist.make_grad_allreduce_hook(peer=peer),
                           metrics_hook=pdist.allreduce_mean_scalars)
        update, finish = train.make_runner(seed_keys(0, 1)[0])
        drv = update.driver
        seen, alive = [], []
        prev = _lib.get_option("peer_timeout_s")
        for u in range(5):
            if flip and u == 3:
                _lib.set_option("peer_timeout_s", prev + 1)    # every rank, same update: the re-capture is collective
            update(u)
            g = drv.whole if drv.whole is not None else (drv.graphs[0] if drv.graphs else None)
            if g is not None and not any(g is x for x in alive):
                alive.append(g)            # kept alive: a dropped graph's address could be handed to its successor
            seen.append(None if g is None else [i for i, x in enumerate(alive) if x is g][0])
        _lib.set_option("peer_timeout_s", prev)
        out = finish()
        torch.cuda.synchronize()
        thetas.append(out["runner_state"]["theta"].cpu().numpy())
        graphs.append(seen)
        dist.barrier()
    q.put((rank, thetas, graphs, type(drv).__name__))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("peer", [True, False])
def test_env_shard_driver_recaptures_its_graphs_when_an_option_changes(gpu, peer):
    """EnvShardDriver watches pqn_options_epoch like UpdateDriver does.  Two ranks x 32 envs, 5 updates: update 1
    captures (one whole-update graph with the in-graph peer all-reduce / nine per-segment graphs around the host collective),
    update 2 replays, an option changes on both ranks before update 3 -- that update is enqueued and captured afresh -- and the
    parameters equal the undisturbed run's bit for bit on both ranks."""
    from purejaxql_amd.networks import QNetwork
    world = 2
    theta0 = QNetwork("cnn", (10, 10, 4), 3, device=gpu).init(17).cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_recapture_main, args=(r, world, port, q, theta0, peer)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for _rank, thetas, graphs, name in res:
        assert name == "EnvShardDriver"
        plain, flipped = graphs
        assert plain[1] is not None and plain[1] == plain[2] == plain[3] == plain[4], plain
        assert flipped[1] == flipped[2] and flipped[3] is not None and flipped[3] != flipped[2] and flipped[4] == flipped[3], flipped
        np.testing.assert_array_equal(thetas[0], thetas[1])
    np.testing.assert_array_equal(res[0][1][0], res[1][1][0])


def _peer_main(rank, world, port, q, n, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), PQN_DIST_BACKEND="gloo", PQN_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.set_num_threads(2)      # the ranks share the host: one thread per core EACH (the default) thrashes
    from purejaxql_amd import dist as pdist
    pdist.init_from_env()
    dev = torch.device("cuda:0")
    import time
    t_start = time.time()
    par = pdist.PeerAllReduce(n, dev)
    ok = par.setup()
    t_setup = time.time() - t_start
    outs, refs = [], []
    if ok:
        g = torch.Generator(device="cpu")
        side = torch.cuda.Stream()
        for s in range(steps):
            # every rank can rebuild every rank's bucket: the expected mean needs no second collective
            vs = []
            for r in range(world):
                g.manual_seed(1000 * s + r)
                vs.append(torch.randn(n, generator=g))
            mine = vs[rank].to(dev)
            if s == steps // 2:
                time_skew = torch.randn(4096, 4096, device=dev)      # one rank arrives late at this step
                for _ in range(20 if rank == 0 else 0):
                    time_skew = time_skew @ time_skew * 1e-3
            if s % 3 == 2:
                with torch.cuda.stream(side):                        # the collective follows the caller's current stream
                    side.wait_stream(torch.cuda.current_stream())
                    par(mine)
                torch.cuda.current_stream().wait_stream(side)
            else:
                par(mine)
            acc = torch.zeros(n)
            for r in range(world):
                acc += vs[r]
            outs.append(mine.cpu())
            refs.append(acc * (1.0 / world))
        par.check()
    if rank == 0:
        print(f"peer all-reduce, world {world}: setup {t_setup:.1f} s, {steps} steps {time.time() - t_start - t_setup:.1f} s", flush=True)
    q.put((rank, ok, [o.numpy() for o in outs], [r.numpy() for r in refs]))
    dist.barrier()
    par.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 132475), (2, 7), (8, 132475)])
def test_peer_allreduce_processes_on_one_gpu(gpu, world, n):
    """dist.PeerAllReduce by itself: `world` processes map each other's staging regions through hipIpc and average a bucket
    of the CNN's size (odd length: the scalar tail) over 24 consecutive steps (double-buffer reuse, one rank arriving late,
    calls from a side stream): bit-identical on all ranks and equal to the sum in rank order times 1 / world.  world = 8 is
    PQN_PEER_MAX, the node size of BASELINE.json's configs[3] (eight processes on the one GPU of the box here)."""
    steps = 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_peer_main, args=(r, world, port, q, n, steps)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _o, _r in res), "hipIpc peer mapping is unavailable on this box"
    _, _, o0, r0 = res[0]
    for s in range(steps):
        np.testing.assert_array_equal(o0[s], r0[s], err_msg=f"step {s}")
        for rk in range(1, world):
            np.testing.assert_array_equal(o0[s], res[rk][2][s], err_msg=f"step {s} rank {rk}")


def _peer_timeout_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), PQN_DIST_BACKEND="gloo", PQN_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import time
    import torch.distributed as dist
    torch.set_num_threads(2)      # the ranks share the host: one thread per core EACH (the default) thrashes
    from purejaxql_amd import _lib
    from purejaxql_amd import dist as pdist
    pdist.init_from_env()
    dev = torch.device("cuda:0")
    n = 4099
    par = pdist.PeerAllReduce(n, dev)
    ok = par.setup()            # includes the 3-step self-test through the real kernels
    res = {"ok": ok}
    if ok:
        _lib.set_option("peer_timeout_s", 2)
        x = torch.full((n,), float(rank + 1), device=dev)
        par(x)                  # step 4: both ranks take part
        torch.cuda.synchronize()
        res["step_ok"] = bool(torch.equal(x.cpu(), torch.full((n,), 1.5)))
        if rank == 0:           # rank 1 never publishes step 5: rank 0's wait must end after ~2 s of WALL CLOCK with the error word set
            t0 = time.time()
            par(x)
            torch.cuda.synchronize()
            res["waited_s"] = time.time() - t0
            res["poisoned"] = bool(torch.isnan(x).all())   # no plausible mean of stale buffers: the bucket is NaN
            par.poll()          # queues the asynchronous copy of the error word ...
            torch.cuda.synchronize()
            try:
                par.poll()      # ... and the next call sees it
                res["poll_raised"] = False
            except RuntimeError as exc:
                res["poll_raised"] = "rank 1" in str(exc)
            try:
                par.check()
                res["check_raised"] = False
            except RuntimeError as exc:
                res["check_raised"] = "rank 1" in str(exc)
            t1 = time.time()
            par(x)              # after a time-out every later collective fails fast
            torch.cuda.synchronize()
            res["fail_fast_s"] = time.time() - t1
    q.put((rank, res))
    dist.barrier()
    par.close()
    dist.destroy_process_group()


def test_peer_allreduce_time_out_is_wall_clock_and_reported(gpu):
    """The in-graph peer all-reduce with a rank that never publishes (round 4 hardening): the waiting rank gives up after the
    configured WALL-CLOCK time (option peer_timeout_s = 2 here; rounds 1-3 counted polls), writes NaN over its gradient bucket
    instead of a mean of stale staging buffers (round 5), leaves a sticky error word that names the missing rank, PeerAllReduce.poll() -- the training loop's once-per-update, non-blocking look at it -- and
    check() raise, and later collectives fail fast instead of waiting again.  setup() has by then passed its self-test
    (three all-reduces of a known bucket through the real kernels, verdict all-gathered)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_peer_timeout_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0]["ok"] and res[1]["ok"], "hipIpc peer mapping is unavailable on this box"
    assert res[0]["step_ok"] and res[1]["step_ok"]
    assert 1.5 <= res[0]["waited_s"] <= 6.0, res[0]
    assert res[0]["poisoned"] is True, res[0]
    assert res[0]["poll_raised"] is True and res[0]["check_raised"] is True, res[0]
    assert res[0]["fail_fast_s"] < 1.0, res[0]


@pytest.mark.parametrize("gpus,mode", [(2, "seeds"), (2, "envs"), (8, "envs")])
def test_bench_gpus_n_launches_its_own_ranks(gpu, gpus, mode):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (the driver's command line) re-executes itself
    under torch.distributed.run with N ranks -- all on the one GPU of the box here (PQN_BENCH_ONE_GPU=1, gloo) -- and
    rank 0 prints ONE JSON line with n_gpus = N, the whole-job rate, max-over-ranks timing; --mode envs also reports
    which gradient all-reduce ran (the in-graph peer kernels) and how many ranks took part in it.  N = 8 is the node size of
    BASELINE.json's configs[3]: control flow of the 8-rank launch, not a scaling measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PQN_BENCH_ONE_GPU="1", PQN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--no-extras",
           "--no-cpu-baseline", "--seeds-per-gpu", "2", "--mode", mode]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == gpus and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    c = d["config"]
    assert c["dist_backend"] == "gloo" and c["rccl_ranks"] == 0 and c["gpus_visible"] == 1
    assert len(c["per_rank_ms_per_step"]) == gpus and abs(max(c["per_rank_ms_per_step"]) - d["ms_per_step"]) < 1e-6 and 0 <= c["per_rank_spread"] < 1
    if mode == "seeds":
        assert d["scaling"] == "weak" and c["seeds_total"] == 4 and c["env_steps_per_step"] == 4 * 4096 * 32
        assert abs(d["value"] - c["env_steps_per_step"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    else:
        assert d["scaling"] == "strong" and c["grad_allreduce"] == "peer" and c["driver"] == "hipGraph replay"
        assert c["ranks_in_allreduce"] == gpus
