"""The configuration bench.py times, under the oracle: MATMUL_DTYPE=bf16x3, 16 seeds x 4096 envs per GPU (BASELINE.json
configs[3]'s per-GPU share) -- i.e. the POSITION-PARALLEL form of the training step (csrc/pqn_qnet_pos.hip: pos_gather_kernel,
cnn_pos_fwd_kernel, cnn_pos_bwd_kernel; round 5) and the same structure's rollout kernel (cnn_pos_rollout_kernel), plus -- option bwd_pos = 0 -- the pair form of the training kernel that launches of 2 .. 9
seeds still take (qnet_cnn_train_pair_kernel, qnet_fc1_wgrad_x3_kernel).
Every test asserts in-process (pqn_cnn_last_kernel_form) that those are the kernels that ran.
Reference lines: value_and_grad(_loss_fn) pqn_minatar.py:271-297 under the seeds vmap :459-461; _update_step :176-369."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

S, N, T, MB, EP = 16, 4096, 32, 32, 2


def _np(t):
    return t.detach().cpu().numpy()


def _pack_bits(obs):
    n, c = obs.shape[0], obs.shape[-1]
    ow = (((100 * c + 31) // 32) + 3) // 4 * 4
    padded = np.zeros((n, ow * 32), np.uint64)
    padded[:, :100 * c] = obs.reshape(n, -1)
    return (padded.reshape(n, ow, 32) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)


GAMES = {"Breakout-MinAtar": (4, 3), "Asterix-MinAtar": (4, 5), "SpaceInvaders-MinAtar": (6, 4), "Freeway-MinAtar": (7, 3)}


def _cfg(n_upd, env="Breakout-MinAtar", dtype="bf16x3", **extra):
    from purejaxql_amd.config_loader import flatten, load_config
    cfg = flatten(load_config(["+alg=pqn_minatar", f"alg.ENV_NAME={env}", f"alg.NUM_ENVS={N}",
                               "alg.TEST_DURING_TRAINING=False", f"alg.MATMUL_DTYPE={dtype}"]))
    assert (cfg["NUM_STEPS"], cfg["NUM_MINIBATCHES"], cfg["NUM_EPOCHS"]) == (T, MB, EP)
    cfg["TOTAL_TIMESTEPS"] = n_upd * N * T
    cfg["TOTAL_TIMESTEPS_DECAY"] = 30 * N * T
    cfg.update(extra)
    return cfg


def _assert_grad_close(g, g_ref, shapes, what):
    """The tolerance of test_cnn_grad_vs_oracle (rtol 2e-3, atol 3e-6 max|g|) -- except for relu decisions AT the threshold:
    a launch of 16 seeds x 4096 samples evaluates 8.4e6 hidden units of Dense_0, and a unit whose LayerNorm_1 output is zero
    to f32 rounding is "on" in one implementation and "off" in the other.  Each such flip moves ONE output column of the
    Dense_0 kernel gradient (and that unit's bias / LayerNorm_1 entries, and through dz a trace in the conv block) by one
    sample's contribution out of 4096.  So: either the plain tolerance holds, or the violating Dense_0 entries are confined
    to at most two output columns, the worst entry is within 2e-3 of max|g| and the whole gradient within 1e-3 in L2."""
    atol = 3e-6 * np.abs(g_ref).max() + 1e-9
    d = np.abs(g - g_ref)
    bad = d > atol + 2e-3 * np.abs(g_ref)
    if not bad.any():
        return
    off = 0
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        if k.endswith("Dense_0/kernel") and tuple(shp) == (1024, 128):
            cols = np.unique(np.nonzero(bad[off:off + n].reshape(1024, 128))[1])
            assert len(cols) <= 2, (what, "Dense_0 kernel gradient off in columns", cols[:10])
        off += n
    assert d.max() <= 2e-3 * np.abs(g_ref).max() and np.linalg.norm(g - g_ref) <= 1e-3 * np.linalg.norm(g_ref), \
        (what, float(d.max()), float(np.abs(g_ref).max()), int(bad.sum()))


@pytest.mark.parametrize("stacked,c,a,form,dtype", [(False, 4, 3, "pos", "bf16x3"), (True, 4, 3, "pos", "bf16x3"), (False, 6, 4, "pos", "bf16x3"),
                                                    (False, 7, 3, "pos", "bf16x3"), (False, 4, 3, "pos", "f16x2"), (True, 4, 3, "pos", "f16x2"),
                                                    (False, 6, 4, "pos", "f16x2"), (False, 7, 3, "pos", "f16x2"),
                                                    (False, 4, 3, "pair", "bf16x3"), (True, 4, 3, "pair", "bf16x3"), (False, 6, 4, "pair", "bf16x3"),
                                                    (False, 7, 3, "pair", "bf16x3"), (False, 4, 3, "pair", "f16x2")])
def test_headline_launch_gradient_vs_oracle(gpu, oracle, stacked, c, a, form, dtype):
    """vmap(value_and_grad(_loss_fn)) at the bench's launch shape -- 16 seeds x 4096-sample minibatches in one launch,
    bf16x3, seeds remapped over the XCDs -- against the oracle's numpy backward, seed by seed (own parameters, own
    minibatch), with the tolerances of test_cnn_grad_vs_oracle; for Breakout (C = 4, 3 actions), SpaceInvaders (C = 6, 4)
    and Freeway (C = 7, 3) -- the shapes bench.py's minatar_suite times.  form = "pos": the default of such a launch, the
    position-parallel kernels (256 forward + 256 backward workgroups); form = "pair" (option bwd_pos = 0): 2048 workgroups
    of the pair kernel + the fc1 weight-gradient kernel, what launches of fewer than 10 seeds take.  stacked=True reads the
    samples out of a stacked [T][S*N] record (the layout pqn_cnn_update_seeds trains from), False from a shared pool.
    The same seed launched ALONE in the same form gives the same bits (the summation orders do not depend on the launch).
    dtype = f16x2: the position-parallel kernels on two fp16 pieces per operand, same bounds; its "pair" case runs bf16x3 (the mode
    exists in the position form only -- every other kernel form of an f16x2 layout is the bf16x3 one)."""
    from purejaxql_amd import _lib
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, cnn_grad_seeds, matmul_mode
    rng = np.random.default_rng(2026 + stacked + 10 * c)
    torch.manual_seed(7)
    nb = 4096
    n_env, t_len = (512, 12) if stacked else (20000, 1)
    rows = n_env * t_len * (S if stacked else 1)
    obs = (rng.random((rows, 10, 10, c)) < 0.12).astype(np.float32)
    bits = torch.from_numpy(_pack_bits(obs).view(np.int32)).to(gpu)
    action = rng.integers(0, a, rows).astype(np.int32)
    target = rng.standard_normal(rows).astype(np.float32)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    lay = CnnKernelLayout(c, a, matmul_f16=matmul_mode(dtype))
    stride = (lay.alloc + 3) // 4 * 4
    thetas = [net.init(100 + s) + 0.05 * torch.randn(net.num_params, device=gpu) for s in range(S)]
    theta_k = torch.zeros((S, stride), dtype=torch.float32, device=gpu)
    for s in range(S):
        theta_k[s, :lay.alloc] = lay.to_kernel(thetas[s])
    idx = np.stack([rng.permutation(n_env * t_len)[:nb] for _ in range(S)]).astype(np.int64)   # per-seed transition indices
    with _lib.options(bwd_pos=1 if form == "pos" else 0):
        grad, loss, qv = cnn_grad_seeds(lay, theta_k, torch.from_numpy(idx).to(gpu), bits, torch.from_numpy(action).to(gpu),
                                        torch.from_numpy(target).to(gpu), n_env, n_env * S if stacked else n_env)
        assert _lib.last_kernel_form()[0] == form
    shapes = oracle.cnn_shapes((10, 10, c), a)
    for s in range(S):
        j = idx[s]
        row = (j // n_env) * (n_env * S) + s * n_env + j % n_env if stacked else j    # pqn_hotpath.h: pqn_qnet_cnn_grad_seeds
        p = oracle.unflatten(_np(thetas[s]), shapes)
        lo, chosen, g_ref = oracle.net_loss_grad("cnn", p, shapes, obs[row], action[row], target[row])
        assert abs(float(loss[s]) - lo) <= 1e-4 * max(1.0, abs(lo)), s
        assert abs(float(qv[s]) - chosen.mean()) <= 1e-4, s
        g = _np(lay.to_flax(grad[s]))
        _assert_grad_close(g, g_ref, shapes, f"seed {s}")
    if not stacked:   # the same seed alone: the same form forced ("pos"), or the single-tile kernel whose sums run in the pair kernel's order
        with _lib.options(bwd_pos=2 if form == "pos" else 0):
            g1, l1, _ = cnn_grad_seeds(lay, theta_k[5:6].contiguous(), torch.from_numpy(idx[5:6]).to(gpu), bits,
                                       torch.from_numpy(action).to(gpu), torch.from_numpy(target).to(gpu), n_env, n_env)
            assert _lib.last_kernel_form()[0] == ("pos" if form == "pos" else "single")
        assert torch.equal(g1[0, :lay.total], grad[5, :lay.total]) and float(l1[0]) == float(loss[5])


def test_single_seed_8192_sample_minibatch_takes_the_pair_kernel(gpu, oracle):
    """One seed with 256 pairs per launch (nb = 8192) selects the pair kernel by itself; against the oracle."""
    from purejaxql_amd import _lib
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer, matmul_mode
    rng = np.random.default_rng(8192)
    torch.manual_seed(3)
    pool, nb = 30000, 8192
    obs = (rng.random((pool, 10, 10, 4)) < 0.12).astype(np.float32)
    bits = torch.from_numpy(_pack_bits(obs).view(np.int32)).to(gpu)
    action = rng.integers(0, 3, pool).astype(np.int32)
    target = rng.standard_normal(pool).astype(np.float32)
    idx = rng.permutation(pool)[:nb].astype(np.int64)
    net = QNetwork("cnn", (10, 10, 4), 3, device=gpu)
    lay = CnnKernelLayout(4, 3, matmul_f16=matmul_mode("bf16x3"))
    theta = net.init(11) + 0.05 * torch.randn(net.num_params, device=gpu)
    tr = CnnTrainer(lay, theta, 5e-4, 10.0, max_minibatch=nb)
    lo_t = torch.zeros(1, device=gpu)
    g = tr.compute_grad(torch.from_numpy(idx).to(gpu), bits, torch.from_numpy(action).to(gpu), torch.from_numpy(target).to(gpu), lo_t)
    assert _lib.last_kernel_form()[0] == "pair"
    shapes = oracle.cnn_shapes((10, 10, 4), 3)
    lo, _chosen, g_ref = oracle.net_loss_grad("cnn", oracle.unflatten(_np(theta), shapes), shapes, obs[idx], action[idx], target[idx])
    assert abs(float(lo_t) - lo) <= 1e-4 * max(1.0, abs(lo))
    np.testing.assert_allclose(_np(lay.to_flax(g)), g_ref, rtol=2e-3, atol=3e-6 * np.abs(g_ref).max() + 1e-9)


@pytest.mark.parametrize("pinned,dtype", [(True, "bf16x3"), (False, "bf16x3"), (True, "f16x2")])
def test_headline_16_seeds_bf16x3_against_solo_runs(gpu, pinned, dtype):
    """The bench configuration (16 seeds x 4096 envs, bf16x3, hipGraph replay) for 2 updates against the solo runs of seeds
    0 / 7 / 15.  The batch trains through the position-parallel kernels (one workgroup per 256 samples / per 8 conv
    positions: they need >= 10 seeds to fill the chip); a solo run by default takes the single-tile kernels (256 workgroups of
    16 samples) -- other summation orders, so the two agree to f32 rounding amplified by RAdam, stated here as a bound.
    SEED_BATCH_BIT_IDENTICAL=True takes the form from the minibatch size alone in BOTH runs (the solo run is then slow:
    32 workgroups) and they are bit-identical: metrics, parameters, optimizer state, env state.  pqn_minatar.py:459-461."""
    from purejaxql_amd import _lib
    from purejaxql_amd.pqn import make_train, seed_keys, vmap_train
    cfg = _cfg(2, dtype=dtype, SEED_BATCH_BIT_IDENTICAL=pinned)
    keys = seed_keys(0, S)
    outs = vmap_train(make_train(dict(cfg), device="cuda:0"), keys)
    assert _lib.last_kernel_form() == ("pos", "pos")
    rs = outs["runner_state"]
    assert len(rs) == S and rs[0]["seed_batch"] == S and rs[0]["driver"] == "graph", rs[0]["driver_graph_error"]
    for s in (0, 7, 15):
        solo = make_train(dict(cfg), device="cuda:0")(keys[s])
        assert _lib.last_kernel_form() == (("pos", "pos") if pinned else ("single", "single"))
        if pinned:
            for k in ("td_loss", "qvals", "returned_episode_returns", "returned_episode_lengths", "returned_episode", "timestep"):
                assert torch.equal(outs["metrics"][k][s], solo["metrics"][k]), (s, k)
            assert torch.equal(rs[s]["theta"], solo["runner_state"]["theta"]), s
            assert torch.equal(rs[s]["opt_mu"], solo["runner_state"]["opt_mu"][:rs[s]["opt_mu"].numel()]), s
            assert torch.equal(rs[s]["env_state"], solo["runner_state"]["env_state"]), s
        else:
            # first update: same parameters, same rollout -- identical data, td_loss / qvals to f32 rounding; after two
            # updates the parameters sit within a few learning-rate steps of each other (RAdam is scale-free: an entry whose
            # gradient is rounding noise moves by +-lr in either run; 128 optimizer steps)
            for k in ("td_loss", "qvals"):
                a0, b0 = float(outs["metrics"][k][s][0]), float(solo["metrics"][k][0])
                assert abs(a0 - b0) <= 1e-4 * max(1.0, abs(b0)), (s, k, a0, b0)
            d = (rs[s]["theta"] - solo["runner_state"]["theta"]).abs()
            assert float(d.max()) <= 16 * cfg["LR"] and float(d.mean()) <= 0.2 * cfg["LR"], (s, float(d.max()), float(d.mean()))
    assert not torch.equal(outs["metrics"]["td_loss"][0], outs["metrics"]["td_loss"][1])


@pytest.mark.parametrize("tail", ["graph", "eager"])
def test_seed_groups_pipeline_is_bit_identical_to_one_batch(gpu, tail):
    """pqn_cnn_update_seed_groups (two groups of 8 seeds, the tails of one group on a second stream under the training
    kernel of the other -- the default of the bench configuration) against ONE 16-seed batch through
    pqn_cnn_update_seeds: every seed, every metric, parameters, optimizer state and env state bit for bit, as a replayed
    two-branch hipGraph and as the eager two-stream enqueue; the graph is re-captured when pqn_options_epoch moves between two
    updates.  (vmap over seeds, pqn_minatar.py:459-461.)"""
    from purejaxql_amd import _lib
    from purejaxql_amd.pqn import make_train, seed_keys
    keys = seed_keys(3, S)

    def run(groups):
        # (pinned form: a group of 8 seeds would otherwise fall back to the pair kernels and no longer equal the 16-seed batch)
        cfg = _cfg(3, SEED_GROUPS=groups, _SEED_GROUPS_TAIL=tail, SEED_BATCH_BIT_IDENTICAL=True)
        update, finish = make_train(cfg, device="cuda:0").make_batch_runner(keys)
        update(0); update(1)
        g1 = update.driver.graph
        # an option changes mid-run (one that leaves the results alone): every driver -- SeedGroupsDriver included, ADVICE r4 --
        # drops its graph and captures again, so that options held by value in the captured launches follow pqn_set_option
        prev = _lib.get_option("peer_timeout_s")
        try:
            _lib.set_option("peer_timeout_s", prev + 1)
            update(2)
        finally:
            _lib.set_option("peer_timeout_s", prev)
        torch.cuda.synchronize()
        assert (g1 is None and update.driver.graph is None) or update.driver.graph is not g1
        return finish(), update.driver

    one, d1 = run(1)
    two, d2 = run(2)
    assert _lib.last_kernel_form() == ("pos", "pos")
    assert type(d2).__name__ == "SeedGroupsDriver" and len(d2.drivers) == 2 and type(d1).__name__ == "SeedsUpdateDriver"
    assert two[0]["runner_state"]["seed_groups"] == 2 and two[0]["runner_state"]["seed_batch"] == S
    if tail == "graph":
        assert d2.graph is not None, d2.graph_error
    else:
        assert d2.graph is None
    for s in range(S):
        a, b = one[s], two[s]
        for k in a["metrics"]:
            assert torch.equal(a["metrics"][k], b["metrics"][k]), (s, k)
        for k in ("theta", "opt_mu", "opt_nu", "env_state", "opt_count"):
            assert torch.equal(a["runner_state"][k], b["runner_state"][k]), (s, k)


_ORACLE_UPDATES = {}   # (game, seed key) -> the oracle's whole update from net.init(123): metrics, theta, float64 learn phase

@pytest.mark.parametrize("env_name,seeds_checked,dtype", [("Breakout-MinAtar", (0, 7, 15), "bf16x3"), ("SpaceInvaders-MinAtar", (7,), "bf16x3"),
                                                         ("Freeway-MinAtar", (7,), "bf16x3"), ("Asterix-MinAtar", (7,), "bf16x3"),
                                                         ("Breakout-MinAtar", (0, 15), "f16x2"), ("SpaceInvaders-MinAtar", (7,), "f16x2"),
                                                         ("Freeway-MinAtar", (7,), "f16x2"), ("Asterix-MinAtar", (7,), "f16x2"),
                                                         ("Breakout-MinAtar", (0, 7), "f16x2/8"), ("Breakout-MinAtar", (0,), "f16x2/4"),   # (seeds the 16-seed cases already ran through the oracle)
                                                         ("SpaceInvaders-MinAtar", (7,), "f16x2/8")])
def test_headline_whole_update_vs_oracle(gpu, oracle, env_name, seeds_checked, dtype):
    """ONE whole update of the bench workload -- 16 seeds batched into the launches, bf16x3, pair rollout + position-parallel
    training kernels -- against oracle.make_train, for seeds 0 / 7 / 15 of Breakout (first, middle and last XCD group) and
    seed 7 of the other three games of bench.py's minatar_suite (C = 6 / 7 channels, 4 / 3 / 5 actions), from shared
    initial parameters: metrics to 1e-3, the update vector by the size-aware criterion of
    test_make_train_end_to_end_vs_oracle (cosine > 0.998, relative L2 < 6e-2, < 1 % of entries outside rtol 2e-3, worst
    entry < 4 lr; backed in the benched operand mode and kernel form by the same-theta trajectory test of
    tests/test_fullsize_gpu.py).  dtype "f16x2/8", "f16x2/4": the same with 8 / 4 seeds in the launches -- the finer cuts of round 6
    (4- / 2-wave forward and rollout workgroups, 4 / 8 backward chunks: pos_plan) are what runs there.
    (The numpy-f32 oracle is itself 3.9e-2 from float64; which seeds pass its 1 %-of-entries criterion is the ORACLE's noise: SpaceInvaders seed 2
    has 3.5 % of its entries outside rtol 2e-3 with bf16x3 / 16 seeds, f16x2 / 16 and f16x2 / 8 alike -- cosine 0.998883 in all three.)"""
    from purejaxql_amd import _lib
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.pqn import make_train, seed_keys
    dtype, _, ns = dtype.partition("/")
    ns = int(ns) if ns else S
    cfg = _cfg(1, env=env_name, dtype=dtype)
    ocfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
    c, a = GAMES[env_name]
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    theta0 = net.init(123)
    cfg["_INIT_PARAMS"] = theta0
    keys = seed_keys(0, ns)
    train = make_train(cfg, device="cuda:0")
    update, finish = train.make_batch_runner(keys)
    update(0)
    outs = finish()
    assert _lib.last_kernel_form() == ("pos", "pos") and cfg["NUM_UPDATES"] == 1
    otrain = oracle.make_train(ocfg)
    th0 = _np(theta0)
    for s in seeds_checked:
        # the oracle's update of (game, seed key) does not depend on the operand mode or on how many seeds share the GPU launches: computed
        # once per session (15 s of numpy each, + 15 s for the float64 learn phase) and shared by the parametrisations
        want64 = env_name == "Breakout-MinAtar" and s == seeds_checked[0]
        ck = (env_name, int(keys[s]))
        if ck not in _ORACLE_UPDATES or (want64 and _ORACLE_UPDATES[ck]["th64"] is None):
            oo = otrain(keys[s], th0)
            th64_c = None
            if want64:
                import pqn_oracle_f64 as o64
                sh = oo["shards"][0]
                th64_c = o64.learn_phase(ocfg, otrain.shapes, th0, sh["of"], sh["af"], sh["tf"], oracle.fold_in(int(keys[s]) & 0xFFFFFFFFFFFFFFFF, 4))[0]
            _ORACLE_UPDATES[ck] = {"metrics": oo["metrics"], "theta": oo["theta"], "th64": th64_c, "th0": th0.copy()}
            del oo
        oout = _ORACLE_UPDATES[ck]
        assert np.array_equal(oout["th0"], th0)
        om = oout["metrics"][0]
        m = outs[s]["metrics"]
        for k in ("env_step", "update_steps", "grad_steps"):
            assert float(m[k][0]) == om[k], (s, k)
        for k in ("td_loss", "qvals", "returned_episode_returns", "returned_episode_lengths", "timestep", "returned_episode",
                  "discount"):
            assert abs(float(m[k][0]) - om[k]) <= 1e-3 * max(1.0, abs(om[k])), (s, k, float(m[k][0]), om[k])
        th, oth = _np(outs[s]["runner_state"]["theta"]), oout["theta"]
        d = np.abs(th - oth)
        bad = d > (2e-5 + 2e-3 * np.abs(oth))
        upd, oupd = th - th0, oth - th0
        cos = float(np.dot(upd, oupd) / (np.linalg.norm(upd) * np.linalg.norm(oupd)))
        rel = float(np.linalg.norm(upd - oupd) / np.linalg.norm(oupd))
        assert np.isfinite(oth).all() and np.isfinite(th).all()
        print(f"\n[{dtype}/{ns}] {env_name} seed {s}: cos {cos:.6f} rel {rel:.4f} bad {float(bad.mean()):.4f} dmax/lr {float(d.max()) / cfg['LR']:.2f}")
        assert cos > 0.998 and rel < 6e-2 and bad.mean() < 1e-2 and d.max() < 4 * cfg["LR"], (s, cos, rel, float(bad.mean()), float(d.max()))
        if env_name == "Breakout-MinAtar" and s == seeds_checked[0]:
            # Which side is nearer to exact arithmetic (VERDICT r5 weak point 2)?  The 64 optimizer steps in FLOAT64 on the oracle's own
            # rollout record (oracle/pqn_oracle_f64.py): the kernels' update must be at least as near to it as the numpy-f32 oracle's
            # is, and inside the band the f32 oracle itself keeps against it.
            th64 = oout["th64"]
            u64 = th64 - th0
            rel_hip = float(np.linalg.norm(upd - u64) / np.linalg.norm(u64))
            rel_np = float(np.linalg.norm(oupd - u64) / np.linalg.norm(u64))
            print(f"\n[{dtype}] update vector vs the float64 learn phase: HIP rel-L2 {rel_hip:.3e}, numpy-f32 oracle rel-L2 {rel_np:.3e}, HIP vs numpy-f32 {rel:.3e}")
            # measured (round 6, profiles/r06_v4_f64_learn_phase.txt): HIP 2.1e-4, numpy-f32 oracle 3.9e-2 -- the numpy oracle is the
            # outlier (its f32 matmuls sum 4096-sample columns in one f32 chain), not the kernels.  Against the float64 reference the
            # criterion is 30x tighter than the 6e-2 the f32-vs-f32 comparison above has to allow.
            cos64 = float(np.dot(upd, u64) / (np.linalg.norm(upd) * np.linalg.norm(u64)))
            d64 = np.abs(th - th64)
            # f16x2 (measured 2.4e-3, cosine 0.999997): where the f32 fma-chain kernels of this library sit (2.4e-3, tests/test_parity_gpu.py,
            # bound 6e-3) -- 64 RAdam steps amplify WHICH roundings happen, not only how large they are; a single gradient of the mode is
            # nearer to float64 than bf16x3's (tests/test_qnet_gpu.py::test_operand_modes_of_the_position_form_against_float64)
            lim, cmin = (2e-3, 0.999995) if dtype == "bf16x3" else (6e-3, 0.99999)
            assert np.isfinite(th64).all() and rel_hip < lim and cos64 > cmin and rel_hip < 0.2 * rel_np, (rel_hip, rel_np, rel, cos64)
            assert (d64 > 2e-5 + 2e-3 * np.abs(th64)).mean() < 1e-4 and d64.max() < cfg["LR"], (float(d64.max()),)


def test_f16x2_update_leaves_both_plane_sets_of_the_fc1_kernel_valid(gpu):
    """An f16x2 layout carries the fc1 kernel three times: the f32 parameters, six bf16 planes (every kernel form but the
    position-parallel one) and four fp16 planes (the position-parallel kernels).  Inside an update only the fp16 set follows the
    optimizer steps (pqn_update.hip upd_apply: copy_mode 4); the update's LAST step rewrites both.  After two updates of the bench shape
    (the second a hipGraph replay) every seed's planes must be exactly what pqn_qnet_cnn_pack_w1b derives from its parameters -- i.e.
    whatever runs between updates (evaluation rollouts in the 16-env form, cnn_forward on the returned parameters) reads current
    weights -- and a bf16x3-form forward on the driver's buffer equals the forward on a freshly packed copy."""
    from purejaxql_amd import _lib
    from purejaxql_amd.pqn import make_train, seed_keys
    from purejaxql_amd.qnet import cnn_forward
    cfg = _cfg(2, dtype="f16x2")
    train = make_train(cfg, device="cuda:0")
    update, finish = train.make_batch_runner(seed_keys(0, S))
    update(0)
    update(1)
    torch.cuda.synchronize()
    assert _lib.last_kernel_form() == ("pos", "pos")
    drv = update.driver
    lay = drv.layout
    assert lay.pos_f16x2 and lay.alloc == lay.total + 5 * 1024 * 128
    rng = np.random.default_rng(3)
    obs = (rng.random((64, 10, 10, 4)) < 0.12).astype(np.float32)
    bits = torch.from_numpy(_pack_bits(obs).view(np.int32)).to(gpu)
    for s in (0, 9, 15):
        tk = drv.theta[s, :lay.alloc].clone()
        fresh = tk.clone()
        fresh[lay.total:] = 0
        lay.refresh_copies(fresh)
        torch.cuda.synchronize()
        assert torch.equal(tk[:lay.total], fresh[:lay.total])
        assert torch.equal(tk[lay.total:].view(torch.int32), fresh[lay.total:].view(torch.int32)), s     # all ten planes, bit for bit
        q0, _, _ = cnn_forward(lay, bits, tk)
        q1, _, _ = cnn_forward(lay, bits, fresh)
        assert torch.equal(q0, q1) and bool(torch.isfinite(q0).all())


@pytest.mark.parametrize("nb,seeds", [(4096, 16), (512, 2), (272, 3), (16, 1)])
def test_fc1_weight_gradient_without_split_k_partials_is_bit_identical(gpu, nb, seeds):
    """qnet_fc1_wgrad_x3_kernel<true> (round 4: one workgroup per (row block, seed) walks over every 256-sample slab and
    adds the per-slab tiles in slab order -- no split-K partials in HBM) against the partial-slab form + the reduction's
    fold: the whole flat gradient bit for bit, at the bench's launch shape (16 seeds x 4096 samples, where it is the
    default), at a ragged minibatch (272 = one slab + one tile: the zeroed tail of the dz planes) and at one tile, from a
    NaN-poisoned workspace.  fc1 part of value_and_grad(_loss_fn), pqn_minatar.py:271-291."""
    from purejaxql_amd import _lib
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, cnn_grad_seeds, matmul_mode
    rng = np.random.default_rng(nb + seeds)
    c, a, rows = 4, 3, 6000
    obs = (rng.random((rows, 10, 10, c)) < 0.12).astype(np.float32)
    bits = torch.from_numpy(_pack_bits(obs).view(np.int32)).to(gpu)
    action = torch.from_numpy(rng.integers(0, a, rows).astype(np.int32)).to(gpu)
    target = torch.from_numpy(rng.standard_normal(rows).astype(np.float32)).to(gpu)
    net = QNetwork("cnn", (10, 10, c), a, device=gpu)
    lay = CnnKernelLayout(c, a, matmul_f16=matmul_mode("bf16x3"))
    stride = (lay.alloc + 3) // 4 * 4
    theta_k = torch.zeros((seeds, stride), dtype=torch.float32, device=gpu)
    for s in range(seeds):
        th = lay.to_kernel(net.init(40 + s) + 0.05 * torch.randn(net.num_params, device=gpu))
        theta_k[s, :th.numel()] = th
    idx = torch.from_numpy(np.stack([rng.permutation(rows)[:nb] for _ in range(seeds)]).astype(np.int64)).to(gpu).contiguous()
    out = {}
    for acc in (0, 2, 1):
        with _lib.options(t2_acc=acc):
            g, lo, qv = cnn_grad_seeds(lay, theta_k, idx, bits, action, target, rows, _ws_fill=float("nan"))
        torch.cuda.synchronize()
        out[acc] = (g.clone(), lo.clone(), qv.clone())
        assert torch.isfinite(g).all() and torch.isfinite(lo).all()
    for acc in (2, 1):
        for x, y in zip(out[0], out[acc]):
            assert torch.equal(x, y), (nb, seeds, acc)
    assert float(out[0][0][:, lay.struct.off_w1:lay.struct.off_w1 + 1024 * 128].abs().max()) > 0


def test_kernel_selection_options_reach_a_running_graph_driver(gpu):
    """A captured hipGraph replays the kernels chosen at capture time; the update drivers therefore watch pqn_options_epoch and
    re-capture when a kernel-selection option changed (an option set after the capture used to be ignored while
    pqn_cnn_last_kernel_form kept reporting the last ENQUEUE).  Small Breakout run, bf16x3: pair form forced, two updates (the
    second captured), then the option flips to the single-tile form mid-run -- the very next update is enqueued and captured
    again in that form -- and back; the metrics equal an undisturbed run bit for bit (the two forms sum in the same order)."""
    from purejaxql_amd import _lib
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train, seed_keys

    def cfg():
        c = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", "alg.NUM_ENVS=64", "alg.TEST_DURING_TRAINING=False",
                                 "alg.MATMUL_DTYPE=bf16x3"]))
        c["TOTAL_TIMESTEPS"] = 5 * 64 * c["NUM_STEPS"]
        return c

    key = seed_keys(9, 1)[0]
    prev = _lib.get_option("t1_pair")
    try:
        _lib.set_option("t1_pair", 2)
        ref_update, ref_finish = make_train(cfg(), device="cuda:0").make_runner(key)
        for u in range(5):
            ref_update(u)
        ref = ref_finish()
        update, finish = make_train(cfg(), device="cuda:0").make_runner(key)
        update(0); update(1)
        assert update.driver.graph is not None and _lib.last_kernel_form()[0] == "pair"
        g1 = update.driver.graph
        _lib.set_option("t1_pair", 0)
        update(2)
        assert _lib.last_kernel_form()[0] == "single" and update.driver.graph is not None and update.driver.graph is not g1
        g2 = update.driver.graph
        update(3)
        assert update.driver.graph is g2          # no option changed: plain replay
        _lib.set_option("t1_pair", 2)
        update(4)
        assert _lib.last_kernel_form()[0] == "pair" and update.driver.graph is not g2
        out = finish()
    finally:
        _lib.set_option("t1_pair", prev)
    for k in ("td_loss", "qvals", "returned_episode_returns"):
        assert torch.equal(out["metrics"][k], ref["metrics"][k]), k
    assert torch.equal(out["runner_state"]["theta"], ref["runner_state"]["theta"])


def test_forced_pair_forms_at_small_sizes_in_process(gpu):
    """The pair forms forced at sizes where they are not selected by default (pqn_set_option, in-process): the training
    kernel at 2, 8 and 256 pairs (C = 4), and -- round 4, head-parameter block of the LDS plan sized by the action count --
    at C = 6 (SpaceInvaders' shape) and C = 7 (Freeway's); C = 10 with 6 actions still exceeds the 160 KB and must run the
    single-tile kernel.  Repeats bit-identical, equal to the single-tile bf16x3 kernel bit for bit and to the f32-MFMA mode
    to f32 rounding."""
    from purejaxql_amd import _lib
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer
    for c, a, nb, pool in ((4, 3, 64, 256), (4, 3, 256, 1000), (4, 3, 8192, 20000), (6, 4, 1024, 2000), (7, 3, 512, 1000),
                           (4, 5, 512, 1000), (10, 6, 512, 1000)):
        rng = np.random.default_rng(nb + c)
        torch.manual_seed(1234)
        net = QNetwork("cnn", (10, 10, c), a, device=gpu)
        theta = net.init(11) + 0.05 * torch.randn(net.num_params, device=gpu)
        obs = (rng.random((pool, 10, 10, c)) < 0.12).astype(np.float32)
        bits = torch.from_numpy(_pack_bits(obs).view(np.int32)).to(gpu)
        action = torch.from_numpy(rng.integers(0, a, pool).astype(np.int32)).to(gpu)
        target = torch.from_numpy(rng.standard_normal(pool).astype(np.float32)).to(gpu)
        idx = torch.from_numpy(rng.permutation(pool)[:nb].astype(np.int64)).to(gpu)
        res = {}
        for name, mode, pair in (("f32", 0, 0), ("single", 2, 0), ("pair", 2, 2)):
            with _lib.options(t1_pair=pair):
                lay = CnnKernelLayout(c, a, matmul_f16=mode)
                tr = CnnTrainer(lay, theta, 5e-4, 10.0, max_minibatch=nb)
                reps = [tr.compute_grad(idx, bits, action, target)[:lay.total].clone() for _ in range(3)]
                # C = 10: the pair kernel's LDS plan exceeds the 160 KB of a CU, so the switch cannot select it
                # (and the f32 operand mode takes its K-split form at minibatches of at most 256 samples)
                want = "pair" if (pair and c != 10) else ("ksplit" if (mode == 0 and nb <= 256) else "single")
                assert _lib.last_kernel_form()[0] == want, (name, c, nb)
                assert torch.equal(reps[0], reps[1]) and torch.equal(reps[0], reps[2]), (name, c, nb)
                res[name] = reps[0]
        assert torch.equal(res["single"], res["pair"]), (c, nb)
        scale = float(res["f32"].abs().max())
        assert float((res["f32"] - res["pair"]).abs().max()) <= 2e-5 * scale, (c, nb)


def test_forced_rollout_pair_form_is_bit_identical_in_process(gpu):
    from purejaxql_amd import _lib
    from purejaxql_amd.envs import LogWrapper, make
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, cnn_rollout, matmul_mode
    lib = _lib.load()
    for name, c, a, n, t in (("Breakout-MinAtar", 4, 3, 64, 40), ("Asterix-MinAtar", 4, 5, 96, 30), ("SpaceInvaders-MinAtar", 6, 4, 32, 20)):
        env, params = make(name, device=gpu)
        env = LogWrapper(env)
        net = QNetwork("cnn", (10, 10, c), a, device=gpu)
        lay = CnnKernelLayout(c, a, matmul_f16=matmul_mode("bf16x3"))
        torch.manual_seed(0)
        theta_k = lay.to_kernel(net.init(3) + 0.05 * torch.randn(net.num_params, device=gpu))
        (_o, bits0), state = env.reset(11, params, n, want_obs=False, want_bits=True)
        for i in range(30):
            (_o, bits0), state, *_ = env.step(500 + i, state, torch.randint(0, a, (n,), dtype=torch.int32, device=gpu), params,
                                              want_obs=False, want_bits=True)
        keys = torch.empty(t, dtype=torch.int64, device=gpu)
        _lib.check(lib.pqn_fold_in_range(0x77, 7, t, _lib.ptr(keys), _lib.stream_ptr()), "pqn_fold_in_range")
        eps = torch.full((1,), 0.3, dtype=torch.float32, device=gpu)
        outs = []
        for pair in (0, 2):
            with _lib.options(rollout_pair=pair):
                words = state.words.clone()
                ob = torch.zeros((t + 1, n, bits0.shape[1]), dtype=bits0.dtype, device=gpu)
                ob[0] = bits0
                rec = cnn_rollout(lay, env._env.env_id, words, ob, theta_k, keys, eps)
                assert _lib.last_kernel_form()[1] == ("pair" if pair else "single")
                outs.append((rec, words, ob))
        for k in outs[0][0]:
            assert torch.equal(outs[0][0][k], outs[1][0][k]), (name, k)
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2]), name
        assert int(outs[0][0]["done"].sum()) > 0, name
