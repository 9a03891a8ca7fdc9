"""GPU tests at BASELINE.json's full shape (Breakout-MinAtar, 4096 envs x 32 steps, minibatch 4096) through
size-independent properties, plus error-path / ragged-shape checks of the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _rollout_bits(gpu, n, t):
    from purejaxql_amd.envs import make
    env, params = make("Breakout-MinAtar", device=gpu)
    (_, bits), state = env.reset(1, params, n, want_bits=True, want_obs=False)
    out, rewards, dones = [], [], []
    g = torch.Generator(device=gpu)
    g.manual_seed(0)
    for i in range(t + 20):
        a = torch.randint(0, 3, (n,), dtype=torch.int32, device=gpu, generator=g)
        (_, bits), state, r, d, _ = env.step(100 + i, state, a, params, want_bits=True, want_obs=False, inplace=True)
        if i >= 20:
            out.append(bits)
            rewards.append(r)
            dones.append(d.view(torch.uint8))
    return torch.stack(out), torch.stack(rewards), torch.stack(dones)


def test_full_size_gradient_is_linear_in_the_minibatch(gpu):
    """grad(minibatch of 4096) == mean of the gradients of its two halves (loss is a mean over samples):
    exercises every kernel of the optimizer step at the bench shape, incl. the 16-slab split-K reduction."""
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer
    n, t = 4096, 32
    bits, _, _ = _rollout_bits(gpu, n, t)
    bits = bits.reshape(n * t, -1).contiguous()
    net = QNetwork("cnn", (10, 10, 4), 3, device=gpu)
    lay = CnnKernelLayout(4, 3)
    tr = CnnTrainer(lay, net.init(3) + 0.02 * torch.randn(net.num_params, device=gpu), 5e-4, 10.0)
    g = torch.Generator(device=gpu)
    g.manual_seed(1)
    idx = torch.randperm(n * t, device=gpu, generator=g)[:4096].contiguous()
    act = torch.randint(0, 3, (n * t,), dtype=torch.int32, device=gpu, generator=g)
    tgt = torch.randn(n * t, device=gpu, generator=g)
    loss = torch.zeros(3, device=gpu)
    g_full = tr.compute_grad(idx, bits, act, tgt, loss[0:1]).clone()
    g_a = tr.compute_grad(idx[:2048].contiguous(), bits, act, tgt, loss[1:2]).clone()
    g_b = tr.compute_grad(idx[2048:].contiguous(), bits, act, tgt, loss[2:3]).clone()
    ref = 0.5 * (g_a + g_b)
    scale = float(ref.abs().max())
    # f32 with different summation trees: 1e-5 of the largest gradient entry
    assert float((g_full - ref).abs().max()) <= 1e-5 * scale
    assert abs(float(loss[0]) - 0.5 * float(loss[1] + loss[2])) <= 1e-5 * max(1.0, float(loss[0]))
    # permuting the minibatch changes nothing but the summation order
    g_perm = tr.compute_grad(idx.flip(0).contiguous(), bits, act, tgt).clone()
    assert float((g_full - g_perm).abs().max()) <= 1e-5 * scale


def test_full_size_forward_is_per_sample(gpu):
    """q of a sample does not depend on its batch neighbours or position (tiles of 16, 4096 rows)."""
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, cnn_forward
    bits, _, _ = _rollout_bits(gpu, 4096, 1)
    bits = bits[0].contiguous()
    net = QNetwork("cnn", (10, 10, 4), 3, device=gpu)
    lay = CnnKernelLayout(4, 3)
    th = lay.to_kernel(net.init(9))
    q, _, _ = cnn_forward(lay, bits, th)
    perm = torch.randperm(4096, device=gpu)
    qp, _, _ = cnn_forward(lay, bits[perm].contiguous(), th)
    # the fc1 K order is rotated per workgroup, so rows agree to f32 rounding, not bitwise
    torch.testing.assert_close(qp, q[perm], rtol=1e-5, atol=1e-6)
    q1, _, _ = cnn_forward(lay, bits[:33].contiguous(), th)          # ragged tail tile
    torch.testing.assert_close(q1, q[:33], rtol=1e-5, atol=1e-6)


def test_full_size_q_lambda_properties(gpu, oracle):
    from purejaxql_amd import ops
    _, r, d = _rollout_bits(gpu, 4096, 32)
    r, d = r.contiguous(), d.contiguous()
    qm = torch.randn(32, 4096, device=gpu)
    lq = torch.randn(4096, device=gpu)
    tgt = ops.q_lambda(r, d, qm, lq, 0.99, 0.65)
    np.testing.assert_array_equal(_np(tgt), oracle.q_lambda(_np(r), _np(d), _np(qm), _np(lq), 0.99, 0.65))
    # linearity in (reward, qmax, last_q) for a fixed done pattern
    t2 = ops.q_lambda(2 * r, d, 2 * qm, 2 * lq, 0.99, 0.65)
    torch.testing.assert_close(t2, 2 * tgt, rtol=1e-6, atol=1e-6)
    # lambda = 0 is the 1-step target; an episode end cuts the bootstrap
    t0 = ops.q_lambda(r, d, qm, lq, 0.99, 0.0)
    nxt = torch.cat([qm[1:], lq[None]], 0)
    nxt[30] = lq * (1 - d[31].float())      # the reference's T-2 quirk
    torch.testing.assert_close(t0, r + 0.99 * (1 - d.float()) * nxt, rtol=1e-6, atol=1e-6)
    assert torch.equal(tgt[d.bool()], r[d.bool()])


def test_full_size_shuffle_is_a_bijection_and_update_is_deterministic(gpu):
    from purejaxql_amd import ops
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train, seed_keys
    p = ops.shuffle_permutation(77, 131072, gpu)
    assert torch.equal(torch.sort(p).values, torch.arange(131072, device=gpu))
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", "alg.NUM_ENVS=4096",
                               "alg.TEST_DURING_TRAINING=False"]))
    cfg["TOTAL_TIMESTEPS"] = 3 * 4096 * 32
    outs = [make_train(dict(cfg), device="cuda:0")(seed_keys(5, 1)[0]) for _ in range(2)]
    assert outs[0]["runner_state"]["driver"] == "graph"
    for k in ("td_loss", "qvals", "returned_episode_returns", "returned_episode"):
        assert torch.equal(outs[0]["metrics"][k], outs[1]["metrics"][k]), k      # hipGraph replay is bit-reproducible
    assert torch.equal(outs[0]["runner_state"]["theta"], outs[1]["runner_state"]["theta"])
    m = outs[0]["metrics"]
    assert m["env_step"].tolist() == [131072.0, 262144.0, 393216.0] and m["grad_steps"].tolist() == [64.0, 128.0, 192.0]
    assert abs(float(m["timestep"][2]) - (64 + 16.5)) < 1e-3 and float(m["returned_episode"][0]) > 0.05


@pytest.mark.parametrize("impl", [2, 0, 512])   # 512: the library's sort with its in-LDS capacity lowered to 512 keys -- every bucket takes the rank-sort path
@pytest.mark.parametrize("n_envs,seeds", [(128, 1), (112, 1), (4096, 1), (1008, 2), (4096, 16), (32768, 4)])
def test_update_permutation_equals_the_full_width_sort_of_its_keys(gpu, n_envs, seeds, impl):
    """The epoch shuffle's sort (pqn_update.hip pqn_sort_keys; jax.random.permutation's stand-in, pqn_minatar.py:299-315) must leave the
    permutation the full-width sort of the unique keys gives.  impl 2 (the default rule takes it above 4096 keys per seed): the library's own sort -- one workgroup per seed in LDS up
    to 4096 keys (3584 = not a power of two), above that buckets by the leading random bits + a workgroup per bucket (B = 32 ... 1024).
    impl 0: rocPRIM's radix sort over the random + seed bits only, which relies on stability for transitions that drew the same 31 bits
    (~4 pairs per seed at 131,072 transitions, ~256 at 2^20) -- on its merge-sort path (one seed) and its onesweep path (seed batches)."""
    from purejaxql_amd import _lib
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train, seed_keys
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", f"alg.NUM_ENVS={n_envs}", "alg.TEST_DURING_TRAINING=False"]))
    cfg["TOTAL_TIMESTEPS"] = 3 * n_envs * 32
    if impl == 512 and n_envs * 32 * seeds > (1 << 18):
        pytest.skip("the rank-sort path is quadratic: small shapes only")
    with _lib.options(sort_impl=2 if impl else 0, sort_cap=impl if impl > 2 else 0):
        tr = make_train(cfg, device="cuda:0")
        upd, _ = tr.make_batch_runner(seed_keys(3, seeds)) if seeds > 1 else tr.make_runner(seed_keys(3, 1)[0])
        upd(0)
        torch.cuda.synchronize()
    drv = upd.driver
    tn = n_envs * 32
    mask = (1 << max(1, (tn - 1).bit_length())) - 1
    k_in, k_out = drv.sk_in.clone(), drv.sk_out.clone()
    assert k_in.numel() == seeds * tn
    ref = torch.sort(k_in).values
    same_random_bits = int(((ref[1:] >> (tn - 1).bit_length()) == (ref[:-1] >> (tn - 1).bit_length())).sum())
    if tn * seeds >= 1 << 21:
        assert same_random_bits > 0        # the case the stability argument is about occurs at this size
    assert torch.equal(k_out & mask, ref & mask), same_random_bits
    if seeds > 1:   # every seed's segment is a permutation of its own transitions
        seg = (k_out & mask).view(seeds, tn)
        assert torch.equal(torch.sort(seg, dim=1).values, torch.arange(tn, device=gpu).expand(seeds, tn))


@pytest.mark.parametrize("env_name,n_envs,seeds,mode,form", [
    ("Breakout-MinAtar", 128, 1, "auto", "ksplit"),        # yaml default: K-split kernels, f32 operands, 64 + 8 partial records
    ("Breakout-MinAtar", 1024, 1, "auto", "single"),       # C2: single-tile bf16x3 kernels, split-K slabs of the fc1 weight gradient
    ("Breakout-MinAtar", 4096, 2, "bf16x3", "pair"),       # pair form, accumulating fc1 weight gradient or slabs
    ("Breakout-MinAtar", 4096, 4, "f16x2", "pos"),         # position-parallel form: chunk slabs + position records, fp16 planes only until the last step
    ("SpaceInvaders-MinAtar", 1024, 3, "f32", "single"),   # other channel / action counts: another parameter layout, w1b mirror without planes
    ("Freeway-MinAtar", 512, 2, "f16", "single"),          # fp16 operand copies
])
def test_one_launch_fold_clip_radam_is_bit_identical_to_the_two_launch_form(gpu, env_name, n_envs, seeds, mode, form):
    """Option fold_apply (2 = at any launch size, 1 = the default rule; pqn_fold.h): the optimizer kernel of a fused update folds the gradient partials itself -- same blocks,
    loads and order of additions as qnet_grad_reduce_kernel + radam_apply_kernel (optax.chain(clip_by_global_norm, radam),
    pqn_minatar.py:159-162,293-297): parameters, both moments, the step count, the metrics rows and the last minibatch's gradient
    must be bit-identical after 3 updates (192 optimizer steps, the last 2 updates as hipGraph replays) in every kernel form."""
    from purejaxql_amd import _lib
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train, seed_keys
    outs = []
    for fold in (0, 2):
        with _lib.options(fold_apply=fold):
            cfg = flatten(load_config(["+alg=pqn_minatar", f"alg.ENV_NAME={env_name}", f"alg.NUM_ENVS={n_envs}", "alg.TEST_DURING_TRAINING=False",
                                       f"alg.MATMUL_DTYPE={mode}"]))
            cfg["TOTAL_TIMESTEPS"] = 3 * n_envs * 32
            tr = make_train(cfg, device="cuda:0")
            upd, fin = tr.make_batch_runner(seed_keys(11, seeds)) if seeds > 1 else tr.make_runner(seed_keys(11, 1)[0])
            for u in range(3):
                upd(u)
            torch.cuda.synchronize()
            assert _lib.last_kernel_form()[0] == form
            drv = upd.driver
            tn = drv if seeds > 1 else drv._keep[0]   # SeedsUpdateDriver holds the stacked buffers itself, UpdateDriver its trainer
            outs.append({"theta": tn.theta.clone(), "m": tn.m.clone(), "v": tn.v.clone(), "count": tn.count.clone(), "grad": tn.grad.clone(),
                         "metrics": drv.metrics.clone()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert int(outs[0]["count"].reshape(-1)[0]) == 192 and torch.isfinite(outs[0]["theta"]).all() and float(outs[0]["grad"].abs().max()) > 0


@pytest.mark.parametrize("env_name,n_envs,seeds", [("Breakout-MinAtar", 4096, 4), ("SpaceInvaders-MinAtar", 1024, 16), ("Freeway-MinAtar", 2048, 8)])
def test_grouped_epoch_gather_moves_the_same_bytes(gpu, env_name, n_envs, seeds):
    """pos_gather_kernel<C, 4> (four 32-sample super-tiles per workgroup, all key and row loads requested before any is consumed) against
    <C, 1>: the gathered rows / bit transposes / actions / targets feed the position-parallel kernels (the shuffled minibatches of
    pqn_minatar.py:299-320), so parameters and metrics after 2 updates must be bit-identical (13 / 20 / 24 packed words per row)."""
    from purejaxql_amd import _lib
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train, seed_keys
    outs = []
    for group in (1, 4):
        with _lib.options(gather_group=group, bwd_pos=2):
            cfg = flatten(load_config(["+alg=pqn_minatar", f"alg.ENV_NAME={env_name}", f"alg.NUM_ENVS={n_envs}", "alg.TEST_DURING_TRAINING=False",
                                       "alg.MATMUL_DTYPE=f16x2"]))
            cfg["TOTAL_TIMESTEPS"] = 2 * n_envs * 32
            upd, _ = make_train(cfg, device="cuda:0").make_batch_runner(seed_keys(21, seeds))
            upd(0); upd(1)
            torch.cuda.synchronize()
            assert _lib.last_kernel_form()[0] == "pos"
            drv = upd.driver
            outs.append((drv.theta.clone(), drv.m.clone(), drv.metrics.clone()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert torch.isfinite(outs[0][0]).all()


def test_c_abi_argument_errors(gpu):
    """Bad arguments are rejected on the host with a negative code and a message (no launch)."""
    from purejaxql_amd import _lib
    from purejaxql_amd.qnet import CnnKernelLayout
    lib = _lib.load()
    lay = CnnKernelLayout(4, 3)
    x = torch.zeros(64, dtype=torch.float32, device=gpu)
    assert lib.pqn_env_step(0, 16, 1, None, None, None, None, None) == -1
    assert b"NULL" in lib.pqn_last_error()
    assert lib.pqn_env_reset(0, 0, 1, x.data_ptr(), None, None, None) == -1
    assert lib.pqn_q_lambda(None, None, None, None, 0.99, 0.65, 4, 4, 1, None, None) == -1
    assert lib.pqn_eps_greedy(x.data_ptr(), 0, 3, 0.1, 1, x.data_ptr(), None, None) == -1
    assert lib.pqn_env_reset(99, 4, 1, x.data_ptr(), None, None, None) == -3          # unknown env
    assert lib.pqn_env_reset(1, 4, 1, x.data_ptr(), None, x.data_ptr(), None) == -1   # CartPole has no packed obs
    bad = _lib.EnvSpec()
    assert lib.pqn_env_spec(42, ctypes.byref(bad)) == -3
    # minibatch sizes must be multiples of the 16-sample MFMA tile
    rc = lib.pqn_qnet_cnn_grad(ctypes.byref(lay.struct), 24, *([x.data_ptr()] * 11), None)
    assert rc == -1 and b"multiple of 16" in lib.pqn_last_error()
    bad_layout = _lib.c_int32(0)
    from purejaxql_amd.qnet import CnnLayoutStruct
    s = CnnLayoutStruct()
    assert lib.pqn_cnn_layout(5, 3, ctypes.byref(s)) == -1 and lib.pqn_cnn_layout(4, 9, ctypes.byref(s)) == -1


def test_c4_per_gpu_share_16_seeds_x_4096_envs_equals_solo_runs(gpu):
    """BASELINE.json configs[3] (128 seeds x 4096 envs over 8 GPUs) = 16 seeds per GPU: all 16 seeds of the headline
    shape advance in the same launches (pqn_cnn_update_seeds, grid.y = seed; pqn_minatar.py:459-461) for 2 updates
    (the second one a hipGraph replay); seeds 0 / 7 / 15 must be bit-identical to their solo runs."""
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train, seed_keys, vmap_train
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", "alg.NUM_ENVS=4096",
                               "alg.TEST_DURING_TRAINING=False", "alg.MATMUL_DTYPE=f32"]))   # (auto = bf16x3 here: batch and solo then take
    cfg["TOTAL_TIMESTEPS"] = 2 * 4096 * 32                                                   #  different kernel forms; test_headline_gpu.py's pinned case)
    keys = seed_keys(0, 16)
    outs = vmap_train(make_train(dict(cfg), device="cuda:0"), keys)
    rs = outs["runner_state"]
    assert len(rs) == 16 and rs[0]["seed_batch"] == 16 and rs[0]["driver"] == "graph", rs[0]["driver_graph_error"]
    assert outs["metrics"]["td_loss"].shape == (16, 2)
    for s in (0, 7, 15):
        solo = make_train(dict(cfg), device="cuda:0")(keys[s])
        for k in ("td_loss", "qvals", "returned_episode_returns", "returned_episode_lengths", "returned_episode", "timestep"):
            assert torch.equal(outs["metrics"][k][s], solo["metrics"][k]), (s, k)
        assert torch.equal(rs[s]["theta"], solo["runner_state"]["theta"]), s
        assert torch.equal(rs[s]["opt_mu"], solo["runner_state"]["opt_mu"][:rs[s]["opt_mu"].numel()]), s
        assert torch.equal(rs[s]["env_state"], solo["runner_state"]["env_state"]), s
    # different seeds are different runs
    assert not torch.equal(outs["metrics"]["td_loss"][0], outs["metrics"]["td_loss"][1])


@pytest.mark.parametrize("mode,opts,form", [("f32", {}, "single"), ("bf16x3", {"bwd_pos": 0}, "single"), ("bf16x3", {"bwd_pos": 2}, "pos")])
def test_full_size_sgd_trajectory_gradients_match_oracle_at_same_theta(gpu, oracle, mode, opts, form):
    """(mode / kernel form: the package default f32-MFMA mode; the bf16x3 mode bench.py reports, through the single-tile
    kernels a solo run takes and -- option bwd_pos = 2 -- through the position-parallel kernels the benched 16-seed launches
    take: the same 64-step same-theta proof in the benched operand mode AND kernel form.)
    The headline shape's optimizer loop (Breakout 4096 envs x 32 steps, 32 minibatches of 4096, 2 epochs), oracle and
    fused HIP kernels side by side on the SAME rollout data and permutations: at every optimizer step the HIP gradient
    evaluated AT THE ORACLE'S parameters equals the oracle's (<= 2e-4 of the largest entry, every one of the 64 steps,
    both epochs), and the free-running HIP trajectory stays within 6e-2 of the oracle's update (cosine > 0.998) --
    the residual is RAdam's amplification of f32 rounding noise in cancelling gradient entries, see
    test_make_train_end_to_end_vs_oracle."""
    from purejaxql_amd import _lib
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.qnet import CnnKernelLayout, CnnTrainer, matmul_mode
    O = oracle
    N, T, MB, EP = 4096, 32, 32, 2
    env = O.OracleEnv("Breakout-MinAtar")
    net = QNetwork("cnn", (10, 10, 4), 3, device=gpu)
    theta0 = net.init(123)
    shapes = O.cnn_shapes((10, 10, 4), 3)
    th = _np(theta0).copy()
    p = O.unflatten(th, shapes)
    obs, st = env.reset(7, N)
    Obs = np.zeros((T + 1, N, 10, 10, 4), np.float32)
    Obs[0] = obs
    A, R = np.zeros((T, N), np.int32), np.zeros((T, N), np.float32)
    D, QM = np.zeros((T, N), bool), np.zeros((T, N), np.float32)
    for t in range(T):        # eps = 1 at update 0 (pqn_minatar.yaml:7): uniformly random actions
        A[t], QM[t] = O.eps_greedy(O.net_forward("cnn", p, Obs[t]), np.float32(1.0), 100 + t)
        Obs[t + 1], st, R[t], D[t], _ = env.step(100 + t, st, A[t])
    tgt = O.q_lambda(R, D, QM, O.net_forward("cnn", p, Obs[T]).max(-1), 0.99, 0.65)
    of, af, tf = Obs[:T].reshape(T * N, 10, 10, 4), A.reshape(-1), tgt.reshape(-1)
    padded = np.zeros((T * N, 512), np.uint64)
    padded[:, :400] = of.reshape(T * N, -1)
    words = (padded.reshape(-1, 16, 32) << np.arange(32, dtype=np.uint64)).sum(-1).astype(np.uint32)
    bits = torch.from_numpy(words.view(np.int32)).to(gpu)
    act_t, tgt_t = torch.from_numpy(af).to(gpu), torch.from_numpy(tf).to(gpu)
    lay = CnnKernelLayout(4, 3, matmul_f16=matmul_mode(mode))
    B, lr_steps = T * N // MB, 30 * MB * EP
    free = CnnTrainer(lay, theta0, 5e-4, 10.0, lr_decay_steps=float(lr_steps), max_minibatch=B)
    sync = CnnTrainer(lay, theta0, 5e-4, 10.0, lr_decay_steps=float(lr_steps), max_minibatch=B)
    m, v = np.zeros_like(th), np.zeros_like(th)
    step, worst = 0, 0.0
    prev = {k: _lib.get_option(k) for k in opts}
    for k, val in opts.items():
        _lib.set_option(k, val)
    try:
        worst = _trajectory(O, lay, free, sync, of, af, tf, bits, act_t, tgt_t, th, p, shapes, m, v, T, N, MB, EP, B, lr_steps, form)
    finally:
        for k, val in prev.items():
            _lib.set_option(k, val)
    upd, oupd = _np(free.theta_flax()) - _np(theta0), th - _np(theta0)
    cos = float(np.dot(upd, oupd) / (np.linalg.norm(upd) * np.linalg.norm(oupd)))
    rel = float(np.linalg.norm(upd - oupd) / np.linalg.norm(oupd))
    assert cos > 0.998 and rel < 6e-2, (cos, rel, worst)


def _trajectory(O, lay, free, sync, of, af, tf, bits, act_t, tgt_t, th, p, shapes, m, v, T, N, MB, EP, B, lr_steps, form):
    from purejaxql_amd import _lib
    gpu = bits.device
    step, worst = 0, 0.0
    for ep in range(EP):
        perm = O.permutation(O.fold_in(99, ep), T * N)
        for mb in range(MB):
            idx = perm[mb * B:(mb + 1) * B]
            _loss, _chosen, g = O.net_loss_grad("cnn", p, shapes, of[idx], af[idx], tf[idx])
            idx_t = torch.from_numpy(idx.astype(np.int64)).to(gpu)
            sync.theta.copy_(lay.to_kernel(torch.from_numpy(th).to(gpu)))
            lay.refresh_copies(sync.theta, sync.w1b)
            g_gpu = _np(lay.to_flax(sync.compute_grad(idx_t, bits, act_t, tgt_t)))
            assert _lib.last_kernel_form()[0] == form
            err = float(np.abs(g_gpu - g).max() / np.abs(g).max())
            worst = max(worst, err)
            assert err <= 2e-4, (ep, mb, err)
            free.compute_grad(idx_t, bits, act_t, tgt_t)
            free.apply()
            O.radam_clip_step(th, g, m, v, step, np.float32(O.linear_schedule(5e-4, 1e-20, lr_steps, step)), 10.0)
            step += 1
    return worst
