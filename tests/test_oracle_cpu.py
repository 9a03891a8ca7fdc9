"""CPU tests (-m "not gpu"): the oracle against the known-answer vectors
(tests/golden/ka_vectors.json), and the oracle's numpy
network against a plain PyTorch fp32 autograd reference."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "ka_vectors.json")))


def test_threefry_known_answers(oracle):
    # Random123 / jax tests/random_test.py known answers for threefry2x32-20
    for v in KA["threefry2x32"]:
        assert list(oracle.threefry2x32(v["key"], v["ctr"])) == v["out"]


def test_fold_in_is_counter_0_d(oracle):
    key = 0x0123456789ABCDEF
    o = oracle.threefry2x32([key >> 32, key & 0xFFFFFFFF], [0, 77])
    assert oracle.fold_in(key, 77) == (o[0] << 32) | o[1]


def test_q_lambda_known_answers(oracle):  # KA1
    ka = KA["q_lambda"]
    r = np.array(ka["reward"], np.float32)[:, None]
    qm = np.array(ka["qmax"], np.float32)[:, None]
    lq = np.array([ka["last_q"]], np.float32)
    for case in ka["cases"]:
        d = np.array(case["done"], np.uint8)[:, None]
        got = oracle.q_lambda(r, d, qm, lq, ka["gamma"], ka["lambda"], quirk=True)[:, 0]
        np.testing.assert_allclose(got, case["minatar"], rtol=1e-6, atol=1e-6)
        got0 = oracle.q_lambda(r, d, qm, lq, ka["gamma"], ka["lambda"], quirk=False)[:, 0]
        np.testing.assert_allclose(got0, case["atari"], rtol=1e-6, atol=1e-6)


def test_q_lambda_affine_form(oracle):
    # [DERIVED] G_t = d ? r : r + g(1-l) nq + g l G_{t+1}
    rng = np.random.default_rng(0)
    T, M = 32, 50
    r = rng.random((T, M)).astype(np.float32)
    d = (rng.random((T, M)) < 0.1).astype(np.uint8)
    qm = rng.standard_normal((T, M)).astype(np.float32)
    lq = rng.standard_normal(M).astype(np.float32)
    g, l = 0.99, 0.65
    got = oracle.q_lambda(r, d, qm, lq, g, l, quirk=True).astype(np.float64)
    exp = np.zeros((T, M))
    lq2 = lq * (1 - d[-1])
    exp[-1] = r[-1] + g * lq2
    nq = lq2.astype(np.float64)
    for t in range(T - 2, -1, -1):
        exp[t] = np.where(d[t] == 1, r[t], r[t] + g * (1 - l) * nq + g * l * exp[t + 1])
        nq = qm[t]
    np.testing.assert_allclose(got, exp, rtol=2e-5, atol=2e-5)


def test_num_updates_and_schedules(oracle):  # KA2, KA3
    from purejaxql_amd.pqn import derive_config, linear_schedule
    for v in KA["num_updates"]:
        cfg = derive_config({"TOTAL_TIMESTEPS": v["total"], "TOTAL_TIMESTEPS_DECAY": v["total"],
                             "NUM_STEPS": v["steps"], "NUM_ENVS": v["envs"], "NUM_MINIBATCHES": 1})
        assert cfg["NUM_UPDATES"] == v["updates"]
    assert int(76 * 0.05) == 3 and int(2441 * 0.05) == 122
    eps = linear_schedule(1.0, 0.05, 0.1 * 2441)
    assert eps(0) == 1.0 and eps(245) == 0.05 and eps(1e9) == 0.05
    assert abs(eps(122.05) - (0.95 * 0.5 + 0.05)) < 1e-12
    for c in (0, 10, 244.1, 300):
        assert abs(eps(c) - oracle.linear_schedule(1.0, 0.05, 244.1, c)) < 1e-12
    lr_n = 2441 * 32 * 2
    assert oracle.linear_schedule(5e-4, 1e-20, lr_n, 0) == 5e-4
    assert oracle.linear_schedule(5e-4, 1e-20, lr_n, lr_n + 5) == 1e-20


def test_log_wrapper_known_answer(oracle):  # KA4
    ka = KA["log_wrapper"]
    n = 1
    st = dict(ep_ret=np.zeros(n, np.float32), ep_len=np.zeros(n, np.int32), ret_ret=np.zeros(n, np.float32),
              ret_len=np.zeros(n, np.int32), timestep=np.zeros(n, np.int32))
    for r, d in zip(ka["reward"], ka["done"]):
        oracle.lib().pqn_oracle_log_step(n, oracle._p(np.array([r], np.float32)), oracle._p(np.array([d], np.uint8)),
                                         oracle._p(st["ep_ret"]), oracle._p(st["ep_len"]), oracle._p(st["ret_ret"]),
                                         oracle._p(st["ret_len"]), oracle._p(st["timestep"]))
    a = ka["after"]
    assert st["ret_ret"][0] == a["returned_episode_returns"] and st["ret_len"][0] == a["returned_episode_lengths"]
    assert st["ep_ret"][0] == a["episode_returns"] and st["ep_len"][0] == a["episode_lengths"]
    assert st["timestep"][0] == a["timestep"]


def test_eps_greedy_rules(oracle):  # KA8
    rng = np.random.default_rng(1)
    q = rng.standard_normal((4000, 3)).astype(np.float32)
    q[:100] = 1.0  # ties -> first index
    a, qm = oracle.eps_greedy(q, 0.0, key=123)
    assert (a == q.argmax(-1)).all() and (a[:100] == 0).all()
    np.testing.assert_array_equal(qm, q.max(-1))
    a1, _ = oracle.eps_greedy(q, 1.0, key=123)
    assert set(np.unique(a1)) == {0, 1, 2}
    counts = np.bincount(a1, minlength=3) / len(a1)
    assert np.abs(counts - 1 / 3).max() < 0.03
    a5, _ = oracle.eps_greedy(q[100:], 0.3, key=9)
    frac_greedy = (a5 == q[100:].argmax(-1)).mean()
    assert abs(frac_greedy - (0.7 + 0.3 / 3)) < 0.03


def test_shuffle_is_a_permutation_shared_by_leaves(oracle):  # KA7
    T, N = 4, 8
    perm = oracle.permutation(key=42, n=T * N)
    assert sorted(perm.tolist()) == list(range(T * N))
    assert (perm != np.arange(T * N)).any()
    x = np.arange(T * N).reshape(T, N)          # flatten index t*N + e
    assert x.reshape(-1)[perm[0]] == perm[0] and x[perm[0] // N, perm[0] % N] == perm[0]
    assert (oracle.permutation(42, T * N) == perm).all() and (oracle.permutation(43, T * N) != perm).any()


def test_breakout_hand_derived_trajectories(oracle):
    env = oracle.OracleEnv("Breakout-MinAtar")
    for tr in KA["breakout_trajectories"]:
        key = next(k for k in range(1000) if (oracle.env_bits(k, 0, 1)[0] & 1) == tr["start"])
        obs, st = env.reset(key, 1)
        assert obs.shape == (1, 10, 10, 4) and obs.sum() == 1 + 1 + 1 + 30
        assert st["si"][0, 1] == (9 if tr["start"] else 0) and st["si"][0, 2] == (3 if tr["start"] else 2)
        for i, a in enumerate(tr["actions"]):
            obs, st, r, d, info = env.step(0, st, np.array([a]), autoreset=False)
            assert [int(st["si"][0, 1]), int(st["si"][0, 0])] == tr["ball"][i], (i, st["si"][0, :9])
            assert r[0] == tr["reward"][i] and int(d[0]) == tr["terminal"][i]
            assert obs[0, st["si"][0, 0], st["si"][0, 1], 1] == 1 and obs[0, 9, st["si"][0, 3], 0] == 1
        if "final" in tr:
            f = tr["final"]
            assert st["si"][0, 2] == f["dir"] and st["si"][0, 3] == f["pos"] and st["si"][0, 4] == f["strike"]
            y, x = f["brick_cleared"]
            assert st["si"][0, 9 + y * 10 + x] == 0 and st["si"][0, 9:].sum() == 29


def test_breakout_invariants_random_play(oracle):
    env = oracle.OracleEnv("Breakout-MinAtar")
    n = 256
    rng = np.random.default_rng(3)
    obs, st = env.reset(7, n)
    total_r = np.zeros(n)
    for t in range(1200):
        before = st["si"][:, 9:].sum(1).copy()
        a = rng.integers(0, 3, n)
        obs, st, r, d, info = env.step(1000 + t, st, a)
        bm = st["si"][:, 9:].reshape(n, 10, 10)
        assert bm[:, 0].sum() == 0 and bm[:, 4:].sum() == 0          # bricks only in rows 1..3
        assert ((r == 0) | (r == 1)).all()
        after = st["si"][:, 9:].sum(1)
        ok = d | (after == before - r) | (after == 30)                  # one brick per reward (or respawn/reset)
        assert ok.all()
        assert (st["si"][d, 7] == 0).all() and (st["si"][:, 7] <= 1000).all()   # auto-reset zeroes time
        assert (obs.reshape(n, -1).sum(1) == 3 + after).all()
        assert (info["discount"] == 1 - d).all()
    assert st["timestep"].min() == 1200


def test_cartpole_oracle_basic(oracle):
    env = oracle.OracleEnv("CartPole-v1")
    obs, st = env.reset(5, 64)
    assert obs.shape == (64, 4) and np.abs(obs).max() <= 0.05
    done_seen = 0
    for t in range(600):
        obs, st, r, d, info = env.step(t, st, np.zeros(64, np.int32))   # always push left -> falls
        assert (r == 1.0).all()                                          # reward 1 - prev_terminal (auto-reset => 1)
        done_seen += d.sum()
    assert done_seen > 64 and info["returned_episode_lengths"].max() < 100


def test_acrobot_oracle_basic(oracle):
    """Acrobot-v1 restatement (gymnax acrobot.py, book dynamics + RK4): reset range, observation layout, the -1 reward,
    the 500-step time limit under zero torque (a hanging acrobot that is barely perturbed never reaches the goal
    height), energy sanity (zero torque from rest at small angles keeps |theta| small), and that pumping the second
    joint in phase with its velocity does reach the goal."""
    env = oracle.OracleEnv("Acrobot-v1")
    n = 64
    obs, st = env.reset(11, n)
    assert obs.shape == (n, 6) and np.abs(st["sf"]).max() <= 0.1
    np.testing.assert_allclose(obs[:, 0], np.cos(st["sf"][:, 0]), atol=2e-7)
    np.testing.assert_allclose(obs[:, 3], np.sin(st["sf"][:, 1]), atol=2e-7)
    lens = []
    for t in range(501):
        obs, st, r, d, info = env.step(100 + t, st, np.ones(n, np.int32))      # action 1 = zero torque
        if t < 499:
            assert (r == -1.0).all() and not d.any() and np.abs(st["sf"][:, :2]).max() < 0.3
        if d.any():
            lens.append(int(info["returned_episode_lengths"][d].max()))
    assert lens and max(lens) == 500                                            # time limit, auto-reset afterwards
    # energy pumping: torque along the second joint's velocity swings the tip above the bar
    obs, st = env.reset(12, n)
    done_any = np.zeros(n, bool)
    for t in range(400):
        a = np.where(st["sf"][:, 3] >= 0, 2, 0).astype(np.int32)
        obs, st, r, d, info = env.step(300 + t, st, a)
        done_any |= d
    assert done_any.mean() > 0.9 and info["returned_episode_lengths"].max() < 400


@pytest.mark.parametrize("kind", ["cnn", "mlp"])
def test_oracle_network_matches_torch_autograd(oracle, kind):
    """numpy fwd/bwd restatement vs plain PyTorch fp32 autograd of the same net."""
    from purejaxql_amd.networks import QNetwork
    torch.manual_seed(0)
    if kind == "cnn":
        net = QNetwork("cnn", (10, 10, 4), 3, device="cpu")
        x = (torch.rand(32, 10, 10, 4) < 0.2).float()
        shapes = oracle.cnn_shapes((10, 10, 4), 3)
        assert net.num_params == KA["breakout_param_count"]  # KA6
    else:
        net = QNetwork("mlp", (4,), 2, hidden_size=64, num_layers=2, device="cpu")
        x = torch.randn(32, 4)
        shapes = oracle.mlp_shapes(4, 2, 64, 2)
    theta = net.init(3)
    theta += 0.05 * torch.randn_like(theta)  # move LN scales/biases off their init
    leaves = {k: v.clone().requires_grad_(True) for k, v in net.views(theta).items()}
    action = torch.randint(0, net.action_dim, (32,))
    target = torch.randn(32)
    q = net.apply(leaves, x)
    chosen = q.gather(1, action[:, None])[:, 0]
    loss = 0.5 * torch.square(chosen - target).mean()  # KA5
    loss.backward()
    p = oracle.unflatten(theta.numpy().copy(), shapes)
    qo = oracle.net_forward(kind, p, x.numpy())
    np.testing.assert_allclose(qo, q.detach().numpy(), rtol=2e-4, atol=2e-5)
    lo, ch, g = oracle.net_loss_grad(kind, p, shapes, x.numpy(), action.numpy(), target.numpy())
    assert abs(lo - loss.item()) < 1e-5
    gt = torch.cat([(leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])).reshape(-1)
                    for k in shapes]).numpy()
    np.testing.assert_allclose(g, gt, rtol=2e-3, atol=2e-6)
    assert np.abs(g[:2 * x.shape[-1]]).max() == 0  # dummy input BatchNorm never gets gradient


def test_oracle_radam_matches_formula(oracle):
    """optax.radam + clip_by_global_norm formula in float64."""
    rng = np.random.default_rng(0)
    n = 1000
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    p64, m64, v64 = p.astype(np.float64), np.zeros(n), np.zeros(n)
    b1, b2, eps, lr, max_norm = 0.9, 0.999, 1e-8, 1e-3, 10.0
    ro_inf = 2 / (1 - b2) - 1
    for count in range(12):
        g = (rng.standard_normal(n) * (3.0 if count % 3 == 0 else 0.1)).astype(np.float32)
        gn = oracle.radam_clip_step(p, g, m, v, count, lr, max_norm)
        g64 = g.astype(np.float64)
        norm = np.sqrt((g64 ** 2).sum())
        assert abs(gn - norm) / norm < 1e-5
        if not norm < max_norm:
            g64 = g64 / norm * max_norm
        t = count + 1
        m64 = b1 * m64 + (1 - b1) * g64
        v64 = b2 * v64 + (1 - b2) * g64 ** 2
        ro = ro_inf - 2 * t * b2 ** t / (1 - b2 ** t)
        mh, vh = m64 / (1 - b1 ** t), v64 / (1 - b2 ** t)
        if ro >= 5.0:
            r = np.sqrt((ro - 4) * (ro - 2) * ro_inf / ((ro_inf - 4) * (ro_inf - 2) * ro))
            u = r * mh / (np.sqrt(vh) + eps)
        else:
            u = mh
        p64 = p64 - lr * u
        np.testing.assert_allclose(p, p64, rtol=1e-5, atol=1e-6)
    # the rectified branch is reached from t=6 on (ro_6 = 5.0...): make sure both were exercised
    assert ro_inf - 2 * 5 * b2 ** 5 / (1 - b2 ** 5) < 5.0 <= ro_inf - 2 * 6 * b2 ** 6 / (1 - b2 ** 6)


def _torch_radam_reference(p0, grads, lr, max_norm=None):
    """torch.optim.RAdam (an independent implementation of Liu et al. 2020, the algorithm optax.radam implements) fed the
    same gradients; optax.clip_by_global_norm restated as its two lines (g if norm < max else g / norm * max)."""
    pt = torch.nn.Parameter(torch.from_numpy(p0.astype(np.float64)))
    opt = torch.optim.RAdam([pt], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
    out = []
    for g in grads:
        g64 = torch.from_numpy(g.astype(np.float64))
        if max_norm is not None:
            norm = g64.norm()
            if not norm < max_norm:
                g64 = g64 / norm * max_norm
        pt.grad = g64
        opt.step()
        out.append(pt.detach().numpy().copy())
    return out


def test_oracle_radam_vs_torch_optim_radam(oracle):
    """pqn_oracle_radam_clip_step (optax.chain(clip_by_global_norm, radam), pqn_minatar.py:159-162) against
    torch.optim.RAdam over 200 steps, through the rho <= 5 warm-up (SGD-with-momentum steps 1-5) into the rectified
    branch.  The two libraries differ only in where eps enters -- optax: r m_hat / (sqrt(v_hat) + eps); torch:
    r m_hat sqrt(bc2) / (sqrt(v) + eps) = r m_hat / (sqrt(v_hat) + eps / sqrt(bc2)) -- i.e. by <= eps / sqrt(v_hat)
    relative (1e-5 for |g| ~ 1e-2 at step 6, shrinking with bc2 -> 1), and torch rectifies for rho > 5 where optax
    takes rho >= 5 (rho_5 = 4.99, rho_6 = 5.98: the same steps).  An independent witness of the published algorithm,
    not a reference golden."""
    rng = np.random.default_rng(11)
    n, steps, lr = 4096, 200, 5e-4
    p0 = rng.standard_normal(n).astype(np.float32)
    grads = [(rng.standard_normal(n) * (0.3 if t % 7 else 3.0) * np.exp(-t / 120.0)).astype(np.float32) for t in range(steps)]
    for max_norm in (1e9, 10.0):           # never clipped / clipped on the large-gradient steps
        p, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
        ref = _torch_radam_reference(p0, grads, lr, max_norm)
        clipped = 0
        for t, g in enumerate(grads):
            gn = oracle.radam_clip_step(p, g, m, v, t, np.float32(lr), max_norm)
            clipped += int(not gn < max_norm)
            # the oracle keeps p, m, v in f32 (as optax does), the torch run is f64: per step one f32 rounding of p (random
            # walk: ~ulp * sqrt(steps)) plus the eps-placement difference of <= 1e-5 of a step of size ~lr
            np.testing.assert_allclose(p, ref[t], rtol=2e-7 * np.sqrt(t + 1.0), atol=1e-7 + 2e-5 * lr * (t + 1), err_msg=f"step {t}")
        assert (clipped > 20) == (max_norm == 10.0)
    # warm-up really is the unrectified branch in both: after 5 steps p moved by lr * sum of bias-corrected momenta
    m64, acc = np.zeros(n), np.zeros(n)
    for t in range(5):
        m64 = 0.9 * m64 + 0.1 * grads[t].astype(np.float64)
        acc += m64 / (1 - 0.9 ** (t + 1))
    np.testing.assert_allclose(_torch_radam_reference(p0, grads[:5], lr)[-1], p0 - lr * acc, rtol=0, atol=1e-9)


def test_oracle_layernorm_and_conv_vs_torch_functional(oracle):
    """The oracle's LayerNorm (flax: eps 1e-6 inside the rsqrt, fast variance) and its 3x3 VALID NHWC convolution with an
    HWIO kernel (nn.Conv(16, (3, 3), padding="VALID"), pqn_minatar.py:28-36) against torch.nn.functional.layer_norm /
    conv2d, forward and input / parameter gradients -- library implementations this build did not write."""
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    x = rng.standard_normal((64, 8, 8, 16)).astype(np.float32) * 3 + 1
    sc, bi = rng.standard_normal(16).astype(np.float32), rng.standard_normal(16).astype(np.float32)
    y, cache = oracle._ln_fwd(x, sc, bi)
    xt, st, bt = (torch.from_numpy(a.copy()).requires_grad_(True) for a in (x, sc, bi))
    yt = F.layer_norm(xt, (16,), st, bt, 1e-6)
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-5, atol=2e-6)
    dy = rng.standard_normal(y.shape).astype(np.float32)
    yt.backward(torch.from_numpy(dy))
    dx, dsc, dbi = oracle._ln_bwd(dy, sc, cache)
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dsc, st.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(dbi, bt.grad.numpy(), rtol=2e-4, atol=2e-4)
    for c in (4, 6, 7, 10):
        obs = (rng.random((32, 10, 10, c)) < 0.2).astype(np.float32)
        k = (rng.standard_normal((3, 3, c, 16)) * 0.2).astype(np.float32)          # flax HWIO
        b = rng.standard_normal(16).astype(np.float32)
        out = oracle._patches(np.ascontiguousarray(obs)) @ k.reshape(-1, 16) + b    # the oracle's conv (net_forward)
        kt = torch.from_numpy(k.copy()).requires_grad_(True)
        ot = F.conv2d(torch.from_numpy(obs).permute(0, 3, 1, 2), kt.permute(3, 2, 0, 1), torch.from_numpy(b)).permute(0, 2, 3, 1)
        np.testing.assert_allclose(out, ot.detach().numpy(), rtol=1e-5, atol=1e-5)
        d = rng.standard_normal(out.shape).astype(np.float32)
        ot.backward(torch.from_numpy(d))
        dk = oracle._patches(np.ascontiguousarray(obs)).reshape(-1, 9 * c).T @ d.reshape(-1, 16)   # _net_backward's wgrad
        np.testing.assert_allclose(dk.reshape(3, 3, c, 16), kt.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_cpu_baseline_loop_equals_the_oracle_loop(oracle):
    """bench.py's cpu_baseline (oracle/pqn_cpu_torch.py: the oracle loop with the Q-network on torch-CPU autograd) against
    oracle.make_train on a small Breakout shape: same key schedule, same C env / eps-greedy / Q(lambda) / RAdam -- the
    parameters after two updates agree to f32 rounding, so the rate the bench prints is the rate of the same algorithm
    (pqn_minatar.py:176-369)."""
    import importlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    cpu_loop = importlib.import_module("pqn_cpu_torch")
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.networks import QNetwork
    cfg = flatten(load_config(["+alg=pqn_minatar", "alg.ENV_NAME=Breakout-MinAtar", "alg.NUM_ENVS=32", "alg.TEST_DURING_TRAINING=False"]))
    cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
    cfg["NUM_MINIBATCHES"] = 4
    cfg["TOTAL_TIMESTEPS"] = cfg["TOTAL_TIMESTEPS_DECAY"] = 6 * 32 * cfg["NUM_STEPS"]
    th0 = QNetwork("cnn", (10, 10, 4), 3, device="cpu").init(1).numpy()
    a = oracle.make_train(dict(cfg))(7, th0, max_updates=2)
    b = cpu_loop.make_train(dict(cfg), threads=2)(7, th0, max_updates=2)
    assert len(b["seconds_per_update"]) == 2
    np.testing.assert_allclose(b["theta"], a["theta"], rtol=0, atol=2e-5)
    upd = np.linalg.norm(a["theta"] - th0)
    assert np.linalg.norm(a["theta"] - b["theta"]) < 1e-3 * upd
    for k in ("td_loss", "qvals"):
        assert abs(a["metrics"][-1][k] - b["metrics"][-1][k]) < 1e-5 * max(1.0, abs(a["metrics"][-1][k]))


def test_oracle_regression_pins():
    """The oracle's outputs on seeded inputs still hash to the committed digests (tests/golden/regression_pins.json):
    env rules + RNG streams of all five envs, eps-greedy, shuffle, Q(lambda) both forms, fold_in.  Self-regression
    pins, not reference goldens (the reference cannot run here)."""
    import importlib.util
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_regression_pins", os.path.join(here, "golden", "make_regression_pins.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "golden", "regression_pins.json")))
    got = mod.pins()
    assert got == want


def test_minatar_suite_hand_derived_steps(oracle):
    """Known answers derived by hand from the published MinAtar rules (Young & Tian 2019: asterix.py, freeway.py,
    space_invaders.py) for the deterministic part of each game -- the pieces that need no random draw.
    Pins of the oracle, not reference-generated goldens."""
    # --- Asterix: player starts at (5,5); spawn timer 10, move timer 5, ramp timer 100; moves clamp to x 0..9, y 1..8
    env = oracle.OracleEnv("Asterix-MinAtar")
    _obs, st = env.reset(1, 1)
    s = st["si"][0]
    assert list(s[:11]) == [5, 5, 0, 10, 10, 5, 5, 100, 0, 0, 0] and s[11:].sum() == 0
    for t in range(5):
        _o, st, r, d, _i = env.step(10 + t, st, np.array([0], np.int32), autoreset=False)
        assert r[0] == 0 and not d[0]
    s = st["si"][0]
    assert (s[4], s[6], s[9]) == (5, 0, 5)                     # spawn timer 10-5, move timer 5-5, time
    for a, want in ((1, (4, 5)), (1, (3, 5)), (2, (3, 4)), (4, (3, 5)), (3, (4, 5))):    # left, left, up, down, right
        _o, st, _r, _d, _i = env.step(99, st, np.array([a], np.int32), autoreset=False)
        assert (st["si"][0][0], st["si"][0][1]) == want
    for _ in range(12):                                          # up is clamped at row 1, left at column 0
        _o, st, _r, d, _i = env.step(99, st, np.array([2], np.int32), autoreset=False)
        if d[0]:
            break
    assert d[0] or st["si"][0][1] == 1
    # --- Freeway: chicken starts at row 9 with a 3-frame move cooldown, terminate timer 2500; 'up' (action 1) moves
    # one row every 4th frame at most; rows stay in 0..9; reward only on reaching row 0
    env = oracle.OracleEnv("Freeway-MinAtar")
    _obs, st = env.reset(5, 1)
    s = st["si"][0]
    assert (s[0], s[1], s[2], s[3], s[4]) == (9, 3, 2500, 0, 0)
    speeds = s[5 + 2:29:3]
    assert all(1 <= abs(int(v)) <= 5 for v in speeds) and list(s[5:29:3]) == [0] * 8      # 8 cars at x=0, speed 1..5
    rows = []
    for t in range(8):
        _o, st, r, d, _i = env.step(50 + t, st, np.array([1], np.int32), autoreset=False)
        rows.append(int(st["si"][0][0]))
        assert r[0] == 0 and not d[0]
    # cooldown 3 -> the first move happens on the 4th 'up' frame, the next one 4 frames later (unless a car on row 8/7
    # at x=4 sends the chicken back to row 9, which cannot happen in the first 8 frames: cars start at x=0 and need
    # >= 4 moves of >= 1 frame each... the fastest car reaches x=4 at frame 4, on its own row only)
    assert rows[:3] == [9, 9, 9] and rows[3] in (8, 9) and all(7 <= v <= 9 for v in rows)
    assert st["si"][0][2] == 2500 - 8 and st["si"][0][3] == 8
    # --- SpaceInvaders: cannon at x=5, 4x6 alien block at rows 0..3, columns 2..7, moving left (-1) every 12 frames;
    # 'fire' (minimal action set: 0 noop, 1 left, 2 right, 3 fire) puts a friendly bullet on row 8 above the cannon
    env = oracle.OracleEnv("SpaceInvaders-MinAtar")
    _obs, st = env.reset(9, 1)
    s = st["si"][0]
    assert (s[0], s[1], s[2], s[3], s[4]) == (5, -1, 12, 12, 10)
    al = s[9:109].reshape(10, 10)
    assert al.sum() == 24 and al[:4, 2:8].all()
    _o, st, r, d, _i = env.step(1, st, np.array([3], np.int32), autoreset=False)
    s = st["si"][0]
    fb = s[109:209].reshape(10, 10)
    assert fb.sum() == 1 and fb[8, 5] == 1 and r[0] == 0 and not d[0]
    for t in range(4):                                           # the bullet climbs one row per frame
        _o, st, _r, _d, _i = env.step(2 + t, st, np.array([0], np.int32), autoreset=False)
    fb = st["si"][0][109:209].reshape(10, 10)
    assert fb.sum() == 1 and fb[4, 5] == 1
    _o, st, r, _d, _i = env.step(7, st, np.array([0], np.int32), autoreset=False)
    s = st["si"][0]
    assert r[0] == 1.0 and s[109:209].sum() == 0 and s[9:109].sum() == 23     # hits the alien at (3,5): +1, both removed


def test_third_party_rule_known_answers_timers_waves_thresholds(oracle):
    """More hand-derived known answers for the third-party rules SURVEY Appendix B states from recollection (they narrow
    what recollection alone carries; they are pins of the oracle, not reference-generated goldens).  Each expectation
    below is worked out on paper from the published MinAtar / gymnax rule quoted beside it.
      * Asterix (asterix.py): `ramp_timer` counts 100, 99, ..., 0, -1 and the ramp fires on the step that FINDS it
        negative, i.e. on step 102, 204, ...; every ramp lowers spawn_speed, odd ramp indices also lower move_speed.
      * SpaceInvaders (space_invaders.py): shooting the last alien refills rows 0-3 x columns 2-7 in the same step and
        lowers enemy_move_interval by one (12 -> 11, floor 6), ramp_index + 1.
      * Freeway (freeway.py): a car [x, timer, speed] waits while timer > 0 and moves (one cell, wrapping at 0 / 9) on the
        frame that finds timer == 0, reloading timer = |speed|: period |speed| + 1 frames.
      * CartPole (gymnax cartpole.py): done = |x| > 2.4 or |theta| > 12 * 2 pi / 360 (strict), reward = 1 - prev_terminal."""
    noop = np.zeros(1, np.int32)
    # ---- Asterix ramp ----
    env = oracle.OracleEnv("Asterix-MinAtar")
    _o, st = env.reset(1, 1)
    s = st["si"][0]
    s[4] = 100000                       # spawn timer far away: an empty board, the player cannot die
    assert (s[3], s[5], s[7], s[8]) == (10, 5, 100, 0)
    for t in range(1, 205):
        _o, st, r, d, _i = env.step(t, st, noop, autoreset=False)
        s = st["si"][0]
        assert r[0] == 0 and not d[0]
        if t == 101:
            assert (s[3], s[5], s[7], s[8]) == (10, 5, -1, 0)
        if t == 102:
            assert (s[3], s[5], s[7], s[8]) == (9, 5, 100, 1)          # ramp index 0 (even): spawn speed only
        if t == 203:
            assert (s[3], s[5], s[7], s[8]) == (9, 5, -1, 1)
        if t == 204:
            assert (s[3], s[5], s[7], s[8]) == (8, 4, 100, 2)          # ramp index 1 (odd): move speed too
    # the move timer reloads from move_speed when it reaches 0: 204 frames = 40 reloads of 5 + 4 spent of the current
    # period before the reload value changed ... checked structurally instead: 0 <= timer < move_speed + 1
    assert 0 <= st["si"][0][6] <= 5
    # ---- SpaceInvaders wave reset ----
    env = oracle.OracleEnv("SpaceInvaders-MinAtar")
    _o, st = env.reset(9, 1)
    s = st["si"][0]
    s[9:109] = 0
    s[9 + 3 * 10 + 5] = 1               # one alien left, at row 3 above the cannon (x = 5)
    s[3] = 1000; s[4] = 1000            # no alien move, no enemy shot while the bullet flies
    _o, st, r, d, _i = env.step(1, st, np.array([3], np.int32), autoreset=False)      # fire: bullet on row 8
    for t in range(4):
        _o, st, r, d, _i = env.step(2 + t, st, noop, autoreset=False)                 # rows 7, 6, 5, 4
        assert r[0] == 0 and st["si"][0][9:109].sum() == 1
    _o, st, r, d, _i = env.step(7, st, noop, autoreset=False)                          # row 3: hit
    s = st["si"][0]
    al = s[9:109].reshape(10, 10)
    assert r[0] == 1.0 and not d[0] and al.sum() == 24 and al[:4, 2:8].all()           # refilled in the same step
    assert (s[2], s[6]) == (11, 1) and s[109:209].sum() == 0                           # move interval 12 -> 11, ramp index 1
    # ---- Freeway car period ----
    env = oracle.OracleEnv("Freeway-MinAtar")
    _o, st = env.reset(5, 1)
    s = st["si"][0]
    s[5:29] = 0
    s[5 + 0:5 + 3] = (0, 2, 2)          # car of row 1: x = 0, timer 2, speed +2 (rightwards)
    s[5 + 3:5 + 6] = (1, 0, -4)         # car of row 2: x = 1, timer 0, speed -4 (leftwards): moves at once, wraps 0 -> 9
    for i in range(2, 8):
        s[5 + 3 * i:5 + 3 * i + 3] = (0, 1000, 1)    # parked
    xs1, xs2 = [], []
    for t in range(12):
        _o, st, r, d, _i = env.step(30 + t, st, noop, autoreset=False)
        xs1.append(int(st["si"][0][5])); xs2.append(int(st["si"][0][8]))
    assert xs1 == [0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4]            # frames 3, 6, 9, 12: period |2| + 1
    assert xs2 == [0, 0, 0, 0, 0, 9, 9, 9, 9, 9, 8, 8]            # frames 1, 6, 11: period |-4| + 1, wrap below 0
    # ---- CartPole thresholds ----
    env = oracle.OracleEnv("CartPole-v1")
    _o, st = env.reset(3, 4)
    thr = np.float32(12 * 2 * np.pi / 360)
    st["sf"][:] = 0.0
    st["sf"][0, 0] = 2.4                 # |x| == 2.4 is NOT out of bounds (strict >) ...
    st["sf"][1, 0] = np.nextafter(np.float32(2.4), np.float32(3.0))      # ... the next float is
    st["sf"][2, 2] = thr
    st["sf"][3, 2] = np.nextafter(thr, np.float32(1.0))
    st["si"][:, 0] = 0
    a = np.ones(4, np.int32)
    _o, st2, r, d, _i = env.step(1, {k: v.copy() for k, v in st.items()}, a, autoreset=False)
    # env 1 / 3 were terminal BEFORE the step: reward 1 - prev_terminal = 0.  The push (+10 N) gives x_dot = tau * xacc > 0
    # but positions advance with the OLD velocities (explicit Euler), so x and theta are unchanged after one step
    assert list(r) == [1.0, 0.0, 1.0, 0.0] and list(d) == [False, True, False, True]
    np.testing.assert_array_equal(st2["sf"][:, 0], st["sf"][:, 0])
    np.testing.assert_array_equal(st2["sf"][:, 2], st["sf"][:, 2])
    assert (st2["sf"][[0, 1], 1] > 0).all()


def test_float64_learn_phase_twin_tracks_the_f32_oracle(oracle):
    """oracle/pqn_oracle_f64.py (the float64 learn phase the headline tests measure both sides against) on the f32 oracle's own
    rollout record of one small update: gradient of the first minibatch to f32 rounding, parameters after the whole update within
    a few lr of the f32 oracle's (RAdam turns rounding-level gradient differences of near-zero elements into lr-sized steps), the
    update vectors aligned.  purejaxql/pqn_minatar.py:263-327."""
    import pqn_oracle_f64 as o64
    cfg = {"ENV_NAME": "Breakout-MinAtar", "NUM_ENVS": 64, "NUM_STEPS": 16, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2, "TOTAL_TIMESTEPS": 64 * 16,
           "TOTAL_TIMESTEPS_DECAY": 64 * 16 * 10, "EPS_START": 1.0, "EPS_FINISH": 0.05, "EPS_DECAY": 0.1, "GAMMA": 0.99, "LAMBDA": 0.65,
           "LR": 5e-4, "LR_LINEAR_DECAY": True, "MAX_GRAD_NORM": 10.0, "NORM_TYPE": "layer_norm", "NORM_INPUT": False, "REW_SCALE": 1.0,
           "TEST_DURING_TRAINING": False}
    train = oracle.make_train(cfg)
    shapes = train.shapes
    n = sum(int(np.prod(s)) for s in shapes.values())
    rng = np.random.default_rng(3)
    theta0 = (rng.standard_normal(n) * 0.05).astype(np.float32)
    p0 = oracle.unflatten(theta0, shapes)
    for k in p0:
        if k.endswith("/scale"):
            p0[k][...] = 1.0
    key = 0x1234567
    out = train(key, theta0)
    sh = out["shards"][0]
    k_shuf = oracle.fold_in(key & 0xFFFFFFFFFFFFFFFF, 4)
    # one gradient, same inputs
    idx = oracle.permutation(oracle.fold_in(k_shuf, 0), 64 * 16)[:256]
    _l32, _c32, g32 = oracle.net_loss_grad("cnn", p0, shapes, sh["of"][idx], sh["af"][idx], sh["tf"][idx])
    p64 = oracle.unflatten(theta0.astype(np.float64), shapes)
    _l64, _c64, g64 = o64.cnn_loss_grad(p64, shapes, sh["of"][idx], sh["af"][idx], sh["tf"][idx])
    assert np.isfinite(g32).all() and np.isfinite(g64).all()
    assert np.abs(g32 - g64).max() <= 2e-6 * np.abs(g64).max()
    th64, _m, _v = o64.learn_phase(cfg, shapes, theta0, sh["of"], sh["af"], sh["tf"], k_shuf)
    upd32, upd64 = out["theta"].astype(np.float64) - theta0, th64 - theta0
    cos = float(np.dot(upd32, upd64) / (np.linalg.norm(upd32) * np.linalg.norm(upd64)))
    assert cos > 0.999 and np.abs(out["theta"] - th64).max() < 4 * cfg["LR"], (cos, float(np.abs(out["theta"] - th64).max()))
