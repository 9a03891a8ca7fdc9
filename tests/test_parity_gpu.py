"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against
the CPU oracle on the same seeded inputs.  Integer/index work must be bit-exact;
float tolerances are written at each assert."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KA = json.load(open(os.path.join(HERE, "golden", "ka_vectors.json")))


def _np(t):
    return t.detach().cpu().numpy()


def _check_state(env, oenv, state, ost):
    si, sf, log = env.export_state(state)
    np.testing.assert_array_equal(_np(si), ost["si"])
    np.testing.assert_array_equal(_np(log).view(np.uint32), oenv.log_words(ost))
    return sf


@pytest.mark.parametrize("n", [1, 5, 16, 1000, 4096])
def test_breakout_step_bit_exact_vs_oracle(gpu, oracle, n):
    from purejaxql_amd.envs import LogWrapper, make
    env, params = make("Breakout-MinAtar", device=gpu)
    env = LogWrapper(env)
    oenv = oracle.OracleEnv("Breakout-MinAtar")
    (obs, bits), state = env.reset(11, params, n, want_bits=True)
    oobs, ost = oenv.reset(11, n)
    np.testing.assert_array_equal(_np(obs), oobs)
    _check_state(env, oenv, state, ost)
    rng = np.random.default_rng(n)
    steps = 1500 if n <= 1000 else 300
    for t in range(steps):
        a = rng.integers(0, 3, n).astype(np.int32)
        if t % 7 == 0:  # follow the ball sometimes so episodes last and bricks clear
            a = np.where(ost["si"][:, 1] < ost["si"][:, 3], 1, np.where(ost["si"][:, 1] > ost["si"][:, 3], 2, 0)).astype(np.int32)
        key = 5000 + t
        (obs, bits), state, r, d, info = env.step(key, state, torch.from_numpy(a).to(gpu), params, want_bits=True)
        oobs, ost, orr, od, oinfo = oenv.step(key, ost, a)
        np.testing.assert_array_equal(_np(r), orr)
        np.testing.assert_array_equal(_np(d), od)
        if t % 50 == 0 or t == steps - 1:
            np.testing.assert_array_equal(_np(obs), oobs)
            _check_state(env, oenv, state, ost)
            for k in oinfo:
                np.testing.assert_array_equal(_np(info[k]), oinfo[k], err_msg=k)
            # packed observation == f32 observation, bit (y*10+x)*C+c
            b = _np(bits).view(np.uint32)
            flat = oobs.reshape(n, -1).astype(np.uint8)
            unpacked = ((b[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, -1)[:, :400]
            np.testing.assert_array_equal(unpacked, flat)
            assert (b[:, 13:] == 0).all() and ((b[:, 12] >> 16) == 0).all()
    assert ost["ret_len"].max() > 0  # episodes actually finished


@pytest.mark.parametrize("name,c,a,n,steps", [("Asterix-MinAtar", 4, 5, 1024, 1500), ("Freeway-MinAtar", 7, 3, 512, 2700),
                                               ("SpaceInvaders-MinAtar", 6, 4, 1024, 1500), ("Asterix-MinAtar", 4, 5, 37, 400),
                                               # BASELINE.json configs[2]: the suite at 4096 envs each
                                               ("Asterix-MinAtar", 4, 5, 4096, 300), ("Freeway-MinAtar", 7, 3, 4096, 200),
                                               ("SpaceInvaders-MinAtar", 6, 4, 4096, 300)])
def test_minatar_suite_step_bit_exact_vs_oracle(gpu, oracle, name, c, a, n, steps):
    """Asterix / Freeway / SpaceInvaders: HIP packed-state kernels vs the C oracle (MinAtar rules), bit-exact
    on reward, done, observation (f32 and packed), full state and LogWrapper record."""
    from purejaxql_amd.envs import LogWrapper, make
    env, params = make(name, device=gpu)
    env = LogWrapper(env)
    oenv = oracle.OracleEnv(name)
    assert env.obs_shape == (10, 10, c) and env.num_actions == a and params.max_steps_in_episode == oenv.max_steps
    (obs, bits), state = env.reset(21, params, n, want_bits=True)
    oobs, ost = oenv.reset(21, n)
    np.testing.assert_array_equal(_np(obs), oobs)
    _check_state(env, oenv, state, ost)
    rng = np.random.default_rng(n + c)
    ow = bits.shape[1]
    for t in range(steps):
        act = rng.integers(0, a, n).astype(np.int32)
        if name.startswith("Freeway") and t % 3:      # mostly "up" so chickens cross and cars get re-randomised
            act[: n // 2] = 1
        if name.startswith("SpaceInvaders") and t % 2:  # fire a lot so waves get cleared (ramping)
            act[: n // 2] = 3
        key = 9000 + t
        (obs, bits), state, r, d, info = env.step(key, state, torch.from_numpy(act).to(gpu), params, want_bits=True)
        oobs, ost, orr, od, oinfo = oenv.step(key, ost, act)
        np.testing.assert_array_equal(_np(r), orr, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(_np(d), od, err_msg=f"done t={t}")
        if t % 25 == 0 or t == steps - 1:
            np.testing.assert_array_equal(_np(obs), oobs, err_msg=f"obs t={t}")
            _check_state(env, oenv, state, ost)
            for k in oinfo:
                np.testing.assert_array_equal(_np(info[k]), oinfo[k], err_msg=k)
            b = _np(bits).view(np.uint32)
            unpacked = ((b[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(n, -1)
            np.testing.assert_array_equal(unpacked[:, :100 * c], oobs.reshape(n, -1).astype(np.uint8))
            assert unpacked[:, 100 * c:].sum() == 0
    if not (name.startswith("Freeway") and steps < 2500):   # Freeway episodes only end at the 2500-step limit
        assert ost["ret_len"].max() > 0 and ost["ret_ret"].max() > 0
    else:
        assert ost["ep_ret"].max() > 0                       # chickens did cross


def test_breakout_hand_derived_trajectories_on_gpu(gpu, oracle):
    from purejaxql_amd.envs import make
    env, params = make("Breakout-MinAtar", device=gpu)
    oenv = oracle.OracleEnv("Breakout-MinAtar")
    for tr in KA["breakout_trajectories"]:
        key = next(k for k in range(1000) if (oracle.env_bits(k, 0, 1)[0] & 1) == tr["start"])
        _, ost = oenv.reset(key, 1)
        obs, state = env.reset(key, params, 1)
        for i, a in enumerate(tr["actions"]):
            # step_env only (auto-reset off in the oracle) == the product while no episode ends
            obs, state, r, d, _ = env.step(0, state, torch.tensor([a], dtype=torch.int32, device=gpu), params)
            si, _, _ = env.export_state(state)
            si = _np(si)[0]
            if not tr["terminal"][i]:
                assert [int(si[1]), int(si[0])] == tr["ball"][i]
            assert float(r[0]) == tr["reward"][i] and int(d[0]) == tr["terminal"][i]
        if "final" in tr:
            f = tr["final"]
            assert si[2] == f["dir"] and si[3] == f["pos"] and si[4] == f["strike"]
            assert si[9 + f["brick_cleared"][0] * 10 + f["brick_cleared"][1]] == 0 and si[9:].sum() == 29


def test_breakout_import_export_round_trip_and_edge_states(gpu, oracle):
    """Crafted states: last brick about to be cleared (respawn), time limit, ball in corners."""
    from purejaxql_amd.envs import make
    env, params = make("Breakout-MinAtar", device=gpu)
    oenv = oracle.OracleEnv("Breakout-MinAtar")
    rng = np.random.default_rng(0)
    n = 2048
    si = np.zeros((n, 109), np.int32)
    si[:, 0] = rng.integers(0, 9, n)    # ball_y (row 9 is unreachable: the ball bounces or the episode ends)
    si[:, 1] = rng.integers(0, 10, n)   # ball_x
    si[:, 2] = rng.integers(0, 4, n)
    si[:, 3] = rng.integers(0, 10, n)
    si[:, 4] = rng.integers(0, 2, n)
    si[:, 5] = rng.integers(0, 10, n)
    si[:, 6] = rng.integers(0, 10, n)
    si[:, 7] = rng.choice([0, 5, 998, 999], n)
    dens = rng.choice([0.0, 0.03, 0.5, 1.0], n)[:, None]
    si[:, 9 + 10:9 + 40] = rng.random((n, 30)) < dens
    state = env.import_state(torch.from_numpy(si))
    si2, _, log = env.export_state(state)
    np.testing.assert_array_equal(_np(si2), si)
    ost = {"si": si.copy(), "sf": np.zeros((n, 1), np.float32), "ep_ret": np.zeros(n, np.float32),
           "ep_len": np.zeros(n, np.int32), "ret_ret": np.zeros(n, np.float32), "ret_len": np.zeros(n, np.int32),
           "timestep": np.zeros(n, np.int32)}
    for t in range(40):
        a = rng.integers(0, 3, n).astype(np.int32)
        obs, state, r, d, _ = env.step(77 + t, state, torch.from_numpy(a).to(gpu), params)
        oobs, ost, orr, od, _ = oenv.step(77 + t, ost, a)
        np.testing.assert_array_equal(_np(r), orr)
        np.testing.assert_array_equal(_np(d), od)
        np.testing.assert_array_equal(_np(obs), oobs)
        _check_state(env, oenv, state, ost)


def test_cartpole_step_vs_oracle(gpu, oracle):
    """f32 dynamics, bit-exact TRAJECTORIES: sin / cos are a fixed sequence of IEEE f32 operations shared by the rule
    (pqn_env_rules.h pqn_sincos_f32) and the oracle's restatement, divisions are correctly rounded on both sides and
    contraction is off, so 700 free-running steps of 512 envs (no re-synchronisation) agree in every bit: observations,
    rewards, dones, info, state."""
    from purejaxql_amd.envs import FlattenObservationWrapper, LogWrapper, make
    env, params = make("CartPole-v1", device=gpu)
    env = LogWrapper(FlattenObservationWrapper(env))
    oenv = oracle.OracleEnv("CartPole-v1")
    n = 512
    obs, state = env.reset(3, params, n)
    oobs, ost = oenv.reset(3, n)
    np.testing.assert_array_equal(_np(obs), oobs)
    rng = np.random.default_rng(0)
    for t in range(700):
        a = rng.integers(0, 2, n).astype(np.int32)
        obs, state, r, d, info = env.step(900 + t, state, torch.from_numpy(a).to(gpu), params)
        oobs, ost, orr, od, oinfo = oenv.step(900 + t, ost, a)
        np.testing.assert_array_equal(_np(d), od, err_msg=f"done t={t}")
        np.testing.assert_array_equal(_np(obs), oobs, err_msg=f"obs t={t}")
        np.testing.assert_array_equal(_np(r), orr)
        if t % 50 == 0 or t == 699:
            for k in oinfo:
                np.testing.assert_array_equal(_np(info[k]), oinfo[k], err_msg=k)
            si, sf, log = env.export_state(state)
            np.testing.assert_array_equal(_np(si), ost["si"])
            np.testing.assert_array_equal(_np(sf), ost["sf"])
            np.testing.assert_array_equal(_np(log).view(np.uint32), oenv.log_words(ost))
    assert ost["ret_len"].max() > 5


def test_acrobot_step_vs_oracle(gpu, oracle):
    """Acrobot-v1 (the alternative env of config/alg/pqn_cartpole.yaml:24): RK4 of the book dynamics in f32 on the shared
    explicit sin / cos -- 1200 free-running steps of 256 envs agree with the oracle in every bit (observations, rewards,
    dones incl. the 500-step limit and goal terminations under a pumping policy, info, exported state)."""
    from purejaxql_amd.envs import FlattenObservationWrapper, LogWrapper, make
    env, params = make("Acrobot-v1", device=gpu)
    env = LogWrapper(FlattenObservationWrapper(env))
    assert env.action_space(params).n == 3 and tuple(env.observation_space(params).shape) == (6,)
    oenv = oracle.OracleEnv("Acrobot-v1")
    n = 256
    obs, state = env.reset(3, params, n)
    oobs, ost = oenv.reset(3, n)
    np.testing.assert_array_equal(_np(obs), oobs)
    rng = np.random.default_rng(0)
    ndone = 0
    for t in range(1200):
        a = np.where(ost["sf"][:, 3] >= 0, 2, 0).astype(np.int32)            # pump the second joint ...
        a[: n // 2] = rng.integers(0, 3, n // 2).astype(np.int32)            # ... half of the envs act randomly
        obs, state, r, d, info = env.step(900 + t, state, torch.from_numpy(a).to(gpu), params)
        oobs, ost, orr, od, oinfo = oenv.step(900 + t, ost, a)
        np.testing.assert_array_equal(_np(d), od, err_msg=f"done t={t}")
        np.testing.assert_array_equal(_np(obs), oobs, err_msg=f"obs t={t}")
        np.testing.assert_array_equal(_np(r), orr)
        ndone += int(od.sum())
        if t % 100 == 0 or t == 1199:
            for k in oinfo:
                np.testing.assert_array_equal(_np(info[k]), oinfo[k], err_msg=k)
            si, sf, log = env.export_state(state)
            np.testing.assert_array_equal(_np(si), ost["si"])
            np.testing.assert_array_equal(_np(sf), ost["sf"])
            np.testing.assert_array_equal(_np(log).view(np.uint32), oenv.log_words(ost))
    assert ndone > n                                                          # goal terminations and time limits both occurred


@pytest.mark.parametrize("m,a", [(1, 2), (1000, 3), (4096, 3), (70001, 6)])
def test_eps_greedy_bit_exact(gpu, oracle, m, a):
    from purejaxql_amd import ops
    rng = np.random.default_rng(m)
    q = rng.standard_normal((m, a)).astype(np.float32)
    q[: m // 10] = 0.5  # ties -> first index
    qt = torch.from_numpy(q).to(gpu)
    for eps in (0.0, 0.05, 0.5, 1.0):
        act, qmax = ops.eps_greedy(qt, eps, key=1234 + m)
        oa, oq = oracle.eps_greedy(q, eps, key=1234 + m)
        np.testing.assert_array_equal(_np(act), oa)
        np.testing.assert_array_equal(_np(qmax), oq)


def test_q_lambda_known_answers_and_oracle(gpu, oracle):
    from purejaxql_amd import ops
    ka = KA["q_lambda"]
    r = torch.tensor(ka["reward"], dtype=torch.float32, device=gpu)[:, None].contiguous()
    qm = torch.tensor(ka["qmax"], dtype=torch.float32, device=gpu)[:, None].contiguous()
    lq = torch.tensor([ka["last_q"]], dtype=torch.float32, device=gpu)
    for case in ka["cases"]:
        d = torch.tensor(case["done"], dtype=torch.uint8, device=gpu)[:, None].contiguous()
        np.testing.assert_allclose(_np(ops.q_lambda(r, d, qm, lq, ka["gamma"], ka["lambda"], quirk=True))[:, 0],
                                   case["minatar"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(_np(ops.q_lambda(r, d, qm, lq, ka["gamma"], ka["lambda"], quirk=False))[:, 0],
                                   case["atari"], rtol=1e-6, atol=1e-6)
    rng = np.random.default_rng(0)
    for T, M in [(1, 7), (2, 64), (32, 4096), (32, 65536 + 3), (128, 100)]:
        rr = (rng.random((T, M)) < 0.05).astype(np.float32) * rng.random((T, M)).astype(np.float32)
        dd = (rng.random((T, M)) < 0.02).astype(np.uint8)
        qq = rng.standard_normal((T, M)).astype(np.float32)
        ll = rng.standard_normal(M).astype(np.float32)
        for quirk in (True, False):
            got = ops.q_lambda(torch.from_numpy(rr).to(gpu), torch.from_numpy(dd).to(gpu), torch.from_numpy(qq).to(gpu),
                               torch.from_numpy(ll).to(gpu), 0.99, 0.65, quirk=quirk)
            exp = oracle.q_lambda(rr, dd, qq, ll, 0.99, 0.65, quirk=quirk)
            np.testing.assert_array_equal(_np(got), exp)   # same f32 op order, contraction off on both sides


def test_shuffle_permutation_bit_exact(gpu, oracle):
    from purejaxql_amd import ops
    for n in (1, 2, 1000, 131072):
        p = ops.shuffle_permutation(99 + n, n, gpu)
        np.testing.assert_array_equal(_np(p), oracle.permutation(99 + n, n))


def test_radam_clip_vs_oracle(gpu, oracle):
    """f32 elementwise identical; the global norm is a different summation order
    (tree in f32 vs sequential f64) -> params agree to rtol 2e-6 / atol 1e-7 per step."""
    from purejaxql_amd import ops
    rng = np.random.default_rng(0)
    for n, lr_steps in [(5, 0.0), (132475, 200.0), (1_000_003, 7.0)]:
        p = rng.standard_normal(n).astype(np.float32)
        pt = torch.from_numpy(p.copy()).to(gpu)
        opt = ops.FlatRAdam(pt, 5e-4, 10.0, lr_decay_steps=lr_steps)
        m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
        for count in range(10):
            scale = 1.0 if count % 2 else 1e-3   # alternate clipped / unclipped steps
            g = (rng.standard_normal(n) * scale).astype(np.float32)
            opt.step(torch.from_numpy(g).to(gpu))
            lr = oracle.linear_schedule(5e-4, 1e-20, lr_steps, count) if lr_steps > 0 else 5e-4
            gn = oracle.radam_clip_step(p, g, m, v, count, np.float32(lr), 10.0)
            assert abs(float(opt.gnorm[0]) - gn) <= 2e-6 * gn
            np.testing.assert_allclose(_np(pt), p, rtol=2e-6, atol=1e-7)
            # clipped g carries the 2e-6 norm difference; m = 0.1 g + 0.9 m can cancel -> atol scaled to |m|max
            np.testing.assert_allclose(_np(opt.m), m, rtol=2e-6, atol=4e-6 * np.abs(m).max())
            np.testing.assert_allclose(_np(opt.v), v, rtol=4e-6, atol=1e-12)
        assert int(opt.count[0]) == 10


def test_radam_kernel_vs_torch_optim_radam(gpu):
    """radam_apply_kernel (pqn_radam_clip_step: optax.chain(clip_by_global_norm, radam), pqn_minatar.py:159-162) against
    torch.optim.RAdam in f64 -- an independent implementation of the published algorithm -- over 200 steps, through the
    rho <= 5 warm-up into the rectified branch, clipped and unclipped.  The libraries differ only by where eps enters
    (<= eps / sqrt(v_hat) relative per step; see tests/test_oracle_cpu.py::test_oracle_radam_vs_torch_optim_radam)."""
    from purejaxql_amd import ops
    from tests.test_oracle_cpu import _torch_radam_reference
    rng = np.random.default_rng(11)
    n, steps, lr = 132475, 200, 5e-4
    p0 = rng.standard_normal(n).astype(np.float32)
    grads = [(rng.standard_normal(n) * (0.03 if t % 7 else 0.3) * np.exp(-t / 120.0)).astype(np.float32) for t in range(steps)]
    for max_norm in (1e9, 10.0):
        ref = _torch_radam_reference(p0, grads, lr, max_norm)
        pt = torch.from_numpy(p0.copy()).to(gpu)
        opt = ops.FlatRAdam(pt, lr, max_norm)
        clipped = 0
        for t, g in enumerate(grads):
            opt.step(torch.from_numpy(g).to(gpu))
            if t in (0, 4, 5, 6, 20, 199):
                clipped += int(not float(opt.gnorm[0]) < max_norm)
                np.testing.assert_allclose(_np(pt), ref[t], rtol=2e-7 * np.sqrt(t + 1.0), atol=1e-7 + 2e-5 * lr * (t + 1),
                                           err_msg=f"step {t} max_norm {max_norm}")
        assert int(opt.count[0]) == steps and (clipped > 0) == (max_norm == 10.0)


def test_product_network_vs_oracle_network(gpu, oracle):
    """torch-on-GPU fp32 network vs the oracle's numpy network (same theta)."""
    from purejaxql_amd.networks import QNetwork
    net = QNetwork("cnn", (10, 10, 4), 3, device=gpu)
    theta = net.init(1)
    x = (torch.rand(256, 10, 10, 4, device=gpu) < 0.15).float()
    q = net.apply(net.views(theta), x)
    p = oracle.unflatten(_np(theta), oracle.cnn_shapes((10, 10, 4), 3))
    np.testing.assert_allclose(_np(q), oracle.net_forward("cnn", p, _np(x)), rtol=1e-4, atol=1e-5)


_ORACLE_RUNS = {}   # (script, game, shape) -> the oracle loop's result from net.init(123), shared by the operand-mode cases of the 4096-env shapes

@pytest.mark.parametrize("alg,env_name,extra", [
    ("pqn_minatar", "Breakout-MinAtar", {"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2,
                                         "_BACKEND": "fused"}),
    ("pqn_minatar", "Breakout-MinAtar", {"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2,
                                         "_BACKEND": "torch"}),
    ("pqn_minatar", "Breakout-MinAtar", {"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2,
                                         "_BACKEND": "fused", "_DRIVER": False}),   # per-kernel Python loop (grad_hook path)
    ("pqn_minatar", "Breakout-MinAtar", {"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2,
                                         "_BACKEND": "fused", "_GRAPH": False}),    # C++ enqueue without hipGraph
    ("pqn_minatar", "Breakout-MinAtar", {"NUM_ENVS": 1024, "NUM_STEPS": 32, "NUM_MINIBATCHES": 32, "NUM_EPOCHS": 2}),
    # bf16x3 split-operand fc1 products: the same tolerances as the f32-MFMA mode
    ("pqn_minatar", "Breakout-MinAtar", {"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2,
                                         "MATMUL_DTYPE": "bf16x3"}),
    ("pqn_minatar", "Breakout-MinAtar", {"NUM_ENVS": 1024, "NUM_STEPS": 32, "NUM_MINIBATCHES": 32, "NUM_EPOCHS": 2,
                                         "MATMUL_DTYPE": "bf16x3"}),
    ("pqn_minatar", "Freeway-MinAtar", {"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2,
                                        "MATMUL_DTYPE": "bf16x3"}),
    # the headline shape itself (BASELINE.json metric; bench.py's workload): one whole update, fused + hipGraph
    ("pqn_minatar", "Breakout-MinAtar", {"NUM_ENVS": 4096, "NUM_STEPS": 32, "NUM_MINIBATCHES": 32, "NUM_EPOCHS": 2}),   # MATMUL_DTYPE auto -> bf16x3
    ("pqn_minatar", "Breakout-MinAtar", {"NUM_ENVS": 4096, "NUM_STEPS": 32, "NUM_MINIBATCHES": 32, "NUM_EPOCHS": 2, "MATMUL_DTYPE": "f32"}),
    ("pqn_minatar", "Asterix-MinAtar", {"NUM_ENVS": 4096, "NUM_STEPS": 32, "NUM_MINIBATCHES": 32, "NUM_EPOCHS": 2}),
    ("pqn_minatar", "Asterix-MinAtar", {"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2}),
    ("pqn_minatar", "Freeway-MinAtar", {"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2}),
    ("pqn_minatar", "SpaceInvaders-MinAtar", {"NUM_ENVS": 64, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2}),
    ("pqn_cartpole", "CartPole-v1", {"NUM_ENVS": 4, "NUM_STEPS": 16, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2}),
    ("pqn_cartpole", "CartPole-v1", {"NUM_ENVS": 4, "NUM_STEPS": 16, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2,
                                     "_BACKEND": "torch"}),
    # the alternative env of the reference's pqn_cartpole.yaml (fused MLP kernels, 6-float observation, 3 actions)
    ("pqn_cartpole", "Acrobot-v1", {"NUM_ENVS": 16, "NUM_STEPS": 16, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2}),
])
def test_make_train_end_to_end_vs_oracle(gpu, oracle, alg, env_name, extra):
    """Whole loop (rollout + Q(lambda) + minibatch updates) vs the oracle loop from the same
    initial parameters and keys.  Tolerances: params rtol 2e-3 / atol 2e-5 (fp32 summation order
    differs between MFMA / rocBLAS / numpy), scalar metrics 1e-3.  Small configs run 3 updates;
    the 1024-env config runs 1: once ~1e5 greedy decisions have been taken, a 1e-6 difference in
    q flips an argmax somewhere and the two trajectories legitimately part ways (RL is chaotic),
    so multi-update comparisons at that size measure chaos, not correctness."""
    n_upd = 3 if extra["NUM_ENVS"] <= 64 else 1
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train, seed_keys
    cfg = flatten(load_config([f"+alg={alg}"]))
    cfg.update(extra)
    cfg.update({"ENV_NAME": env_name, "TOTAL_TIMESTEPS": n_upd * cfg["NUM_ENVS"] * cfg["NUM_STEPS"],
                "TOTAL_TIMESTEPS_DECAY": 30 * cfg["NUM_ENVS"] * cfg["NUM_STEPS"], "TEST_DURING_TRAINING": False})
    ocfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
    key = seed_keys(0, 1)[0]
    torch.manual_seed(0)
    # shared initial parameters
    from purejaxql_amd.networks import QNetwork
    otrain = oracle.make_train(ocfg)
    n_params = sum(int(np.prod(s)) for s in otrain.shapes.values())
    kind = otrain.kind
    oe = oracle.OracleEnv(env_name)
    env_obs = oe.obs_shape
    net = QNetwork(kind, env_obs, oe.num_actions, hidden_size=cfg.get("HIDDEN_SIZE", 128),
                   num_layers=cfg.get("NUM_LAYERS", 2), device=gpu)
    assert net.num_params == n_params
    theta0 = net.init(123)
    cfg["_INIT_PARAMS"] = theta0
    train = make_train(cfg, device="cuda:0")
    assert train.backend == extra.get("_BACKEND", "fused")
    out = train(key)
    if train.backend == "fused" and extra.get("_DRIVER", True):
        want = "eager" if extra.get("_GRAPH", True) is False else "graph"
        assert out["runner_state"]["driver"] == want, out["runner_state"]["driver_graph_error"]
    # the oracle loop does not depend on the operand mode / driver switches of the GPU side: the 4096-env cases (28 s of numpy + float64
    # each) share one run per (script, game, shape)
    ck = (alg, env_name, tuple(sorted((k, str(v)) for k, v in extra.items() if not k.startswith("_") and k != "MATMUL_DTYPE")))
    big = cfg["NUM_ENVS"] * cfg["NUM_STEPS"] >= 100000
    if big and ck in _ORACLE_RUNS:
        oout = _ORACLE_RUNS[ck]
        assert np.array_equal(oout["th0"], _np(theta0))
    else:
        oout = otrain(key, _np(theta0))
        if big:
            oout = {"metrics": oout["metrics"], "theta": oout["theta"], "shards": oout["shards"][:1], "th0": _np(theta0).copy(), "th64": None}
            _ORACLE_RUNS[ck] = oout
    assert cfg["NUM_UPDATES"] == n_upd
    for u in range(n_upd):
        om = oout["metrics"][u]
        for k in ("env_step", "update_steps", "grad_steps"):
            assert float(out["metrics"][k][u]) == om[k]
        for k in ("td_loss", "qvals", "returned_episode_returns", "returned_episode_lengths", "timestep",
                  "returned_episode", "discount"):
            assert abs(float(out["metrics"][k][u]) - om[k]) <= 1e-3 * max(1.0, abs(om[k])), (u, k)
    # RAdam steps are scale-free: an element whose gradient is at rounding-noise level still moves by
    # ~lr*r_t per step, so isolated elements may differ by O(lr) between two f32 implementations.
    th, oth, th0 = _np(out["runner_state"]["theta"]), oout["theta"], _np(theta0)
    d = np.abs(th - oth)
    bad = d > (2e-5 + 2e-3 * np.abs(oth))
    if cfg["NUM_ENVS"] * cfg["NUM_STEPS"] < 100000:
        assert bad.mean() < 1e-3 and d.max() < cfg["LR"], (int(bad.sum()), float(d.max()))
    else:
        # 64 optimizer steps on 4096-sample minibatches: a gradient element that is a cancelling sum of 4096 x 64 terms
        # carries an ABSOLUTE f32 rounding error of ~1e-5 max|g| (measured per step AT THE SAME theta by
        # test_full_size_sgd_trajectory_gradients_match_oracle_at_same_theta: <= 1e-4), which for the smallest elements is
        # a large RELATIVE error; RAdam's m_hat / sqrt(v_hat) turns exactly those into lr-sized steps of implementation-
        # dependent sign.  The fused kernels, the torch-op network and the numpy oracle are three such implementations:
        # the two GPU paths end 2.4e-3 of the update apart, either is 3.9e-2 from numpy (tools/debug_e2e_groups.py,
        # tools/debug_epoch2.py; gpurun_out r2b / r2e).  So the whole-update criterion at this size is on the update
        # vector: direction (cosine), relative L2, the share of elements outside the element-wise band, worst element.
        upd, oupd = th - th0, oth - th0
        cos = float(np.dot(upd, oupd) / (np.linalg.norm(upd) * np.linalg.norm(oupd)))
        rel = float(np.linalg.norm(upd - oupd) / np.linalg.norm(oupd))
        assert cos > 0.998 and rel < 6e-2 and bad.mean() < 1e-2 and d.max() < 4 * cfg["LR"], (cos, rel, float(bad.mean()), float(d.max()))
        if kind == "cnn" and cfg["NORM_TYPE"] == "layer_norm" and not cfg.get("NORM_INPUT", False) and n_upd == 1:
            # round 6: the numpy-f32 oracle is the outlier of that comparison (profiles/r06_v4_f64_learn_phase.txt); against the float64
            # learn phase on the oracle's own rollout record the kernels are held 30x tighter
            import pqn_oracle_f64 as o64
            if oout.get("th64") is None:
                sh = oout["shards"][0]
                th64 = o64.learn_phase(ocfg, otrain.shapes, th0, sh["of"], sh["af"], sh["tf"], oracle.fold_in(int(key) & 0xFFFFFFFFFFFFFFFF, 4))[0]
                if big:
                    oout["th64"], oout["shards"] = th64, None     # the record is not needed again
            else:
                th64 = oout["th64"]
            u64 = th64 - th0
            rel64, rel_np = float(np.linalg.norm(upd - u64) / np.linalg.norm(u64)), float(np.linalg.norm(oupd - u64) / np.linalg.norm(u64))
            # (measured: 2.4e-3 for the single-tile kernels of a one-seed launch, 2.1e-4 for the position-parallel kernels, 3.9e-2 numpy-f32)
            assert np.isfinite(th64).all() and rel64 < 6e-3 and rel64 < 0.2 * rel_np, (rel64, rel_np, rel)


@pytest.mark.parametrize("alg,env_name,norm_type,norm_input", [
    ("pqn_minatar", "Breakout-MinAtar", "batch_norm", False),
    ("pqn_minatar", "Breakout-MinAtar", "batch_norm", True),
    ("pqn_minatar", "Asterix-MinAtar", "none", False),
    ("pqn_cartpole", "CartPole-v1", "batch_norm", True),
    ("pqn_cartpole", "CartPole-v1", "none", True),
])
def test_make_train_norm_variants_vs_oracle(gpu, oracle, alg, env_name, norm_type, norm_input):
    """NORM_TYPE = batch_norm / none and NORM_INPUT=True (pqn_minatar.py:31-36,61-66): these run the torch-op
    Q-network over the HIP env / eps-greedy / Q(lambda) / RAdam kernels; whole loop vs the oracle loop,
    including the batch_stats collection.  2 updates of a small config; BatchNorm on x/255-scaled conv
    outputs cancels in f32 (E[x^2]-E[x]^2), hence the looser scalar tolerance for that case."""
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.pqn import make_train, seed_keys
    cfg = flatten(load_config([f"+alg={alg}"]))
    small = ({"NUM_ENVS": 16, "NUM_STEPS": 8, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2} if alg == "pqn_minatar"
             else {"NUM_ENVS": 8, "NUM_STEPS": 16, "NUM_MINIBATCHES": 4, "NUM_EPOCHS": 2})
    cfg.update(small)
    cfg.update({"ENV_NAME": env_name, "NORM_TYPE": norm_type, "NORM_INPUT": norm_input,
                "TOTAL_TIMESTEPS": 2 * cfg["NUM_ENVS"] * cfg["NUM_STEPS"],
                "TOTAL_TIMESTEPS_DECAY": 30 * cfg["NUM_ENVS"] * cfg["NUM_STEPS"], "TEST_DURING_TRAINING": False})
    ocfg = dict(cfg)
    key = seed_keys(1, 1)[0]
    otrain = oracle.make_train(ocfg)
    oe = oracle.OracleEnv(env_name)
    net = QNetwork(otrain.kind, oe.obs_shape, oe.num_actions, norm_type=norm_type, norm_input=norm_input,
                   hidden_size=cfg.get("HIDDEN_SIZE", 128), num_layers=cfg.get("NUM_LAYERS", 2), device=gpu)
    assert list(net.shapes) == list(otrain.shapes)
    theta0 = net.init(11)
    cfg["_INIT_PARAMS"] = theta0
    train = make_train(cfg, device="cuda:0")
    assert train.backend == "torch"
    out = train(key)
    oout = otrain(key, _np(theta0))
    tol = 5e-3 if (norm_type == "batch_norm" and not norm_input and otrain.kind == "cnn") else 1e-3
    for u in range(2):
        om = oout["metrics"][u]
        for k in ("td_loss", "qvals", "returned_episode_returns", "returned_episode_lengths", "timestep", "discount"):
            assert abs(float(out["metrics"][k][u]) - om[k]) <= tol * max(1.0, abs(om[k])), (u, k)
    d = np.abs(_np(out["runner_state"]["theta"]) - oout["theta"])
    bad = d > (2e-5 + 2e-3 * np.abs(oout["theta"]))
    # (scale-free RAdam steps amplify the f32 variance cancellation of that ill-conditioned case)
    assert bad.mean() < (5e-2 if tol > 1e-3 else 5e-3) and d.max() < 2 * cfg["LR"], (int(bad.sum()), float(d.max()))
    bs = out["runner_state"]["batch_stats"]
    assert sorted(bs) == sorted(oout["batch_stats"])
    for k, v in oout["batch_stats"].items():
        assert np.abs(_np(bs[k]) - v).max() <= 10 * tol * max(np.abs(v).max(), 1e-3), k   # running moments, per-array scale


@pytest.mark.parametrize("name", ["Breakout-MinAtar", "Asterix-MinAtar", "Freeway-MinAtar", "SpaceInvaders-MinAtar", "CartPole-v1",
                                  "Acrobot-v1"])
def test_hip_envs_hash_to_regression_pins(gpu, name):
    """The HIP env kernels (reset / step / auto-reset / LogWrapper, f32 observation surface) reproduce the committed
    SHA-256 digests of tests/golden/regression_pins.json on the pinned keys and actions -- no oracle in the loop.
    (CartPole included: its sin / cos are explicit f32 arithmetic shared with the oracle, see test_cartpole_step_vs_oracle.)"""
    import hashlib
    import importlib.util
    import json
    import os
    from purejaxql_amd import _lib
    from purejaxql_amd.envs import LogWrapper, make
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_regression_pins", os.path.join(here, "golden", "make_regression_pins.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pin = json.load(open(os.path.join(here, "golden", "regression_pins.json")))["envs"][name]
    n, steps, key = pin["n"], pin["steps"], pin["key"]
    env, params = make(name, device=gpu)
    env = LogWrapper(env)
    rng = np.random.default_rng(7)
    obs, state = env.reset(key, params, n)
    h = hashlib.sha256(mod.digest(_np(obs)).encode())
    tot_r, tot_d = 0.0, 0
    num_actions = env.action_space(params).n
    for t in range(steps):
        a = rng.integers(0, num_actions, n).astype(np.int32)
        obs, state, r, d, info = env.step(_lib.fold_in(key, 1 + t), state, torch.from_numpy(a).to(gpu), params, inplace=True)
        h.update(mod.digest(_np(obs), _np(r), _np(d).astype(bool), _np(info["returned_episode_returns"]),
                            _np(info["returned_episode_lengths"]).astype(np.int32),
                            _np(info["timestep"]).astype(np.int32)).encode())
        tot_r += float(r.sum())
        tot_d += int(d.sum())
    assert (h.hexdigest(), tot_r, tot_d) == (pin["sha256"], pin["sum_reward"], pin["num_done"])


@pytest.mark.parametrize("name,n,ratio,steps", [("Breakout-MinAtar", 1024, 16, 400), ("Asterix-MinAtar", 512, 16, 500),
                                                ("Breakout-MinAtar", 64, 64, 300), ("Breakout-MinAtar", 48, 1, 200),
                                                ("CartPole-v1", 256, 8, 300), ("Acrobot-v1", 128, 8, 700)])
def test_optimistic_reset_wrapper_bit_exact_vs_oracle(gpu, oracle, name, n, ratio, steps):
    """OptimisticResetVecEnvWrapper(LogWrapper(env)) (utils/craftax_wrappers.py:83-148, wrapper order of
    pqn_craftax.py:99-108) as an option of the HIP step kernels, vs the oracle's restatement: reward / done / info /
    observation / full state incl. the restarted LogWrapper record, and the reset slot each finished env took.
    ratio = n: a single reset shared by every finished env; ratio = 1: every env has its own default slot."""
    from purejaxql_amd.envs import FlattenObservationWrapper, LogWrapper, OptimisticResetVecEnvWrapper, make
    base, params = make(name, device=gpu)
    flat = len(base.obs_shape) == 1
    inner = LogWrapper(FlattenObservationWrapper(base) if flat else base)
    env = OptimisticResetVecEnvWrapper(inner, num_envs=n, reset_ratio=ratio)
    assert env.num_resets == n // ratio
    oenv = oracle.OracleEnv(name)
    obs, state = env.reset(31, params)
    oobs, ost = oenv.reset(31, n)
    np.testing.assert_array_equal(_np(obs), oobs)
    rng = np.random.default_rng(n + ratio)
    a_n = inner.action_space(params).n
    shared, own = 0, 0
    for t in range(steps):
        act = rng.integers(0, a_n, n).astype(np.int32)
        key = 4000 + t
        obs, state, r, d, info = env.step(key, state, torch.from_numpy(act).to(gpu), params, want_slots=True)
        oobs, ost, orr, od, oinfo = oenv.step_optimistic(key, ost, act, ratio)
        np.testing.assert_array_equal(_np(r), orr)
        np.testing.assert_array_equal(_np(d), od)
        np.testing.assert_array_equal(_np(info["reset_slot"]), oinfo["reset_slot"])
        for k in ("discount", "returned_episode_returns", "returned_episode_lengths", "timestep", "returned_episode"):
            np.testing.assert_array_equal(_np(info[k]), oinfo[k], err_msg=k)
        np.testing.assert_array_equal(_np(obs), oobs)
        if t % 20 == 0 or t == steps - 1:
            sf = _check_state(inner, oenv, state, ost)
        sl = oinfo["reset_slot"][od]
        nd = int(od.sum())
        if nd:
            ranks_own = min(nd, n // ratio)
            assert sorted(sl.tolist())[:0] == []   # (slots are checked element-wise above; count the two kinds)
            own += ranks_own
            shared += nd - ranks_own
            # a finished env's LogWrapper record restarted from zero, a running one keeps counting
            lw = oenv.log_words(ost)
            assert (lw[od] == 0).all() and (lw[~od, 4] > 0).all()
    assert own > 0 and (ratio == 1 or name == "CartPole-v1" or shared + own > 0)
