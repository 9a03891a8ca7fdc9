"""Craftax-Classic-Symbolic-v1 on the GPU (csrc/pqn_craftax.hip) vs the oracle's restatement (oracle/craftax_classic.c),
bit for bit: procedural worlds, the transition rule with its counter-based draws, auto-reset and optimistic resets,
the LogWrapper record, the 1345-float symbolic observation.  (The rules themselves are third-party and restated:
parity with the craftax package is unpinned -- see the oracle's header; the known-answer steps are in
test_craftax_env_cpu.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
NAME = "Craftax-Classic-Symbolic-v1"
S = 4096


def _np(t):
    return t.detach().cpu().numpy()


def _check_state(env, oenv, state, ost):
    si, sf, log = env.export_state(state)
    np.testing.assert_array_equal(_np(si), ost["si"])
    np.testing.assert_array_equal(_np(sf), ost["sf"])
    np.testing.assert_array_equal(_np(log).view(np.uint32), oenv.log_words(ost))


def _policy(rng, n, t):
    """random actions, biased towards moving and interacting so that rules beyond walking get exercised"""
    a = rng.integers(0, 17, n).astype(np.int32)
    m = rng.random(n)
    a = np.where(m < 0.35, rng.integers(1, 5, n), a)
    a = np.where((m >= 0.35) & (m < 0.6), 5, a)
    return a.astype(np.int32)


def test_craftax_reset_world_and_observation(gpu, oracle):
    from purejaxql_amd.envs import LogWrapper, make
    env, params = make(NAME, device=gpu)
    assert env.obs_shape == (1345,) and env.num_actions == 17 and params.max_steps_in_episode == 10000
    env = LogWrapper(env)
    oenv = oracle.OracleEnv(NAME)
    for key, n in ((3, 1), (11, 37), (2026, 256)):
        obs, state = env.reset(key, params, n)
        oobs, ost = oenv.reset(key, n)
        np.testing.assert_array_equal(_np(obs), oobs)
        _check_state(env, oenv, state, ost)


@pytest.mark.parametrize("n,steps", [(256, 600), (33, 300), (1024, 150)])   # 1024 = C5's NUM_ENVS (pqn_craftax.yaml:4)
def test_craftax_step_autoreset_bit_exact_vs_oracle(gpu, oracle, n, steps):
    from purejaxql_amd.envs import LogWrapper, make
    env, params = make(NAME, device=gpu)
    env = LogWrapper(env)
    oenv = oracle.OracleEnv(NAME)
    obs, state = env.reset(5, params, n)
    oobs, ost = oenv.reset(5, n)
    rng = np.random.default_rng(n)
    dones, ach_seen = 0, 0
    for t in range(steps):
        a = _policy(rng, n, t)
        if t == 100:   # hand out tools and materials once so that mining / crafting / placing / fighting rules fire
            si, sf, log = env.export_state(state)
            si = _np(si).copy()
            si[:, S + 8:S + 20] = rng.integers(0, 4, (n, 12))
            state = env.import_state(torch.from_numpy(si), sf, log)
            ost["si"][:] = si
        key = 7000 + t
        obs, state, r, d, info = env.step(key, state, torch.from_numpy(a).to(gpu), params)
        # the achievement mask of the episodes that end with this step: the stepped state BEFORE the reset, from a copy
        ost2 = {k: v.copy() for k, v in ost.items()}
        _o2, ost2, _r2, od2, _i2 = oenv.step(key, ost2, a, autoreset=False)
        exp_ach = np.where(od2, ost2["si"][:, S + 109], 0).astype(np.int32)
        np.testing.assert_array_equal(_np(info["achievements"]), exp_ach, err_msg=f"achievements t={t}")
        ach_seen |= int(np.bitwise_or.reduce(exp_ach))
        oobs, ost, orr, od, oinfo = oenv.step(key, ost, a)
        np.testing.assert_array_equal(_np(d), od, err_msg=f"done t={t}")
        np.testing.assert_array_equal(_np(r), orr, err_msg=f"reward t={t}")
        np.testing.assert_array_equal(_np(obs), oobs, err_msg=f"obs t={t}")
        dones += int(od.sum())
        if t % 20 == 0 or t == steps - 1:
            for k in oinfo:
                np.testing.assert_array_equal(_np(info[k]), oinfo[k], err_msg=k)
            _check_state(env, oenv, state, ost)
    assert dones > 0 and ost["ret_len"].max() > 0
    assert n < 100 or ach_seen != 0            # finished episodes carried achievements in their info mask
    ach = np.bitwise_or.reduce(ost["si"][:, S + 109])
    assert bin(int(ach)).count("1") >= 4       # several different achievements were unlocked along the way


@pytest.mark.parametrize("n", [64, 1024])   # 1024 envs / ratio 16 = the C5 shape (pqn_craftax.yaml:4,25-26)
def test_craftax_optimistic_resets_bit_exact_vs_oracle(gpu, oracle, n):
    from purejaxql_amd.envs import LogWrapper, OptimisticResetVecEnvWrapper, make
    ratio = 16
    base, params = make(NAME, device=gpu)
    inner = LogWrapper(base)
    env = OptimisticResetVecEnvWrapper(inner, num_envs=n, reset_ratio=ratio)
    oenv = oracle.OracleEnv(NAME)
    obs, state = env.reset(9, params)
    oobs, ost = oenv.reset(9, n)
    np.testing.assert_array_equal(_np(obs), oobs)
    # weaken everyone so that episodes end (and several in the same step)
    si, sf, log = inner.export_state(state)
    si = _np(si).copy()
    si[:, S + 3] = 1
    si[:, S + 4:S + 7] = 0
    state = inner.import_state(torch.from_numpy(si), sf, log)
    ost["si"][:] = si
    rng = np.random.default_rng(1)
    shared = 0
    for t in range(120):
        a = _policy(rng, n, t)
        obs, state, r, d, info = env.step(300 + t, state, torch.from_numpy(a).to(gpu), params, want_slots=True)
        oobs, ost, orr, od, oinfo = oenv.step_optimistic(300 + t, ost, a, ratio)
        np.testing.assert_array_equal(_np(d), od)
        np.testing.assert_array_equal(_np(r), orr)
        np.testing.assert_array_equal(_np(info["reset_slot"]), oinfo["reset_slot"])
        np.testing.assert_array_equal(_np(obs), oobs)
        for k in ("returned_episode_returns", "returned_episode_lengths", "timestep"):
            np.testing.assert_array_equal(_np(info[k]), oinfo[k], err_msg=k)
        if od.sum() > n // ratio:
            shared += 1
        if t % 10 == 0:
            _check_state(inner, oenv, state, ost)
    _check_state(inner, oenv, state, ost)
    assert shared > 0        # at least once more envs finished than fresh worlds existed: some shared a reset


def test_craftax_script_runs_on_craftax_classic(gpu):
    """`python -m purejaxql_amd.pqn_craftax alg.ENV_NAME=Craftax-Classic-Symbolic-v1` at a reduced size: the C5 loop
    (1 step x 1 minibatch x 1 epoch, BatchRenorm input, 1-step loss, optimistic resets) end to end on the HIP env."""
    from purejaxql_amd.run import main
    outs = main(["alg.ENV_NAME=Craftax-Classic-Symbolic-v1", "alg.NUM_ENVS=128", "alg.HIDDEN_SIZE=256", "alg.NUM_LAYERS=2",
                 "alg.TOTAL_TIMESTEPS=12800", "alg.TOTAL_TIMESTEPS_DECAY=12800", "SAVE_PATH=null"], "pqn_craftax", script="craftax")
    m = outs["metrics"]
    assert m["td_loss"].shape == (1, 100) and torch.isfinite(m["td_loss"]).all() and float(m["env_step"][0, -1]) == 12800


def test_craftax_functional_step_equals_in_place(gpu):
    """env.step is functional like gymnax's (a new state comes back, the old one is untouched) unless `inplace=True`; the
    map-in-memory env does it by copying its 4.2 KB per env across before stepping the copy.  Same outputs either way, plain
    and under optimistic resets."""
    from purejaxql_amd.envs import LogWrapper, OptimisticResetVecEnvWrapper, make
    n = 64
    base, params = make(NAME, device=gpu)
    for env in (LogWrapper(base), OptimisticResetVecEnvWrapper(LogWrapper(base), num_envs=n, reset_ratio=16)):
        obs, state = env.reset(3, params, n) if isinstance(env, LogWrapper) else env.reset(3, params)
        rng = np.random.default_rng(0)
        for t in range(30):
            a = torch.from_numpy(rng.integers(0, 17, n).astype(np.int32)).to(gpu)
            before = state.words.clone()
            o1, s1, r1, d1, i1 = env.step(100 + t, state, a, params)
            assert torch.equal(state.words, before) and s1.words.data_ptr() != state.words.data_ptr()
            twin = type(state)(before.clone())
            o2, s2, r2, d2, i2 = env.step(100 + t, twin, a, params, inplace=True)
            assert s2.words.data_ptr() == twin.words.data_ptr()
            assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2) and torch.equal(s1.words, s2.words)
            for k in i1:
                assert torch.equal(i1[k], i2[k]), k
            state = s1
