"""The oracle against fixtures generated FROM THE REFERENCE ITSELF (tests/golden/make_reference_fixtures.py: gymnax
0.0.6 + purejaxql/pqn_minatar.py under jax).  The fixtures cannot be produced in the build image (no jax there); every
test here SKIPS while its .npz is absent and pins the oracle the moment the files are committed."""
import numpy as np

from tests.reference_fixture_maps import (breakout_canon, cartpole_canon, flax_grads, flax_params_to_theta, load,
                                          spaceinvaders_canon)


def test_q_lambda_vs_reference(oracle):
    rec = load("ref_qlambda.npz")
    ka = {"reward": [1, 0, 2, 1], "qmax": [5, 6, 7, 8]}
    for i, d in enumerate(([0, 0, 0, 0], [0, 1, 0, 1])):
        got = oracle.q_lambda(np.asarray(ka["reward"], np.float32)[:, None], np.asarray(d, np.uint8)[:, None],
                              np.asarray(ka["qmax"], np.float32)[:, None], np.asarray([9.0], np.float32), 0.99, 0.65)
        np.testing.assert_allclose(got, rec[f"ka1_{i}"], rtol=1e-6, atol=1e-6)
    got = oracle.q_lambda(rec["reward"], rec["done"], rec["qmax"], rec["last_q"], 0.99, 0.65, quirk=True)
    np.testing.assert_allclose(got, rec["target"], rtol=1e-6, atol=1e-6)


def test_radam_clip_and_schedules_vs_reference(oracle):
    rec = load("ref_radam.npz")
    p = rec["p0"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    for i in range(rec["grads"].shape[0]):
        lr = oracle.linear_schedule(5e-4, 1e-20, 7, i)
        assert abs(lr - rec["lr"][i]) <= 1e-12 + 1e-7 * rec["lr"][i]
        oracle.radam_clip_step(p, rec["grads"][i], m, v, i, np.float32(lr), 10.0)
        np.testing.assert_allclose(p, rec["params"][i], rtol=2e-6, atol=1e-7)
    for c, e in zip((0, 1, 100, 244, 245, 3000), rec["eps"]):
        assert abs(oracle.linear_schedule(1.0, 0.05, 0.1 * 2441, c) - e) <= 1e-6
    assert abs(oracle.linear_schedule(1.0, 0.05, 0, 7) - float(rec["eps_degenerate"])) <= 1e-12


def test_qnetwork_forward_and_gradient_vs_reference(oracle):
    rec = load("ref_qnet.npz")
    obs, action, target = rec["obs"], rec["action"], rec["target"]
    for norm_type, norm_input in (("layer_norm", False), ("batch_norm", False), ("layer_norm", True)):
        tag = f"{norm_type}_{int(norm_input)}"
        shapes = oracle.cnn_shapes((10, 10, 4), 3, norm_type)
        theta = flax_params_to_theta(rec, tag, shapes)
        p = oracle.unflatten(theta, shapes)
        stats = oracle.init_batch_stats("cnn", (10, 10, 4), 128, 2, norm_type, norm_input)
        q_eval = oracle.net_forward("cnn", p, obs, norm_type == "layer_norm", 2, norm_type=norm_type, norm_input=norm_input,
                                    train=False, stats=stats)
        np.testing.assert_allclose(q_eval, rec[f"{tag}/q_eval"], rtol=1e-4, atol=1e-5)
        new_stats = {}
        loss, chosen, g = oracle.net_loss_grad("cnn", p, shapes, obs, action, target, norm_type == "layer_norm", 2,
                                               norm_type=norm_type, norm_input=norm_input, stats=stats, new_stats=new_stats)
        assert abs(float(loss) - float(rec[f"{tag}/loss"])) <= 1e-5 * max(1.0, abs(float(loss)))
        np.testing.assert_allclose(chosen, rec[f"{tag}/chosen"], rtol=1e-4, atol=1e-5)
        gr = flax_grads(rec, tag, shapes)
        np.testing.assert_allclose(g, gr, rtol=2e-3, atol=1e-5 * np.abs(gr).max() + 1e-9)
        for k, v in new_stats.items():      # running moments after one train-mode call (flax momentum 0.99)
            np.testing.assert_allclose(v, rec[f"{tag}/new_batch_stats/{k}"], rtol=1e-4, atol=1e-6)


def _step_env_check(oracle, name, canon):
    rec = load(f"ref_env_{name}.npz")
    oenv = oracle.OracleEnv(name)
    assert oenv.max_steps == int(rec["max_steps_in_episode"])
    n_steps, n = rec["actions"].shape
    np.testing.assert_array_equal(oenv.reset(0, n)[0].shape, rec["obs0"].shape)
    for t in range(n_steps):
        si = canon(rec, "before", t)
        st = {"si": np.ascontiguousarray(si), "sf": np.zeros((n, 1), np.float32), "ep_ret": np.zeros(n, np.float32),
              "ep_len": np.zeros(n, np.int32), "ret_ret": np.zeros(n, np.float32), "ret_len": np.zeros(n, np.int32),
              "timestep": np.zeros(n, np.int32)}
        obs, st, r, d, _ = oenv.step(0, st, rec["actions"][t], autoreset=False)     # step_env alone
        np.testing.assert_array_equal(st["si"], canon(rec, "after", t), err_msg=f"{name} state after step {t}")
        np.testing.assert_array_equal(obs, rec["step_env_obs"][t], err_msg=f"{name} obs after step {t}")
        np.testing.assert_array_equal(r, rec["reward"][t])
        np.testing.assert_array_equal(d, rec["done"][t])


def test_breakout_step_env_vs_gymnax(oracle):
    _step_env_check(oracle, "Breakout-MinAtar", breakout_canon)


def test_spaceinvaders_step_env_vs_gymnax(oracle):
    _step_env_check(oracle, "SpaceInvaders-MinAtar", spaceinvaders_canon)


def test_cartpole_step_env_vs_gymnax(oracle):
    rec = load("ref_env_CartPole-v1.npz")
    oenv = oracle.OracleEnv("CartPole-v1")
    n_steps, n = rec["actions"].shape
    for t in range(n_steps):
        si, sf = cartpole_canon(rec, "before", t)
        st = {"si": np.ascontiguousarray(si), "sf": np.ascontiguousarray(sf), "ep_ret": np.zeros(n, np.float32),
              "ep_len": np.zeros(n, np.int32), "ret_ret": np.zeros(n, np.float32), "ret_len": np.zeros(n, np.int32),
              "timestep": np.zeros(n, np.int32)}
        obs, st, r, d, _ = oenv.step(0, st, rec["actions"][t], autoreset=False)
        si2, sf2 = cartpole_canon(rec, "after", t)
        np.testing.assert_array_equal(st["si"], si2)
        np.testing.assert_allclose(st["sf"], sf2, rtol=2e-6, atol=2e-6)     # f32 sin / cos of two libms
        np.testing.assert_array_equal(r, rec["reward"][t])
        np.testing.assert_array_equal(d, rec["done"][t])


def test_craftax_qnetwork_batchrenorm_one_step_loss_vs_reference(oracle):
    """pqn_craftax.py's QNetwork (BatchRenorm input, LayerNorm MLP, :33-62) and the `Q_LAMBDA: False` loss (:287-304) from
    the reference itself: oracle.net_forward / net_loss_grad_1step / brn_fwd bookkeeping (utils/batch_renorm.py:95-116), cold
    and warm statistics.  Skips while ref_craftax_qnet.npz is absent."""
    rec = load("ref_craftax_qnet.npz")
    obs, nxt, action, reward, done = rec["obs"], rec["next_obs"], rec["action"], rec["reward"], rec["done"]
    nb, d = obs.shape
    a = int(rec["cold/all_q"].shape[1])
    h = int(rec["cold/params/Dense_0/kernel"].shape[1])
    layers = sum(1 for k in rec.files if k.startswith("cold/params/Dense_") and k.endswith("/kernel")) - 1
    shapes = oracle.mlp_shapes(d, a, h, layers, "layer_norm", True)
    nkw = dict(layers=layers, norm_type="layer_norm", norm_input=True, renorm=True)
    for tag in ("cold", "warm"):
        theta = flax_params_to_theta(rec, tag, shapes)
        p = oracle.unflatten(theta, shapes)
        stats = {k[len(tag) + len("/batch_stats/"):]: np.asarray(rec[k]) for k in rec.files if k.startswith(f"{tag}/batch_stats/")}
        q_eval = oracle.net_forward("mlp", p, obs, train=False, stats=stats, **nkw)
        np.testing.assert_allclose(q_eval, rec[f"{tag}/q_eval"], rtol=1e-4, atol=1e-5)
        new_stats = {}
        loss, chosen, g = oracle.net_loss_grad_1step("mlp", p, shapes, obs, nxt, action, reward, done, float(rec["gamma"]),
                                                     stats=stats, new_stats=new_stats, **nkw)
        assert abs(float(loss) - float(rec[f"{tag}/loss"])) <= 1e-5 * max(1.0, abs(float(loss)))
        np.testing.assert_allclose(chosen, rec[f"{tag}/chosen"], rtol=1e-4, atol=1e-5)
        gr = flax_grads(rec, tag, shapes)
        np.testing.assert_allclose(g, gr, rtol=2e-3, atol=1e-5 * np.abs(gr).max() + 1e-9)
        for k, v in new_stats.items():
            np.testing.assert_allclose(np.asarray(v, np.float64), np.asarray(rec[f"{tag}/new_batch_stats/{k}"], np.float64), rtol=1e-4, atol=1e-6)


def test_acrobot_step_env_vs_gymnax(oracle):
    """Acrobot-v1 (config/alg/pqn_cartpole.yaml:24): the oracle's restatement of gymnax's rk4 dynamics on the reference's own
    states.  Skips while ref_env_Acrobot-v1.npz is absent."""
    rec = load("ref_env_Acrobot-v1.npz")
    oenv = oracle.OracleEnv("Acrobot-v1")
    n_steps, n = rec["actions"].shape
    f = ["joint_angle1", "joint_angle2", "velocity_1", "velocity_2"]
    have = sorted(k[len("before/"):] for k in rec.files if k.startswith("before/"))
    assert all(f"before/{k}" in rec.files for k in f + ["time"]), f"fixture state fields {have}: update the field list here"
    for t in range(n_steps):
        sf = np.stack([np.asarray(rec[f"before/{k}"][t], np.float32) for k in f], axis=1)
        si = np.asarray(rec["before/time"][t], np.int32).reshape(-1, 1)
        st = {"si": np.ascontiguousarray(si), "sf": np.ascontiguousarray(sf), "ep_ret": np.zeros(n, np.float32),
              "ep_len": np.zeros(n, np.int32), "ret_ret": np.zeros(n, np.float32), "ret_len": np.zeros(n, np.int32),
              "timestep": np.zeros(n, np.int32)}
        obs, st, r, d, _ = oenv.step(0, st, rec["actions"][t], autoreset=False)
        sf2 = np.stack([np.asarray(rec[f"after/{k}"][t], np.float32) for k in f], axis=1)
        np.testing.assert_allclose(st["sf"], sf2, rtol=2e-5, atol=2e-5)       # rk4 in f32: two libms, four stages
        np.testing.assert_array_equal(r, rec["reward"][t])
        np.testing.assert_array_equal(d, rec["done"][t])


def test_optimistic_reset_wrapper_vs_reference(oracle):
    """OptimisticResetVecEnvWrapper(LogWrapper(CartPole-v1), 16, 4).step (utils/craftax_wrappers.py:83-148): from the
    reference's own pre-step states, the envs that finish, the rewards, and that every finished env continues from ONE of the
    num_envs / reset_ratio freshly reset states the reference drew (which one follows jax.random.choice and is not
    reproduced).  Skips while ref_optimistic.npz is absent."""
    rec = load("ref_optimistic.npz")
    n_steps, n = rec["actions"].shape
    oenv = oracle.OracleEnv("CartPole-v1")
    for t in range(n_steps):
        sf = np.stack([np.asarray(rec[f"before/env_state/{k}"][t], np.float32) for k in ("x", "x_dot", "theta", "theta_dot")], axis=1)
        si = np.asarray(rec["before/env_state/time"][t], np.int32).reshape(-1, 1)
        st = {"si": np.ascontiguousarray(si), "sf": np.ascontiguousarray(sf), "ep_ret": np.zeros(n, np.float32),
              "ep_len": np.zeros(n, np.int32), "ret_ret": np.zeros(n, np.float32), "ret_len": np.zeros(n, np.int32),
              "timestep": np.zeros(n, np.int32)}
        _obs, st, r, d, _ = oenv.step(0, st, rec["actions"][t], autoreset=False)
        np.testing.assert_array_equal(d, rec["done"][t], err_msg=f"step {t}")
        np.testing.assert_array_equal(r, rec["reward"][t], err_msg=f"step {t}")
        live = ~np.asarray(rec["done"][t], bool)
        after = np.stack([np.asarray(rec[f"after/env_state/{k}"][t], np.float32) for k in ("x", "x_dot", "theta", "theta_dot")], axis=1)
        np.testing.assert_allclose(st["sf"][live], after[live], rtol=2e-6, atol=2e-6)
        fresh = after[~live]
        assert len(np.unique(fresh, axis=0)) <= n // int(rec["reset_ratio"]), "more distinct reset states than num_resets"
        assert (np.abs(fresh) <= 0.05 + 1e-6).all()              # CartPole's reset distribution: U(-0.05, 0.05)^4
