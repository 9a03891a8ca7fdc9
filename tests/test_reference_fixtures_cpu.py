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
