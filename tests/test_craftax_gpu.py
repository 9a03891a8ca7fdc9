"""GPU tests of the Craftax-script twin of the hot path (purejaxql/pqn_craftax.py:82-468) on envs the oracle can run:
the loop differences -- wrapper-batched env with / without optimistic resets (:96-114), BatchRenorm Q-network (:33-62),
the Q_LAMBDA switch of the loss (:277-304), done-weighted metric means (:364-369) -- against the oracle's restatement
of the same script.  NUM_STEPS = NUM_MINIBATCHES = NUM_EPOCHS = 1 with Q_LAMBDA=False is exactly the loop shape of
config/alg/pqn_craftax.yaml."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


CASES = [
    # (env, overrides, updates)
    ("CartPole-v1", dict(NUM_ENVS=16, NUM_STEPS=1, NUM_MINIBATCHES=1, NUM_EPOCHS=1, Q_LAMBDA=False, NORM_INPUT=True,
                         NORM_TYPE="layer_norm", USE_OPTIMISTIC_RESETS=False), 40),   # T = 1: episodes need ~10-40 updates to end
    ("CartPole-v1", dict(NUM_ENVS=16, NUM_STEPS=8, NUM_MINIBATCHES=2, NUM_EPOCHS=2, Q_LAMBDA=True, NORM_INPUT=True,
                         NORM_TYPE="batch_norm", USE_OPTIMISTIC_RESETS=False), 3),
    ("Breakout-MinAtar", dict(NUM_ENVS=32, NUM_STEPS=8, NUM_MINIBATCHES=2, NUM_EPOCHS=1, Q_LAMBDA=False, NORM_INPUT=True,
                              NORM_TYPE="layer_norm", USE_OPTIMISTIC_RESETS=True, OPTIMISTIC_RESET_RATIO=8), 4),
    ("Breakout-MinAtar", dict(NUM_ENVS=32, NUM_STEPS=4, NUM_MINIBATCHES=1, NUM_EPOCHS=1, Q_LAMBDA=False, NORM_INPUT=False,
                              NORM_TYPE="none", USE_OPTIMISTIC_RESETS=True, OPTIMISTIC_RESET_RATIO=16), 4),
]


@pytest.mark.parametrize("env_name,over,n_upd", CASES)
def test_craftax_script_loop_vs_oracle(gpu, oracle, env_name, over, n_upd):
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.pqn import make_train, seed_keys
    cfg = flatten(load_config(["+alg=pqn_craftax"]))
    assert cfg["Q_LAMBDA"] is False and cfg["NUM_STEPS"] == 1 and cfg["NORM_INPUT"] is True   # pqn_craftax.yaml:5,11,20
    cfg.update(over)
    n, t = cfg["NUM_ENVS"], cfg["NUM_STEPS"]
    cfg.update({"ENV_NAME": env_name, "HIDDEN_SIZE": 64, "NUM_LAYERS": 2, "TOTAL_TIMESTEPS": n_upd * n * t,
                "TOTAL_TIMESTEPS_DECAY": 30 * n * t, "TEST_DURING_TRAINING": True, "TEST_INTERVAL": 0.5,
                "TEST_NUM_ENVS": 16, "TEST_NUM_STEPS": 40, "EPS_START": 0.5, "LR": 5e-4})
    ocfg = dict(cfg)
    key = seed_keys(3, 1)[0]
    otrain = oracle.make_train(ocfg, script="craftax")
    oe = oracle.OracleEnv(env_name)
    d_obs = int(np.prod(oe.obs_shape))
    net = QNetwork("mlp", (d_obs,), oe.num_actions, norm_type=cfg["NORM_TYPE"], norm_input=cfg["NORM_INPUT"], hidden_size=64,
                   num_layers=2, device=gpu, renorm=True)
    assert list(net.shapes) == list(otrain.shapes) and "BatchRenorm_0/scale" in net.shapes
    theta0 = net.init(21)
    cfg["_INIT_PARAMS"] = theta0
    train = make_train(cfg, device="cuda:0", script="craftax")
    assert train.backend == "torch"
    out = train(key)
    oout = otrain(key, _np(theta0))
    assert cfg["NUM_UPDATES"] == n_upd == len(oout["metrics"])
    saw_done = False
    for u in range(n_upd):
        om = oout["metrics"][u]
        assert float(out["metrics"]["env_step"][u]) == om["env_step"] and float(out["metrics"]["grad_steps"][u]) == om["grad_steps"]
        for k in ("td_loss", "qvals"):
            assert abs(float(out["metrics"][k][u]) - om[k]) <= 1e-3 * max(1.0, abs(om[k])), (u, k, float(out["metrics"][k][u]), om[k])
        # done-weighted means: NaN while no episode has finished inside the update (pqn_craftax.py:364-369)
        for k in ("returned_episode_returns", "returned_episode_lengths", "timestep", "returned_episode", "discount"):
            a, b = float(out["metrics"][k][u]), om[k]
            assert (math.isnan(a) and math.isnan(b)) or abs(a - b) <= 1e-3 * max(1.0, abs(b)), (u, k, a, b)
            saw_done = saw_done or not math.isnan(b)
        for k in ("test/returned_episode_returns", "test/returned_episode_lengths", "test/timestep"):
            a, b = float(out["metrics"][k][u]), om[k]
            assert (math.isnan(a) and math.isnan(b)) or abs(a - b) <= 1e-3 * max(1.0, abs(b)), (u, k, a, b)
    assert saw_done
    d = np.abs(_np(out["runner_state"]["theta"]) - oout["theta"])
    bad = d > (2e-5 + 2e-3 * np.abs(oout["theta"]))
    assert bad.mean() < 5e-3 and d.max() < 2 * cfg["LR"], (int(bad.sum()), float(d.max()))
    bs = out["runner_state"]["batch_stats"]
    assert sorted(bs) == sorted(oout["batch_stats"])
    for k, v in oout["batch_stats"].items():
        if k.endswith("/steps"):
            assert int(bs[k]) == int(v) == n_upd * cfg["NUM_MINIBATCHES"] * cfg["NUM_EPOCHS"]
        else:
            assert np.abs(_np(bs[k]) - v).max() <= 1e-3 * max(np.abs(v).max(), 1e-3), k


def test_c5_yaml_shape_loop_vs_oracle_on_craftax_classic(gpu, oracle):
    """BASELINE.json configs[4] at its own shape: `+alg=pqn_craftax` exactly as config/alg/pqn_craftax.yaml states it
    (1024 envs, 1 step x 1 minibatch x 1 epoch, BatchRenorm input + 4 x 1024 LayerNorm MLP, 1-step loss on
    concat(obs, next_obs), optimistic resets ratio 16, clip 1.0) on Craftax-Classic-Symbolic-v1, whole loop vs
    oracle.make_train(script="craftax") from shared initial parameters for 4 updates (pqn_craftax.py:96-114,277-304)."""
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.networks import QNetwork
    from purejaxql_amd.pqn import make_train, seed_keys
    n_upd = 4
    cfg = flatten(load_config(["+alg=pqn_craftax", "alg.ENV_NAME=Craftax-Classic-Symbolic-v1"]))
    assert (cfg["NUM_ENVS"], cfg["NUM_STEPS"], cfg["HIDDEN_SIZE"], cfg["NUM_LAYERS"]) == (1024, 1, 1024, 4)
    assert cfg["NORM_INPUT"] is True and cfg["NORM_TYPE"] == "layer_norm" and cfg["USE_OPTIMISTIC_RESETS"] is True
    cfg.update({"TOTAL_TIMESTEPS": n_upd * 1024, "TOTAL_TIMESTEPS_DECAY": 30 * 1024, "TEST_DURING_TRAINING": False})
    ocfg = dict(cfg)
    key = seed_keys(5, 1)[0]
    otrain = oracle.make_train(ocfg, script="craftax")
    net = QNetwork("mlp", (1345,), 17, norm_type="layer_norm", norm_input=True, hidden_size=1024, num_layers=4, device=gpu,
                   renorm=True)
    assert list(net.shapes) == list(otrain.shapes) and net.num_params == 4555411
    theta0 = net.init(21)
    cfg["_INIT_PARAMS"] = theta0
    train = make_train(cfg, device="cuda:0", script="craftax")
    out = train(key)
    oout = otrain(key, _np(theta0))
    assert cfg["NUM_UPDATES"] == n_upd == len(oout["metrics"])
    for u in range(n_upd):
        om = oout["metrics"][u]
        assert float(out["metrics"]["env_step"][u]) == om["env_step"] and float(out["metrics"]["grad_steps"][u]) == om["grad_steps"]
        for k in ("td_loss", "qvals"):
            assert abs(float(out["metrics"][k][u]) - om[k]) <= 1e-3 * max(1.0, abs(om[k])), (u, k, float(out["metrics"][k][u]), om[k])
        for k in ("returned_episode_returns", "returned_episode_lengths", "timestep", "returned_episode"):
            a, b = float(out["metrics"][k][u]), om[k]
            assert (math.isnan(a) and math.isnan(b)) or abs(a - b) <= 1e-3 * max(1.0, abs(b)), (u, k, a, b)
    th, oth, th0 = _np(out["runner_state"]["theta"]), oout["theta"], _np(theta0)
    d = np.abs(th - oth)
    bad = d > (2e-5 + 2e-3 * np.abs(oth))
    upd, oupd = th - th0, oth - th0
    cos = float(np.dot(upd, oupd) / (np.linalg.norm(upd) * np.linalg.norm(oupd)))
    # 4.5 M parameters, RAdam's first steps are sign-like (m_hat only): entries whose gradient is rounding noise move by
    # +-lr in either implementation; the criterion is therefore on the update vector plus a cap on the worst entry
    assert cos > 0.99 and bad.mean() < 2e-2 and d.max() < 2 * n_upd * cfg["LR"], (cos, float(bad.mean()), float(d.max()))
    bs = out["runner_state"]["batch_stats"]
    for k, v in oout["batch_stats"].items():
        if k.endswith("/steps"):
            assert int(bs[k]) == int(v) == n_upd
        else:
            assert np.abs(_np(bs[k]) - v).max() <= 1e-3 * max(np.abs(v).max(), 1e-3), k


def test_c5_trajectory_gradients_match_oracle_at_same_theta(gpu, oracle):
    """The same-theta counterpart of test_c5_yaml_shape_loop_vs_oracle_on_craftax_classic (whose whole-loop criterion is a
    cosine: RAdam's sign-like first steps amplify rounding-level gradient entries): the oracle walks 6 updates of the
    yaml's C5 loop on Craftax-Classic (1024 envs, 1 step, optimistic resets ratio 16, BatchRenorm input, 1-step loss on
    concat(obs, next_obs), clip 1.0 + RAdam) and at EVERY update the wide-MLP kernels (pqn_bigmlp_grad) are evaluated AT THE
    ORACLE'S parameters and running statistics on the oracle's own transition batch: loss to 1e-4, every gradient entry
    to rtol 2e-3 (given the kernel's relu decisions, which may differ from the oracle's only at |h| <= 1e-4), updated
    BatchRenorm statistics to 1e-5.  Reference lines: pqn_craftax.py:181-224,277-312; utils/batch_renorm.py:95-116."""
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.networks import QNetwork, bn_module
    from purejaxql_amd.qnet import BigMlpKernelLayout, BigMlpTrainer
    O = oracle
    cfg = flatten(load_config(["+alg=pqn_craftax", "alg.ENV_NAME=Craftax-Classic-Symbolic-v1"]))
    N, d, h, layers, a = int(cfg["NUM_ENVS"]), 1345, int(cfg["HIDDEN_SIZE"]), int(cfg["NUM_LAYERS"]), 17
    assert (N, cfg["NUM_STEPS"], cfg["NUM_MINIBATCHES"], cfg["NUM_EPOCHS"], h, layers) == (1024, 1, 1, 1, 1024, 4)
    ratio, gamma, lr, clip = int(cfg["OPTIMISTIC_RESET_RATIO"]), float(cfg["GAMMA"]), float(cfg["LR"]), float(cfg["MAX_GRAD_NORM"])
    n_upd = 6
    env = O.OracleEnv("Craftax-Classic-Symbolic-v1")
    net = QNetwork("mlp", (d,), a, norm_type="layer_norm", norm_input=True, hidden_size=h, num_layers=layers, device=gpu, renorm=True)
    lay = BigMlpKernelLayout(d, h, layers, a, 2)
    theta0 = net.init(21)
    tr = BigMlpTrainer(lay, theta0, lr, clip)
    shapes = O.mlp_shapes(d, a, h, layers, "layer_norm", True)
    bn0 = bn_module(True) + "_0"
    th = _np(theta0).copy()
    p = O.unflatten(th, shapes)
    m, v = np.zeros_like(th), np.zeros_like(th)
    stats = O.init_batch_stats("mlp", (d,), h, layers, "layer_norm", True, True)
    nkw = dict(layers=layers, norm_type="layer_norm", norm_input=True, renorm=True)
    obs, st = env.reset(77, N)
    obs = obs.reshape(N, -1)
    worst = 0.0
    # Batch moments of the BatchRenorm input layer in f64 on the oracle's side: a Craftax observation is full of
    # near-constant columns, for which flax's f32 fast variance E[x^2] - E[x]^2 cancels -- the oracle's own f32 restatement
    # and the kernels (which accumulate these moments in f64) then differ by up to 2.5e-3 in h_0 with
    # neither being wrong.  With exact moments the comparison isolates everything else at the usual tolerances.
    O.MOMENTS_DTYPE = np.float64
    try:
        worst = _c5_trajectory(O, env, tr, lay, shapes, p, th, m, v, stats, nkw, obs, st, cfg, N, layers, ratio, gamma, lr, clip, bn0,
                               n_upd, gpu)
    finally:
        O.MOMENTS_DTYPE = np.float32
    assert worst < 2e-3, worst


def _c5_trajectory(O, env, tr, lay, shapes, p, th, m, v, stats, nkw, obs, st, cfg, N, layers, ratio, gamma, lr, clip, bn0, n_upd, gpu):
    worst = 0.0
    for u in range(n_upd):
        eps = O.linear_schedule(cfg["EPS_START"], cfg["EPS_FINISH"], 10.0, u)
        q = O.net_forward("mlp", p, obs, stats=stats, **nkw)
        act, _qm = O.eps_greedy(q, np.float32(eps), 1000 + u)
        o_next, st, rew, done, _info = env.step_optimistic(1000 + u, st, act, ratio)
        o_next = o_next.reshape(N, -1)
        obs_all = np.concatenate((obs, o_next)).astype(np.float32)            # the [T+1][N] record with T = 1
        idx = O.permutation(O.fold_in(55, u), N).astype(np.int64)
        # --- the kernels at the oracle's theta / statistics
        tr.theta.copy_(lay.to_kernel(torch.from_numpy(th).to(gpu)))
        tr.refresh_planes()
        tr.in_mean.copy_(torch.from_numpy(np.asarray(stats[bn0 + "/mean"], np.float32)))
        tr.in_var.copy_(torch.from_numpy(np.asarray(stats[bn0 + "/var"], np.float32)))
        tr.in_steps[0] = int(stats[bn0 + "/steps"])
        lo_t, qv_t = torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)
        g = tr.compute_grad(torch.from_numpy(idx).to(gpu), torch.from_numpy(obs_all).to(gpu), torch.from_numpy(act).to(gpu),
                            reward=torch.from_numpy(rew).to(gpu), done=torch.from_numpy(done.astype(np.uint8)).to(gpu), gamma=gamma,
                            next_offset=N, loss_out=lo_t, qv_out=qv_t).clone()
        # --- the oracle's value_and_grad, spelled out so that its backward takes the kernel's relu decisions
        new_stats = {}
        xx = np.concatenate((obs_all[idx], obs_all[idx + N])).astype(np.float32)
        q_all, cache = O.net_forward("mlp", p, xx, want_cache=True, train=True, stats=stats, new_stats=new_stats, **nkw)
        flips = 0
        for l in range(layers):
            hk, ho = _np(tr.intermediate(2 * N, N, "h", l)), cache["hs"][l + 1]
            np.testing.assert_allclose(hk, ho, rtol=1e-4, atol=1e-4, err_msg=f"update {u} h_{l}")
            mism = (hk > 0) != (ho > 0)
            flips += int(mism.sum())
            assert not mism.any() or float(np.maximum(hk, ho)[mism].max()) <= 1e-4, (u, l)
            cache["hs"][l + 1] = hk.copy()
        assert flips <= 64, (u, flips)
        qo, q_next = q_all[:N], q_all[N:]
        tgt = (rew[idx] + (np.float32(1) - done[idx].astype(np.float32)) * np.float32(gamma) * q_next.max(-1)).astype(np.float32)
        chosen = qo[np.arange(N), act[idx]]
        diff = (chosen - tgt).astype(np.float32)
        lo = np.float32(0.5) * np.mean(diff * diff, dtype=np.float32)
        dq = np.zeros_like(q_all)
        dq[np.arange(N), act[idx]] = diff / np.float32(N)
        g_ref = O._net_backward("mlp", p, shapes, xx, cache, dq, layers, True)
        assert abs(float(lo_t) - lo) <= 1e-4 * max(1.0, abs(lo)), (u, float(lo_t), lo)
        g_flax = _np(lay.to_flax(g))
        np.testing.assert_allclose(g_flax, g_ref, rtol=2e-3, atol=1e-5 * np.abs(g_ref).max() + 1e-9, err_msg=f"update {u}")
        worst = max(worst, float(np.abs(g_flax - g_ref).max() / np.abs(g_ref).max()))
        np.testing.assert_allclose(_np(tr.in_mean), new_stats[bn0 + "/mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(_np(tr.in_var), new_stats[bn0 + "/var"], rtol=1e-5, atol=1e-6)
        assert int(tr.in_steps[0]) == int(new_stats[bn0 + "/steps"]) == u + 1
        # --- the oracle moves on (its own gradient: the trajectory is the oracle's)
        _lo2, _ch2, g_or = O.net_loss_grad_1step("mlp", p, shapes, obs_all[idx], obs_all[idx + N], act[idx], rew[idx], done[idx],
                                                 gamma, stats=stats, new_stats={}, **nkw)
        O.radam_clip_step(th, g_or, m, v, u, np.float32(lr), np.float32(clip))
        stats.update(new_stats)
        obs = o_next
    return worst


@pytest.mark.parametrize("env_name,n,t,upd", [("Breakout-MinAtar", 64, 4, 40), ("Craftax-Classic-Symbolic-v1", 128, 1, 30)])
def test_craftax_script_seeds_as_concurrent_streams_equal_solo_runs(gpu, env_name, n, t, upd):
    """NUM_SEEDS > 1 on the Craftax script runs the seeds as concurrent HIP streams (the torch-op network has no
    seed-batched kernels): every seed must be bit-identical to its solo run although all seeds share one env / wrapper
    object -- the optimistic-reset sort keys and Craftax's reset-slot scratch are per stream (round-2 advisor finding:
    they were shared, so one seed's reset kernels could overwrite another's between its step and world kernels)."""
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train, seed_keys, vmap_train
    cfg = flatten(load_config(["+alg=pqn_craftax"]))
    cfg.update({"ENV_NAME": env_name, "NUM_ENVS": n, "NUM_STEPS": t, "HIDDEN_SIZE": 64, "NUM_LAYERS": 2, "OPTIMISTIC_RESET_RATIO": 8,
                "TOTAL_TIMESTEPS": upd * n * t, "TOTAL_TIMESTEPS_DECAY": upd * n * t, "EPS_START": 1.0, "TEST_DURING_TRAINING": False})
    keys = seed_keys(11, 3)
    train = make_train(dict(cfg), device="cuda:0", script="craftax")
    assert not train.can_batch_seeds
    both = vmap_train(train, keys, concurrent="streams")
    torch.cuda.synchronize()
    for s, k in enumerate(keys):
        solo = make_train(dict(cfg), device="cuda:0", script="craftax")(k)
        for name in ("td_loss", "qvals", "returned_episode", "returned_episode_returns"):
            a, b = both["metrics"][name][s], solo["metrics"][name]
            assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)), (s, name)
        assert torch.equal(both["runner_state"][s]["theta"], solo["runner_state"]["theta"]), s
        assert torch.equal(both["runner_state"][s]["env_state"], solo["runner_state"]["env_state"]), s
    if env_name.startswith("Breakout"):
        assert float(torch.nan_to_num(both["metrics"]["returned_episode"]).sum()) > 0   # episodes ended: resets were handed out


def test_craftax_config_group_and_entry_point(gpu):
    """`+alg=pqn_craftax` carries the values of config/alg/pqn_craftax.yaml:1-35; the entry module runs a tiny job
    end to end (wrapper-batched env, optimistic resets, 1-step loss, per-seed checkpoint in the reference's format)."""
    import os
    import tempfile
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.run import main
    c = flatten(load_config(["+alg=pqn_craftax"]))
    assert (c["NUM_ENVS"], c["NUM_STEPS"], c["NUM_MINIBATCHES"], c["NUM_EPOCHS"]) == (1024, 1, 1, 1)
    assert (c["HIDDEN_SIZE"], c["NUM_LAYERS"], c["MAX_GRAD_NORM"], c["LR"]) == (1024, 4, 1.0, 1e-4)
    assert c["USE_OPTIMISTIC_RESETS"] is True and c["OPTIMISTIC_RESET_RATIO"] == 16 and c["TOTAL_TIMESTEPS"] == 1e9
    with tempfile.TemporaryDirectory() as tmp:
        outs = main(["alg.ENV_NAME=Breakout-MinAtar", "alg.NUM_ENVS=64", "alg.HIDDEN_SIZE=128", "alg.NUM_LAYERS=2",
                     "alg.TOTAL_TIMESTEPS=1280", "alg.TOTAL_TIMESTEPS_DECAY=1280", f"SAVE_PATH={tmp}", "NUM_SEEDS=2"],
                    "pqn_craftax", script="craftax")
        assert outs["metrics"]["td_loss"].shape == (2, 20) and torch.isfinite(outs["metrics"]["td_loss"]).all()
        files = sorted(os.listdir(os.path.join(tmp, "Breakout-MinAtar")))
        assert files == ["pqn_craftax_Breakout-MinAtar_seed0_config.yaml", "pqn_craftax_Breakout-MinAtar_seed0_vmap0.safetensors",
                         "pqn_craftax_Breakout-MinAtar_seed0_vmap1.safetensors"]
        from safetensors import safe_open
        with safe_open(os.path.join(tmp, "Breakout-MinAtar", files[1]), "pt") as f:
            keys = set(f.keys())
        assert {"BatchRenorm_0,scale", "Dense_0,kernel", "LayerNorm_1,bias", "Dense_2,kernel"} <= keys


DRIVER_CASES = [
    # (script, env, overrides, updates)
    ("craftax", "Craftax-Classic-Symbolic-v1", dict(NUM_ENVS=128, NUM_STEPS=1, NUM_MINIBATCHES=1, NUM_EPOCHS=1, Q_LAMBDA=False,
                                                    USE_OPTIMISTIC_RESETS=True, OPTIMISTIC_RESET_RATIO=16), 700),
    ("craftax", "Craftax-Classic-Symbolic-v1", dict(NUM_ENVS=64, NUM_STEPS=4, NUM_MINIBATCHES=2, NUM_EPOCHS=2, Q_LAMBDA=True,
                                                    USE_OPTIMISTIC_RESETS=False), 100),
    ("craftax", "Breakout-MinAtar", dict(NUM_ENVS=64, NUM_STEPS=4, NUM_MINIBATCHES=2, NUM_EPOCHS=1, Q_LAMBDA=False,
                                         USE_OPTIMISTIC_RESETS=True, OPTIMISTIC_RESET_RATIO=8, REW_SCALE=0.5), 25),
    ("gymnax", "Craftax-Classic-Symbolic-v1", dict(NUM_ENVS=64, NUM_STEPS=8, NUM_MINIBATCHES=4, NUM_EPOCHS=2), 6),
]


@pytest.mark.parametrize("script,env_name,over,n_upd", DRIVER_CASES)
def test_wide_mlp_whole_update_enqueue_equals_the_stepwise_loop(gpu, script, env_name, over, n_upd):
    """pqn_bigmlp_update (one C call per update, replayed as a hipGraph, keys / eps / metrics derived on the device)
    against the same loop enqueued piece by piece from Python (`_DRIVER: False` -- the path the oracle tests above and
    tests/test_bigmlp_gpu.py cover): identical parameters, optimizer state, batch statistics, env state and last
    observation, bit for bit; metric rows equal up to the summation order of their f64 means.  Both wrappers, both
    branches of the loss, a non-Craftax env under the Craftax script, and the gymnax script with a wide network."""
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train, seed_keys
    cfg = flatten(load_config(["+alg=pqn_craftax" if script == "craftax" else "+alg=pqn_cartpole"]))
    cfg.update(over)
    n, t = cfg["NUM_ENVS"], cfg["NUM_STEPS"]
    cfg.update({"ENV_NAME": env_name, "HIDDEN_SIZE": 256, "NUM_LAYERS": 2, "NORM_TYPE": "layer_norm", "TOTAL_TIMESTEPS": n_upd * n * t,
                "TOTAL_TIMESTEPS_DECAY": 30 * n * t, "TEST_DURING_TRAINING": False, "EPS_START": 0.5})
    key = seed_keys(11, 1)[0]
    outs = []
    for drv in (True, False):
        c = dict(cfg)
        c["_DRIVER"] = drv
        train = make_train(c, device="cuda:0", script=script)
        assert train.backend == "fused_big"
        out = train(key)
        rs = out["runner_state"]
        assert rs["driver"] == ("graph" if drv else None), (rs["driver"], rs["driver_graph_error"])
        outs.append(out)
    a, b = outs
    ra, rb = a["runner_state"], b["runner_state"]
    for k in ("theta", "opt_mu", "opt_nu", "env_state", "last_obs"):
        assert torch.equal(ra[k], rb[k]), k
    assert int(ra["opt_count"]) == int(rb["opt_count"]) == n_upd * cfg["NUM_MINIBATCHES"] * cfg["NUM_EPOCHS"]
    for k, v in rb["batch_stats"].items():
        assert torch.equal(ra["batch_stats"][k], v), k
    saw_done = False
    for k, vb in b["metrics"].items():
        va = a["metrics"][k]
        assert va.shape == vb.shape == (n_upd,), k
        na, nb = torch.isnan(va), torch.isnan(vb)
        assert torch.equal(na, nb), k
        assert torch.allclose(va[~na], vb[~nb], rtol=1e-6, atol=1e-7), (k, va, vb)
        if k == "returned_episode":
            saw_done = bool((va[~na] > 0).any())
    # Breakout episodes end within the run for certain (the optimistic-reset pass with the step key in device memory);
    # Craftax-Classic episodes of a half-random policy usually do within 400-700 steps, which is reported, not required
    print(f"{script} {env_name}: finished episodes inside the run: {saw_done}")
    assert saw_done or "Craftax" in env_name, "no finished episode inside the run: the reset paths were not exercised"


def test_wide_mlp_update_side_stream_option_is_bit_identical(gpu):
    """Option upd_overlap (bit 0: the first epoch's permutation on a side stream beside the rollout, bit 1: the
    input-gradient plane copy of the last optimizer step beside the closing bookkeeping; pqn_bigmlp_refresh_planes_streams)
    only moves launches between streams: parameters, optimizer state, statistics, env state and metrics of a run must not
    change by a bit, with several minibatches and epochs per update (the later permutations stay on the main stream)."""
    from purejaxql_amd import _lib
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.pqn import make_train, seed_keys
    cfg = flatten(load_config(["+alg=pqn_craftax"]))
    cfg.update(dict(NUM_ENVS=64, NUM_STEPS=4, NUM_MINIBATCHES=2, NUM_EPOCHS=2, Q_LAMBDA=True, USE_OPTIMISTIC_RESETS=True,
                    OPTIMISTIC_RESET_RATIO=8))
    n, t, n_upd = cfg["NUM_ENVS"], cfg["NUM_STEPS"], 40
    cfg.update({"ENV_NAME": "Craftax-Classic-Symbolic-v1", "HIDDEN_SIZE": 256, "NUM_LAYERS": 2, "NORM_TYPE": "layer_norm",
                "TOTAL_TIMESTEPS": n_upd * n * t, "TOTAL_TIMESTEPS_DECAY": 30 * n * t, "TEST_DURING_TRAINING": False, "EPS_START": 0.5})
    key = seed_keys(5, 1)[0]
    prev = _lib.get_option("upd_overlap")
    outs = []
    try:
        for ov in (0, 3, 1):
            _lib.set_option("upd_overlap", ov)
            out = make_train(dict(cfg), device="cuda:0", script="craftax")(key)
            assert out["runner_state"]["driver"] == "graph", out["runner_state"]["driver_graph_error"]
            outs.append(out)
    finally:
        _lib.set_option("upd_overlap", prev)
    ref = outs[0]["runner_state"]
    for o in outs[1:]:
        rs = o["runner_state"]
        for k in ("theta", "opt_mu", "opt_nu", "env_state", "last_obs"):
            assert torch.equal(rs[k], ref[k]), k
        for k, v in ref["batch_stats"].items():
            assert torch.equal(rs["batch_stats"][k], v), k
        for k, v in outs[0]["metrics"].items():
            assert torch.equal(torch.nan_to_num(o["metrics"][k]), torch.nan_to_num(v)), k


@pytest.mark.parametrize("driver", [True, False])
def test_log_achievements_adds_the_done_weighted_achievement_means(gpu, driver):
    """`LOG_ACHIEVEMENTS: True` (pqn_craftax.py:384-387 keeps the info["Achievements/<name>"] keys in the metrics): 22
    columns `Achievements/<name>` = (done * unlocked * 100 * returned_episode).sum() / returned_episode.sum(), NaN exactly
    where no episode finished, within [0, 100] otherwise; the per-step masks themselves are checked against the oracle in
    tests/test_craftax_env_gpu.py.  Round 4: the whole-update enqueue (pqn_bigmlp_update, one hipGraph per update) reduces
    the columns on the device (update_ach_means_kernel) instead of dropping to the stepwise loop; both paths must give the
    same columns bit for bit (same f64 sums of 0 / 100 values over the same masks)."""
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.envs import CRAFTAX_CLASSIC_ACHIEVEMENTS
    from purejaxql_amd.pqn import make_train, seed_keys
    cfg = flatten(load_config(["+alg=pqn_craftax", "alg.ENV_NAME=Craftax-Classic-Symbolic-v1"]))
    n_upd = 500
    cfg.update({"NUM_ENVS": 128, "HIDDEN_SIZE": 256, "NUM_LAYERS": 2, "TOTAL_TIMESTEPS": n_upd * 128, "TOTAL_TIMESTEPS_DECAY": n_upd * 128,
                "LOG_ACHIEVEMENTS": True, "EPS_START": 1.0, "EPS_FINISH": 1.0, "_DRIVER": driver})
    out = make_train(cfg, device="cuda:0", script="craftax")(seed_keys(2, 1)[0])
    assert (out["runner_state"]["driver"] == "graph") == driver and len(CRAFTAX_CLASSIC_ACHIEVEMENTS) == 22
    m = out["metrics"]
    nan_ref = torch.isnan(m["returned_episode_returns"])
    for name in CRAFTAX_CLASSIC_ACHIEVEMENTS:
        v = m[f"Achievements/{name}"]
        assert v.shape == (n_upd,) and torch.equal(torch.isnan(v), nan_ref), name
        ok = v[~nan_ref]
        assert bool(((ok >= 0) & (ok <= 100)).all()), name
    assert int((~nan_ref).sum()) > 0 and float(torch.nan_to_num(m["Achievements/collect_wood"], nan=0.0).max()) > 0
    key = ("ach", tuple(float(x) for x in torch.nan_to_num(m["Achievements/collect_wood"], nan=-1.0)[:200]))
    _ACH_RUNS.setdefault("ref", key)
    assert _ACH_RUNS["ref"] == key, "the graph driver and the stepwise loop disagree on Achievements/collect_wood"


_ACH_RUNS = {}


def test_done_weighted_means_are_global_ratios_under_a_metrics_hook(gpu):
    """Env-sharded mode: the Craftax script's done-weighted metrics -- (x * returned_episode).sum() /
    returned_episode.sum(), pqn_craftax.py:364-369, and the Achievements/* columns -- go to the cross-shard hook as
    NUMERATORS and the finished-episode COUNT, and are divided afterwards, so that the result is the ratio over all envs
    and a shard without a finished episode contributes 0 / 0 to nothing.  A hook that stands for "the other shard finished no
    episode" (it halves every entry: the mean of this shard's sums and a shard of zeros) must leave every done-weighted
    column exactly as the undisturbed run has it -- with the ratio taken per shard first it was (x + NaN) / 2 = NaN -- while
    the plain means (td_loss, qvals) are halved."""
    from purejaxql_amd.config_loader import flatten, load_config
    from purejaxql_amd.envs import CRAFTAX_CLASSIC_ACHIEVEMENTS
    from purejaxql_amd.pqn import make_train, seed_keys
    n_upd = 300

    def run(hook):
        cfg = flatten(load_config(["+alg=pqn_craftax", "alg.ENV_NAME=Craftax-Classic-Symbolic-v1"]))
        cfg.update({"NUM_ENVS": 128, "HIDDEN_SIZE": 256, "NUM_LAYERS": 2, "TOTAL_TIMESTEPS": n_upd * 128, "TOTAL_TIMESTEPS_DECAY": n_upd * 128,
                    "LOG_ACHIEVEMENTS": True, "EPS_START": 1.0, "EPS_FINISH": 1.0})
        return make_train(cfg, device="cuda:0", script="craftax", metrics_hook=hook)(seed_keys(2, 1)[0])["metrics"]

    plain, halved = run(None), run(lambda v: 0.5 * v)
    fin = ~torch.isnan(plain["returned_episode_returns"])
    assert int(fin.sum()) > 10
    for k in ["returned_episode_returns", "returned_episode_lengths", "timestep", "discount"] + [f"Achievements/{a}" for a in CRAFTAX_CLASSIC_ACHIEVEMENTS]:
        assert torch.equal(torch.isnan(halved[k]), ~fin), k
        torch.testing.assert_close(halved[k][fin], plain[k][fin], rtol=1e-6, atol=0)
    torch.testing.assert_close(halved["td_loss"], 0.5 * plain["td_loss"], rtol=1e-6, atol=0)
