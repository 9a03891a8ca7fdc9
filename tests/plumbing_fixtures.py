"""PLUMBING ONLY -- NOT PARITY, NOT REFERENCE DATA.

tests/golden/make_reference_fixtures.py cannot run in the build image (no jax / gymnax there), so the consumers of its
.npz files (tests/test_reference_fixtures_{cpu,gpu}.py, tests/reference_fixture_maps.py) would never execute before the
day the fixtures arrive.  This module writes files with THE SAME NAMES, KEY LAYOUT, SHAPES AND DTYPES the generator
writes -- but with the expected outputs computed by the ORACLE itself -- into a caller-supplied temporary directory, and
tests/test_fixture_plumbing_{cpu,gpu}.py run the consumers against them.  That proves the key names, the canonical-state
maps and the shapes line up end to end; it proves nothing about the oracle (oracle vs oracle) and is never reported as
parity.  Nothing here is written under tests/golden/, and nothing written here may be committed.

The inverse maps below (canonical state words -> gymnax 0.0.6 EnvState field names) mirror reference_fixture_maps.py; a
field-name correction there needs the same correction here."""
import os

import numpy as np

BREAKOUT_FIELDS = ["ball_y", "ball_x", "ball_dir", "pos", "strike", "last_y", "last_x", "time", "terminal"]
SPACEINVADERS_FIELDS = ["pos", "alien_dir", "enemy_move_interval", "alien_move_timer", "alien_shot_timer", "shot_timer",
                        "ramp_index", "time", "terminal"]
CARTPOLE_FIELDS = ["x", "x_dot", "theta", "theta_dot"]
ACROBOT_FIELDS = ["joint_angle1", "joint_angle2", "velocity_1", "velocity_2"]


def _fields_of(name, si, sf):
    """canonical (si, sf) of n envs -> {gymnax field: array} as make_reference_fixtures._flat(EnvState) lays them out"""
    out = {}
    if name == "Breakout-MinAtar":
        for j, k in enumerate(BREAKOUT_FIELDS):
            out[k] = si[:, j].astype(np.bool_ if k in ("strike", "terminal") else np.int32)
        out["brick_map"] = si[:, 9:109].reshape(-1, 10, 10).astype(np.bool_)
    elif name == "SpaceInvaders-MinAtar":
        for j, k in enumerate(SPACEINVADERS_FIELDS):
            out[k] = si[:, j].astype(np.bool_ if k == "terminal" else np.int32)
        for m, k in enumerate(("alien_map", "f_bullet_map", "e_bullet_map")):
            out[k] = si[:, 9 + 100 * m:109 + 100 * m].reshape(-1, 10, 10).astype(np.bool_)
    elif name in ("CartPole-v1", "Acrobot-v1"):
        for j, k in enumerate(CARTPOLE_FIELDS if name == "CartPole-v1" else ACROBOT_FIELDS):
            out[k] = sf[:, j].astype(np.float32)
        out["time"] = si[:, 0].astype(np.int32)
    else:
        raise KeyError(name)
    return out


def _copy_state(st):
    return {k: v.copy() for k, v in st.items()}


def env_trace(oracle, out_dir, name, n_envs=8, n_steps=120):
    """ref_env_<name>.npz: the generator's env loop (step_env alone for before / after, env.step to continue)."""
    oenv = oracle.OracleEnv(name)
    rng = np.random.default_rng(len(name))
    actions = rng.integers(0, oenv.num_actions, size=(n_steps, n_envs)).astype(np.int32)
    obs, st = oenv.reset(0, n_envs)
    rec = {"actions": actions, "max_steps_in_episode": np.int32(oenv.max_steps), "obs0": obs.copy()}
    before, after, st_obs, ob, rw, dn = [], [], [], [], [], []
    for t in range(n_steps):
        before.append(_fields_of(name, st["si"], st["sf"]))
        s2 = _copy_state(st)
        o_se, s2, _r, _d, _ = oenv.step(0, s2, actions[t], autoreset=False)
        after.append(_fields_of(name, s2["si"], s2["sf"]))
        st_obs.append(o_se.copy())
        o, st, r, d, _ = oenv.step(oracle.fold_in(1000, t), st, actions[t], autoreset=True)
        ob.append(o.copy()); rw.append(r.copy()); dn.append(d.copy())
    for k in before[0]:
        rec[f"before/{k}"] = np.stack([b[k] for b in before])
        rec[f"after/{k}"] = np.stack([a[k] for a in after])
    rec.update(step_env_obs=np.stack(st_obs), obs=np.stack(ob), reward=np.stack(rw), done=np.stack(dn))
    np.savez_compressed(os.path.join(out_dir, f"ref_env_{name}.npz"), **rec)


def qlambda(oracle, out_dir):
    rec = {}
    r = np.asarray([1, 0, 2, 1], np.float32)[:, None]
    q = np.asarray([5, 6, 7, 8], np.float32)[:, None]
    for i, d in enumerate(([0, 0, 0, 0], [0, 1, 0, 1])):
        rec[f"ka1_{i}"] = oracle.q_lambda(r, np.asarray(d, np.uint8)[:, None], q, np.asarray([9.0], np.float32), 0.99, 0.65)
    rng = np.random.default_rng(2)
    rr = ((rng.random((32, 64)) < 0.05) * rng.random((32, 64))).astype(np.float32)
    dd = rng.random((32, 64)) < 0.02
    qq = rng.standard_normal((32, 64)).astype(np.float32)
    ll = rng.standard_normal(64).astype(np.float32)
    rec.update(reward=rr, done=dd, qmax=qq, last_q=ll, target=oracle.q_lambda(rr, dd, qq, ll, 0.99, 0.65, quirk=True))
    np.savez_compressed(os.path.join(out_dir, "ref_qlambda.npz"), **rec)


def radam(oracle, out_dir):
    rng = np.random.default_rng(1)
    n, steps = 1000, 5
    p0 = rng.standard_normal(n).astype(np.float32)
    grads = np.stack([(rng.standard_normal(n) * (30.0 if i % 2 else 1e-3)).astype(np.float32) for i in range(steps)])
    p, m, v = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    traj, lrs = [], []
    for i in range(steps):
        lrs.append(oracle.linear_schedule(5e-4, 1e-20, 7, i))
        oracle.radam_clip_step(p, grads[i], m, v, i, np.float32(lrs[-1]), 10.0)
        traj.append(p.copy())
    np.savez_compressed(os.path.join(out_dir, "ref_radam.npz"), p0=p0, grads=grads, params=np.stack(traj),
                        lr=np.asarray(lrs, np.float64),
                        eps=np.asarray([oracle.linear_schedule(1.0, 0.05, 0.1 * 2441, c) for c in (0, 1, 100, 244, 245, 3000)], np.float64),
                        eps_degenerate=np.float64(oracle.linear_schedule(1.0, 0.05, 0, 7)))


def _random_theta(rng, shapes):
    parts = {}
    for k, s in shapes.items():
        fan_in = int(np.prod(s[:-1])) if len(s) > 1 else 1
        x = rng.standard_normal(s).astype(np.float32) / np.float32(np.sqrt(max(fan_in, 1)))
        parts[k] = (1.0 + 0.1 * x).astype(np.float32) if k.endswith("scale") else x
    return parts


def qnet(oracle, out_dir):
    rng = np.random.default_rng(0)
    b, a = 32, 3
    obs = (rng.random((b, 10, 10, 4)) < 0.12).astype(np.float32)
    action = rng.integers(0, a, b).astype(np.int32)
    target = rng.standard_normal(b).astype(np.float32)
    rec = {"obs": obs, "action": action, "target": target}
    for norm_type, norm_input in (("layer_norm", False), ("batch_norm", False), ("layer_norm", True)):
        tag = f"{norm_type}_{int(norm_input)}"
        shapes = oracle.cnn_shapes((10, 10, 4), a, norm_type)
        p = _random_theta(rng, shapes)
        stats = oracle.init_batch_stats("cnn", (10, 10, 4), 128, 2, norm_type, norm_input)
        kw = dict(norm_type=norm_type, norm_input=norm_input)
        q_eval = oracle.net_forward("cnn", p, obs, norm_type == "layer_norm", 2, train=False, stats=stats, **kw)
        new_stats = {}
        loss, chosen, g = oracle.net_loss_grad("cnn", p, shapes, obs, action, target, norm_type == "layer_norm", 2, stats=stats,
                                               new_stats=new_stats, **kw)
        gp = oracle.unflatten(np.asarray(g, np.float32), shapes)
        for k in shapes:
            rec[f"{tag}/params/{k}"] = p[k]
            rec[f"{tag}/grads/{k}"] = gp[k]
        for k, v in new_stats.items():
            rec[f"{tag}/new_batch_stats/{k}"] = np.asarray(v)
        rec[f"{tag}/loss"], rec[f"{tag}/chosen"], rec[f"{tag}/q_eval"] = np.float32(loss), np.asarray(chosen), np.asarray(q_eval)
    np.savez_compressed(os.path.join(out_dir, "ref_qnet.npz"), **rec)


def craftax_qnet(oracle, out_dir):
    rng = np.random.default_rng(3)
    nb, d, a, h, layers, gamma = 64, 40, 5, 64, 2, 0.99
    obs = (rng.standard_normal((nb, d)) * (0.3 + rng.random(d)) + 0.3 * rng.standard_normal(d)).astype(np.float32)
    nxt = (obs + 0.1 * rng.standard_normal((nb, d))).astype(np.float32)
    action = rng.integers(0, a, nb).astype(np.int32)
    reward = rng.standard_normal(nb).astype(np.float32)
    done = rng.random(nb) < 0.2
    rec = {"obs": obs, "next_obs": nxt, "action": action, "reward": reward, "done": done, "gamma": np.float32(gamma)}
    shapes = oracle.mlp_shapes(d, a, h, layers, "layer_norm", True)
    nkw = dict(layers=layers, norm_type="layer_norm", norm_input=True, renorm=True)
    p = _random_theta(rng, shapes)
    for tag, steps in (("cold", 0), ("warm", 2000)):
        stats = oracle.init_batch_stats("mlp", (d,), h, layers, "layer_norm", True, renorm=True)
        if steps:
            for k in list(stats):
                if k.endswith("steps"):
                    stats[k] = steps
                elif k.endswith("mean"):
                    stats[k] = (stats[k] + 0.1 * rng.standard_normal(stats[k].shape)).astype(np.float32)
                else:
                    stats[k] = (stats[k] * (0.6 + rng.random(stats[k].shape))).astype(np.float32)
        q_eval = oracle.net_forward("mlp", p, obs, train=False, stats=stats, **nkw)
        all_q = oracle.net_forward("mlp", p, np.concatenate((obs, nxt)), train=True, stats=dict(stats), new_stats={}, **nkw)
        new_stats = {}
        loss, chosen, g = oracle.net_loss_grad_1step("mlp", p, shapes, obs, nxt, action, reward, done, gamma, stats=stats,
                                                     new_stats=new_stats, **nkw)
        gp = oracle.unflatten(np.asarray(g, np.float32), shapes)
        for k in shapes:
            rec[f"{tag}/params/{k}"] = p[k]
            rec[f"{tag}/grads/{k}"] = gp[k]
        for k, v in stats.items():
            rec[f"{tag}/batch_stats/{k}"] = np.asarray(v)
        for k, v in new_stats.items():
            rec[f"{tag}/new_batch_stats/{k}"] = np.asarray(v)
        rec[f"{tag}/loss"], rec[f"{tag}/chosen"] = np.float32(loss), np.asarray(chosen)
        rec[f"{tag}/all_q"], rec[f"{tag}/q_eval"] = np.asarray(all_q), np.asarray(q_eval)
    np.savez_compressed(os.path.join(out_dir, "ref_craftax_qnet.npz"), **rec)


def optimistic(oracle, out_dir):
    n_envs, ratio, n_steps = 16, 4, 300
    oenv = oracle.OracleEnv("CartPole-v1")
    rng = np.random.default_rng(4)
    actions = rng.integers(0, 2, size=(n_steps, n_envs)).astype(np.int32)
    obs, st = oenv.reset(0, n_envs)
    rec = {"actions": actions, "obs0": obs.copy(), "num_envs": np.int32(n_envs), "reset_ratio": np.int32(ratio)}
    before, after, ob, rw, dn, rer, rel, ts = [], [], [], [], [], [], [], []
    for t in range(n_steps):
        before.append(_fields_of("CartPole-v1", st["si"], st["sf"]))
        o, st, r, d, info = oenv.step_optimistic(oracle.fold_in(5000, t), st, actions[t], ratio)
        after.append(_fields_of("CartPole-v1", st["si"], st["sf"]))
        ob.append(o.copy()); rw.append(r.copy()); dn.append(d.copy())
        rer.append(info["returned_episode_returns"].copy()); rel.append(info["returned_episode_lengths"].copy())
        ts.append(info["timestep"].copy())
    for k in before[0]:   # LogEnvState.env_state.<field>, as _flat() of the wrapped state names them
        rec[f"before/env_state/{k}"] = np.stack([b[k] for b in before])
        rec[f"after/env_state/{k}"] = np.stack([a[k] for a in after])
    rec.update(obs=np.stack(ob), reward=np.stack(rw), done=np.stack(dn), returned_episode_returns=np.stack(rer),
               returned_episode_lengths=np.stack(rel), timestep=np.stack(ts))
    np.savez_compressed(os.path.join(out_dir, "ref_optimistic.npz"), **rec)


def write_all(oracle, out_dir):
    out_dir = str(out_dir)
    assert os.path.basename(os.path.normpath(out_dir)) != "golden", "plumbing files never go under tests/golden/"
    for name in ("Breakout-MinAtar", "SpaceInvaders-MinAtar", "CartPole-v1", "Acrobot-v1"):
        env_trace(oracle, out_dir, name)
    qlambda(oracle, out_dir)
    radam(oracle, out_dir)
    qnet(oracle, out_dir)
    craftax_qnet(oracle, out_dir)
    optimistic(oracle, out_dir)
    with open(os.path.join(out_dir, "PLUMBING_ONLY.txt"), "w") as f:
        f.write("Written by tests/plumbing_fixtures.py from the oracle: key-layout stand-ins, not reference data.\n")
