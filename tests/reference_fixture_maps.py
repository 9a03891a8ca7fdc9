"""gymnax 0.0.6 EnvState field names -> this build's canonical state words (oracle/pqn_oracle.c, Env::to_canon).
Used by the reference-fixture tests; the field names are recollection of the un-vendored dependency --
if a fixture holds different names, the test fails with the list of fields it found, which is the edit to make here."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    path = os.path.join(GOLDEN, name)
    if not os.path.exists(path):
        import pytest
        pytest.skip(f"{name} absent: run tests/golden/make_reference_fixtures.py in the reference's jax environment")
    return np.load(path)


def _need(rec, prefix, fields):
    have = sorted(k[len(prefix) + 1:] for k in rec.files if k.startswith(prefix + "/"))
    missing = [f for f in fields if f"{prefix}/{f}" not in rec.files]
    assert not missing, f"fixture state fields {have} do not contain {missing}: update tests/reference_fixture_maps.py"


def breakout_canon(rec, prefix, t=None):
    """[ball_y, ball_x, ball_dir, pos, strike, last_y, last_x, time, terminal, brick_map[100]] (Breakout::to_canon)"""
    f = ["ball_y", "ball_x", "ball_dir", "pos", "strike", "last_y", "last_x", "time", "terminal", "brick_map"]
    _need(rec, prefix, f)
    g = lambda k: rec[f"{prefix}/{k}"] if t is None else rec[f"{prefix}/{k}"][t]
    cols = [np.asarray(g(k)).astype(np.int32).reshape(-1, 1) for k in f[:-1]]
    bm = np.asarray(g("brick_map")).astype(np.int32)
    return np.concatenate(cols + [bm.reshape(bm.shape[0], -1)], axis=1)


def cartpole_canon(rec, prefix, t=None):
    """si = [time], sf = [x, x_dot, theta, theta_dot] (CartPole::to_canon)"""
    f = ["x", "x_dot", "theta", "theta_dot", "time"]
    _need(rec, prefix, f)
    g = lambda k: rec[f"{prefix}/{k}"] if t is None else rec[f"{prefix}/{k}"][t]
    si = np.asarray(g("time")).astype(np.int32).reshape(-1, 1)
    sf = np.stack([np.asarray(g(k)).astype(np.float32) for k in f[:4]], axis=1)
    return si, sf


def spaceinvaders_canon(rec, prefix, t=None):
    """[pos, alien_dir, enemy_move_interval, alien_move_timer, alien_shot_timer, shot_timer, ramp_index, time, terminal,
    alien_map[100], f_bullet_map[100], e_bullet_map[100]]"""
    f = ["pos", "alien_dir", "enemy_move_interval", "alien_move_timer", "alien_shot_timer", "shot_timer", "ramp_index", "time",
         "terminal", "alien_map", "f_bullet_map", "e_bullet_map"]
    _need(rec, prefix, f)
    g = lambda k: rec[f"{prefix}/{k}"] if t is None else rec[f"{prefix}/{k}"][t]
    cols = [np.asarray(g(k)).astype(np.int32).reshape(-1, 1) for k in f[:9]]
    maps = [np.asarray(g(k)).astype(np.int32) for k in f[9:]]
    return np.concatenate(cols + [m.reshape(m.shape[0], -1) for m in maps], axis=1)


def flax_params_to_theta(rec, tag, shapes):
    """flat parameter vector in this build's flax order from the fixture's `<tag>/params/<name>` leaves"""
    parts = []
    for k, s in shapes.items():
        key = f"{tag}/params/{k}"
        assert key in rec.files, f"{key} not in fixture: {[x for x in rec.files if x.startswith(tag + '/params/')]}"
        assert tuple(rec[key].shape) == tuple(s), (k, rec[key].shape, s)
        parts.append(np.asarray(rec[key], np.float32).reshape(-1))
    return np.concatenate(parts)


def flax_grads(rec, tag, shapes):
    return np.concatenate([np.asarray(rec[f"{tag}/grads/{k}"], np.float32).reshape(-1) for k in shapes])
