"""CPU ORACLE, Python side -- TEST INFRASTRUCTURE ONLY.

ctypes loader for oracle/libpqn_oracle.so (plain-C restatement, see
pqn_oracle.h) plus a numpy restatement of the Q-networks and of the whole
make_train loop of the reference:

    purejaxql/pqn_minatar.py:24-69,89-431   (CNN + loop)
    purejaxql/pqn_gymnax.py:29-58,78-424    (MLP + loop)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  purejaxql_amd never does.

PARITY STATUS: the reference has no tests/golden vectors and cannot be imported
in the build container (no jax/flax/optax/gymnax).  Algorithm pieces are pinned
by the hand-derived known-answer vectors of SURVEY.md 8(c) (tests/golden/);
env dynamics and flax/optax numerics are restated from third-party recollection
-> **parity unpinned** for those.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from collections import OrderedDict
from typing import Any, Dict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpqn_oracle.so")
_lib = None

ENV_IDS = {"Breakout-MinAtar": 0, "CartPole-v1": 1, "Asterix-MinAtar": 2, "Freeway-MinAtar": 3,
           "SpaceInvaders-MinAtar": 4, "Craftax-Classic-Symbolic-v1": 5, "Acrobot-v1": 6}


class Spec(C.Structure):
    _fields_ = [("obs_dim", C.c_int32 * 3), ("obs_size", C.c_int32), ("num_actions", C.c_int32),
                ("max_steps", C.c_int32), ("si", C.c_int32), ("sf", C.c_int32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.pqn_oracle_fold_in.restype = C.c_uint64
        L.pqn_oracle_fold_in.argtypes = [C.c_uint64, C.c_uint32]
        L.pqn_oracle_bits_to_uniform.restype = C.c_float
        L.pqn_oracle_bits_to_uniform.argtypes = [C.c_uint32]
        L.pqn_oracle_linear_schedule.restype = C.c_double
        L.pqn_oracle_linear_schedule.argtypes = [C.c_double] * 4
        L.pqn_oracle_radam_clip_step.restype = C.c_float
        L.pqn_oracle_radam_clip_step.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int64, C.c_float, C.c_float]
        L.pqn_oracle_env_reset.argtypes = [C.c_int, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pqn_oracle_env_step.argtypes = [C.c_int, C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pqn_oracle_env_step_optimistic.argtypes = [C.c_int, C.c_int32, C.c_uint64, C.c_int32] + [C.c_void_p] * 16
        L.pqn_oracle_env_obs.argtypes = [C.c_int, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pqn_oracle_log_step.argtypes = [C.c_int32] + [C.c_void_p] * 7
        L.pqn_oracle_eps_greedy.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_void_p,
                                            C.c_void_p]
        L.pqn_oracle_q_lambda.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32,
                                                             C.c_void_p]
        L.pqn_oracle_sort_keys.argtypes = [C.c_uint64, C.c_int32, C.c_void_p]
        L.pqn_oracle_env_bits.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.pqn_oracle_threefry2x32.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.pqn_oracle_env_spec.argtypes = [C.c_int, C.POINTER(Spec)]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- PRNG ---------------------------------------------------------------------------
def threefry2x32(key, ctr):
    k = (C.c_uint32 * 2)(*key)
    c = (C.c_uint32 * 2)(*ctr)
    o = (C.c_uint32 * 2)()
    lib().pqn_oracle_threefry2x32(k, c, o)
    return int(o[0]), int(o[1])


def fold_in(key: int, data: int) -> int:
    return int(lib().pqn_oracle_fold_in(C.c_uint64(key & 0xFFFFFFFFFFFFFFFF), C.c_uint32(data & 0xFFFFFFFF)))


def env_bits(key, index, stream):
    o = (C.c_uint32 * 2)()
    lib().pqn_oracle_env_bits(C.c_uint64(key), index, stream, o)
    return int(o[0]), int(o[1])


def permutation(key: int, n: int) -> np.ndarray:
    keys = np.empty(n, dtype=np.int64)
    lib().pqn_oracle_sort_keys(C.c_uint64(key), n, _p(keys))
    return np.argsort(keys, kind="stable")


# ---- envs ---------------------------------------------------------------------------
class OracleEnv:
    """Batched gymnax-style env + LogWrapper on the C oracle (canonical state)."""

    def __init__(self, name: str):
        self.name = name
        self.env_id = ENV_IDS[name]
        sp = Spec()
        assert lib().pqn_oracle_env_spec(self.env_id, C.byref(sp)) == 0, f"oracle has no env {name}"
        self.spec = sp
        d = tuple(int(x) for x in sp.obs_dim)
        self.obs_shape = d if d[1] > 0 else (d[0],)
        self.obs_size = int(sp.obs_size)
        self.num_actions = int(sp.num_actions)
        self.max_steps = int(sp.max_steps)

    def reset(self, key: int, n: int):
        st = {"si": np.zeros((n, int(self.spec.si)), np.int32),
              "sf": np.zeros((n, max(int(self.spec.sf), 1)), np.float32),
              "ep_ret": np.zeros(n, np.float32), "ep_len": np.zeros(n, np.int32),
              "ret_ret": np.zeros(n, np.float32), "ret_len": np.zeros(n, np.int32),
              "timestep": np.zeros(n, np.int32)}
        obs = np.zeros((n, *self.obs_shape), np.float32)
        lib().pqn_oracle_env_reset(self.env_id, n, C.c_uint64(key), _p(st["si"]), _p(st["sf"]), _p(obs))
        return obs, st

    def step(self, key: int, st, action, autoreset: bool = True):
        n = st["si"].shape[0]
        action = np.ascontiguousarray(action, dtype=np.int32)
        obs = np.zeros((n, *self.obs_shape), np.float32)
        reward = np.zeros(n, np.float32)
        done = np.zeros(n, np.uint8)
        disc = np.zeros(n, np.float32)
        lib().pqn_oracle_env_step(self.env_id, n, C.c_uint64(key), _p(st["si"]), _p(st["sf"]), _p(action),
                                  1 if autoreset else 0, _p(obs), _p(reward), _p(done), _p(disc))
        lib().pqn_oracle_log_step(n, _p(reward), _p(done), _p(st["ep_ret"]), _p(st["ep_len"]), _p(st["ret_ret"]),
                                  _p(st["ret_len"]), _p(st["timestep"]))
        info = {"discount": disc, "returned_episode_returns": st["ret_ret"].copy(),
                "returned_episode_lengths": st["ret_len"].copy(), "timestep": st["timestep"].copy(),
                "returned_episode": done.astype(bool)}
        return obs, st, reward, done.astype(bool), info

    def step_optimistic(self, key: int, st, action, reset_ratio: int):
        """OptimisticResetVecEnvWrapper(LogWrapper(env), n, reset_ratio).step (utils/craftax_wrappers.py:83-148)."""
        n = st["si"].shape[0]
        action = np.ascontiguousarray(action, dtype=np.int32)
        obs = np.zeros((n, *self.obs_shape), np.float32)
        reward, done, disc = np.zeros(n, np.float32), np.zeros(n, np.uint8), np.zeros(n, np.float32)
        irr, irl, its, slot = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        rc = lib().pqn_oracle_env_step_optimistic(self.env_id, n, C.c_uint64(key), int(reset_ratio), _p(st["si"]), _p(st["sf"]),
                                                  _p(action), _p(obs), _p(reward), _p(done), _p(disc), _p(st["ep_ret"]),
                                                  _p(st["ep_len"]), _p(st["ret_ret"]), _p(st["ret_len"]), _p(st["timestep"]),
                                                  _p(irr), _p(irl), _p(its), _p(slot))
        assert rc == 0, "reset ratio must perfectly divide num envs"
        info = {"discount": disc, "returned_episode_returns": irr, "returned_episode_lengths": irl, "timestep": its,
                "returned_episode": done.astype(bool), "reset_slot": slot}
        return obs, st, reward, done.astype(bool), info

    def log_words(self, st) -> np.ndarray:
        """[n,5] u32 bit patterns of the LogWrapper record (product's export layout)."""
        out = np.zeros((st["si"].shape[0], 5), np.uint32)
        out[:, 0] = st["ep_ret"].view(np.uint32)
        out[:, 1] = st["ep_len"].view(np.uint32)
        out[:, 2] = st["ret_ret"].view(np.uint32)
        out[:, 3] = st["ret_len"].view(np.uint32)
        out[:, 4] = st["timestep"].view(np.uint32)
        return out


def eps_greedy(q, eps, key):
    q = np.ascontiguousarray(q, np.float32)
    m, a = q.shape
    action = np.zeros(m, np.int32)
    qmax = np.zeros(m, np.float32)
    lib().pqn_oracle_eps_greedy(_p(q), m, a, C.c_float(eps), C.c_uint64(key), _p(action), _p(qmax))
    return action, qmax


def q_lambda(reward, done, qmax, last_q, gamma, lam, quirk=True):
    reward = np.ascontiguousarray(reward, np.float32)
    done = np.ascontiguousarray(done).astype(np.uint8)
    qmax = np.ascontiguousarray(qmax, np.float32)
    last_q = np.ascontiguousarray(last_q, np.float32)
    t, m = reward.shape
    target = np.zeros((t, m), np.float32)
    lib().pqn_oracle_q_lambda(_p(reward), _p(done), _p(qmax), _p(last_q), C.c_float(gamma), C.c_float(lam), t, m,
                              1 if quirk else 0, _p(target))
    return target


def linear_schedule(init, end, steps, count):
    return float(lib().pqn_oracle_linear_schedule(float(init), float(end), float(steps), float(count)))


def radam_clip_step(p, g, m, v, count, lr, max_norm):
    """In-place on p, m, v (float32 contiguous).  Returns the pre-clip global norm."""
    g = np.ascontiguousarray(g, np.float32).copy()
    return float(lib().pqn_oracle_radam_clip_step(_p(p), _p(g), _p(m), _p(v), p.size, int(count), C.c_float(lr),
                                                  C.c_float(max_norm)))


# ---- networks (numpy, float32; flax semantics of SURVEY Appendix A) ----------------------
LN_EPS = np.float32(1e-6)


BN_EPS = np.float32(1e-5)      # flax nn.BatchNorm defaults (A.4): epsilon 1e-5, momentum 0.99
BN_MOMENTUM = np.float32(0.99)


def _norm_kind(norm_type):
    return norm_type if norm_type in ("layer_norm", "batch_norm") else "none"    # pqn_minatar.py:31-36


def cnn_norm_names(norm_type):
    k = _norm_kind(norm_type)
    if k == "layer_norm":
        return "CNN_0/LayerNorm_0", "CNN_0/LayerNorm_1"
    if k == "batch_norm":
        return "CNN_0/BatchNorm_0", "CNN_0/BatchNorm_1"
    return "", ""


def bn_module(renorm):
    # nn.BatchNorm in pqn_minatar.py / pqn_gymnax.py, BatchRenorm in pqn_craftax.py:33-62 (flax auto-name stems)
    return "BatchRenorm" if renorm else "BatchNorm"


def mlp_norm_name(norm_type, l, renorm=False):
    # BatchNorm_0 is the input / dummy BatchNorm (pqn_gymnax.py:39-43): hidden layers get BatchNorm_1..
    k = _norm_kind(norm_type)
    return f"LayerNorm_{l}" if k == "layer_norm" else (f"{bn_module(renorm)}_{l + 1}" if k == "batch_norm" else "")


def cnn_shapes(obs_shape, a, norm_type="layer_norm"):
    h, w, c = obs_shape
    n0, n1 = cnn_norm_names(norm_type)
    s = OrderedDict([("BatchNorm_0/scale", (c,)), ("BatchNorm_0/bias", (c,)),
                     ("CNN_0/Conv_0/kernel", (3, 3, c, 16)), ("CNN_0/Conv_0/bias", (16,))])
    if n0:
        s[n0 + "/scale"], s[n0 + "/bias"] = (16,), (16,)
    s["CNN_0/Dense_0/kernel"], s["CNN_0/Dense_0/bias"] = ((h - 2) * (w - 2) * 16, 128), (128,)
    if n1:
        s[n1 + "/scale"], s[n1 + "/bias"] = (128,), (128,)
    s["Dense_0/kernel"], s["Dense_0/bias"] = (128, a), (a,)
    return s


def mlp_shapes(d, a, hidden, layers, norm_type="layer_norm", renorm=False):
    s = OrderedDict([(bn_module(renorm) + "_0/scale", (d,)), (bn_module(renorm) + "_0/bias", (d,))])
    for l in range(layers):
        s[f"Dense_{l}/kernel"] = (d, hidden)
        s[f"Dense_{l}/bias"] = (hidden,)
        n = mlp_norm_name(norm_type, l, renorm)
        if n:
            s[n + "/scale"] = (hidden,)
            s[n + "/bias"] = (hidden,)
        d = hidden
    s[f"Dense_{layers}/kernel"] = (d, a)
    s[f"Dense_{layers}/bias"] = (a,)
    return s


def init_batch_stats(kind, obs_shape, hidden, layers, norm_type, norm_input, renorm=False):
    """variables["batch_stats"]: running mean 0 / var 1 of every BatchNorm whose output is used (the
    dummy input BatchNorm of NORM_INPUT=False, pqn_minatar.py:63-65, is dead state and not tracked)."""
    feats = OrderedDict()
    if norm_input:
        feats[bn_module(renorm) + "_0"] = int(obs_shape[-1])
    if _norm_kind(norm_type) == "batch_norm":
        if kind == "cnn":
            n0, n1 = cnn_norm_names(norm_type)
            feats[n0], feats[n1] = 16, 128
        else:
            for l in range(layers):
                feats[mlp_norm_name(norm_type, l, renorm)] = hidden
    st = OrderedDict()
    for name, f in feats.items():
        st[name + "/mean"] = np.zeros(f, np.float32)
        st[name + "/var"] = np.ones(f, np.float32)
        if renorm:
            st[name + "/steps"] = 0      # utils/batch_renorm.py:71-76
    return st


def unflatten(theta, shapes):
    out, off = {}, 0
    for k, s in shapes.items():
        n = int(np.prod(s))
        out[k] = theta[off:off + n].reshape(s)
        off += n
    assert off == theta.size
    return out


def _ln_fwd(x, scale, bias):
    # flax LayerNorm: mean, var = E[x^2]-E[x]^2 (clamped >= 0), eps inside rsqrt (A.3)
    mu = x.mean(-1, keepdims=True, dtype=np.float32)
    var = np.maximum((x * x).mean(-1, keepdims=True, dtype=np.float32) - mu * mu, np.float32(0))
    rstd = (np.float32(1) / np.sqrt(var + LN_EPS)).astype(np.float32)
    xhat = (x - mu) * rstd
    return xhat * scale + bias, (xhat, rstd)


def _ln_bwd(dy, scale, cache):
    xhat, rstd = cache
    dscale = (dy * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    dbias = dy.reshape(-1, xhat.shape[-1]).sum(0)
    dxh = dy * scale
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xhat * (dxh * xhat).mean(-1, keepdims=True))
    return dx.astype(np.float32), dscale.astype(np.float32), dbias.astype(np.float32)


def _patches(x):
    # NHWC 3x3 VALID windows -> [B, H-2, W-2, 3*3*C] in (ky,kx,c) order
    b, h, w, c = x.shape
    s = x.strides
    v = np.lib.stride_tricks.as_strided(x, (b, h - 2, w - 2, 3, 3, c), (s[0], s[1], s[2], s[1], s[2], s[3]))
    return v.reshape(b, h - 2, w - 2, 9 * c)


def _bn_fwd(x, scale, bias, name, train, stats, new_stats):
    """flax nn.BatchNorm(use_running_average=not train) (A.4): moments over every axis but the last, fast
    variance E[x^2]-E[x]^2 clamped at 0; running <- 0.99 running + 0.01 batch."""
    f = x.shape[-1]
    x2 = x.reshape(-1, f)
    if train:
        mean = x2.mean(0, dtype=np.float32)
        var = np.maximum((x2 * x2).mean(0, dtype=np.float32) - mean * mean, np.float32(0))
        if new_stats is not None:
            new_stats[name + "/mean"] = (BN_MOMENTUM * stats[name + "/mean"] + (np.float32(1) - BN_MOMENTUM) * mean).astype(np.float32)
            new_stats[name + "/var"] = (BN_MOMENTUM * stats[name + "/var"] + (np.float32(1) - BN_MOMENTUM) * var).astype(np.float32)
    else:
        mean, var = stats[name + "/mean"], stats[name + "/var"]
    rstd = (np.float32(1) / np.sqrt(var + BN_EPS)).astype(np.float32)
    xhat = ((x - mean) * rstd).astype(np.float32)
    return (xhat * scale + bias).astype(np.float32), (xhat, rstd, train)


def _bn_bwd(dy, scale, cache):
    """Backward of the train-mode BatchNorm (batch moments are functions of x)."""
    xhat, rstd, train = cache
    f = xhat.shape[-1]
    dy2, xh2 = dy.reshape(-1, f), xhat.reshape(-1, f)
    dscale = (dy2 * xh2).sum(0)
    dbias = dy2.sum(0)
    dxh = dy2 * scale
    if train:
        dx = rstd * (dxh - dxh.mean(0, dtype=np.float32) - xh2 * (dxh * xh2).mean(0, dtype=np.float32))
    else:
        dx = rstd * dxh
    return dx.reshape(dy.shape).astype(np.float32), dscale.astype(np.float32), dbias.astype(np.float32)


# BatchRenorm (purejaxql/utils/batch_renorm.py:19-131, used by pqn_craftax.py:44-53): eps 1e-3, momentum 0.999,
# r_max 3, d_max 5, renormalisation active once steps >= 1000.  r and d carry no gradient (stop_gradient, :100-103).
BRN_EPS, BRN_MOMENTUM, BRN_R_MAX, BRN_D_MAX, BRN_WARMUP = np.float32(1e-3), np.float32(0.999), np.float32(3.0), np.float32(5.0), 1000


MOMENTS_DTYPE = np.float32   # batch moments of brn_fwd: f32 = the flax restatement (default); see the note inside


def brn_fwd(x, scale, bias, stats, train, new_stats=None):
    f = x.shape[-1]
    x2 = x.reshape(-1, f).astype(np.float32)
    if not train:
        mean, var = stats["mean"], stats["var"]
        cache = None
    else:
        # flax's fast variance E[x^2] - E[x]^2 in f32 (MOMENTS_DTYPE).  For near-constant columns (a symbolic observation has
        # many) it cancels catastrophically: two correct f32 implementations then differ at the 1e-3 level.  Tests that need
        # to tell such conditioning apart from an error set MOMENTS_DTYPE = np.float64 -- the value both approximate.
        md = MOMENTS_DTYPE
        bmean64 = x2.mean(0, dtype=md)
        bvar = np.maximum((x2.astype(md) * x2.astype(md)).mean(0, dtype=md) - bmean64 * bmean64, 0).astype(np.float32)
        bmean = bmean64.astype(np.float32)
        mean, var = bmean, bvar
        r = d = None
        if int(stats["steps"]) >= BRN_WARMUP:
            ra_std = np.sqrt(stats["var"] + BRN_EPS)
            r = np.clip(np.sqrt(bvar + BRN_EPS) / ra_std, 1 / BRN_R_MAX, BRN_R_MAX).astype(np.float32)   # :100-101
            d = np.clip((bmean - stats["mean"]) / ra_std, -BRN_D_MAX, BRN_D_MAX).astype(np.float32)      # :102-103
            var = (bvar / (r * r)).astype(np.float32)                                                    # :104
            mean = (bmean - d * np.sqrt(bvar) / r).astype(np.float32)                                    # :105
        if new_stats is not None:                                                                         # :112-116
            new_stats["mean"] = (BRN_MOMENTUM * stats["mean"] + (1 - BRN_MOMENTUM) * bmean).astype(np.float32)
            new_stats["var"] = (BRN_MOMENTUM * stats["var"] + (1 - BRN_MOMENTUM) * bvar).astype(np.float32)
            new_stats["steps"] = int(stats["steps"]) + 1
        cache = (x2, bmean, bvar, mean, var, r, d)
    k = (1.0 / np.sqrt(var + BRN_EPS)).astype(np.float32)
    y = ((x2 - mean) * (k * scale) + bias).astype(np.float32)
    return y.reshape(x.shape), cache


def brn_bwd(dy, scale, cache, need_dx=True):
    """d/dx, d/dscale, d/dbias of the train-mode BatchRenorm (batch moments are functions of x; r, d are constants).
    need_dx=False (the INPUT normalisation: nothing consumes d/dx) returns None for it -- its d var / d x term divides by
    sqrt(batch variance), which is 0 for a constant observation column: NaNs that nobody reads, but that np.testing treats as equal."""
    x2, bmean, bvar, cm, cv, r, d = cache
    n = x2.shape[0]
    dy2 = dy.reshape(x2.shape).astype(np.float32)
    k = 1.0 / np.sqrt(cv + BRN_EPS)
    xc = x2 - cm
    dscale = (dy2 * xc * k).sum(0)
    dbias = dy2.sum(0)
    if not need_dx:
        return None, dscale.astype(np.float32), dbias.astype(np.float32)
    dyh = dy2 * scale
    g_cm = -k * dyh.sum(0)
    g_cv = -0.5 * k ** 3 * (dyh * xc).sum(0)
    if r is None:
        g_mean, g_var = g_cm, g_cv
    else:
        g_mean = g_cm
        g_var = g_cv / (r * r) + g_cm * (-(d / r) / (2.0 * np.sqrt(bvar)))
    dx = dyh * k + g_mean / n + g_var * 2.0 * (x2 - bmean) / n
    return dx.reshape(dy.shape).astype(np.float32), dscale.astype(np.float32), dbias.astype(np.float32)


def _brn_named_fwd(x, scale, bias, name, train, stats, new_stats):
    """BatchRenorm module `name` of the Craftax network (pqn_craftax.py:43-51) on the flat batch_stats dict."""
    st = {k: stats[name + "/" + k] for k in ("mean", "var", "steps")}
    ns = {} if (train and new_stats is not None) else None
    y, cache = brn_fwd(x, scale, bias, st, train, ns)
    if ns:
        new_stats.update({name + "/" + k: v for k, v in ns.items()})
    return y, ("brn", cache, train)


def _batchnorm_fwd(x, scale, bias, name, train, stats, new_stats, renorm):
    if renorm:
        return _brn_named_fwd(x, scale, bias, name, train, stats, new_stats)
    return _bn_fwd(x, scale, bias, name, train, stats, new_stats)


def _batchnorm_bwd(dy, scale, cache, need_dx=True):
    if isinstance(cache, tuple) and len(cache) == 3 and isinstance(cache[0], str) and cache[0] == "brn":
        _tag, c, train = cache
        if not train:
            raise ValueError("backward through an eval-mode BatchRenorm is not used on this path")
        return brn_bwd(dy, scale, c, need_dx)
    return _bn_bwd(dy, scale, cache)


def _norm_fwd(norm, x, p, name, train, stats, new_stats, renorm=False):
    if norm == "layer_norm":
        return _ln_fwd(x, p[name + "/scale"], p[name + "/bias"])
    if norm == "batch_norm":
        return _batchnorm_fwd(x, p[name + "/scale"], p[name + "/bias"], name, train, stats, new_stats, renorm)
    return x, None


def _norm_bwd(norm, dy, p, name, cache, g):
    if norm == "layer_norm":
        dy, g[name + "/scale"], g[name + "/bias"] = _ln_bwd(dy, p[name + "/scale"], cache)
    elif norm == "batch_norm":
        dy, g[name + "/scale"], g[name + "/bias"] = _batchnorm_bwd(dy, p[name + "/scale"], cache)
    return dy


def net_forward(kind, p, x, use_ln=True, layers=2, want_cache=False, norm_type=None, norm_input=False,
                train=False, stats=None, new_stats=None, renorm=False):
    """QNetwork.apply({"params": p, "batch_stats": stats}, x, train) (pqn_minatar.py:54-69,
    pqn_gymnax.py:29-58).  x float32.  norm_type defaults to layer_norm / none by `use_ln`."""
    norm = _norm_kind(norm_type) if norm_type is not None else ("layer_norm" if use_ln else "none")
    cache = {}
    x = x.astype(np.float32)
    cin = None
    bn0 = bn_module(renorm) + "_0"
    if norm_input:                                                         # :61-62 (and no /255 on this branch)
        x, cin = _batchnorm_fwd(x, p[bn0 + "/scale"], p[bn0 + "/bias"], bn0, train, stats, new_stats, renorm)
    if kind == "cnn":
        b = x.shape[0]
        xs = x if norm_input else (x / np.float32(255.0)).astype(np.float32)   # pqn_minatar.py:66
        n0, n1 = cnn_norm_names(norm)
        pt = _patches(np.ascontiguousarray(xs))                                 # [B,8,8,9C]
        y = pt @ p["CNN_0/Conv_0/kernel"].reshape(-1, 16) + p["CNN_0/Conv_0/bias"]
        y, c0 = _norm_fwd(norm, y, p, n0, train, stats, new_stats)
        h1 = np.maximum(y, 0).reshape(b, -1)                               # (h,w,c) flatten, :47
        z = h1 @ p["CNN_0/Dense_0/kernel"] + p["CNN_0/Dense_0/bias"]
        z, c1 = _norm_fwd(norm, z, p, n1, train, stats, new_stats)
        h2 = np.maximum(z, 0)
        q = h2 @ p["Dense_0/kernel"] + p["Dense_0/bias"]
        if want_cache:
            cache = dict(pt=pt, c0=c0, h1=h1, c1=c1, h2=h2, cin=cin, norm=norm)
        return (q.astype(np.float32), cache) if want_cache else q.astype(np.float32)
    hs, cs = [x], []
    y = hs[0]
    for l in range(layers):
        y = y @ p[f"Dense_{l}/kernel"] + p[f"Dense_{l}/bias"]
        y, c = _norm_fwd(norm, y, p, mlp_norm_name(norm, l, renorm), train, stats, new_stats, renorm)
        cs.append(c)
        y = np.maximum(y, 0)
        hs.append(y)
    q = y @ p[f"Dense_{layers}/kernel"] + p[f"Dense_{layers}/bias"]
    if want_cache:
        return q.astype(np.float32), dict(hs=hs, cs=cs, cin=cin, norm=norm, renorm=renorm)
    return q.astype(np.float32)


def net_loss_grad(kind, p, shapes, x, action, target, use_ln=True, layers=2, norm_type=None, norm_input=False,
                  stats=None, new_stats=None, renorm=False):
    """loss = 0.5*mean((q[a]-target)^2) and d loss / d theta (flat), pqn_minatar.py:271-291 (train=True)."""
    q, cache = net_forward(kind, p, x, use_ln, layers, want_cache=True, norm_type=norm_type, norm_input=norm_input,
                           train=True, stats=stats, new_stats=new_stats, renorm=renorm)
    b = x.shape[0]
    chosen = q[np.arange(b), action]
    diff = (chosen - target).astype(np.float32)
    loss = np.float32(0.5) * np.mean(diff * diff, dtype=np.float32)
    dq = np.zeros_like(q)
    dq[np.arange(b), action] = diff / np.float32(b)
    return loss, chosen, _net_backward(kind, p, shapes, x, cache, dq, layers, norm_input)


def net_loss_grad_1step(kind, p, shapes, x, x_next, action, reward, done, gamma, use_ln=True, layers=2, norm_type=None,
                        norm_input=False, stats=None, new_stats=None, renorm=False):
    """The `Q_LAMBDA: False` branch of _loss_fn (pqn_craftax.py:287-304): obs and next_obs go through the train-mode
    network as ONE batch (batch statistics over both halves), q_next carries no gradient,
    target = reward + (1 - done) * gamma * max_a q_next, loss = 0.5 * mean((q[a] - target)^2) over the obs half."""
    b = x.shape[0]
    xx = np.concatenate((x, x_next)).astype(np.float32)                   # :296
    q_all, cache = net_forward(kind, p, xx, use_ln, layers, want_cache=True, norm_type=norm_type, norm_input=norm_input,
                               train=True, stats=stats, new_stats=new_stats, renorm=renorm)
    q, q_next = q_all[:b], q_all[b:]                                       # :300
    target = (reward + (np.float32(1) - done.astype(np.float32)) * np.float32(gamma) * q_next.max(-1)).astype(np.float32)   # :301-306
    chosen = q[np.arange(b), action]
    diff = (chosen - target).astype(np.float32)
    loss = np.float32(0.5) * np.mean(diff * diff, dtype=np.float32)
    dq = np.zeros_like(q_all)                                              # stop_gradient(q_next): zero rows
    dq[np.arange(b), action] = diff / np.float32(b)
    return loss, chosen, _net_backward(kind, p, shapes, xx, cache, dq, layers, norm_input)


def _net_backward(kind, p, shapes, x, cache, dq, layers, norm_input):
    """d (sum of q * dq) / d theta, flat in `shapes` order, from the forward cache of net_forward(want_cache=True)."""
    norm = cache["norm"]
    renorm = cache.get("renorm", False)
    b = x.shape[0]
    g = {k: np.zeros(s, np.float32) for k, s in shapes.items()}
    if kind == "cnn":
        n0, n1 = cnn_norm_names(norm)
        g["Dense_0/kernel"] = cache["h2"].T @ dq
        g["Dense_0/bias"] = dq.sum(0)
        dz = (dq @ p["Dense_0/kernel"].T) * (cache["h2"] > 0)
        dz = _norm_bwd(norm, dz, p, n1, cache["c1"], g)
        g["CNN_0/Dense_0/kernel"] = cache["h1"].T @ dz
        g["CNN_0/Dense_0/bias"] = dz.sum(0)
        dh1 = (dz @ p["CNN_0/Dense_0/kernel"].T) * (cache["h1"] > 0)
        dy = dh1.reshape(b, x.shape[1] - 2, x.shape[2] - 2, 16)
        dy = _norm_bwd(norm, dy, p, n0, cache["c0"], g)
        pt = cache["pt"].reshape(-1, cache["pt"].shape[-1])
        g["CNN_0/Conv_0/kernel"] = (pt.T @ dy.reshape(-1, 16)).reshape(shapes["CNN_0/Conv_0/kernel"])
        g["CNN_0/Conv_0/bias"] = dy.reshape(-1, 16).sum(0)
        if norm_input:   # d loss / d (input-BatchNorm output) = conv transposed: scatter the window gradients back
            wk = p["CNN_0/Conv_0/kernel"].reshape(-1, 16)
            dpt = (dy.reshape(-1, 16) @ wk.T).reshape(b, x.shape[1] - 2, x.shape[2] - 2, 3, 3, x.shape[3])
            dxn = np.zeros(x.shape, np.float32)
            for ky in range(3):
                for kx in range(3):
                    dxn[:, ky:ky + x.shape[1] - 2, kx:kx + x.shape[2] - 2, :] += dpt[:, :, :, ky, kx, :]
            _dx, g["BatchNorm_0/scale"], g["BatchNorm_0/bias"] = _bn_bwd(dxn, p["BatchNorm_0/scale"], cache["cin"])
    else:
        bn0 = bn_module(renorm) + "_0"
        hs, cs = cache["hs"], cache["cs"]
        g[f"Dense_{layers}/kernel"] = hs[-1].T @ dq
        g[f"Dense_{layers}/bias"] = dq.sum(0)
        d = dq @ p[f"Dense_{layers}/kernel"].T
        for l in reversed(range(layers)):
            d = d * (hs[l + 1] > 0)
            d = _norm_bwd(norm, d, p, mlp_norm_name(norm, l, renorm), cs[l], g)
            g[f"Dense_{l}/kernel"] = hs[l].T @ d
            g[f"Dense_{l}/bias"] = d.sum(0)
            d = d @ p[f"Dense_{l}/kernel"].T
        if norm_input:
            _dx, g[bn0 + "/scale"], g[bn0 + "/bias"] = _batchnorm_bwd(d, p[bn0 + "/scale"], cache["cin"], need_dx=False)
    return np.concatenate([g[k].reshape(-1).astype(np.float32) for k in shapes])


# ---- whole loop -----------------------------------------------------------------------------
INFO_KEYS = ("discount", "returned_episode_returns", "returned_episode_lengths", "timestep", "returned_episode")


def make_train(config: Dict[str, Any], script: str = "gymnax"):
    """numpy/C restatement of make_train (pqn_minatar.py:89-431) with the SAME key
    schedule as purejaxql_amd.pqn (documented there).
    script="craftax": the twin of pqn_craftax.py:82-468 -- flat observations into the BatchRenorm MLP (:33-62), the env
    batched by OptimisticResetVecEnvWrapper(LogWrapper(env)) or BatchEnvWrapper(LogWrapper(auto-reset env)) (:96-114), the
    `Q_LAMBDA` switch of the loss (:277-304) and done-weighted info means (:364-369, :433-437)."""
    craftax = script == "craftax"
    config["NUM_UPDATES"] = config["TOTAL_TIMESTEPS"] // config["NUM_STEPS"] // config["NUM_ENVS"]
    config["NUM_UPDATES_DECAY"] = config["TOTAL_TIMESTEPS_DECAY"] // config["NUM_STEPS"] // config["NUM_ENVS"]
    assert (config["NUM_STEPS"] * config["NUM_ENVS"]) % config["NUM_MINIBATCHES"] == 0
    env = OracleEnv(config["ENV_NAME"])
    kind = "cnn" if len(env.obs_shape) == 3 and not craftax else "mlp"
    obs_shape = env.obs_shape if kind == "cnn" else (int(np.prod(env.obs_shape)),)   # the Craftax network sees flat observations
    N, T = int(config["NUM_ENVS"]), int(config["NUM_STEPS"])
    NU, MB, EP = int(config["NUM_UPDATES"]), int(config["NUM_MINIBATCHES"]), int(config["NUM_EPOCHS"])
    B = N * T // MB
    A = env.num_actions
    layers = int(config.get("NUM_LAYERS", 2))
    use_ln = config["NORM_TYPE"] == "layer_norm"
    norm_type, norm_input = config["NORM_TYPE"], bool(config.get("NORM_INPUT", False))
    hidden = int(config.get("HIDDEN_SIZE", 128))
    shapes = cnn_shapes(env.obs_shape, A, norm_type) if kind == "cnn" else mlp_shapes(obs_shape[0], A, hidden, layers, norm_type, craftax)
    nkw = dict(norm_type=norm_type, norm_input=norm_input, renorm=craftax)
    test_on = bool(config.get("TEST_DURING_TRAINING", False))
    test_steps = env.max_steps if (kind == "cnn" and not craftax) else int(config.get("TEST_NUM_STEPS", env.max_steps))
    q_lambda_loss = bool(config.get("Q_LAMBDA", False)) if craftax else True      # pqn_craftax.py:277
    optimistic = craftax and bool(config.get("USE_OPTIMISTIC_RESETS", False))
    ratio = int(config.get("OPTIMISTIC_RESET_RATIO", 16))

    def flat(o):
        return o.reshape(o.shape[0], -1) if kind == "mlp" else o

    def env_step(key, st, action, n):
        """the batched env of pqn_craftax.py:96-114 / the vmapped gymnax env of pqn_minatar.py:110-112"""
        if optimistic:
            return env.step_optimistic(key, st, action, min(ratio, n))
        return env.step(key, st, action)
    gamma, lam, rs = float(config["GAMMA"]), float(config["LAMBDA"]), float(config.get("REW_SCALE", 1))

    def train(rng: int, init_theta: np.ndarray, max_updates: int = None, shard_world: int = 1):
        """shard_world > 1: the envs of this ONE seed sharded over that many ranks (`config` is the
        per-rank config, NUM_ENVS = the rank's share): every rank rolls out its own envs under the shared
        parameters with rank-distinct env / shuffle streams (fold_in(K, 1000 + rank), as purejaxql_amd.pqn), and
        every optimizer step uses the gradient averaged over the ranks' minibatches (pqn_minatar.py:159-162,285-292
        on the global minibatch).  Metric means are means over the shards, step counts are over all envs."""
        K = int(rng) & 0xFFFFFFFFFFFFFFFF
        K_init, K_reset, K_test, K_roll, K_shuf = (fold_in(K, i) for i in range(5))
        W = int(shard_world)
        if W > 1:
            Kr = [tuple(fold_in(k, 1000 + r) for k in (K_reset, K_test, K_roll, K_shuf)) for r in range(W)]
        else:
            Kr = [(K_reset, K_test, K_roll, K_shuf)]
        theta = np.ascontiguousarray(init_theta, np.float32).copy()
        m, v = np.zeros_like(theta), np.zeros_like(theta)
        p = unflatten(theta, shapes)
        stats = init_batch_stats(kind, obs_shape, hidden, layers, norm_type, norm_input, craftax)   # train_state.batch_stats
        lr_steps = config["NUM_UPDATES_DECAY"] * MB * EP
        n_updates = grad_steps = timesteps = 0
        runs = [0]

        def test_metrics_fn():
            if not test_on:
                return None
            k = fold_in(Kr[0][1], runs[0])
            runs[0] += 1
            nt = int(config["TEST_NUM_ENVS"])
            obs, st = env.reset(fold_in(k, 0), nt)
            sums = {kk: 0.0 for kk in INFO_KEYS}
            cnt = 0.0
            for t in range(test_steps):
                sk = fold_in(k, 1 + t)
                a, _ = eps_greedy(net_forward(kind, p, flat(obs), use_ln, layers, stats=stats, **nkw), float(config["EPS_TEST"]), sk)
                obs, st, _r, done, info = env_step(sk, st, a, nt)
                cnt += float(done.sum())
                for kk in INFO_KEYS:
                    sums[kk] += float((info[kk].astype(np.float64) * done).sum())
            return {kk: np.float32(sums[kk] / cnt) if cnt > 0 else np.float32(np.nan) for kk in INFO_KEYS}

        tm = test_metrics_fn()
        shards = []
        for r in range(W):
            obs, st = env.reset(Kr[r][0], N)
            shards.append({"obs": obs, "st": st})
        nu = NU if max_updates is None else min(NU, max_updates)
        metrics = []
        period = int(NU * config["TEST_INTERVAL"]) if test_on else 0
        for u in range(nu):
            eps = linear_schedule(config["EPS_START"], config["EPS_FINISH"], config["EPS_DECAY"] * config["NUM_UPDATES_DECAY"], n_updates)
            info_means = {kk: [] for kk in INFO_KEYS}
            for r, sh in enumerate(shards):
                O = np.zeros((T + 1, N, *obs_shape), np.float32)
                O[0] = flat(sh["obs"])
                Aa = np.zeros((T, N), np.int32)
                R = np.zeros((T, N), np.float32)
                D = np.zeros((T, N), bool)
                QM = np.zeros((T, N), np.float32)
                infos = {kk: [] for kk in INFO_KEYS}
                for t in range(T):
                    sk = fold_in(Kr[r][2], u * T + t)
                    q = net_forward(kind, p, O[t], use_ln, layers, stats=stats, **nkw)
                    Aa[t], QM[t] = eps_greedy(q, np.float32(eps), sk)
                    o_next, sh["st"], rr, D[t], info = env_step(sk, sh["st"], Aa[t], N)
                    O[t + 1] = flat(o_next)
                    R[t] = np.float32(rs) * rr if rs != 1.0 else rr
                    for kk in INFO_KEYS:
                        infos[kk].append(info[kk])
                last_q = net_forward(kind, p, O[T], use_ln, layers, stats=stats, **nkw).max(-1)
                tgt = q_lambda(R, D, QM, last_q, gamma, lam, quirk=True)
                sh["of"], sh["af"], sh["tf"] = O[:T].reshape(T * N, *obs_shape), Aa.reshape(-1), tgt.reshape(-1)
                sh["nf"], sh["rf"], sh["df"] = O[1:].reshape(T * N, *obs_shape), R.reshape(-1), D.reshape(-1)
                sh["obs"] = O[T]
                dmask = np.stack(infos["returned_episode"]).astype(np.float64)
                for kk in INFO_KEYS:
                    x = np.stack(infos[kk])
                    if craftax:   # (x * returned_episode).sum() / returned_episode.sum()  (pqn_craftax.py:364-369)
                        with np.errstate(invalid="ignore", divide="ignore"):
                            info_means[kk].append(float((x.astype(np.float64) * dmask).sum() / dmask.sum()))
                    else:
                        info_means[kk].append(float(np.mean(x.astype(np.float32))))
            timesteps += T * N * W
            losses, qvs = [], []
            for ep in range(EP):
                perms = [permutation(fold_in(Kr[r][3], u * EP + ep), T * N) for r in range(W)]
                for mb in range(MB):
                    gs, ls, cs = [], [], []
                    for r, sh in enumerate(shards):
                        idx = perms[r][mb * B:(mb + 1) * B]
                        new_stats = {}
                        if q_lambda_loss:
                            loss, chosen, g = net_loss_grad(kind, p, shapes, sh["of"][idx], sh["af"][idx], sh["tf"][idx], use_ln,
                                                            layers, stats=stats, new_stats=new_stats, **nkw)
                        else:
                            loss, chosen, g = net_loss_grad_1step(kind, p, shapes, sh["of"][idx], sh["nf"][idx], sh["af"][idx],
                                                                  sh["rf"][idx], sh["df"][idx], gamma, use_ln, layers,
                                                                  stats=stats, new_stats=new_stats, **nkw)
                        gs.append(g)
                        ls.append(loss)
                        cs.append(chosen.mean(dtype=np.float32))
                    stats.update(new_stats)                                # mutable=["batch_stats"] (:272-277,296)
                    g = gs[0] if W == 1 else (np.sum(gs, axis=0, dtype=np.float32) / np.float32(W)).astype(np.float32)
                    lr = linear_schedule(config["LR"], 1e-20, lr_steps, grad_steps) if config.get("LR_LINEAR_DECAY", False) else config["LR"]
                    radam_clip_step(theta, g, m, v, grad_steps, np.float32(lr), np.float32(config["MAX_GRAD_NORM"]))
                    grad_steps += 1
                    losses.append(ls[0] if W == 1 else np.float32(np.mean(ls)))
                    qvs.append(cs[0] if W == 1 else np.float32(np.mean(cs)))
            n_updates += 1
            mm = {"env_step": timesteps, "update_steps": n_updates, "grad_steps": grad_steps,
                  "td_loss": float(np.mean(losses)), "qvals": float(np.mean(qvs))}
            if kind == "cnn":
                mm["env_frame"] = timesteps * env.obs_shape[-1]
            for kk in INFO_KEYS:
                mm[kk] = float(np.mean(info_means[kk]))
            if test_on:
                if period > 0 and n_updates % period == 0:
                    tm = test_metrics_fn()
                mm.update({f"test/{k}": float(v2) for k, v2 in tm.items()})
            metrics.append(mm)
        return {"theta": theta, "metrics": metrics, "env_state": shards[0]["st"], "last_obs": shards[0]["obs"],
                "batch_stats": stats, "shards": shards, "opt_mu": m, "opt_nu": v}

    train.shapes = shapes
    train.kind = kind
    return train
