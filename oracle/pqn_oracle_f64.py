"""float64 twin of the oracle's learn phase -- TEST INFRASTRUCTURE (only tests/ import it), never the product.

Why it exists (VERDICT r5, weak point 2): at the headline size the whole-update test accepts cosine > 0.998 / rel-L2 < 6e-2 between
the HIP kernels and the numpy-f32 oracle, with the two GPU paths 2.4e-3 apart and either 3.9e-2 from numpy.  To show which side is
nearer to exact arithmetic, the 64 optimizer steps of one update (purejaxql/pqn_minatar.py:263-327: per epoch one permutation, per
minibatch value_and_grad(_loss_fn) :271-291 and optax.chain(clip_by_global_norm, radam) :159-162,292) are run here in float64 on
the f32 oracle's OWN rollout record (observations, actions, Q(lambda) targets: the inputs of the learn phase, so that the eps-greedy
argmax cannot send the two precisions down different trajectories).  MinAtar CNN with NORM_TYPE=layer_norm only."""
from __future__ import annotations

import numpy as np

import pqn_oracle as oracle

F = np.float64
LN_EPS = 1e-6


def _ln_fwd(x, scale, bias):
    mu = x.mean(-1, keepdims=True)
    var = np.maximum((x * x).mean(-1, keepdims=True) - mu * mu, 0.0)     # flax: E[x^2] - E[x]^2, clamped
    rstd = 1.0 / np.sqrt(var + LN_EPS)
    xhat = (x - mu) * rstd
    return xhat * scale + bias, (xhat, rstd)


def _ln_bwd(dy, scale, cache):
    xhat, rstd = cache
    dscale = (dy * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    dbias = dy.reshape(-1, xhat.shape[-1]).sum(0)
    dxh = dy * scale
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xhat * (dxh * xhat).mean(-1, keepdims=True))
    return dx, dscale, dbias


def cnn_loss_grad(p, shapes, x, action, target):
    """loss = 0.5 * mean((q[a] - target)^2) and d loss / d theta (flat, `shapes` order) of QNetwork(CNN, layer_norm) in float64
    (pqn_minatar.py:24-69, 271-291); same structure as pqn_oracle.net_loss_grad / _net_backward."""
    x = np.asarray(x, F)
    b = x.shape[0]
    pt = oracle._patches(np.ascontiguousarray(x / 255.0))
    y = pt @ p["CNN_0/Conv_0/kernel"].reshape(-1, 16) + p["CNN_0/Conv_0/bias"]
    y, c0 = _ln_fwd(y, p["CNN_0/LayerNorm_0/scale"], p["CNN_0/LayerNorm_0/bias"])
    h1 = np.maximum(y, 0).reshape(b, -1)
    z = h1 @ p["CNN_0/Dense_0/kernel"] + p["CNN_0/Dense_0/bias"]
    z, c1 = _ln_fwd(z, p["CNN_0/LayerNorm_1/scale"], p["CNN_0/LayerNorm_1/bias"])
    h2 = np.maximum(z, 0)
    q = h2 @ p["Dense_0/kernel"] + p["Dense_0/bias"]
    chosen = q[np.arange(b), action]
    diff = chosen - np.asarray(target, F)
    loss = 0.5 * np.mean(diff * diff)
    dq = np.zeros_like(q)
    dq[np.arange(b), action] = diff / b
    g = {k: np.zeros(s, F) for k, s in shapes.items()}
    g["Dense_0/kernel"] = h2.T @ dq
    g["Dense_0/bias"] = dq.sum(0)
    dz = (dq @ p["Dense_0/kernel"].T) * (h2 > 0)
    dz, g["CNN_0/LayerNorm_1/scale"], g["CNN_0/LayerNorm_1/bias"] = _ln_bwd(dz, p["CNN_0/LayerNorm_1/scale"], c1)
    g["CNN_0/Dense_0/kernel"] = h1.T @ dz
    g["CNN_0/Dense_0/bias"] = dz.sum(0)
    dy = ((dz @ p["CNN_0/Dense_0/kernel"].T) * (h1 > 0)).reshape(b, x.shape[1] - 2, x.shape[2] - 2, 16)
    dy, g["CNN_0/LayerNorm_0/scale"], g["CNN_0/LayerNorm_0/bias"] = _ln_bwd(dy, p["CNN_0/LayerNorm_0/scale"], c0)
    g["CNN_0/Conv_0/kernel"] = (pt.reshape(-1, pt.shape[-1]).T @ dy.reshape(-1, 16)).reshape(shapes["CNN_0/Conv_0/kernel"])
    g["CNN_0/Conv_0/bias"] = dy.reshape(-1, 16).sum(0)
    return loss, chosen, np.concatenate([g[k].reshape(-1) for k in shapes])


def radam_clip_step(p, g, m, v, count, lr, max_norm):
    """optax.chain(clip_by_global_norm(max_norm), radam(lr)) in float64, in place on p, m, v: the arithmetic of
    oracle/pqn_oracle.c:pqn_oracle_radam_clip_step (b1 .9, b2 .999, eps 1e-8 outside the root, threshold 5)."""
    b1, b2, eps, threshold = 0.9, 0.999, 1e-8, 5.0
    gnorm = float(np.sqrt(np.sum(g * g)))
    if not gnorm < max_norm:
        g = (g / gnorm) * max_norm
    t = float(count + 1)
    b1t, b2t = b1 ** t, b2 ** t
    ro_inf = 2.0 / (1.0 - b2) - 1.0
    ro = ro_inf - 2.0 * t * b2t / (1.0 - b2t)
    m[:] = (1.0 - b1) * g + b1 * m
    v[:] = (1.0 - b2) * (g * g) + b2 * v
    mh, vh = m / (1.0 - b1t), v / (1.0 - b2t)
    if ro >= threshold:
        r = np.sqrt((ro - 4.0) * (ro - 2.0) * ro_inf / ((ro_inf - 4.0) * (ro_inf - 2.0) * ro))
        p -= lr * (r * mh / (np.sqrt(vh) + eps))
    else:
        p -= lr * mh
    return gnorm


def learn_phase(config, shapes, theta0, obs_flat, action_flat, target_flat, k_shuf, update_index=0, grad_steps0=0, m0=None, v0=None):
    """The NUM_EPOCHS x NUM_MINIBATCHES optimizer steps of update `update_index` in float64 from a rollout record in the oracle's
    flattened order (index t * N + e): returns (theta, m, v) as float64 arrays.  Key schedule, permutation, LR schedule and step
    count are the oracle's own (pqn_oracle.make_train)."""
    mb_n, ep_n = int(config["NUM_MINIBATCHES"]), int(config["NUM_EPOCHS"])
    tn = obs_flat.shape[0]
    b = tn // mb_n
    theta = np.asarray(theta0, F).copy()
    m = np.zeros_like(theta) if m0 is None else np.asarray(m0, F).copy()
    v = np.zeros_like(theta) if v0 is None else np.asarray(v0, F).copy()
    p = oracle.unflatten(theta, shapes)       # views into theta
    lr_steps = config["NUM_UPDATES_DECAY"] * mb_n * ep_n
    grad_steps = int(grad_steps0)
    for ep in range(ep_n):
        perm = oracle.permutation(oracle.fold_in(k_shuf, update_index * ep_n + ep), tn)
        for mb in range(mb_n):
            idx = perm[mb * b:(mb + 1) * b]
            _loss, _chosen, g = cnn_loss_grad(p, shapes, obs_flat[idx], action_flat[idx], target_flat[idx])
            lr = oracle.linear_schedule(config["LR"], 1e-20, lr_steps, grad_steps) if config.get("LR_LINEAR_DECAY", False) else config["LR"]
            radam_clip_step(theta, g, m, v, grad_steps, float(lr), float(config["MAX_GRAD_NORM"]))
            grad_steps += 1
    return theta, m, v
