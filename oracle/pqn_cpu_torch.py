"""CPU baseline of the PQN hot path for bench.py's `cpu_baseline` leg (kind = "port") -- TEST / MEASUREMENT infrastructure,
never imported by the product (purejaxql_amd).

The loop is oracle.make_train's (pqn_oracle.py: the restatement of purejaxql/pqn_minatar.py:176-369, same key schedule)
for the MinAtar CNN with the Q(lambda) loss; only the Q-network is evaluated differently: torch-CPU (oneDNN / MKL
convolution, matmul, layer_norm and autograd on EVERY host core, torch.set_num_threads(os.cpu_count())) instead of the
oracle's numpy restatement, as BASELINE.md section 4 plans ("C/OpenMP env step + LogWrapper + eps-greedy + Q(lambda), torch-CPU
Q-network, optax-exact RAdam").  The env step / auto-reset / LogWrapper (OpenMP), the eps-greedy draw, the Q(lambda) scan,
the shuffle and clip + RAdam are the C oracle's.  tests/test_oracle_cpu.py holds this loop to the numpy oracle loop on a
small shape, so the number bench.py prints is the rate of the SAME algorithm.

Network: QNetwork(CNN) of pqn_minatar.py:24-69 -- x / 255 -> Conv 3x3 VALID (HWIO kernel) -> LayerNorm(16) -> relu ->
flatten (h, w, c) -> Dense 128 -> LayerNorm(128) -> relu -> Dense A; loss 0.5 * mean((q_a - target)^2) (:271-285).
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.nn.functional as F

import pqn_oracle as oracle

LN_EPS = 1e-6   # flax nn.LayerNorm


def _net(p, x):
    """p: dict of torch tensors (flax names / layouts), x: float32 [B, 10, 10, C] in {0, 1} (MinAtar observation)."""
    y = F.conv2d((x / 255.0).permute(0, 3, 1, 2), p["CNN_0/Conv_0/kernel"].permute(3, 2, 0, 1), p["CNN_0/Conv_0/bias"])
    y = y.permute(0, 2, 3, 1)                                                             # NHWC: flatten is (h, w, c)
    y = F.relu(F.layer_norm(y, (16,), p["CNN_0/LayerNorm_0/scale"], p["CNN_0/LayerNorm_0/bias"], LN_EPS))
    z = y.reshape(y.shape[0], -1) @ p["CNN_0/Dense_0/kernel"] + p["CNN_0/Dense_0/bias"]
    z = F.relu(F.layer_norm(z, (128,), p["CNN_0/LayerNorm_1/scale"], p["CNN_0/LayerNorm_1/bias"], LN_EPS))
    return z @ p["Dense_0/kernel"] + p["Dense_0/bias"]


def make_train(config, threads=None):
    """train(rng, init_theta, max_updates) -> dict(theta, metrics, seconds_per_update[list]) for a MinAtar CNN config with
    NORM_TYPE layer_norm, NORM_INPUT False (the yaml default the bench runs)."""
    config = dict(config)
    assert config["NORM_TYPE"] == "layer_norm" and not config.get("NORM_INPUT", False)
    threads = int(threads or os.cpu_count() or 1)
    torch.set_num_threads(threads)
    config["NUM_UPDATES"] = config["TOTAL_TIMESTEPS"] // config["NUM_STEPS"] // config["NUM_ENVS"]
    config["NUM_UPDATES_DECAY"] = config["TOTAL_TIMESTEPS_DECAY"] // config["NUM_STEPS"] // config["NUM_ENVS"]
    env = oracle.OracleEnv(config["ENV_NAME"])
    assert len(env.obs_shape) == 3, "MinAtar CNN path only"
    N, T = int(config["NUM_ENVS"]), int(config["NUM_STEPS"])
    NU, MB, EP = int(config["NUM_UPDATES"]), int(config["NUM_MINIBATCHES"]), int(config["NUM_EPOCHS"])
    B = N * T // MB
    shapes = oracle.cnn_shapes(env.obs_shape, env.num_actions, "layer_norm")
    gamma, lam, rs = float(config["GAMMA"]), float(config["LAMBDA"]), float(config.get("REW_SCALE", 1))

    def train(rng, init_theta, max_updates=None, time_budget=None, min_updates=1):
        """time_budget (seconds): stop after the first update that ends past it, once min_updates have run"""
        K = int(rng) & 0xFFFFFFFFFFFFFFFF
        _k_init, k_reset, _k_test, k_roll, k_shuf = (oracle.fold_in(K, i) for i in range(5))
        theta = np.ascontiguousarray(init_theta, np.float32).copy()
        m, v = np.zeros_like(theta), np.zeros_like(theta)
        views = oracle.unflatten(theta, shapes)                      # numpy views into theta ...
        p = {k: torch.from_numpy(a).requires_grad_(True) for k, a in views.items()}   # ... shared with torch: RAdam updates both
        names = list(shapes)
        lr_steps = config["NUM_UPDATES_DECAY"] * MB * EP
        obs, st = env.reset(k_reset, N)
        n_updates = grad_steps = 0
        nu = NU if max_updates is None else min(NU, max_updates)
        metrics, secs = [], []
        for u in range(nu):
            t0 = time.perf_counter()
            eps = oracle.linear_schedule(config["EPS_START"], config["EPS_FINISH"], config["EPS_DECAY"] * config["NUM_UPDATES_DECAY"], n_updates)
            O = np.zeros((T + 1, N, *env.obs_shape), np.float32)
            O[0] = obs
            A, R = np.zeros((T, N), np.int32), np.zeros((T, N), np.float32)
            D, QM = np.zeros((T, N), bool), np.zeros((T, N), np.float32)
            with torch.no_grad():
                for t in range(T):
                    sk = oracle.fold_in(k_roll, u * T + t)
                    q = _net(p, torch.from_numpy(O[t])).numpy()
                    A[t], QM[t] = oracle.eps_greedy(q, np.float32(eps), sk)
                    O[t + 1], st, rr, D[t], _info = env.step(sk, st, A[t])
                    R[t] = np.float32(rs) * rr if rs != 1.0 else rr
                last_q = _net(p, torch.from_numpy(O[T])).numpy().max(-1)
            tgt = oracle.q_lambda(R, D, QM, last_q, gamma, lam, quirk=True)
            of = torch.from_numpy(O[:T].reshape(T * N, *env.obs_shape))
            af, tf = torch.from_numpy(A.reshape(-1).astype(np.int64)), torch.from_numpy(tgt.reshape(-1))
            obs = O[T]
            losses, qvs = [], []
            for ep in range(EP):
                perm = torch.from_numpy(oracle.permutation(oracle.fold_in(k_shuf, u * EP + ep), T * N).astype(np.int64))
                for mb in range(MB):
                    idx = perm[mb * B:(mb + 1) * B]
                    q = _net(p, of[idx])
                    chosen = q.gather(1, af[idx][:, None])[:, 0]
                    loss = 0.5 * torch.square(chosen - tf[idx]).mean()
                    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
                    g = torch.cat([(gi if gi is not None else torch.zeros_like(p[k])).reshape(-1)
                                   for k, gi in zip(names, grads)]).numpy()
                    lr = (oracle.linear_schedule(config["LR"], 1e-20, lr_steps, grad_steps)
                          if config.get("LR_LINEAR_DECAY", False) else config["LR"])
                    oracle.radam_clip_step(theta, g, m, v, grad_steps, np.float32(lr), np.float32(config["MAX_GRAD_NORM"]))
                    grad_steps += 1
                    losses.append(float(loss.detach()))
                    qvs.append(float(chosen.detach().mean()))
            n_updates += 1
            metrics.append({"env_step": n_updates * T * N, "update_steps": n_updates, "grad_steps": grad_steps,
                            "td_loss": float(np.mean(losses)), "qvals": float(np.mean(qvs))})
            secs.append(time.perf_counter() - t0)
            if time_budget is not None and len(secs) >= min_updates and sum(secs) > time_budget:
                break
        return {"theta": theta, "metrics": metrics, "seconds_per_update": secs, "threads": threads}

    return train
