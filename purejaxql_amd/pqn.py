"""make_train(config) -- the PQN hot path (rollout + Q(lambda) + minibatch update).

Same surface and control flow as the reference closures
  purejaxql/pqn_minatar.py:89-431   (CNN, MinAtar)
  purejaxql/pqn_gymnax.py:78-424    (MLP, gymnax classic control)
but built MI355X-first: env dynamics, eps-greedy, Q(lambda), shuffle keys and
clip+RAdam are HIP kernels behind the C ABI (include/pqn_hotpath.h), state is
SoA in HBM, the rollout record is time-major [T, N, ...] (the reference's scan
stacking, :214-219), and randomness is counter-based threefry keyed by
(seed, purpose, update*T+t, env) instead of split/carry chains.

    train = make_train(config)          # config: flat dict, UPPER_CASE keys
    out = train(seed_key)               # {"runner_state": ..., "metrics": {name: tensor[NUM_UPDATES]}}
    outs = vmap_train(train, keys)      # jax.vmap(make_train(config))(rngs): leading [S] axis

Key schedule (this build's own; jax streams cannot be reproduced):
    K_init = fold_in(K, 0); K_reset = fold_in(K, 1); K_test = fold_in(K, 2)
    K_roll = fold_in(K, 3) -> step key fold_in(K_roll, u*T + t)
    K_shuf = fold_in(K, 4) -> epoch key fold_in(K_shuf, u*NUM_EPOCHS + ep)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Callable, Dict, List, Optional

import torch

from . import _lib, ops
from .envs import BatchEnvWrapper, FlattenObservationWrapper, LogWrapper, OptimisticResetVecEnvWrapper, make
from .networks import FlatParams, QNetwork

INFO_KEYS = ("discount", "returned_episode_returns", "returned_episode_lengths", "timestep", "returned_episode")


def linear_schedule(init_value: float, end_value: float, transition_steps: float) -> Callable[[float], float]:
    """optax.linear_schedule, used for eps at pqn_minatar.py:134-138."""

    def f(count: float) -> float:
        if transition_steps <= 0:      # optax.polynomial_schedule: a constant schedule at init_value
            return init_value
        c = min(max(float(count), 0.0), float(transition_steps))
        return (init_value - end_value) * (1.0 - c / float(transition_steps)) + end_value

    return f


def derive_config(config: Dict[str, Any]) -> Dict[str, Any]:
    """The in-place derivations of make_train (pqn_minatar.py:91-101)."""
    config["NUM_UPDATES"] = config["TOTAL_TIMESTEPS"] // config["NUM_STEPS"] // config["NUM_ENVS"]
    config["NUM_UPDATES_DECAY"] = config["TOTAL_TIMESTEPS_DECAY"] // config["NUM_STEPS"] // config["NUM_ENVS"]
    assert (config["NUM_STEPS"] * config["NUM_ENVS"]) % config["NUM_MINIBATCHES"] == 0, \
        "NUM_MINIBATCHES must divide NUM_STEPS*NUM_ENVS"
    return config


class TrainState:
    """The fields of the reference's CustomTrainState a caller reads (pqn_minatar.py:82-86): params (flax tree keyed by
    "/"-joined names), batch_stats, timesteps, n_updates, grad_steps, plus the optimizer state of this build."""

    def __init__(self, rs):
        self.params, self.batch_stats = rs["params"], rs.get("batch_stats", {})
        self.timesteps, self.n_updates, self.grad_steps = rs["timesteps"], rs["n_updates"], rs["grad_steps"]
        self.opt_state = {k: rs[k] for k in ("opt_count", "opt_mu", "opt_nu") if k in rs}


class RunnerState(tuple):
    """train()'s `runner_state`: the reference's 4-tuple (pqn_minatar.py:420-424,
    `(train_state, (obs, env_state), test_metrics, rng)`) -- a real tuple, so `runner_state[0].params`,
    `train_state, expl_state, test_metrics, rng = runner_state`, len() and iteration behave exactly as they do on the
    reference's value -- that additionally answers this build's named entries by STRING key (`rs["theta"]`,
    `"driver" in rs`, `rs.get(...)`, `rs.entries` = the dict itself)."""

    def __new__(cls, entries):
        entries = dict(entries)
        self = super().__new__(cls, (TrainState(entries), (entries["last_obs"], entries["env_state"]),
                                     entries.get("test_metrics"), entries.get("rng")))
        self.entries = entries
        return self

    def __getitem__(self, k):
        if isinstance(k, str):
            return self.entries[k]
        return tuple.__getitem__(self, k)

    def __contains__(self, k):
        return k in self.entries if isinstance(k, str) else tuple.__contains__(self, k)

    def get(self, k, default=None):
        return self.entries.get(k, default)

    def as_tuple(self):
        return tuple(self)


class _TestRows:
    """metrics["test/<k>"][u] = the result of the LATEST evaluation at update u (pqn_minatar.py:340-350: test_metrics is carried
    through the scan and refreshed every NUM_UPDATES * TEST_INTERVAL updates).  Kept as (first update, values) spans and
    written out once, in rows(): the update drivers replay one hipGraph per update, and NOTHING should be enqueued between
    two replays that does not have to be -- a per-update torch.stack + slice assignment here (~100 small launches behind a 27 ms
    graph at 16 seeds x 4096 envs) ended in a GPU memory access fault after ~16 updates on ROCm 7.2, while the same launches
    behind a stream synchronisation, or the same update enqueued eagerly, ran clean (tools/learn_headline.py found it)."""

    def __init__(self, num_updates: int):
        self.num_updates, self.spans = int(num_updates), []

    def note(self, u: int, values: torch.Tensor) -> None:
        """`values` ([..., len(INFO_KEYS)]) holds from update u on (until the next note)."""
        self.spans.append((int(u), values))

    def rows(self) -> torch.Tensor:
        """[..., NUM_UPDATES, len(INFO_KEYS)]"""
        first = self.spans[0][1]
        out = torch.zeros((*first.shape[:-1], self.num_updates, first.shape[-1]), dtype=torch.float32, device=first.device)
        for i, (u0, v) in enumerate(self.spans):
            u1 = self.spans[i + 1][0] if i + 1 < len(self.spans) else self.num_updates
            if u1 > u0:
                out[..., u0:u1, :] = v.to(torch.float32).unsqueeze(-2)
        return out


def _streamed(stream, fn, join: bool):
    """Runs fn with `stream` as PyTorch's current stream: the stream first waits for the caller's current stream (inputs prepared
    there), and with join=True the caller's stream waits for `stream` afterwards (results are then ordered for the caller).
    Attributes set on fn (update.driver) are kept."""
    if stream is None:
        return fn

    def wrapped(*args, **kwargs):
        cur = torch.cuda.current_stream(stream.device)
        if cur == stream:
            return fn(*args, **kwargs)
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            out = fn(*args, **kwargs)
        if join:
            cur.wait_stream(stream)
        return out

    wrapped.__dict__.update(fn.__dict__)
    wrapped.__doc__ = fn.__doc__
    return wrapped


def _streamed_factory(stream, factory):
    """A runner factory (make_runner / make_batch_runner) whose construction, update() and finish() all run on `stream`."""
    if stream is None:
        return factory

    def build(*args, **kwargs):
        update, finish = _streamed(stream, factory, join=True)(*args, **kwargs)
        return _streamed(stream, update, join=False), _streamed(stream, finish, join=True)

    build.__doc__ = factory.__doc__
    return build


def _sync_behind_graph(drv) -> None:
    """Host-side wait for the stream before eager work is enqueued behind a replayed update graph -- only when the run sits on the
    legacy NULL stream (config _WORK_STREAM=False): that is the one combination that faults (see make_train); on the run's own
    stream nothing waits."""
    if drv is not None and getattr(drv, "graph", None) is not None and torch.cuda.current_stream().cuda_stream == 0:
        torch.cuda.current_stream().synchronize()


class _Rollout:
    """Time-major rollout record of one update: the reference's `Transition`
    (pqn_minatar.py:72-79) with next_obs folded into slot T of the observation buffer and q_val
    reduced to max_a q (its only consumer is :249).  MinAtar runs keep the observation bit-packed
    (64 B instead of 1600 B per Breakout frame); flat-obs envs keep f32."""

    def __init__(self, t, n, obs_shape, obs_words, device):
        f32, i32 = torch.float32, torch.int32
        if obs_words:
            self.bits = torch.empty((t + 1, n, obs_words), dtype=i32, device=device)
            self.obs = None
        else:
            self.bits = None
            self.obs = torch.empty((t + 1, n, *obs_shape), dtype=f32, device=device)
        self.action = torch.empty((t, n), dtype=i32, device=device)
        self.reward = torch.empty((t, n), dtype=f32, device=device)
        self.done = torch.empty((t, n), dtype=torch.uint8, device=device)
        self.qmax = torch.empty((t, n), dtype=f32, device=device)
        self.discount = torch.empty((t, n), dtype=f32, device=device)
        self.rer = torch.empty((t, n), dtype=f32, device=device)
        self.rel = torch.empty((t, n), dtype=i32, device=device)
        self.ts = torch.empty((t, n), dtype=i32, device=device)
        self.target = torch.empty((t, n), dtype=f32, device=device)
        self.last_q = torch.empty(n, dtype=f32, device=device)


class _TorchPolicy:
    """Q-network through torch ops + autograd (plumbing path; the MLP of pqn_gymnax.py:29-58, and the
    CNN when config["_BACKEND"] == "torch").  Parameters live in one flat buffer, optimizer = HIP RAdam."""
    packed = False

    def __init__(self, network, theta, config, lr_steps, grad_hook):
        self.net = network
        self.fp = FlatParams(network, theta)
        self.opt = ops.FlatRAdam(self.fp.theta, config["LR"], config["MAX_GRAD_NORM"], lr_decay_steps=lr_steps)
        self.grad_hook = grad_hook
        # train_state.batch_stats (pqn_minatar.py:81-86,169): running moments of the BatchNorm layers
        self.stats = network.init_batch_stats() if network.has_batch_stats else None
        if self.stats is not None and grad_hook is not None:
            raise NotImplementedError("BatchNorm with the envs of one seed sharded over ranks would need the batch "
                                      "moments all-reduced as well; use seed sharding (dist.partition_seeds)")

    def q_values(self, obs):
        with torch.no_grad():   # train=False: running averages (pqn_minatar.py:184-192)
            return self.net.apply(self.fp.leaves, obs, train=False, stats=self.stats)

    def act(self, obs, eps, key, action, qmax):
        ops.eps_greedy(self.q_values(obs), eps, key, action, qmax)

    def max_q(self, obs, out):
        out.copy_(self.q_values(obs).max(dim=-1).values)

    def sgd_step(self, idx, obs_flat, act_flat, tgt_flat, loss_out, qv_out):
        self.fp.zero_grad()
        new_stats = {} if self.stats is not None else None   # train=True, mutable=["batch_stats"] (:272-277)
        qv = self.net.apply(self.fp.leaves, obs_flat[idx], train=True, stats=self.stats, new_stats=new_stats)
        if new_stats:
            self.stats = {**self.stats, **new_stats}          # (:296)
        chosen = qv.gather(1, act_flat[idx].to(torch.int64).unsqueeze(1)).squeeze(1)
        loss = 0.5 * torch.square(chosen - tgt_flat[idx]).mean()
        loss.backward()
        if self.grad_hook is not None:
            self.grad_hook(self.fp.grad)
        self.opt.step(self.fp.grad)
        loss_out.copy_(loss.detach())
        qv_out.copy_(chosen.detach().mean())

    def sgd_step_1step(self, idx, obs_all_flat, n_env, act_flat, rew_flat, done_flat, gamma, loss_out, qv_out):
        """The `Q_LAMBDA: False` branch of _loss_fn (pqn_craftax.py:287-304): obs and next_obs through the train-mode
        network as ONE batch, q_next without gradient, target = reward + (1 - done) * gamma * max_a q_next.
        obs_all_flat: the [T+1][N] observation record flattened -- next_obs of transition j is row j + N."""
        self.fp.zero_grad()
        new_stats = {} if self.stats is not None else None
        b = idx.numel()
        x = torch.cat((obs_all_flat[idx], obs_all_flat[idx + n_env]))                                   # (:296)
        q_all = self.net.apply(self.fp.leaves, x, train=True, stats=self.stats, new_stats=new_stats)
        if new_stats:
            self.stats = {**self.stats, **new_stats}
        qv, q_next = q_all[:b], q_all[b:].detach()                                                      # (:300-301)
        target = rew_flat[idx] + (1.0 - done_flat[idx].to(torch.float32)) * gamma * q_next.max(dim=-1).values   # (:302-306)
        chosen = qv.gather(1, act_flat[idx].to(torch.int64).unsqueeze(1)).squeeze(1)
        loss = 0.5 * torch.square(chosen - target).mean()
        loss.backward()
        if self.grad_hook is not None:
            self.grad_hook(self.fp.grad)
        self.opt.step(self.fp.grad)
        loss_out.copy_(loss.detach())
        qv_out.copy_(chosen.detach().mean())

    def theta_flax(self):
        return self.fp.theta

    def opt_state(self):
        return {"opt_count": self.opt.count, "opt_mu": self.opt.m, "opt_nu": self.opt.v,
                "batch_stats": self.stats if self.stats is not None else {}}


class _FusedCnnPolicy:
    """The MinAtar CNN through the fused HIP kernels (csrc/pqn_qnet.hip): packed observations in,
    forward + eps-greedy in one launch, forward+backward+RAdam in four."""
    packed = True

    def __init__(self, network, theta, config, lr_steps, grad_hook, max_mb):
        from .qnet import CnnKernelLayout, CnnTrainer, cnn_forward, matmul_mode
        self.net = network
        self.layout = CnnKernelLayout(network.obs_shape[-1], network.action_dim,
                                      matmul_f16=matmul_mode(config.get("MATMUL_DTYPE", "auto"), max_mb))
        self.tr = CnnTrainer(self.layout, theta, config["LR"], config["MAX_GRAD_NORM"], lr_decay_steps=lr_steps,
                             max_minibatch=max_mb)
        self.fwd = cnn_forward
        self.grad_hook = grad_hook

    def q_values(self, bits):
        return self.fwd(self.layout, bits, self.tr.theta)[0]

    def act(self, bits, eps, key, action, qmax):
        self.fwd(self.layout, bits, self.tr.theta, want_q=False, eps=eps, key=key, action=action, qmax=qmax)

    def max_q(self, bits, out):
        self.fwd(self.layout, bits, self.tr.theta, want_q=False, qmax=out)

    def sgd_step(self, idx, bits_flat, act_flat, tgt_flat, loss_out, qv_out):
        self.tr.compute_grad(idx, bits_flat, act_flat, tgt_flat, loss_out, qv_out)
        if self.grad_hook is not None:
            self.grad_hook(self.tr.grad)
        self.tr.apply(recompute_norm=self.grad_hook is not None)

    def theta_flax(self):
        return self.tr.theta_flax()

    def opt_state(self):
        return {"opt_count": self.tr.count, "opt_mu": self.tr.m, "opt_nu": self.tr.v,
                "kernel_layout": self.layout}


class _FusedMlpPolicy:
    """The gymnax MLP Q-network (pqn_gymnax.py:29-58) through the fused HIP kernels of csrc/pqn_mlp.hip."""
    packed = False

    def __init__(self, network, theta, config, lr_steps, grad_hook, max_mb):
        from .qnet import MlpKernelLayout, MlpTrainer, mlp_forward
        self.net = network
        self.layout = MlpKernelLayout(network.obs_shape[0], network.hidden, network.layers, network.action_dim)
        self.tr = MlpTrainer(self.layout, theta, config["LR"], config["MAX_GRAD_NORM"], lr_decay_steps=lr_steps,
                             max_minibatch=max_mb)
        self.fwd = mlp_forward
        self.grad_hook = grad_hook

    def q_values(self, obs):
        return self.fwd(self.layout, obs, self.tr.theta)[0]

    def act(self, obs, eps, key, action, qmax):
        self.fwd(self.layout, obs, self.tr.theta, want_q=False, eps=eps, key=key, action=action, qmax=qmax)

    def max_q(self, obs, out):
        self.fwd(self.layout, obs, self.tr.theta, want_q=False, qmax=out)

    def sgd_step(self, idx, obs_flat, act_flat, tgt_flat, loss_out, qv_out):
        self.tr.compute_grad(idx, obs_flat, act_flat, tgt_flat, loss_out, qv_out)
        if self.grad_hook is not None:
            self.grad_hook(self.tr.grad)
        self.tr.apply(recompute_norm=self.grad_hook is not None)

    def theta_flax(self):
        return self.tr.theta_flax()

    def opt_state(self):
        return {"opt_count": self.tr.count, "opt_mu": self.tr.m, "opt_nu": self.tr.v, "kernel_layout": self.layout}


class _FusedBigMlpPolicy:
    """The wide LayerNorm MLP of the Craftax script (pqn_craftax.py:33-62 with NORM_TYPE = layer_norm; C5 = 1345 ->
    4 x 1024 -> 17, BatchRenorm on the input) through the tiled bf16x3 MFMA GEMM kernels of csrc/pqn_bigmlp.hip: forward
    (+ eps-greedy), value_and_grad of both branches of _loss_fn (:277-312), clip + RAdam on the flat buffer."""
    packed = False

    def __init__(self, network, theta, config, lr_steps, grad_hook):
        from .qnet import BigMlpKernelLayout, BigMlpTrainer
        self.net = network
        norm_input = (2 if network.renorm else 1) if network.norm_input else 0
        self.layout = BigMlpKernelLayout(network.obs_shape[0], network.hidden, network.layers, network.action_dim, norm_input)
        self.tr = BigMlpTrainer(self.layout, theta, config["LR"], config["MAX_GRAD_NORM"], lr_decay_steps=lr_steps)
        self.grad_hook = grad_hook
        if norm_input and grad_hook is not None:
            raise NotImplementedError("input BatchNorm / BatchRenorm with the envs of one seed sharded over ranks would need "
                                      "the batch moments all-reduced as well; use seed sharding (dist.partition_seeds)")

    def q_values(self, obs):
        return self.tr.forward(obs)[0]

    def act(self, obs, eps, key, action, qmax):
        self.tr.forward(obs, want_q=False, eps=eps, key=key, action=action, qmax=qmax)

    def max_q(self, obs, out):
        self.tr.forward(obs, want_q=False, qmax=out)

    def _step(self):
        if self.grad_hook is not None:
            self.grad_hook(self.tr.grad)
        self.tr.apply()

    def sgd_step(self, idx, obs_flat, act_flat, tgt_flat, loss_out, qv_out):
        self.tr.compute_grad(idx, obs_flat, act_flat, target=tgt_flat, loss_out=loss_out, qv_out=qv_out)
        self._step()

    def sgd_step_1step(self, idx, obs_all_flat, n_env, act_flat, rew_flat, done_flat, gamma, loss_out, qv_out):
        """`Q_LAMBDA: False` (pqn_craftax.py:287-304): next_obs of transition j is row j + n_env of the flattened [T+1][N]
        observation record; obs and next_obs form ONE batch inside pqn_bigmlp_grad."""
        self.tr.compute_grad(idx, obs_all_flat, act_flat, reward=rew_flat, done=done_flat, gamma=gamma, next_offset=n_env,
                             loss_out=loss_out, qv_out=qv_out)
        self._step()

    def theta_flax(self):
        return self.tr.theta_flax()

    def opt_state(self):
        from .networks import bn_module
        stats = {}
        if self.layout.norm_input:
            name = bn_module(self.net.renorm) + "_0"
            stats = {name + "/mean": self.tr.in_mean, name + "/var": self.tr.in_var}
            if self.net.renorm:
                stats[name + "/steps"] = self.tr.in_steps[0]     # [1] is kernel scratch
        return {"opt_count": self.tr.count, "opt_mu": self.tr.m, "opt_nu": self.tr.v, "kernel_layout": self.layout,
                "batch_stats": stats}


def _mlp_fits_fused(obs_dim: int, hidden: int, layers: int) -> bool:
    """LDS footprint of mlp_train_kernel (csrc/pqn_mlp.hip) must stay under 160 KB."""
    if hidden % 16 or hidden < 16 or hidden > 1024 or layers < 1 or layers > 4:
        return False
    floats = 16 * (obs_dim + 1) + (2 * layers + 2) * 16 * (hidden + 4) + layers * 16 + 32
    return floats * 4 <= 160 * 1024


def make_train(config: Dict[str, Any], device: Optional[str] = None, grad_hook: Optional[Callable] = None,
               metrics_hook: Optional[Callable] = None, script: str = "gymnax"):
    """Returns train(key).  `grad_hook(flat_grad)` (optional) runs between backward
    and the optimizer step -- the RCCL all-reduce of env-sharded mode plugs in here.
    `metrics_hook(values)` (optional; dist.allreduce_mean_scalars) turns the per-rank metric means of an update
    into means over all env shards (pqn_minatar.py:330-338 take them over ALL envs).
    script="craftax": the third twin, purejaxql/pqn_craftax.py:82-468 -- the env batched by
    OptimisticResetVecEnvWrapper(LogWrapper(env)) / BatchEnvWrapper(LogWrapper(env)) (:96-114), the BatchRenorm MLP
    (:33-62), the `Q_LAMBDA` switch of the loss (:277-304: with False, obs and next_obs go through the train-mode
    network as one batch and the 1-step target carries no gradient), done-weighted info means (:364-369)."""
    craftax = script == "craftax"
    if script not in ("gymnax", "craftax"):
        raise ValueError(f"unknown script variant {script!r}")
    lib = _lib.load()
    derive_config(config)
    dev = torch.device(device or "cuda")
    if dev.type != "cuda":
        raise RuntimeError("purejaxql_amd runs on the GPU only (no CPU fallback); got device=%s" % dev)
    # Everything a run enqueues goes to ONE created stream, never to the legacy NULL stream (PyTorch's default current stream).
    # Round 6 root cause of the round-5 fault (16 seeds x 4096 envs with evaluations died after ~16 updates): the HIP runtime
    # bundled in the PyTorch wheel (ROCm 7.0.51831 in torch/lib, the one a Python process loads) faults after ~25 replays of a
    # long hipGraph launched on the NULL stream when eager launches are queued behind the replay; on a created stream, or
    # with /opt/rocm's 7.2 runtime, the same program is clean (tools/repro/update_replay.cpp, profiles/r06_v2_null_stream_fault.txt).
    # config _WORK_STREAM=False keeps the caller's current stream (the regression test's negative control).
    work_stream = torch.cuda.Stream(dev) if config.get("_WORK_STREAM", True) else None
    if config.get("SEED_BATCH_BIT_IDENTICAL", False):
        # the evaluation rollouts (pqn_cnn_rollout / pqn_cnn_rollout_seeds) follow the same rule as the update's kernels: form from
        # the per-seed shape alone (library option, process-wide; the update itself carries the flag in pqn_update_args_t.reserved)
        _lib.set_option("pin_form", 1)

    env, env_params = make(config["ENV_NAME"], device=dev, **(config.get("ENV_KWARGS") or {}))
    kind = "cnn" if (len(env.obs_shape) == 3 and not craftax) else "mlp"
    if kind == "mlp":
        env = FlattenObservationWrapper(env)      # pqn_gymnax.py:93 (the Craftax symbolic observation is flat already)
    env = LogWrapper(env)                         # pqn_minatar.py:104 / pqn_craftax.py:100
    base_env = env
    while hasattr(base_env, "_env"):
        base_env = base_env._env
    test_env = None
    if craftax:                                   # pqn_craftax.py:101-114: the env is batched by a wrapper
        n_test = int(config.get("TEST_NUM_ENVS", 128))
        if config.get("USE_OPTIMISTIC_RESETS", False):
            ratio = int(config.get("OPTIMISTIC_RESET_RATIO", 16))
            log_env = env
            env = OptimisticResetVecEnvWrapper(log_env, num_envs=int(config["NUM_ENVS"]),
                                               reset_ratio=min(ratio, int(config["NUM_ENVS"])))
            test_env = OptimisticResetVecEnvWrapper(log_env, num_envs=n_test, reset_ratio=min(ratio, n_test))
        else:
            log_env = env
            env = BatchEnvWrapper(log_env, num_envs=int(config["NUM_ENVS"]))
            test_env = BatchEnvWrapper(log_env, num_envs=n_test)
        config["TEST_NUM_STEPS"] = int(config.get("TEST_NUM_STEPS", env_params.max_steps_in_episode))
    elif kind == "cnn":
        config["TEST_NUM_STEPS"] = env_params.max_steps_in_episode                      # pqn_minatar.py:105
    else:
        config["TEST_NUM_STEPS"] = config.get("TEST_NUM_STEPS", env_params.max_steps_in_episode)  # pqn_gymnax.py:95-97

    N, T = int(config["NUM_ENVS"]), int(config["NUM_STEPS"])
    NUM_UPDATES = int(config["NUM_UPDATES"])
    MB, EPOCHS = int(config["NUM_MINIBATCHES"]), int(config["NUM_EPOCHS"])
    B = (N * T) // MB
    A = env.action_space(env_params).n
    obs_shape = env.observation_space(env_params).shape
    gamma, lam = float(config["GAMMA"]), float(config["LAMBDA"])
    rew_scale = float(config.get("REW_SCALE", 1))
    test_on = bool(config.get("TEST_DURING_TRAINING", False))
    sp = _lib.stream_ptr
    q_lambda_loss = bool(config.get("Q_LAMBDA", False)) if craftax else True     # pqn_craftax.py:277
    backend = config.get("_BACKEND")
    from .qnet import bigmlp_supported
    # wide LayerNorm MLPs (the Craftax yaml: 4 x 1024, input BatchRenorm): tiled MFMA GEMM kernels, csrc/pqn_bigmlp.hip
    big_ok = (kind == "mlp" and config["NORM_TYPE"] == "layer_norm" and
              bigmlp_supported(obs_shape[0], int(config.get("HIDDEN_SIZE", 128)), int(config.get("NUM_LAYERS", 2)), A) and
              not (grad_hook is not None and config.get("NORM_INPUT", False)))
    if craftax:          # wrapper-batched env, 1-step loss: the wide-MLP kernels, else the torch-op network (hidden BatchRenorm,
        if backend not in (None, "torch", "fused_big"):   # narrow layers) -- over the HIP env / eps-greedy / RAdam kernels either way
            raise ValueError("the Craftax script variant runs the wide-MLP kernels or the torch-op network")
        if backend == "fused_big" and not big_ok:
            raise ValueError("_BACKEND=fused_big needs NORM_TYPE=layer_norm and a shape csrc/pqn_bigmlp.hip tiles")
        backend = backend or ("fused_big" if big_ok else "torch")
    if backend is None:  # fused kernels: LayerNorm networks; the CNN also needs 16 | minibatch
        plain_ln = config["NORM_TYPE"] == "layer_norm" and not config.get("NORM_INPUT", False)
        if kind == "cnn":
            backend = "fused" if (plain_ln and B % 16 == 0) else "torch"
        elif plain_ln and _mlp_fits_fused(obs_shape[0], int(config.get("HIDDEN_SIZE", 128)), int(config.get("NUM_LAYERS", 2))):
            backend = "fused"
        else:
            backend = "fused_big" if big_ok else "torch"
    packed = backend == "fused" and kind == "cnn"
    # shape limits of the whole-update C++ enqueue (pqn_cnn_update / pqn_mlp_update: the schedule kernel derives the
    # T + EPOCHS keys of an update with one 1024-thread block) and of seed batching (25 index bits in the shuffle
    # keys); configs outside them run the per-kernel Python loop / per-seed streams instead of failing at run time
    driver_shape_ok = T + EPOCHS <= 1024
    seeds_shape_ok = driver_shape_ok and N % 16 == 0 and T * N <= (1 << 25)

    def env_step_into(key, words, action, obs_out, bits_out, r, d, disc, rer, rel, ts):
        out = _lib.StepOut(obs=_lib.ptr(obs_out), obs_bits=_lib.ptr(bits_out), reward=_lib.ptr(r), done=_lib.ptr(d),
                           discount=_lib.ptr(disc), returned_episode_returns=_lib.ptr(rer),
                           returned_episode_lengths=_lib.ptr(rel), timestep=_lib.ptr(ts))
        _lib.check(lib.pqn_env_step(base_env.env_id, words.shape[1], key, _lib.ptr(words), _lib.ptr(words),
                                    _lib.ptr(action), C.byref(out), sp()), "pqn_env_step")

    def fused_eval(layout, theta_k, k, n_t, steps, buf):
        """The evaluation scan (pqn_minatar.py:380-401) as ONE persistent launch (pqn_cnn_rollout with
        eps = EPS_TEST, nothing recorded but the info arrays), then the masked means (:403-412)."""
        if "keys" not in buf:
            z = lambda dt: torch.empty((steps, n_t), dtype=dt, device=dev)
            buf.update(keys=torch.empty(steps, dtype=torch.int64, device=dev),
                       eps=torch.full((1,), float(config["EPS_TEST"]), dtype=torch.float32, device=dev),
                       done=z(torch.uint8), discount=z(torch.float32), rer=z(torch.float32),
                       rel=z(torch.int32), ts=z(torch.int32))
        b = buf
        (_o, bits), state = env.reset(_lib.fold_in(k, 0), env_params, n_t, want_obs=False, want_bits=True)
        _lib.check(lib.pqn_fold_in_range(k, 1, steps, _lib.ptr(b["keys"]), sp()), "pqn_fold_in_range")
        rec = _lib.StepOut(done=_lib.ptr(b["done"]), discount=_lib.ptr(b["discount"]),
                           returned_episode_returns=_lib.ptr(b["rer"]), returned_episode_lengths=_lib.ptr(b["rel"]),
                           timestep=_lib.ptr(b["ts"]))
        _lib.check(lib.pqn_cnn_rollout(base_env.env_id, C.byref(layout.struct), n_t, steps,
                                       _lib.ptr(state.words), _lib.ptr(bits), 0, _lib.ptr(theta_k),
                                       C.byref(rec), None, None, None, _lib.ptr(b["eps"]), _lib.ptr(b["keys"]),
                                       1.0, sp()), "pqn_cnn_rollout")
        dm = b["done"].to(torch.float64)
        cnt = dm.sum()
        vals = {"discount": b["discount"], "returned_episode_returns": b["rer"], "returned_episode_lengths": b["rel"],
                "timestep": b["ts"], "returned_episode": b["done"]}
        # nanmean(where(returned_episode, x, nan)) (:403-412)
        return {kk: ((vals[kk].to(torch.float64) * dm).sum() / cnt).to(torch.float32) for kk in INFO_KEYS}

    def flat_eval(act, k, n_t, steps, buf):
        """Flat-observation evaluation scan (pqn_gymnax.py:362-404): per step one forward(+eps-greedy) launch
        (`act(obs, eps, key, action_out, qmax_out)`) and one env.step launch writing straight into row t of the
        [steps, n] info record; the masked means are taken once at the end."""
        if "done" not in buf:
            z = lambda dt: torch.empty((steps, n_t), dtype=dt, device=dev)
            buf.update(done=z(torch.uint8), discount=z(torch.float32), rer=z(torch.float32),
                       rel=z(torch.int32), ts=z(torch.int32), reward=torch.empty(n_t, dtype=torch.float32, device=dev),
                       action=torch.empty(n_t, dtype=torch.int32, device=dev),
                       qm=torch.empty(n_t, dtype=torch.float32, device=dev),
                       obs=torch.empty((2, n_t, *obs_shape), dtype=torch.float32, device=dev))
        b = buf
        obs0, state = env.reset(_lib.fold_in(k, 0), env_params, n_t)
        b["obs"][0].copy_(obs0)
        words_t = state.words
        for t in range(steps):
            sk = _lib.fold_in(k, 1 + t)
            cur, nxt = b["obs"][t & 1], b["obs"][(t + 1) & 1]
            act(cur, config["EPS_TEST"], sk, b["action"], b["qm"])
            env_step_into(sk, words_t, b["action"], nxt, None, b["reward"], b["done"][t], b["discount"][t],
                          b["rer"][t], b["rel"][t], b["ts"][t])
        dm = b["done"].to(torch.float64)
        cnt = dm.sum()
        vals = {"discount": b["discount"], "returned_episode_returns": b["rer"], "returned_episode_lengths": b["rel"],
                "timestep": b["ts"], "returned_episode": b["done"]}
        # nanmean(where(returned_episode, x, nan)) (:403-412)
        return {kk: ((vals[kk].to(torch.float64) * dm).sum() / cnt).to(torch.float32) for kk in INFO_KEYS}

    def wrapped_eval(act, k, steps):
        """get_test_metrics of the Craftax script (pqn_craftax.py:399-437): the wrapper-batched test env, TEST_NUM_STEPS
        steps under eps = EPS_TEST, done-weighted info means."""
        obs, state = test_env.reset(_lib.fold_in(k, 0), env_params)
        n_t = test_env.num_envs
        action = torch.empty(n_t, dtype=torch.int32, device=dev)
        qm = torch.empty(n_t, dtype=torch.float32, device=dev)
        sums = {kk: torch.zeros((), dtype=torch.float64, device=dev) for kk in INFO_KEYS}
        cnt = torch.zeros((), dtype=torch.float64, device=dev)
        for t in range(steps):
            sk = _lib.fold_in(k, 1 + t)
            act(obs, config["EPS_TEST"], sk, action, qm)
            obs, state, _r, d, info = test_env.step(sk, state, action, env_params, inplace=True)
            dm = d.to(torch.float64)
            cnt += dm.sum()
            for kk in INFO_KEYS:
                sums[kk] += (info[kk].to(torch.float64) * dm).sum()
        return {kk: (sums[kk] / cnt).to(torch.float32) for kk in INFO_KEYS}

    def _make_runner(rng: int):
        """Builds the per-seed training state and returns (update, finish): update(u) runs ONE
        PQN update (rollout + targets + epochs); finish() returns train()'s result dict."""
        K = int(rng) & 0xFFFFFFFFFFFFFFFF
        K_init, K_reset, K_test, K_roll, K_shuf = (_lib.fold_in(K, i) for i in range(5))
        shard = config.get("_ENV_SHARD")  # (rank, world): envs of ONE seed split over ranks
        if shard is not None:             # same init on every rank, rank-distinct env/shuffle streams
            K_reset, K_test, K_roll, K_shuf = (_lib.fold_in(k, 1000 + int(shard[0])) for k in
                                               (K_reset, K_test, K_roll, K_shuf))

        eps_scheduler = linear_schedule(config["EPS_START"], config["EPS_FINISH"],
                                        config["EPS_DECAY"] * config["NUM_UPDATES_DECAY"])
        lr_steps = (config["NUM_UPDATES_DECAY"] * MB * EPOCHS) if config.get("LR_LINEAR_DECAY", False) else 0.0

        # INIT NETWORK AND OPTIMIZER (pqn_minatar.py:150-173)
        network = QNetwork(kind, obs_shape, A, norm_type=config["NORM_TYPE"],
                           norm_input=config.get("NORM_INPUT", False),
                           hidden_size=config.get("HIDDEN_SIZE", 128), num_layers=config.get("NUM_LAYERS", 2),
                           device=dev, renorm=craftax)
        theta = config.get("_INIT_PARAMS")
        theta = network.init(K_init) if theta is None else theta.to(dev, torch.float32).clone()
        if packed:
            policy = _FusedCnnPolicy(network, theta, config, lr_steps, grad_hook, B)
        elif backend == "fused":
            policy = _FusedMlpPolicy(network, theta, config, lr_steps, grad_hook, B)
        elif backend == "fused_big":
            policy = _FusedBigMlpPolicy(network, theta, config, lr_steps, grad_hook)
        else:
            policy = _TorchPolicy(network, theta, config, lr_steps, grad_hook)
        counters = {"timesteps": 0, "n_updates": 0, "grad_steps": 0}

        # EVAL (pqn_minatar.py:371-413)
        test_runs = [0]

        eval_buf = {}

        def fused_test_metrics(k, n_t, steps):
            return fused_eval(policy.layout, policy.tr.theta, k, n_t, steps, eval_buf)

        def get_test_metrics():
            if not test_on:
                return None
            k = _lib.fold_in(K_test, test_runs[0])
            test_runs[0] += 1
            n_t, steps = int(config["TEST_NUM_ENVS"]), int(config["TEST_NUM_STEPS"])
            if packed:
                return fused_test_metrics(k, n_t, steps)
            if craftax:
                return wrapped_eval(policy.act, k, steps)
            return flat_eval(policy.act, k, n_t, steps, eval_buf)

        tm_box = [get_test_metrics()]

        # reset exploration envs (pqn_minatar.py:418-419)
        ro = _Rollout(T, N, obs_shape, base_env.obs_words if packed else 0, dev)
        obuf = ro.bits if packed else ro.obs
        if craftax:
            o0, state = env.reset(K_reset, env_params)                                    # pqn_craftax.py:446-447
        else:
            o0, state = env.reset(K_reset, env_params, N, want_obs=not packed, want_bits=packed)
        words = state.words
        obuf[0].copy_(o0[1] if packed else o0)

        names = ["env_step", "update_steps", "grad_steps", "td_loss", "qvals"] + list(INFO_KEYS)
        if kind == "cnn":
            names.insert(2, "env_frame")
        # info["Achievements/<name>"] = done * unlocked * 100 of the Craftax envs (pqn_craftax.py:364-369,384-387): logged only
        # with LOG_ACHIEVEMENTS, from the achievement mask the step kernel emits for the episodes that end
        ach_names = list(base_env.achievement_names) if (craftax and config.get("LOG_ACHIEVEMENTS", False)) else []
        ach_buf = torch.zeros((T, N), dtype=torch.int32, device=dev) if ach_names else None
        names += [f"Achievements/{a}" for a in ach_names]
        if test_on:
            names += [f"test/{k}" for k in INFO_KEYS]
        metrics = {k: torch.zeros(NUM_UPDATES, dtype=torch.float32, device=dev) for k in names}
        n_mb_total = MB * EPOCHS
        loss_buf = torch.zeros(n_mb_total, dtype=torch.float32, device=dev)
        qv_buf = torch.zeros(n_mb_total, dtype=torch.float32, device=dev)
        test_period = int(NUM_UPDATES * config["TEST_INTERVAL"]) if test_on else 0
        flat_shape = (T * N, base_env.obs_words) if packed else (T * N, *obs_shape)

        # whole-update C++ enqueue (+ hipGraph replay): the fused CNN / MLP paths without a gradient hook
        driver = None
        if backend == "fused" and grad_hook is None and config.get("_DRIVER", True) and driver_shape_ok:
            from .qnet import UpdateDriver
            dcfg = {"gamma": gamma, "lam": lam, "rew_scale": rew_scale, "eps_start": config["EPS_START"],
                    "eps_finish": config["EPS_FINISH"], "eps_decay_steps": config["EPS_DECAY"] * config["NUM_UPDATES_DECAY"]}
            driver = UpdateDriver(base_env.env_id, N, T, MB, EPOCHS, base_env.obs_words, dcfg, (K_roll, K_shuf),
                                  policy.tr, ro, words, NUM_UPDATES, use_graph=config.get("_GRAPH", True),
                                  pin_form=bool(config.get("SEED_BATCH_BIT_IDENTICAL", False)))
        elif backend == "fused_big" and grad_hook is None and metrics_hook is None and config.get("_DRIVER", True) and driver_shape_ok:
            # the Craftax script's loop (wrapper-batched env, wide MLP) from one C call, replayed as a hipGraph (not with a
            # cross-shard metrics hook: the device computes the done-weighted RATIOS of this shard, the hook needs the sums)
            from .qnet import BigMlpUpdateDriver
            dcfg = {"gamma": gamma, "lam": lam, "rew_scale": rew_scale, "eps_start": config["EPS_START"],
                    "eps_finish": config["EPS_FINISH"], "eps_decay_steps": config["EPS_DECAY"] * config["NUM_UPDATES_DECAY"]}
            ratio = int(env.reset_ratio) if isinstance(env, OptimisticResetVecEnvWrapper) else 0
            driver = BigMlpUpdateDriver(base_env.env_id, N, T, MB, EPOCHS, dcfg, (K_roll, K_shuf), policy.tr, ro, words,
                                        NUM_UPDATES, reset_ratio=ratio, q_lambda=q_lambda_loss, done_weighted_info=craftax,
                                        use_graph=config.get("_GRAPH", True), log_achievements=bool(ach_names))
        elif packed and grad_hook is not None and config.get("_DRIVER", True) and driver_shape_ok:
            # envs of one seed sharded over ranks: the same C++ enqueue, split at the gradient / optimizer boundary
            from .qnet import EnvShardDriver
            dcfg = {"gamma": gamma, "lam": lam, "rew_scale": rew_scale, "eps_start": config["EPS_START"],
                    "eps_finish": config["EPS_FINISH"], "eps_decay_steps": config["EPS_DECAY"] * config["NUM_UPDATES_DECAY"]}
            driver = EnvShardDriver(base_env.env_id, N, T, MB, EPOCHS, base_env.obs_words, dcfg, (K_roll, K_shuf),
                                    policy.tr, ro, words, NUM_UPDATES, use_graph=config.get("_GRAPH", True),
                                    grad_hook=grad_hook)
        test_rows = _TestRows(NUM_UPDATES) if test_on else None
        if test_on:
            test_rows.note(0, torch.stack([tm_box[0][k] for k in INFO_KEYS]))
        shard_world = int(shard[1]) if shard is not None else 1

        def share_metrics_row(row):
            """Env-sharded mode: the means of an update are over ALL env shards, the step counts over all envs."""
            from .qnet import METRIC_NAMES
            i0 = METRIC_NAMES.index("td_loss")
            _sync_behind_graph(driver)
            if metrics_hook is not None:
                row[i0:] = metrics_hook(row[i0:].clone())
            for name in ("env_step", "env_frame"):
                row[METRIC_NAMES.index(name)] *= shard_world

        forms = {}    # kernel forms of this run's launches (asked of the library after the first, eager, enqueue)

        def driver_update(u: int):
            if u != driver.calls:
                raise RuntimeError(f"update({u}) out of order: the device clock is at {driver.calls}")
            driver.update()
            if not forms and packed:
                forms.update(zip(("train", "rollout"), _lib.last_kernel_form()))
                forms["matmul_dtype"] = policy.layout.mode_name   # what MATMUL_DTYPE (auto) resolved to
            if shard_world > 1 or metrics_hook is not None:
                share_metrics_row(driver.metrics[u])     # (waits for the stream first: nothing is enqueued behind a replay in flight)
            if grad_hook is not None and hasattr(grad_hook, "poll"):
                grad_hook.poll()       # in-graph peer all-reduce: a time-out surfaces within an update or two, not at finish()
            counters["timesteps"] += T * N
            counters["n_updates"] += 1
            counters["grad_steps"] += MB * EPOCHS
            if test_on and test_period > 0 and counters["n_updates"] % test_period == 0:
                _sync_behind_graph(driver)
                tm_box[0] = get_test_metrics()
                test_rows.note(u, torch.stack([tm_box[0][k] for k in INFO_KEYS]))
            cb = config.get("_CALLBACK")
            # the Craftax script logs every WANDB_LOG_INTERVAL-th update only (pqn_craftax.py:394-397)
            if cb is not None and (not craftax or counters["n_updates"] % int(config.get("WANDB_LOG_INTERVAL", 128)) == 0):
                row = driver.metrics[u].tolist()   # synchronises: logging is opt-in
                from .qnet import METRIC_NAMES
                m = dict(zip(METRIC_NAMES, row))
                if kind != "cnn":
                    m.pop("env_frame", None)
                if test_on:
                    m.update({f"test/{k}": float(v) for k, v in tm_box[0].items()})
                _log_row(config, K, u, m)

        def update(u: int):
            if driver is not None:
                return driver_update(u)
            # SAMPLE PHASE (_step_env, pqn_minatar.py:181-220)
            eps = eps_scheduler(counters["n_updates"])
            for t in range(T):
                sk = _lib.fold_in(K_roll, u * T + t)
                policy.act(obuf[t], eps, sk, ro.action[t], ro.qmax[t])
                if craftax:   # the wrapper-batched env (pqn_craftax.py:202-204)
                    o_n, _st, r_n, d_n, info_n = env.step(sk, state, ro.action[t], env_params, inplace=True)
                    ro.obs[t + 1].copy_(o_n)
                    ro.reward[t].copy_(r_n)
                    ro.done[t].copy_(d_n.view(torch.uint8))
                    ro.discount[t].copy_(info_n["discount"])
                    ro.rer[t].copy_(info_n["returned_episode_returns"])
                    ro.rel[t].copy_(info_n["returned_episode_lengths"])
                    ro.ts[t].copy_(info_n["timestep"])
                    if ach_names:
                        ach_buf[t].copy_(info_n["achievements"])
                else:
                    env_step_into(sk, words, ro.action[t], None if packed else ro.obs[t + 1],
                                  ro.bits[t + 1] if packed else None, ro.reward[t], ro.done[t], ro.discount[t],
                                  ro.rer[t], ro.rel[t], ro.ts[t])
            counters["timesteps"] += T * N
            info_sums = None
            if craftax:   # (x * returned_episode).sum() / returned_episode.sum()  (pqn_craftax.py:364-369)
                dm = ro.done.to(torch.float64)
                cnt = dm.sum()
                # numerators and the common denominator separately: with the envs sharded over ranks the reference's ratio is
                # sum-over-all-envs / count-over-all-envs, not the mean of the shards' ratios (a shard without a finished
                # episode would contribute 0 / 0)
                info_sums = {kk: (vv.to(torch.float64) * dm).sum() for kk, vv in
                             (("discount", ro.discount), ("returned_episode_returns", ro.rer),
                              ("returned_episode_lengths", ro.rel), ("timestep", ro.ts), ("returned_episode", ro.done))}
                for k_a, a_name in enumerate(ach_names):   # x = done * unlocked * 100, then the same done-weighted mean
                    x = ((ach_buf >> k_a) & 1).to(torch.float64) * 100.0 * dm
                    info_sums[f"Achievements/{a_name}"] = (x * dm).sum()
                info_means = {kk: (vv / cnt).to(torch.float32) for kk, vv in info_sums.items()}
                info_sums["_count"] = cnt
            else:
                info_means = {
                    "discount": ro.discount.mean(), "returned_episode_returns": ro.rer.mean(),
                    "returned_episode_lengths": ro.rel.to(torch.float32).mean(),
                    "timestep": ro.ts.to(torch.float32).mean(), "returned_episode": ro.done.to(torch.float32).mean(),
                }
            if rew_scale != 1.0:
                ro.reward.mul_(rew_scale)      # REW_SCALE*reward (:205); LogWrapper saw the raw reward

            # Q(lambda) TARGETS (pqn_minatar.py:227-260).  The Craftax script traces them with `Q_LAMBDA: False` too
            # (pqn_craftax.py:231-261) but nothing consumes them there, so XLA removes the bootstrap forward and the scan
            # as dead code; they are skipped here for the same reason (no random draw, no state involved).
            if q_lambda_loss:
                policy.max_q(obuf[T], ro.last_q)
                ops.q_lambda(ro.reward, ro.done, ro.qmax, ro.last_q, gamma, lam, quirk=True, target=ro.target)

            # NETWORKS UPDATE (pqn_minatar.py:263-327): one shared permutation per epoch (:299-315),
            # consumed as a gather index -- the shuffled copies are never materialised
            obs_flat = obuf[:T].reshape(flat_shape)
            act_flat = ro.action.reshape(-1)
            tgt_flat = ro.target.reshape(-1)
            i_mb = 0
            for ep in range(EPOCHS):
                perm = ops.shuffle_permutation(_lib.fold_in(K_shuf, u * EPOCHS + ep), T * N, dev)
                for mb in range(MB):
                    if q_lambda_loss:
                        policy.sgd_step(perm[mb * B:(mb + 1) * B], obs_flat, act_flat, tgt_flat,
                                        loss_buf[i_mb:i_mb + 1], qv_buf[i_mb:i_mb + 1])
                    else:
                        policy.sgd_step_1step(perm[mb * B:(mb + 1) * B], obuf.reshape((T + 1) * N, *obs_shape), N, act_flat,
                                              ro.reward.reshape(-1), ro.done.reshape(-1), gamma,
                                              loss_buf[i_mb:i_mb + 1], qv_buf[i_mb:i_mb + 1])
                    counters["grad_steps"] += 1
                    i_mb += 1
            obuf[0].copy_(obuf[T])  # carry last_obs into the next update

            counters["n_updates"] += 1
            m = {"env_step": counters["timesteps"], "update_steps": counters["n_updates"],
                 "grad_steps": counters["grad_steps"], "td_loss": loss_buf.mean(), "qvals": qv_buf.mean()}
            if kind == "cnn":
                m["env_frame"] = counters["timesteps"] * obs_shape[-1]
            m.update(info_means)
            if metrics_hook is not None or shard_world > 1:   # env-sharded mode: means over all shards, counts over all envs
                if info_sums is not None:   # done-weighted means (Craftax script): all-reduce numerators and the denominator
                    sum_keys = [k for k in info_sums if k != "_count"]
                    mean_keys = ["td_loss", "qvals"]
                    vals = torch.stack([torch.as_tensor(m[k], dtype=torch.float32, device=dev) for k in mean_keys] +
                                       [info_sums[k].to(torch.float32) for k in sum_keys] + [info_sums["_count"].to(torch.float32)])
                    if metrics_hook is not None:
                        vals = metrics_hook(vals)       # mean over the shards of every entry: the common factor 1 / world cancels
                    m.update({k: vals[i] for i, k in enumerate(mean_keys)})
                    m.update({k: vals[len(mean_keys) + i] / vals[-1] for i, k in enumerate(sum_keys)})
                else:
                    mean_keys = ["td_loss", "qvals"] + list(INFO_KEYS)
                    vals = torch.stack([torch.as_tensor(m[k], dtype=torch.float32, device=dev) for k in mean_keys])
                    if metrics_hook is not None:
                        vals = metrics_hook(vals)
                    m.update({k: vals[i] for i, k in enumerate(mean_keys)})
                for k in ("env_step", "env_frame"):
                    if k in m:
                        m[k] = m[k] * shard_world
            if test_on:
                if test_period > 0 and counters["n_updates"] % test_period == 0:
                    tm_box[0] = get_test_metrics()
                m.update({f"test/{k}": v for k, v in tm_box[0].items()})
            for k, v in m.items():
                metrics[k][u] = v
            cb = config.get("_CALLBACK")
            # the Craftax script logs every WANDB_LOG_INTERVAL-th update only (pqn_craftax.py:394-397)
            if cb is not None and (not craftax or counters["n_updates"] % int(config.get("WANDB_LOG_INTERVAL", 128)) == 0):
                _log_row(config, K, u, {k: v for k, v in m.items() if config.get("LOG_ACHIEVEMENTS", False) or "achievement" not in k.lower()}
                         if craftax else m)

        def finish():
            if grad_hook is not None and hasattr(grad_hook, "check"):
                grad_hook.check()      # a peer that never arrived inside an in-graph collective: raise, do not return garbage
            if driver is not None:
                from .qnet import METRIC_NAMES
                for j, name in enumerate(METRIC_NAMES):
                    if name in metrics:
                        metrics[name] = driver.metrics[:NUM_UPDATES, j].to(torch.float32)
                if ach_names and getattr(driver, "ach_metrics", None) is not None:   # LOG_ACHIEVEMENTS columns of the whole-update enqueue
                    for k_a, a_name in enumerate(ach_names):
                        metrics[f"Achievements/{a_name}"] = driver.ach_metrics[:NUM_UPDATES, k_a].to(torch.float32)
                if test_on:
                    rows = test_rows.rows()
                    for j, k in enumerate(INFO_KEYS):
                        metrics[f"test/{k}"] = rows[:, j]
            theta_f = policy.theta_flax()
            runner_state = RunnerState({"params": network.views(theta_f), "theta": theta_f, "env_state": words,
                            "last_obs": obuf[0], "test_metrics": tm_box[0], "network": network, "backend": backend,
                            "driver": None if driver is None else ("graph" if driver.graph is not None else "eager"),
                            "driver_graph_error": None if driver is None else driver.graph_error, "rng": K,
                            "driver_graphs": None if driver is None else (1 if getattr(driver, "whole", None) is not None else
                                                                          len(getattr(driver, "graphs", None) or [1])),
                            "allreduce": getattr(grad_hook, "mode", None) if grad_hook is not None else None,
                            "kernel_forms": dict(forms), **policy.opt_state(), **counters})
            return {"runner_state": runner_state, "metrics": metrics}

        update.driver = driver   # bench.py switches graph replay off for its HIP-event timing pass
        return update, finish

    def seed_groups_for(S: int) -> int:
        """How many seed groups a batch of S seeds is cut into (config SEED_GROUPS, else PQN_SEED_GROUPS; default 1).
        With G > 1 one group's HBM-bound optimizer tail runs on a second stream under the next group's compute-bound
        training kernel (qnet.SeedGroupsDriver).  Opt-in: measured SLOWER on the bench workload (2 x 8 seeds: 51.3 ms per
        update as one two-branch hipGraph, 48.1 ms as eager streams, against 44.1 ms for one 16-seed chain of launches;
        profiles/r04_v0_seed_groups_ab.txt) -- the training kernel's workgroups hold every CU's LDS and registers for
        ~54 us each, so the tail kernels wait for CUs instead of running beside it, and their traffic evicts the weight
        planes from the L2s."""
        g = int(config.get("SEED_GROUPS", 0) or 0)
        if g <= 0:
            g = int(os.environ.get("PQN_SEED_GROUPS", "0") or 0)
        if not packed or g < 1 or S % g != 0 or g > 8:
            g = 1
        return g

    make_runner = _streamed_factory(work_stream, _make_runner)

    def make_grouped_runner(rngs: List[int], G: int):
        """make_batch_runner over G seed groups advanced together (pqn_cnn_update_seed_groups): group g holds the seeds
        rngs[g*S/G : (g+1)*S/G] in its own stacked buffers; update / finish behave as make_batch_runner's."""
        from .qnet import SeedGroupsDriver
        per = len(rngs) // G
        subs = [make_batch_runner(rngs[g * per:(g + 1) * per], groups=1) for g in range(G)]
        gd = SeedGroupsDriver([u.driver for u, _f in subs], tail=str(config.get("_SEED_GROUPS_TAIL", os.environ.get("PQN_SEED_GROUPS_TAIL", "graph"))))

        def update(u: int):
            if u != gd.calls:
                raise RuntimeError(f"update({u}) out of order: the device clock is at {gd.calls}")
            gd.update()
            for upd, _f in subs:
                upd(u, enqueue=False)

        def finish():
            outs = []
            for _u, fin in subs:
                outs.extend(fin())
            for o in outs:
                o["runner_state"].entries["seed_batch"] = len(rngs)
                o["runner_state"].entries["seed_groups"] = G
            return outs

        update.driver = gd
        return update, finish

    def _make_batch_runner(rngs: List[int], groups: Optional[int] = None):
        """jax.vmap(train)(rngs) inside the launches: all seeds advance in the same kernels (grid.y = seed,
        pqn_cnn_update_seeds), one hipGraph replay per update for all of them.  Same key schedule, same
        kernels and summation orders as make_runner, so every seed's result is bit-identical to its solo run (one exception:
        f32 mode, minibatches <= 256 samples, when only one of the two launches is small enough for the K-split training
        kernels -- option t1_ksplit_tiles; they then agree to f32 summation order).
        Returns (update, finish); finish() -> list of per-seed result dicts."""
        from .qnet import METRIC_NAMES, CnnKernelLayout, MlpKernelLayout, SeedsUpdateDriver, matmul_mode, mlp_forward
        S = len(rngs)
        if not (backend == "fused" and grad_hook is None and seeds_shape_ok and 1 <= S <= 128):
            raise RuntimeError("seed batching needs a fused path, no gradient hook, NUM_ENVS % 16 == 0, <= 128 seeds")
        G = seed_groups_for(S) if groups is None else int(groups)
        if G > 1:
            return make_grouped_runner(rngs, G)
        if packed:
            layout = CnnKernelLayout(obs_shape[-1], A,
                                     matmul_f16=matmul_mode(config.get("MATMUL_DTYPE", "auto"),
                                                            config["NUM_ENVS"] * config["NUM_STEPS"] // config["NUM_MINIBATCHES"]))
        else:
            layout = MlpKernelLayout(obs_shape[0], int(config.get("HIDDEN_SIZE", 128)), int(config.get("NUM_LAYERS", 2)), A)
        Ks = []
        for rng in rngs:
            K = int(rng) & 0xFFFFFFFFFFFFFFFF
            Ks.append(tuple(_lib.fold_in(K, i) for i in range(5)))     # K_init, K_reset, K_test, K_roll, K_shuf
        lr_steps = (config["NUM_UPDATES_DECAY"] * MB * EPOCHS) if config.get("LR_LINEAR_DECAY", False) else 0.0
        network = QNetwork(kind, obs_shape, A, norm_type=config["NORM_TYPE"], norm_input=False,
                           hidden_size=config.get("HIDDEN_SIZE", 128), num_layers=config.get("NUM_LAYERS", 2), device=dev)
        ro = _Rollout(T, S * N, obs_shape, base_env.obs_words if packed else 0, dev)
        words = None
        dcfg = {"gamma": gamma, "lam": lam, "rew_scale": rew_scale, "eps_start": config["EPS_START"],
                "eps_finish": config["EPS_FINISH"], "eps_decay_steps": config["EPS_DECAY"] * config["NUM_UPDATES_DECAY"]}
        drv = None
        theta_init = config.get("_INIT_PARAMS")
        for s, (K_init, K_reset, _kt, _kr, _ks) in enumerate(Ks):
            o0, st = env.reset(K_reset, env_params, N, want_obs=not packed, want_bits=packed)   # (:418-419)
            if words is None:
                words = torch.empty((st.words.shape[0], S * N), dtype=st.words.dtype, device=dev)
                drv = SeedsUpdateDriver(layout, base_env.env_id, S, N, T, MB, EPOCHS, base_env.obs_words, dcfg,
                                        [k[3] for k in Ks], [k[4] for k in Ks], config["LR"], 1e-20, lr_steps,
                                        config["MAX_GRAD_NORM"], ro, words, NUM_UPDATES, dev,
                                        use_graph=config.get("_GRAPH", True),
                                        pin_form=bool(config.get("SEED_BATCH_BIT_IDENTICAL", False)))
            words[:, s * N:(s + 1) * N] = st.words
            (ro.bits if packed else ro.obs)[0, s * N:(s + 1) * N] = o0[1] if packed else o0
            th = network.init(K_init) if theta_init is None else theta_init.to(dev, torch.float32)   # (:150-173)
            drv.set_params(s, th)
        eval_buf = {}
        runs = [0]

        def test_all():
            """get_test_metrics of every seed (pqn_minatar.py:371-413) in ONE persistent launch: S x TEST_NUM_ENVS
            envs, seed s greedy under its own parameters and keys (pqn_cnn_rollout_seeds)."""
            if not test_on:
                return [None] * S
            n_t, steps = int(config["TEST_NUM_ENVS"]), int(config["TEST_NUM_STEPS"])
            if not packed:      # flat-observation path: the per-seed scan of make_runner, on that seed's parameter slice
                def act_of(s):
                    return lambda obs, eps, key, action, qm: mlp_forward(layout, obs, drv.theta_k(s), want_q=False, eps=eps,
                                                                         key=key, action=action, qmax=qm)
                out = [flat_eval(act_of(s), _lib.fold_in(Ks[s][2], runs[0]), n_t, steps, eval_buf.setdefault(("solo", s), {}))
                       for s in range(S)]
                runs[0] += 1
                return out
            if n_t % 16 != 0:   # ragged test batch: one launch per seed
                out = [fused_eval(layout, drv.theta_k(s), _lib.fold_in(Ks[s][2], runs[0]), n_t, steps,
                                  eval_buf.setdefault(("solo", s), {})) for s in range(S)]
                runs[0] += 1
                return out
            if "keys" not in eval_buf:
                z = lambda dt: torch.empty((steps, S * n_t), dtype=dt, device=dev)
                eval_buf.update(keys=torch.empty((S, steps), dtype=torch.int64, device=dev),
                                eps=torch.full((1,), float(config["EPS_TEST"]), dtype=torch.float32, device=dev),
                                done=z(torch.uint8), discount=z(torch.float32), rer=z(torch.float32),
                                rel=z(torch.int32), ts=z(torch.int32), words=None,
                                bits=torch.empty((S * n_t, base_env.obs_words), dtype=torch.int32, device=dev))
            b = eval_buf
            for s in range(S):
                k = _lib.fold_in(Ks[s][2], runs[0])
                (_o, bits), state = env.reset(_lib.fold_in(k, 0), env_params, n_t, want_obs=False, want_bits=True)
                if b["words"] is None:
                    b["words"] = torch.empty((state.words.shape[0], S * n_t), dtype=state.words.dtype, device=dev)
                b["words"][:, s * n_t:(s + 1) * n_t] = state.words
                b["bits"][s * n_t:(s + 1) * n_t] = bits
                _lib.check(lib.pqn_fold_in_range(k, 1, steps, _lib.ptr(b["keys"][s]), sp()), "pqn_fold_in_range")
            runs[0] += 1
            rec = _lib.StepOut(done=_lib.ptr(b["done"]), discount=_lib.ptr(b["discount"]),
                               returned_episode_returns=_lib.ptr(b["rer"]), returned_episode_lengths=_lib.ptr(b["rel"]),
                               timestep=_lib.ptr(b["ts"]))
            _lib.check(lib.pqn_cnn_rollout_seeds(base_env.env_id, C.byref(layout.struct), S, n_t, steps, _lib.ptr(b["words"]),
                                                 _lib.ptr(b["bits"]), 0, _lib.ptr(drv.theta), drv.stride, C.byref(rec), None,
                                                 None, None, _lib.ptr(b["eps"]), _lib.ptr(b["keys"]), steps, 1.0, sp()),
                       "pqn_cnn_rollout_seeds")
            dm = b["done"].to(torch.float64).view(steps, S, n_t)
            cnt = dm.sum(dim=(0, 2))
            vals = {"discount": b["discount"], "returned_episode_returns": b["rer"], "returned_episode_lengths": b["rel"],
                    "timestep": b["ts"], "returned_episode": b["done"]}
            # nanmean(where(returned_episode, x, nan)) per seed (:403-412)
            means = {kk: ((vals[kk].to(torch.float64).view(steps, S, n_t) * dm).sum(dim=(0, 2)) / cnt).to(torch.float32)
                     for kk in INFO_KEYS}
            return [{kk: means[kk][s] for kk in INFO_KEYS} for s in range(S)]

        tm_box = [test_all()]
        test_period = int(NUM_UPDATES * config["TEST_INTERVAL"]) if test_on else 0
        stack_tm = lambda tm: torch.stack([torch.stack([tm[s][k] for k in INFO_KEYS]) for s in range(S)])   # [S, len(INFO_KEYS)]
        test_rows = _TestRows(NUM_UPDATES) if test_on else None
        if test_on:
            test_rows.note(0, stack_tm(tm_box[0]))
        counters = {"timesteps": 0, "n_updates": 0, "grad_steps": 0}
        forms = {}    # which kernel forms the library took for this batch (asked after the first, eager, enqueue)

        def update(u: int, enqueue: bool = True):
            if enqueue:     # (False: a SeedGroupsDriver already advanced this group's driver)
                if u != drv.calls:
                    raise RuntimeError(f"update({u}) out of order: the device clock is at {drv.calls}")
                drv.update()
            if not forms:
                forms.update(zip(("train", "rollout"), _lib.last_kernel_form()))
                if packed:
                    forms["matmul_dtype"] = layout.mode_name
            counters["timesteps"] += T * N
            counters["n_updates"] += 1
            counters["grad_steps"] += MB * EPOCHS
            if test_on and test_period > 0 and counters["n_updates"] % test_period == 0:
                _sync_behind_graph(drv)
                tm_box[0] = test_all()
                test_rows.note(u, stack_tm(tm_box[0]))

        def finish():
            names = ["env_step", "update_steps", "grad_steps", "td_loss", "qvals"] + list(INFO_KEYS)
            if kind == "cnn":
                names.insert(2, "env_frame")
            outs = []
            rows = test_rows.rows() if test_on else None     # [S, NUM_UPDATES, len(INFO_KEYS)]
            for s in range(S):
                metrics = {name: drv.metrics[s, :NUM_UPDATES, METRIC_NAMES.index(name)].to(torch.float32) for name in names}
                if test_on:
                    for j, k in enumerate(INFO_KEYS):
                        metrics[f"test/{k}"] = rows[s, :, j]
                theta_f = layout.to_flax(drv.theta_k(s))
                sl = slice(s * N, (s + 1) * N)
                runner_state = RunnerState({"params": network.views(theta_f), "theta": theta_f, "env_state": words[:, sl].contiguous(),
                                "last_obs": (ro.bits if packed else ro.obs)[0, sl], "test_metrics": tm_box[0][s],
                                "network": network, "rng": int(rngs[s]) & 0xFFFFFFFFFFFFFFFF,
                                "backend": backend, "driver": "graph" if drv.graph is not None else "eager",
                                "driver_graph_error": drv.graph_error, "opt_count": drv.count[s:s + 1],
                                "opt_mu": drv.m[s, :layout.total], "opt_nu": drv.v[s, :layout.total],
                                "kernel_layout": layout, "seed_batch": S, "kernel_forms": dict(forms), **counters})
                outs.append({"runner_state": runner_state, "metrics": metrics})
            return outs

        update.driver = drv
        return update, finish

    make_batch_runner = _streamed_factory(work_stream, _make_batch_runner)

    def train(rng: int) -> Dict[str, Any]:
        update, finish = make_runner(rng)
        for u in range(NUM_UPDATES):
            update(u)
        return finish()

    train.make_runner = make_runner
    train.make_batch_runner = make_batch_runner
    train.can_batch_seeds = bool(backend == "fused" and grad_hook is None and seeds_shape_ok
                                 and config.get("_DRIVER", True) and config.get("_CALLBACK") is None)
    train.config = config
    train.backend = backend
    train.stream = work_stream
    return train


def _log_row(config: Dict[str, Any], rng: int, u: int, m: Dict[str, Any]) -> None:
    """The reporting callback of the scripts (pqn_minatar.py:353-365, pqn_gymnax.py:346-358, pqn_craftax.py:394-408):
    one call per seed and update; WANDB_LOG_ALL_SEEDS adds every metric a second time under "rng<original_rng>/", where
    original_rng = rng[0] (:132), the first word of the seed's key."""
    if config.get("WANDB_LOG_ALL_SEEDS", False):
        tag = (int(rng) >> 32) & 0xFFFFFFFF
        m = {**m, **{f"rng{tag}/{k}": v for k, v in m.items()}}
    config["_CALLBACK"](u, m)


def vmap_train(train: Callable[[int], Dict[str, Any]], keys: List[int], concurrent: bool = True) -> Dict[str, Any]:
    """jax.vmap(make_train(config))(rngs) (pqn_minatar.py:459-461): independent seeds, outputs stacked on
    a leading [S] axis where they are tensors.

    Seeds share nothing (own parameters, optimizer, envs, RNG streams).  On the fused paths (MinAtar CNN, gymnax
    MLP) they are batched INTO the launches (grid.y = seed, pqn_cnn_update_seeds / pqn_mlp_update_seeds): one
    hipGraph replay advances every seed, so ten 128-env seeds cost about as much as one.  The torch-op network
    path runs the seeds as concurrent HIP streams (`concurrent="streams"` forces that mode).  Either way each seed's result is bit-identical to
    its solo run (`concurrent=False`)."""
    keys = list(keys)
    on_gpu = torch.cuda.is_available()
    if not concurrent or len(keys) <= 1 or not on_gpu or not hasattr(train, "make_runner"):
        outs = [train(k) for k in keys]
    elif concurrent == "streams" or not getattr(train, "can_batch_seeds", False):
        outs = _vmap_streams(train, keys)
    else:
        # the fused paths batch the seeds INTO the launches (grid.y = seed): one hipGraph replay per update
        # advances up to 128 seeds (7 seed bits in the shuffle keys); more seeds run as consecutive groups
        outs = []
        for g in range(0, len(keys), 128):
            update, finish = train.make_batch_runner(keys[g:g + 128])
            for u in range(int(train.config["NUM_UPDATES"])):
                update(u)
            outs.extend(finish())
    metrics = {k: torch.stack([o["metrics"][k] for o in outs]) for k in outs[0]["metrics"]}
    return {"runner_state": [o["runner_state"] for o in outs], "metrics": metrics}


def _vmap_streams(train, keys):
    """Seeds as concurrent HIP streams (paths without seed-batched kernels: torch-op networks)."""
    num_updates = int(train.config["NUM_UPDATES"])
    main = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in keys]
    runners = []
    for s, k in zip(streams, keys):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            runners.append(train.make_runner(k))
    for u in range(num_updates):
        for s, (update, _finish) in zip(streams, runners):
            with torch.cuda.stream(s):
                update(u)
    outs = []
    for s, (_update, finish) in zip(streams, runners):
        with torch.cuda.stream(s):
            outs.append(finish())
        main.wait_stream(s)
    return outs


def seed_keys(seed: int, num_seeds: int) -> List[int]:
    """rngs = split(PRNGKey(SEED), NUM_SEEDS) (pqn_minatar.py:456-459)."""
    base = _lib.prng_key(seed)
    return [_lib.fold_in(base, s) for s in range(int(num_seeds))]
