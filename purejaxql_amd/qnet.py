"""Fused MinAtar CNN Q-network kernels (csrc/pqn_qnet.hip) -- Python binding.

The kernels keep all parameters of one seed in ONE flat fp32 buffer in "kernel
layout" (pqn_cnn_layout_t): flax parameter order, 16-B aligned segments and the
fc1 kernel in MFMA fragment order.  `CnnKernelLayout` converts between that and
the flax-flat order of networks.QNetwork (used for init, checkpoints and the
oracle comparison).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


class CnnLayoutStruct(C.Structure):
    """pqn_cnn_layout_t"""
    _fields_ = [(n, C.c_int32) for n in ("c", "a", "off_bn", "off_wc", "off_bc", "off_ln0s", "off_ln0b", "off_w1",
                                         "off_b1", "off_ln1s", "off_ln1b", "off_w2", "off_b2", "total", "matmul_f16",
                                         "off_w1h", "alloc", "pos_f16x2")]


MATMUL_MODES = {"f32": 0, "float32": 0, "f16": 1, "fp16": 1, "float16": 1, "bf16x3": 2, "f16x2": 3}
MATMUL_MODE_NAMES = ("f32", "f16", "bf16x3", "f16x2")
# MATMUL_DTYPE: auto (the package default since round 6) picks between the f32-grade modes by the minibatch size: f16x2 from 512
# samples on, f32 below.  f16x2 IS bf16x3 in every kernel form but the position-parallel one (which launches that fill the chip take:
# >= 160 workgroups of 256 samples, csrc/pqn_qnet.hip pos_form_taken); there the operands travel as two range-scaled fp16 pieces and a
# product costs 3 matrix instructions instead of 6: 9.9e7 against 7.6e7 env-steps/s at the bench shape in one call, with single
# gradients NEARER to float64 than bf16x3's (profiles/r06_v7_f16x2_accuracy.txt; tests/test_qnet_gpu.py).  Measured whole-loop rates of Breakout (tools/mode_sweep.py, profiles/r06_v1_mode_sweep.txt): minibatch 128 /
# 256 -> f32 1.74e6 / 3.01e6 vs bf16x3 1.05e6 / 2.14e6 env-steps/s (only the f32 mode has the K-split kernels of the small
# launches); 512 / 1024 / 2048 / 4096 -> f32 3.78e6 / 7.46e6 / 1.45e7 / 2.78e7 vs bf16x3 4.28e6 / 8.58e6 / 1.68e7 / 3.05e7 (one seed),
# and 3.6e7 vs 7.55e7 with 16 seeds in the launches.  Both modes are held to the same tolerances against the oracle.
AUTO_BF16X3_MIN_MINIBATCH = 512


def resolve_matmul_dtype(config_value, minibatch=None) -> str:
    """config MATMUL_DTYPE -> the operand mode that runs: auto (or unset) = f16x2 (bf16x3 outside the position-parallel kernels) for
    minibatches of >= 512 samples, else f32."""
    key = str(config_value if config_value is not None else "auto").lower()
    if key == "auto":
        return "f16x2" if (minibatch is not None and int(minibatch) >= AUTO_BF16X3_MIN_MINIBATCH) else "f32"
    if key not in MATMUL_MODES:
        raise ValueError(f"MATMUL_DTYPE={config_value!r}: expected auto or one of {sorted(set(MATMUL_MODES))}")
    return key


def matmul_mode(config_value, minibatch=None) -> int:
    """config MATMUL_DTYPE -> pqn_cnn_layout_t.matmul_f16: 0 = f32-input MFMA (exact f32 fma chains), 1 = fp16 operands
    (opt-in, narrower than the reference's f32), 2 = bf16x3 split operands (f32-grade products on the bf16 matrix core), 3 = f16x2
    (mode 2 + two-piece fp16 operands in the position-parallel kernels); auto: see resolve_matmul_dtype."""
    return MATMUL_MODES[resolve_matmul_dtype(config_value, minibatch)]


class CnnKernelLayout:
    """pqn_cnn_layout_t + the flax <-> kernel index map.  matmul_f16 = the operand mode of the fc1 products (forward,
    input gradient, weight gradient), see matmul_mode(): True / 1 = fp16 operands with f32 accumulation (the parameter
    buffer then carries two fp16 copies of the fc1 kernel behind the `total` parameter floats, `alloc` floats in all);
    2 = bf16x3 split operands (no extra copies: operands are split in registers)."""

    def __init__(self, channels: int, num_actions: int, matmul_f16=False):
        lib = _lib.load()
        self.struct = CnnLayoutStruct()
        _lib.check(lib.pqn_cnn_layout_ex(channels, num_actions, int(matmul_f16), C.byref(self.struct)),
                   "pqn_cnn_layout_ex")
        s = self.struct
        self.c, self.a, self.total = int(s.c), int(s.a), int(s.total)
        self.alloc, self.mode, self.matmul_f16 = int(s.alloc), int(s.matmul_f16), int(s.matmul_f16) == 1
        self.pos_f16x2 = bool(s.pos_f16x2)     # mode 3: bf16x3 (mode 2) everywhere but the position-parallel kernels, which run f16x2
        self.mode_name = "f16x2" if self.pos_f16x2 else MATMUL_MODE_NAMES[self.mode]
        c, a = self.c, self.a
        i = torch.arange(1024).view(-1, 1)
        o = torch.arange(128).view(1, -1)
        fc1 = (((i // 16) * 8 + o // 16) * 64 + ((i % 16) // 4) * 16 + (o % 16)) * 4 + (i % 4)
        parts = [s.off_bn + torch.arange(2 * c), s.off_wc + torch.arange(9 * c * 16 + 48),
                 s.off_w1 + fc1.reshape(-1), s.off_b1 + torch.arange(3 * 128),
                 s.off_w2 + torch.arange(128 * a), s.off_b2 + torch.arange(a)]
        self.kidx = torch.cat(parts).to(torch.int64)       # flax-flat position -> kernel-flat position
        self.num_flax = int(self.kidx.numel())
        assert int(torch.unique(self.kidx).numel()) == self.num_flax

    def to_kernel(self, theta_flax: torch.Tensor) -> torch.Tensor:
        out = torch.zeros(self.alloc, dtype=torch.float32, device=theta_flax.device)
        out[self.kidx.to(theta_flax.device)] = theta_flax.to(torch.float32)
        self.refresh_copies(out)
        return out

    def refresh_copies(self, theta_k: torch.Tensor, w1b: Optional[torch.Tensor] = None):
        """Re-derive the fragment-order copies of the fc1 kernel (w1b: f32 dgrad copy; fp16 copies / bf16x3 planes in
        theta's tail, by operand mode)."""
        if theta_k.is_cuda and (w1b is not None or self.mode != 0):
            _lib.check(_lib.load().pqn_qnet_cnn_pack_w1b(C.byref(self.struct), _lib.ptr(theta_k), _lib.ptr(w1b),
                                                         _lib.stream_ptr()), "pqn_qnet_cnn_pack_w1b")

    def to_flax(self, theta_k: torch.Tensor) -> torch.Tensor:
        return theta_k[self.kidx.to(theta_k.device)]


def cnn_forward(layout: CnnKernelLayout, obs_bits: torch.Tensor, theta_k: torch.Tensor, *, want_q: bool = True,
                eps: Optional[float] = None, key: int = 0, q: Optional[torch.Tensor] = None,
                action: Optional[torch.Tensor] = None, qmax: Optional[torch.Tensor] = None):
    """q = network.apply(params, obs); optionally the eps-greedy action and max_a q in the same launch."""
    lib = _lib.load()
    n = obs_bits.shape[0]
    dev = obs_bits.device
    if want_q and q is None:
        q = torch.empty((n, layout.a), dtype=torch.float32, device=dev)
    if eps is not None and action is None:
        action = torch.empty(n, dtype=torch.int32, device=dev)
    if (eps is not None or not want_q) and qmax is None:
        qmax = torch.empty(n, dtype=torch.float32, device=dev)
    _lib.check(lib.pqn_qnet_cnn_forward(C.byref(layout.struct), n, _lib.ptr(obs_bits), _lib.ptr(theta_k), _lib.ptr(q),
                                        _lib.ptr(action), _lib.ptr(qmax), float(eps or 0.0), key, _lib.stream_ptr()),
               "pqn_qnet_cnn_forward")
    return q, action, qmax


def cnn_rollout(layout: CnnKernelLayout, env_id: int, state_words: torch.Tensor, obs_bits: torch.Tensor,
                theta_k: torch.Tensor, keys: torch.Tensor, eps: torch.Tensor, *, store_obs: bool = True,
                rew_scale: float = 1.0, want_last_q: bool = True):
    """jax.lax.scan(_step_env) + bootstrap forward (pqn_minatar.py:181-235) as ONE persistent launch
    (pqn_cnn_rollout).  state_words u32[W][N] is advanced in place; obs_bits is [T+1][N][OW] (slot 0 = the
    current observation) when store_obs else [1][N][OW]; keys int64[T] device step keys; eps f32[1] device.
    Returns the [T][N] transition record."""
    lib = _lib.load()
    t = int(keys.shape[0])
    n = int(state_words.shape[1])
    dev = state_words.device
    z = lambda dt: torch.empty((t, n), dtype=dt, device=dev)
    rec = {"action": z(torch.int32), "qmax": z(torch.float32), "reward": z(torch.float32), "done": z(torch.uint8),
           "discount": z(torch.float32), "returned_episode_returns": z(torch.float32),
           "returned_episode_lengths": z(torch.int32), "timestep": z(torch.int32),
           "last_q": torch.empty(n, dtype=torch.float32, device=dev) if want_last_q else None}
    out = _lib.StepOut(reward=_lib.ptr(rec["reward"]), done=_lib.ptr(rec["done"]), discount=_lib.ptr(rec["discount"]),
                       returned_episode_returns=_lib.ptr(rec["returned_episode_returns"]),
                       returned_episode_lengths=_lib.ptr(rec["returned_episode_lengths"]),
                       timestep=_lib.ptr(rec["timestep"]))
    _lib.check(lib.pqn_cnn_rollout(env_id, C.byref(layout.struct), n, t, _lib.ptr(state_words), _lib.ptr(obs_bits),
                                   1 if store_obs else 0, _lib.ptr(theta_k), C.byref(out), _lib.ptr(rec["action"]),
                                   _lib.ptr(rec["qmax"]), _lib.ptr(rec["last_q"]), _lib.ptr(eps), _lib.ptr(keys),
                                   float(rew_scale), _lib.stream_ptr()), "pqn_cnn_rollout")
    return rec


class CnnTrainer:
    """Parameters + RAdam state of one seed in kernel layout, and the fused optimizer step
    (train_state.apply_gradients of pqn_minatar.py:289-296) = pqn_qnet_cnn_grad + pqn_qnet_cnn_apply."""

    def __init__(self, layout: CnnKernelLayout, theta_flax: torch.Tensor, lr: float, max_grad_norm: float,
                 lr_decay_steps: float = 0.0, lr_end: float = 1e-20, max_minibatch: int = 4096):
        lib = _lib.load()
        dev = theta_flax.device
        self.layout = layout
        self.theta = layout.to_kernel(theta_flax)
        self.w1b = torch.empty(1024 * 128, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.theta)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lr, self.lr_end, self.lr_steps, self.max_norm = float(lr), float(lr_end), float(lr_decay_steps), float(max_grad_norm)
        self._ws_nb = 0
        self.ws = None
        self._ensure_ws(max_minibatch)
        layout.refresh_copies(self.theta, self.w1b)

    def _ensure_ws(self, nb: int):
        if nb > self._ws_nb:
            n = int(_lib.load().pqn_qnet_cnn_workspace_floats(C.byref(self.layout.struct), nb))
            self.ws = torch.empty(n, dtype=torch.float32, device=self.theta.device)
            self._ws_nb = nb

    def compute_grad(self, idx: torch.Tensor, obs_bits: torch.Tensor, action: torch.Tensor, target: torch.Tensor,
                     loss_out: Optional[torch.Tensor] = None, qv_out: Optional[torch.Tensor] = None):
        lib = _lib.load()
        nb = idx.numel()
        self._ensure_ws(nb)
        assert idx.dtype == torch.int64 and action.dtype == torch.int32 and target.dtype == torch.float32
        _lib.check(lib.pqn_qnet_cnn_grad(C.byref(self.layout.struct), nb, _lib.ptr(idx), _lib.ptr(obs_bits),
                                         _lib.ptr(action), _lib.ptr(target), _lib.ptr(self.theta), _lib.ptr(self.w1b),
                                         _lib.ptr(self.grad), _lib.ptr(self.count), _lib.ptr(self.ws),
                                         _lib.ptr(loss_out), _lib.ptr(qv_out), _lib.stream_ptr()), "pqn_qnet_cnn_grad")
        return self.grad

    def apply(self, recompute_norm: bool = False):
        lib = _lib.load()
        _lib.check(lib.pqn_qnet_cnn_apply(C.byref(self.layout.struct), _lib.ptr(self.theta), _lib.ptr(self.w1b),
                                          _lib.ptr(self.grad), _lib.ptr(self.m), _lib.ptr(self.v), _lib.ptr(self.count),
                                          self.lr, self.lr_end, self.lr_steps, self.max_norm, _lib.ptr(self.ws),
                                          _lib.ptr(self.gnorm), 1 if recompute_norm else 0, _lib.stream_ptr()),
                   "pqn_qnet_cnn_apply")

    def theta_flax(self) -> torch.Tensor:
        return self.layout.to_flax(self.theta)


def cnn_grad_seeds(layout: CnnKernelLayout, theta_k: torch.Tensor, idx: torch.Tensor, obs_bits: torch.Tensor,
                   action: torch.Tensor, target: torch.Tensor, n_env: int, n_env_total: Optional[int] = None,
                   _ws_fill: Optional[float] = None):
    """jax.vmap(value_and_grad(_loss_fn)) over seeds (pqn_minatar.py:271-291 under :459-461) in ONE set of launches
    (pqn_qnet_cnn_grad_seeds): theta_k [S, stride] kernel-layout parameters (operand copies in step, see
    CnnKernelLayout.to_kernel), idx int64 [S, nb] transition indices per seed into the stacked [T][S*N] record (or a
    shared pool when n_env_total == n_env).  Returns (grad [S, stride], loss [S], mean q_a [S])."""
    lib = _lib.load()
    s, nb = int(idx.shape[0]), int(idx.shape[1])
    dev = theta_k.device
    n_env_total = int(n_env if n_env_total is None else n_env_total)
    assert theta_k.dim() == 2 and theta_k.shape[0] == s and theta_k.is_contiguous() and idx.is_contiguous()
    assert idx.dtype == torch.int64 and action.dtype == torch.int32 and target.dtype == torch.float32
    ws_stride = (int(lib.pqn_qnet_cnn_workspace_floats(C.byref(layout.struct), nb)) + 3) // 4 * 4
    ws = torch.empty((s, ws_stride), dtype=torch.float32, device=dev)
    if _ws_fill is not None:      # tests: poison the workspace (every byte the kernels read must have been written by them)
        ws.fill_(_ws_fill)
    w1b = torch.empty((s, 1024 * 128), dtype=torch.float32, device=dev)
    for k in range(s):
        layout.refresh_copies(theta_k[k], w1b[k])
    grad = torch.zeros_like(theta_k)
    count = torch.zeros(s, dtype=torch.int32, device=dev)
    loss = torch.zeros(s, dtype=torch.float32, device=dev)
    qv = torch.zeros(s, dtype=torch.float32, device=dev)
    _lib.check(lib.pqn_qnet_cnn_grad_seeds(C.byref(layout.struct), s, nb, _lib.ptr(idx), int(idx.stride(0)), int(n_env),
                                           n_env_total, _lib.ptr(obs_bits), _lib.ptr(action), _lib.ptr(target),
                                           _lib.ptr(theta_k), int(theta_k.stride(0)), _lib.ptr(w1b), _lib.ptr(grad),
                                           _lib.ptr(count), _lib.ptr(ws), ws_stride, _lib.ptr(loss), _lib.ptr(qv),
                                           _lib.stream_ptr()), "pqn_qnet_cnn_grad_seeds")
    return grad, loss, qv


class UpdateArgs(C.Structure):
    """pqn_update_args_t (include/pqn_hotpath.h)"""
    _fields_ = ([(n, C.c_int32) for n in ("env_id", "num_envs", "num_steps", "num_minibatches", "num_epochs",
                                           "obs_words", "metrics_capacity", "reserved")] +
                [(n, C.c_float) for n in ("gamma", "lam", "rew_scale", "eps_start", "eps_finish",
                                          "lr_init", "lr_end", "max_grad_norm")] +
                [(n, C.c_double) for n in ("eps_decay_steps", "lr_steps")] +
                [(n, C.c_uint64) for n in ("key_roll", "key_shuf", "sort_temp_bytes")] +
                [("layout", CnnLayoutStruct)] +
                [(n, C.c_void_p) for n in ("clock", "sched_keys", "sched_eps", "state", "bits", "action", "reward",
                                           "done", "qmax", "discount", "rer", "rel", "ts", "target", "last_q",
                                           "sort_keys_in", "sort_keys_out", "sort_temp", "theta", "w1b", "grad", "m",
                                           "v", "count", "workspace", "loss_buf", "qv_buf", "metrics")])


METRIC_NAMES = ("env_step", "update_steps", "env_frame", "grad_steps", "td_loss", "qvals", "discount",
                "returned_episode_returns", "returned_episode_lengths", "timestep", "returned_episode")


def _options_epoch_moved(drv) -> bool:
    """Shared by every update driver: a captured hipGraph replays the kernels (and the by-value kernel arguments, e.g. the peer
    time-out) chosen at capture time.  Returns True when a pqn_set_option changed a value since `drv` captured
    (pqn_options_epoch, include/pqn_hotpath.h); the caller then drops its graph(s) and captures again.  Leaves the current
    epoch in drv._epoch_now for the capture that follows."""
    drv._epoch_now = int(_lib.load().pqn_options_epoch())
    return getattr(drv, "_graph_epoch", None) is not None and drv._graph_epoch != drv._epoch_now


class UpdateDriver:
    """Whole-update enqueue (pqn_cnn_update / pqn_mlp_update) with optional hipGraph replay.  Holds the scratch
    buffers the C side needs; all training buffers are the caller's (rollout record, Cnn/MlpTrainer)."""

    def __init__(self, env_id, n, t, mb, epochs, obs_words, cfg, keys, trainer, ro, words, num_updates,
                 use_graph: bool = True, pin_form: bool = False):
        lib = _lib.load()
        dev = trainer.theta.device
        self.dev = dev
        self.mlp = isinstance(trainer, MlpTrainer)
        tn = n * t
        self.clock = torch.zeros(4, dtype=torch.int32, device=dev)
        self.sched_keys = torch.zeros(t + epochs, dtype=torch.int64, device=dev)
        self.sched_eps = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sk_in = torch.empty(tn, dtype=torch.int64, device=dev)
        self.sk_out = torch.empty(tn, dtype=torch.int64, device=dev)
        tb = int(lib.pqn_update_sort_temp_bytes(tn))
        if tb < 0:
            raise RuntimeError("pqn_update_sort_temp_bytes failed")
        self.sort_temp = torch.empty(max(tb, 16), dtype=torch.uint8, device=dev)
        self.loss_buf = torch.zeros(mb * epochs, dtype=torch.float32, device=dev)
        self.qv_buf = torch.zeros(mb * epochs, dtype=torch.float32, device=dev)
        self.metrics = torch.zeros((max(num_updates, 1), len(METRIC_NAMES)), dtype=torch.float64, device=dev)
        trainer._ensure_ws(tn // mb)
        a = MlpUpdateArgs() if self.mlp else UpdateArgs()
        a.env_id, a.num_envs, a.num_steps, a.num_minibatches, a.num_epochs = env_id, n, t, mb, epochs
        a.metrics_capacity = self.metrics.shape[0]
        a.gamma, a.lam, a.rew_scale = cfg["gamma"], cfg["lam"], cfg["rew_scale"]
        a.eps_start, a.eps_finish, a.eps_decay_steps = cfg["eps_start"], cfg["eps_finish"], cfg["eps_decay_steps"]
        a.lr_init, a.lr_end, a.lr_steps, a.max_grad_norm = trainer.lr, trainer.lr_end, trainer.lr_steps, trainer.max_norm
        a.key_roll, a.key_shuf = keys
        a.sort_temp_bytes = tb
        a.layout = trainer.layout.struct
        p = _lib.ptr
        a.clock, a.sched_keys, a.sched_eps = p(self.clock), p(self.sched_keys), p(self.sched_eps)
        a.state = p(words)
        if self.mlp:
            a.obs, a.wt = p(ro.obs), p(trainer.wt)
        else:
            a.obs_words, a.bits, a.w1b = obs_words, p(ro.bits), p(trainer.w1b)
            # bit 1 (config SEED_BATCH_BIT_IDENTICAL): kernel form from the minibatch size alone -- a solo run then takes the
            # kernels a seed batch of the same shape takes and equals its seed of that batch bit for bit
            a.reserved = 2 if pin_form else 0
        a.action, a.reward, a.done, a.qmax = p(ro.action), p(ro.reward), p(ro.done), p(ro.qmax)
        a.discount, a.rer, a.rel, a.ts = p(ro.discount), p(ro.rer), p(ro.rel), p(ro.ts)
        a.target, a.last_q = p(ro.target), p(ro.last_q)
        a.sort_keys_in, a.sort_keys_out, a.sort_temp = p(self.sk_in), p(self.sk_out), p(self.sort_temp)
        a.theta, a.grad, a.m, a.v = p(trainer.theta), p(trainer.grad), p(trainer.m), p(trainer.v)
        a.count, a.workspace = p(trainer.count), p(trainer.ws)
        a.loss_buf, a.qv_buf, a.metrics = p(self.loss_buf), p(self.qv_buf), p(self.metrics)
        self.args = a
        self._keep = (trainer, ro, words)
        self.use_graph = use_graph
        self.graph = None
        self.graph_error = None
        self.calls = 0

    def _enqueue(self):
        lib = _lib.load()
        if self.mlp:
            _lib.check(lib.pqn_mlp_update(C.byref(self.args), _lib.stream_ptr()), "pqn_mlp_update")
        else:
            _lib.check(lib.pqn_cnn_update(C.byref(self.args), _lib.stream_ptr()), "pqn_cnn_update")

    def update(self):
        """Enqueue (or replay) one update; the update index lives on the device (self.clock[0]).  A captured graph replays the
        kernels chosen at capture time: when a kernel-selection option changed since (pqn_options_epoch) it is dropped and
        the update is captured again, so that pqn_set_option / _lib.options(...) take effect on running drivers too."""
        if _options_epoch_moved(self):
            self.graph = None
        epoch = self._epoch_now
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue()
            if self.use_graph and (self.calls == 0 or getattr(self, "_graph_epoch", None) is not None) and self.graph_error is None:
                # first update ran eagerly (kernel attributes set, caches warm); capture the next one
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._enqueue()
                    self.graph = g
                    self._graph_epoch = epoch
                except Exception as exc:  # stay on the eager C++ enqueue (still the HIP path)
                    self.graph_error = repr(exc)
                    torch.cuda.synchronize()
        self.calls += 1


PHASE_BEGIN, PHASE_SHUFFLE, PHASE_GRAD, PHASE_APPLY, PHASE_END = range(5)


class EnvShardDriver(UpdateDriver):
    """One update of ONE seed whose envs are sharded over ranks: the C++ enqueue split at the
    gradient / optimizer boundary of every minibatch (pqn_cnn_update_phase), `grad_hook(flat_grad)` -- the RCCL
    all-reduce -- issued from the host between the segments.  Segment k = [APPLY(k-1)] [SHUFFLE] GRAD(k) ...; each
    segment is captured once in its own hipGraph (the device clock makes every segment replayable), so per update the
    host issues NUM_MINIBATCHES*NUM_EPOCHS + 1 graph launches and as many collectives instead of ~300 kernels."""

    def __init__(self, *args, grad_hook=None, **kw):
        super().__init__(*args, **kw)
        if self.mlp:
            raise RuntimeError("EnvShardDriver: CNN path only")
        if grad_hook is None:
            raise ValueError("EnvShardDriver needs a grad_hook")
        self.grad_hook = grad_hook
        mb, ep = int(self.args.num_minibatches), int(self.args.num_epochs)
        n = mb * ep
        segs = [[(PHASE_BEGIN, 0), (PHASE_SHUFFLE, 0), (PHASE_GRAD, 0)]]
        for i in range(n - 1):
            s = [(PHASE_APPLY, i)]
            if (i + 1) % mb == 0:
                s.append((PHASE_SHUFFLE, (i + 1) // mb))
            s.append((PHASE_GRAD, i + 1))
            segs.append(s)
        segs.append([(PHASE_APPLY, n - 1), (PHASE_END, 0)])
        self.segments = segs
        self.graphs = None
        self.whole = None    # the single graph of the whole update (capturable collective only)

    def _enqueue_segment(self, seg):
        lib = _lib.load()
        for phase, index in seg:
            _lib.check(lib.pqn_cnn_update_phase(C.byref(self.args), phase, index, _lib.stream_ptr()), "pqn_cnn_update_phase")

    def _enqueue_all(self):
        trainer = self._keep[0]
        for k, seg in enumerate(self.segments):
            self._enqueue_segment(seg)
            if k + 1 < len(self.segments):
                self.grad_hook(trainer.grad)

    def update(self):
        trainer = self._keep[0]
        if _options_epoch_moved(self):
            # an option changed (on every rank, by the same code path: the re-capture below meets on the hook's barrier): drop
            # the whole-update graph and the per-segment graphs; this update runs the segments eagerly or re-captures them
            self.whole = self.graphs = self.graph = None
        # a capturable collective (dist.PeerAllReduce: kernels on the training stream) lets the WHOLE update, its
        # NUM_MINIBATCHES*NUM_EPOCHS all-reduces included, be one hipGraph per rank: update 0 runs eagerly (the hook sets
        # itself up there), update 1 is captured and replayed, later ones replay
        if getattr(self.grad_hook, "capturable", False) and self.use_graph and self.graph_error is None and self.graphs is None:
            if self.whole is None and self.calls >= 1:
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._enqueue_all()
                    self.whole = self.graph = g
                    self._graph_epoch = self._epoch_now
                except Exception as exc:  # stay on the per-segment path below (still the HIP path)
                    self.graph_error = repr(exc)
                    torch.cuda.synchronize()
                # capture + instantiate take a rank-dependent time (hundreds of ms): meet on the host before the first
                # replayed collective so that no rank's in-graph wait starts long before its peers can publish
                barrier = getattr(self.grad_hook, "barrier", None)
                if barrier is not None:
                    torch.cuda.synchronize()
                    barrier()
            if self.whole is not None:
                self.whole.replay()
                self.calls += 1
                return
        capture = self.use_graph and self.calls >= 1 and self.graphs is None and self.graph_error is None
        new_graphs = []
        for k, seg in enumerate(self.segments):
            if self.graphs is not None:
                self.graphs[k].replay()
            elif capture and self.graph_error is None:
                # the first update ran eagerly (kernel attributes set, caches warm); capture while running the second
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._enqueue_segment(seg)
                    g.replay()
                    new_graphs.append(g)
                except Exception as exc:  # stay on the eager C++ enqueue (still the HIP path)
                    self.graph_error = repr(exc)
                    torch.cuda.synchronize()
                    self._enqueue_segment(seg)
            else:
                self._enqueue_segment(seg)
            if k + 1 < len(self.segments):
                self.grad_hook(trainer.grad)
        if capture and self.graph_error is None and len(new_graphs) == len(self.segments):
            self.graphs = new_graphs
            self.graph = new_graphs[0]   # "graph" in runner_state["driver"]
            self._graph_epoch = self._epoch_now
        self.calls += 1


class SeedsUpdateDriver:
    """jax.vmap over seeds inside the launches (pqn_cnn_update_seeds / pqn_mlp_update_seeds): S independent seeds
    advance by one update in the same kernels (grid.y = seed) -- one enqueue, one hipGraph.  Owns the stacked
    parameter / optimizer storage ([S, stride]); the caller provides the stacked rollout record ([T][S*N] arrays)
    and env state."""

    def __init__(self, layout, env_id, s, n, t, mb, epochs, obs_words, cfg, keys_roll, keys_shuf,
                 lr, lr_end, lr_steps, max_norm, ro, words, num_updates, device, use_graph: bool = True,
                 pin_form: bool = False):
        lib = _lib.load()
        dev = torch.device(device)
        self.layout, self.s, self.n = layout, int(s), int(n)
        self.mlp = isinstance(layout, MlpKernelLayout)
        tn = n * t
        f32 = torch.float32
        alloc = layout.total if self.mlp else layout.alloc
        self.stride = (alloc + 3) // 4 * 4
        if self.mlp:
            ws = int(lib.pqn_mlp_workspace_floats(C.byref(layout.struct), tn // mb))
        else:   # per-minibatch workspace + the epoch region of the position-parallel form (one gather launch per epoch)
            ws = int(lib.pqn_cnn_update_workspace_floats(C.byref(layout.struct), n, t, mb))
        if ws < 0:
            raise RuntimeError("workspace size query failed")
        self.ws_stride = (ws + 3) // 4 * 4
        self.theta = torch.zeros((s, self.stride), dtype=f32, device=dev)
        self.grad, self.m, self.v = (torch.zeros_like(self.theta) for _ in range(3))
        if self.mlp:
            self.wt_stride = max(layout.layers - 1, 1) * layout.h * layout.h
            self.wt = torch.zeros((s, self.wt_stride), dtype=f32, device=dev)
            self.w1b = None
        else:
            self.w1b = torch.empty((s, 1024 * 128), dtype=f32, device=dev)
        self.count = torch.zeros(s, dtype=torch.int32, device=dev)
        self.ws = torch.empty((s, self.ws_stride), dtype=f32, device=dev)
        self.clock = torch.zeros(4, dtype=torch.int32, device=dev)
        self.sched_keys = torch.zeros((s, t + epochs), dtype=torch.int64, device=dev)
        self.sched_eps = torch.zeros(1, dtype=f32, device=dev)
        self.sk_in = torch.empty(s * tn, dtype=torch.int64, device=dev)
        self.sk_out = torch.empty(s * tn, dtype=torch.int64, device=dev)
        tb = int(lib.pqn_update_sort_temp_bytes(s * tn))
        if tb < 0:
            raise RuntimeError("pqn_update_sort_temp_bytes failed")
        self.sort_temp = torch.empty(max(tb, 16), dtype=torch.uint8, device=dev)
        self.loss_buf = torch.zeros((s, mb * epochs), dtype=f32, device=dev)
        self.qv_buf = torch.zeros((s, mb * epochs), dtype=f32, device=dev)
        self.metrics = torch.zeros((s, max(num_updates, 1), len(METRIC_NAMES)), dtype=torch.float64, device=dev)
        to_i64 = lambda ks: torch.tensor([k - (1 << 64) if k >= (1 << 63) else k for k in ks], dtype=torch.int64, device=dev)
        self.key_roll, self.key_shuf = to_i64(keys_roll), to_i64(keys_shuf)
        a = MlpUpdateArgs() if self.mlp else UpdateArgs()
        a.env_id, a.num_envs, a.num_steps, a.num_minibatches, a.num_epochs = env_id, n, t, mb, epochs
        a.metrics_capacity = self.metrics.shape[1]
        a.gamma, a.lam, a.rew_scale = cfg["gamma"], cfg["lam"], cfg["rew_scale"]
        a.eps_start, a.eps_finish, a.eps_decay_steps = cfg["eps_start"], cfg["eps_finish"], cfg["eps_decay_steps"]
        a.lr_init, a.lr_end, a.lr_steps, a.max_grad_norm = float(lr), float(lr_end), float(lr_steps), float(max_norm)
        a.sort_temp_bytes = tb
        a.layout = layout.struct
        p = _lib.ptr
        a.clock, a.sched_keys, a.sched_eps = p(self.clock), p(self.sched_keys), p(self.sched_eps)
        a.state = p(words)
        if self.mlp:
            a.obs, a.wt = p(ro.obs), p(self.wt)
        else:
            a.obs_words, a.bits, a.w1b = obs_words, p(ro.bits), p(self.w1b)
            a.reserved = 2 if pin_form else 0    # pqn_update_args_t.reserved bit 1: kernel form from the minibatch size alone
        a.action, a.reward, a.done, a.qmax = p(ro.action), p(ro.reward), p(ro.done), p(ro.qmax)
        a.discount, a.rer, a.rel, a.ts = p(ro.discount), p(ro.rer), p(ro.rel), p(ro.ts)
        a.target, a.last_q = p(ro.target), p(ro.last_q)
        a.sort_keys_in, a.sort_keys_out, a.sort_temp = p(self.sk_in), p(self.sk_out), p(self.sort_temp)
        a.theta, a.grad, a.m, a.v = p(self.theta), p(self.grad), p(self.m), p(self.v)
        a.count, a.workspace = p(self.count), p(self.ws)
        a.loss_buf, a.qv_buf, a.metrics = p(self.loss_buf), p(self.qv_buf), p(self.metrics)
        self.args = a
        self._keep = (ro, words)
        self.use_graph, self.graph, self.graph_error, self.calls = use_graph, None, None, 0

    def set_params(self, seed: int, theta_flax: torch.Tensor):
        th = self.layout.to_kernel(theta_flax)
        self.theta[seed, :th.numel()] = th
        if self.mlp:
            _lib.check(_lib.load().pqn_mlp_refresh_transposed(C.byref(self.layout.struct), _lib.ptr(self.theta[seed]),
                                                              _lib.ptr(self.wt[seed]), _lib.stream_ptr()),
                       "pqn_mlp_refresh_transposed")
        else:
            self.layout.refresh_copies(self.theta[seed], self.w1b[seed])

    def theta_k(self, seed: int) -> torch.Tensor:
        return self.theta[seed, :self.layout.total]

    def _enqueue(self):
        lib = _lib.load()
        if self.mlp:
            _lib.check(lib.pqn_mlp_update_seeds(C.byref(self.args), self.s, _lib.ptr(self.key_roll), _lib.ptr(self.key_shuf),
                                                self.stride, self.ws_stride, self.wt_stride, _lib.stream_ptr()),
                       "pqn_mlp_update_seeds")
        else:
            _lib.check(lib.pqn_cnn_update_seeds(C.byref(self.args), self.s, _lib.ptr(self.key_roll), _lib.ptr(self.key_shuf),
                                                self.stride, self.ws_stride, _lib.stream_ptr()), "pqn_cnn_update_seeds")

    update = UpdateDriver.update


class SeedGroupsDriver:
    """G seed groups (one SeedsUpdateDriver each, CNN path) advanced together by pqn_cnn_update_seed_groups: the training
    kernels of all groups back to back on the current stream, every group's HBM-bound tail (fc1 weight gradient, fold,
    clip + RAdam) on a second stream under the NEXT group's training kernel.  The seeds of jax.vmap(make_train)
    (pqn_minatar.py:459-461) are independent, so only the order in time changes: every seed keeps the bits of its
    group's own pqn_cnn_update_seeds.  One hipGraph (two branches) per update by default.

    tail: "graph" (default: capture + replay), "eager" (C++ enqueue on a high-priority tail stream), "masked:<lo>:<hi>" or
    "maskmod:<m>:<r>" (eager, tail stream restricted to CUs [lo, hi) / to CUs with index % m == r, the compute stream to
    the others -- measurement aid)."""

    def __init__(self, drivers, tail: str = "graph"):
        if any(d.mlp for d in drivers):
            raise RuntimeError("SeedGroupsDriver: CNN path only")
        self.drivers = list(drivers)
        self.tail = tail
        g = len(self.drivers)
        self._args = (C.c_void_p * g)(*[C.addressof(d.args) for d in self.drivers])
        self._seeds = (C.c_int32 * g)(*[d.s for d in self.drivers])
        self._kr = (C.c_void_p * g)(*[_lib.ptr(d.key_roll) for d in self.drivers])
        self._ks = (C.c_void_p * g)(*[_lib.ptr(d.key_shuf) for d in self.drivers])
        self._ts = (C.c_int64 * g)(*[d.stride for d in self.drivers])
        self._ws = (C.c_int64 * g)(*[d.ws_stride for d in self.drivers])
        self.use_graph = tail == "graph" and all(d.use_graph for d in self.drivers)
        self.graph, self.graph_error, self.calls = None, None, 0
        self._raw = []          # raw hipStream_t handles made by pqn_stream_create_masked
        self._compute_stream = None
        lib = _lib.load()
        if tail.startswith("mask"):
            kind, lo, hi = tail.split(":")      # masked:<lo>:<hi> = CUs [lo, hi); maskmod:<m>:<r> = CUs with i % m == r
            lo, hi = int(lo), int(hi)
            ncu = torch.cuda.get_device_properties(self.drivers[0].theta.device).multi_processor_count
            words = (ncu + 31) // 32
            tm, cm = [0] * words, [0] * words
            for i in range(ncu):
                on_tail = (i % lo == hi) if kind == "maskmod" else (lo <= i < hi)
                (tm if on_tail else cm)[i // 32] |= 1 << (i % 32)
            self._compute_stream = self._make_raw((C.c_uint32 * words)(*cm), words, 0)
            self._tail_ptr = self._make_raw((C.c_uint32 * words)(*tm), words, 0)
            self._fork = torch.cuda.Event()
        elif tail == "eager":
            self._tail_ptr = self._make_raw(None, 0, 1)
        else:
            self._tail_stream = torch.cuda.Stream(priority=-1)
            self._tail_ptr = self._tail_stream.cuda_stream

    def _make_raw(self, mask, words, prio):
        out = C.c_void_p(0)
        _lib.check(_lib.load().pqn_stream_create_masked(mask, words, prio, C.byref(out)), "pqn_stream_create_masked")
        self._raw.append(out.value)
        return out.value

    def close(self):
        for h in self._raw:
            _lib.load().pqn_stream_destroy(h)
        self._raw = []

    def _enqueue(self):
        lib = _lib.load()
        sc = _lib.stream_ptr()
        if self._compute_stream is not None:
            # the masked compute stream is a raw HIP stream torch does not know: order it behind / ahead of torch's current
            # stream with a device-wide dependency through the legacy default stream semantics is not available, so fence
            # with events recorded by torch and waited on in C++ is overkill for a measurement aid -- synchronise instead
            torch.cuda.current_stream().synchronize()
            sc = self._compute_stream
        _lib.check(lib.pqn_cnn_update_seed_groups(len(self.drivers), self._args, self._seeds, self._kr, self._ks, self._ts,
                                                  self._ws, sc, self._tail_ptr), "pqn_cnn_update_seed_groups")

    def update(self):
        if _options_epoch_moved(self):
            self.graph = None
        if self.graph is not None:
            self.graph.replay()
        else:
            self._enqueue()
            if self.use_graph and (self.calls == 0 or getattr(self, "_graph_epoch", None) is not None) and self.graph_error is None:
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._enqueue()
                    self.graph = g
                    self._graph_epoch = self._epoch_now
                except Exception as exc:  # stay on the eager C++ enqueue (still the HIP path)
                    self.graph_error = repr(exc)
                    torch.cuda.synchronize()
        self.calls += 1
        for d in self.drivers:
            d.calls, d.graph, d.graph_error = self.calls, self.graph, self.graph_error


# ---------------------------------------------------------------------------------------------------------
# fused MLP Q-network (csrc/pqn_mlp.hip)
# ---------------------------------------------------------------------------------------------------------
class MlpLayoutStruct(C.Structure):
    """pqn_mlp_layout_t"""
    _fields_ = [("d", C.c_int32), ("h", C.c_int32), ("layers", C.c_int32), ("a", C.c_int32), ("off_bn", C.c_int32),
                ("off_w", C.c_int32 * 4), ("off_b", C.c_int32 * 4), ("off_lns", C.c_int32 * 4),
                ("off_lnb", C.c_int32 * 4), ("off_wout", C.c_int32), ("off_bout", C.c_int32), ("total", C.c_int32)]


class MlpUpdateArgs(C.Structure):
    """pqn_mlp_update_args_t (include/pqn_hotpath.h)"""
    _fields_ = ([(n, C.c_int32) for n in ("env_id", "num_envs", "num_steps", "num_minibatches", "num_epochs",
                                           "metrics_capacity")] +
                [(n, C.c_float) for n in ("gamma", "lam", "rew_scale", "eps_start", "eps_finish",
                                          "lr_init", "lr_end", "max_grad_norm")] +
                [(n, C.c_double) for n in ("eps_decay_steps", "lr_steps")] +
                [(n, C.c_uint64) for n in ("key_roll", "key_shuf", "sort_temp_bytes")] +
                [("layout", MlpLayoutStruct)] +
                [(n, C.c_void_p) for n in ("clock", "sched_keys", "sched_eps", "state", "obs", "action", "reward",
                                           "done", "qmax", "discount", "rer", "rel", "ts", "target", "last_q",
                                           "sort_keys_in", "sort_keys_out", "sort_temp", "theta", "wt", "grad", "m",
                                           "v", "count", "workspace", "loss_buf", "qv_buf", "metrics")])


class MlpKernelLayout:
    """flax-flat (networks.mlp_param_shapes order) <-> kernel layout (same order, 16-B aligned segments)."""

    def __init__(self, obs_dim: int, hidden: int, layers: int, num_actions: int):
        lib = _lib.load()
        self.struct = MlpLayoutStruct()
        _lib.check(lib.pqn_mlp_layout(obs_dim, hidden, layers, num_actions, C.byref(self.struct)), "pqn_mlp_layout")
        s = self.struct
        self.d, self.h, self.layers, self.a, self.total = int(s.d), int(s.h), int(s.layers), int(s.a), int(s.total)
        parts = [s.off_bn + torch.arange(2 * self.d)]
        ind = self.d
        for l in range(self.layers):
            parts += [s.off_w[l] + torch.arange(ind * self.h), s.off_b[l] + torch.arange(self.h),
                      s.off_lns[l] + torch.arange(self.h), s.off_lnb[l] + torch.arange(self.h)]
            ind = self.h
        parts += [s.off_wout + torch.arange(self.h * self.a), s.off_bout + torch.arange(self.a)]
        self.kidx = torch.cat(parts).to(torch.int64)
        self.num_flax = int(self.kidx.numel())

    def to_kernel(self, theta_flax: torch.Tensor) -> torch.Tensor:
        out = torch.zeros(self.total, dtype=torch.float32, device=theta_flax.device)
        out[self.kidx.to(theta_flax.device)] = theta_flax.to(torch.float32)
        return out

    def to_flax(self, theta_k: torch.Tensor) -> torch.Tensor:
        return theta_k[self.kidx.to(theta_k.device)]


def mlp_forward(layout: MlpKernelLayout, obs: torch.Tensor, theta_k: torch.Tensor, *, want_q: bool = True,
                eps: Optional[float] = None, key: int = 0, q=None, action=None, qmax=None):
    lib = _lib.load()
    n = obs.shape[0]
    dev = obs.device
    if want_q and q is None:
        q = torch.empty((n, layout.a), dtype=torch.float32, device=dev)
    if eps is not None and action is None:
        action = torch.empty(n, dtype=torch.int32, device=dev)
    if (eps is not None or not want_q) and qmax is None:
        qmax = torch.empty(n, dtype=torch.float32, device=dev)
    _lib.check(lib.pqn_mlp_forward(C.byref(layout.struct), n, _lib.ptr(obs), _lib.ptr(theta_k), _lib.ptr(q),
                                   _lib.ptr(action), _lib.ptr(qmax), float(eps or 0.0), key, _lib.stream_ptr()),
               "pqn_mlp_forward")
    return q, action, qmax


class MlpTrainer:
    """Parameters + RAdam state of one seed for the fused MLP; same interface as CnnTrainer."""

    def __init__(self, layout: MlpKernelLayout, theta_flax: torch.Tensor, lr: float, max_grad_norm: float,
                 lr_decay_steps: float = 0.0, lr_end: float = 1e-20, max_minibatch: int = 128):
        lib = _lib.load()
        dev = theta_flax.device
        self.layout = layout
        self.theta = layout.to_kernel(theta_flax)
        self.wt = torch.zeros(max(layout.layers - 1, 1) * layout.h * layout.h, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.theta)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lr, self.lr_end, self.lr_steps, self.max_norm = float(lr), float(lr_end), float(lr_decay_steps), float(max_grad_norm)
        self._ws_nb, self.ws = 0, None
        self._ensure_ws(max_minibatch)
        _lib.check(lib.pqn_mlp_refresh_transposed(C.byref(layout.struct), _lib.ptr(self.theta), _lib.ptr(self.wt),
                                                  _lib.stream_ptr()), "pqn_mlp_refresh_transposed")

    def _ensure_ws(self, nb: int):
        if nb > self._ws_nb:
            n = int(_lib.load().pqn_mlp_workspace_floats(C.byref(self.layout.struct), nb))
            self.ws = torch.empty(n, dtype=torch.float32, device=self.theta.device)
            self._ws_nb = nb

    def compute_grad(self, idx, obs, action, target, loss_out=None, qv_out=None):
        lib = _lib.load()
        nb = idx.numel()
        self._ensure_ws(nb)
        assert idx.dtype == torch.int64 and action.dtype == torch.int32 and obs.dtype == torch.float32
        _lib.check(lib.pqn_mlp_grad(C.byref(self.layout.struct), nb, _lib.ptr(idx), _lib.ptr(obs), _lib.ptr(action),
                                    _lib.ptr(target), _lib.ptr(self.theta), _lib.ptr(self.wt), _lib.ptr(self.grad),
                                    _lib.ptr(self.count), _lib.ptr(self.ws), _lib.ptr(loss_out), _lib.ptr(qv_out),
                                    _lib.stream_ptr()), "pqn_mlp_grad")
        return self.grad

    def apply(self, recompute_norm: bool = False):
        lib = _lib.load()
        _lib.check(lib.pqn_mlp_apply(C.byref(self.layout.struct), _lib.ptr(self.theta), _lib.ptr(self.wt),
                                     _lib.ptr(self.grad), _lib.ptr(self.m), _lib.ptr(self.v), _lib.ptr(self.count),
                                     self.lr, self.lr_end, self.lr_steps, self.max_norm, _lib.ptr(self.ws),
                                     _lib.ptr(self.gnorm), 1 if recompute_norm else 0, _lib.stream_ptr()),
                   "pqn_mlp_apply")

    def theta_flax(self) -> torch.Tensor:
        return self.layout.to_flax(self.theta)


# ---------------------------------------------------------------------------------------------------------
# wide MLP Q-network of the Craftax script (csrc/pqn_bigmlp.hip)
# ---------------------------------------------------------------------------------------------------------
BIGMLP_MAX_LAYERS = 8


class BigMlpLayoutStruct(C.Structure):
    """pqn_bigmlp_layout_t"""
    _fields_ = [("d", C.c_int32), ("h", C.c_int32), ("layers", C.c_int32), ("a", C.c_int32), ("norm_input", C.c_int32),
                ("off_in_scale", C.c_int32), ("off_in_bias", C.c_int32),
                ("off_w", C.c_int32 * (BIGMLP_MAX_LAYERS + 1)), ("off_b", C.c_int32 * (BIGMLP_MAX_LAYERS + 1)),
                ("off_lns", C.c_int32 * BIGMLP_MAX_LAYERS), ("off_lnb", C.c_int32 * BIGMLP_MAX_LAYERS), ("total", C.c_int32)]


def bigmlp_supported(obs_dim: int, hidden: int, layers: int, num_actions: int) -> bool:
    """Shapes csrc/pqn_bigmlp.hip tiles (pqn_bigmlp_layout rejects the rest)."""
    return obs_dim >= 8 and 256 <= hidden <= 2048 and hidden % 256 == 0 and 1 <= layers <= BIGMLP_MAX_LAYERS and \
        1 <= num_actions <= 32


class BigMlpKernelLayout:
    """pqn_bigmlp_layout_t + the flax-flat (networks.mlp_param_shapes order, layer_norm) <-> kernel-layout map: the same
    order with every segment start padded to 16 B.  norm_input: 0 none / 1 nn.BatchNorm / 2 BatchRenorm on the input."""

    def __init__(self, obs_dim: int, hidden: int, layers: int, num_actions: int, norm_input: int):
        lib = _lib.load()
        self.struct = BigMlpLayoutStruct()
        _lib.check(lib.pqn_bigmlp_layout(obs_dim, hidden, layers, num_actions, int(norm_input), C.byref(self.struct)),
                   "pqn_bigmlp_layout")
        s = self.struct
        self.d, self.h, self.layers, self.a, self.total = int(s.d), int(s.h), int(s.layers), int(s.a), int(s.total)
        self.norm_input = int(s.norm_input)
        segs = [(int(s.off_in_scale), self.d), (int(s.off_in_bias), self.d)]
        kin = self.d
        for l in range(self.layers):
            segs += [(int(s.off_w[l]), kin * self.h), (int(s.off_b[l]), self.h), (int(s.off_lns[l]), self.h),
                     (int(s.off_lnb[l]), self.h)]
            kin = self.h
        segs += [(int(s.off_w[self.layers]), self.h * self.a), (int(s.off_b[self.layers]), self.a)]
        self.segments = []          # (flax offset, kernel offset, length)
        off = 0
        for koff, n in segs:
            self.segments.append((off, koff, n))
            off += n
        self.num_flax = off

    def to_kernel(self, theta_flax: torch.Tensor) -> torch.Tensor:
        out = torch.zeros(self.total, dtype=torch.float32, device=theta_flax.device)
        for foff, koff, n in self.segments:
            out[koff:koff + n] = theta_flax[foff:foff + n]
        return out

    def to_flax(self, theta_k: torch.Tensor) -> torch.Tensor:
        return torch.cat([theta_k[koff:koff + n] for _f, koff, n in self.segments])


class BigMlpTrainer:
    """Parameters, RAdam state, running input statistics and workspace of one seed for the wide MLP;
    value_and_grad(_loss_fn) of pqn_craftax.py:277-312 = pqn_bigmlp_grad, apply_gradients = pqn_radam_clip_step."""

    def __init__(self, layout: BigMlpKernelLayout, theta_flax: torch.Tensor, lr: float, max_grad_norm: float,
                 lr_decay_steps: float = 0.0, lr_end: float = 1e-20):
        dev = theta_flax.device
        self.layout = layout
        self.theta = layout.to_kernel(theta_flax)
        self.grad = torch.zeros_like(self.theta)
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.scratch = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.lr, self.lr_end, self.lr_steps, self.max_norm = float(lr), float(lr_end), float(lr_decay_steps), float(max_grad_norm)
        # batch_stats of the input normalisation: running mean 0 / var 1, BatchRenorm's step counter
        self.in_mean = torch.zeros(layout.d, dtype=torch.float32, device=dev)
        self.in_var = torch.ones(layout.d, dtype=torch.float32, device=dev)
        self.in_steps = torch.zeros(2, dtype=torch.int32, device=dev)   # [train-call counter, kernel scratch]
        self._ws = {}
        # bf16 operand planes of the Dense kernels (hi / mid / lo, natural + transposed), refreshed after every step
        n_wp = int(_lib.load().pqn_bigmlp_weight_plane_floats(C.byref(layout.struct)))
        self.wplanes = torch.zeros(n_wp, dtype=torch.float32, device=dev)
        self.refresh_planes()

    def refresh_planes(self):
        _lib.check(_lib.load().pqn_bigmlp_refresh_planes(C.byref(self.layout.struct), _lib.ptr(self.theta), _lib.ptr(self.wplanes),
                                                         _lib.stream_ptr()), "pqn_bigmlp_refresh_planes")

    def _workspace(self, rows: int, nb: int) -> torch.Tensor:
        sid = _lib.stream_ptr()   # one workspace per stream: seeds may run as concurrent streams
        n = int(_lib.load().pqn_bigmlp_workspace_floats(C.byref(self.layout.struct), rows, nb))
        if n < 0:
            raise RuntimeError("pqn_bigmlp_workspace_floats failed")
        ws = self._ws.get(sid)
        if ws is None or ws.numel() < n:
            ws = self._ws[sid] = torch.empty(n, dtype=torch.float32, device=self.theta.device)
        return ws

    def intermediate(self, rows: int, nb: int, what: str, layer: int = 0) -> torch.Tensor:
        """A forward intermediate of the last forward / compute_grad call on this stream (pqn_bigmlp_workspace_view), as an
        f32 tensor: what = "xn" | "z" | "h" | "stat" | "q" (plane triples are summed back: hi + mid + lo is exact)."""
        off, ld = C.c_int64(0), C.c_int64(0)
        code = {"xn": 0, "z": 1, "h": 2, "stat": 3, "q": 4}[what]
        _lib.check(_lib.load().pqn_bigmlp_workspace_view(C.byref(self.layout.struct), rows, nb, code, layer, C.addressof(off),
                                                         C.addressof(ld)), "pqn_bigmlp_workspace_view")
        ws = self._ws[_lib.stream_ptr()]
        if code in (0, 2):
            # fragment-major planes (csrc/pqn_bigmlp.hip, bm_slot): [plane][row block of 16][K block of 32][slot 4 r + (kb ^ (-(r >> 2) & 3))][8]
            rp, k = (rows + 15) // 16 * 16, int(ld.value)
            blk = ws.view(torch.bfloat16)[off.value:off.value + 3 * rp * k].view(3, rp // 16, k // 32, 16, 4, 8)
            r = torch.arange(16, device=blk.device)
            rows_of = torch.stack([blk[:, :, :, r, kb ^ ((-(r >> 2)) & 3), :] for kb in range(4)], dim=3)   # [3][rb][kblk][kb][r][8]
            planes = rows_of.permute(0, 1, 4, 2, 3, 5).reshape(3, rp, k)[:, :rows]
            return planes[0].float() + planes[1].float() + planes[2].float()
        return ws[off.value:off.value + rows * ld.value].view(rows, ld.value)

    def forward(self, obs: torch.Tensor, *, want_q: bool = True, eps: Optional[float] = None, key: int = 0, q=None,
                action=None, qmax=None):
        lib = _lib.load()
        n = int(obs.shape[0])
        dev = obs.device
        assert obs.dtype == torch.float32 and obs.is_contiguous() and obs.shape[1] == self.layout.d
        if want_q and q is None:
            q = torch.empty((n, self.layout.a), dtype=torch.float32, device=dev)
        if eps is not None and action is None:
            action = torch.empty(n, dtype=torch.int32, device=dev)
        if (eps is not None or not want_q) and qmax is None:
            qmax = torch.empty(n, dtype=torch.float32, device=dev)
        ws = self._workspace(n, n)
        use_stats = self.layout.norm_input != 0
        _lib.check(lib.pqn_bigmlp_forward(C.byref(self.layout.struct), n, _lib.ptr(obs), _lib.ptr(self.theta), _lib.ptr(self.wplanes),
                                          _lib.ptr(self.in_mean) if use_stats else None,
                                          _lib.ptr(self.in_var) if use_stats else None, _lib.ptr(ws), _lib.ptr(q),
                                          _lib.ptr(action) if eps is not None else None, _lib.ptr(qmax), float(eps or 0.0), key,
                                          None, None, _lib.stream_ptr()), "pqn_bigmlp_forward")
        return q, action, qmax

    def compute_grad(self, idx: torch.Tensor, obs_flat: torch.Tensor, action: torch.Tensor, *, target=None, reward=None,
                     done=None, gamma: float = 0.0, next_offset: int = 0, loss_out=None, qv_out=None):
        lib = _lib.load()
        nb = int(idx.numel())
        assert idx.dtype == torch.int64 and action.dtype == torch.int32 and obs_flat.dtype == torch.float32
        assert obs_flat.is_contiguous() and obs_flat.shape[1] == self.layout.d
        if done is not None and done.dtype == torch.bool:
            done = done.view(torch.uint8)
        ws = self._workspace(2 * nb if next_offset > 0 else nb, nb)
        use_stats = self.layout.norm_input != 0
        _lib.check(lib.pqn_bigmlp_grad(C.byref(self.layout.struct), nb, _lib.ptr(idx), _lib.ptr(obs_flat), int(next_offset),
                                       _lib.ptr(action), _lib.ptr(target), _lib.ptr(reward), _lib.ptr(done), float(gamma),
                                       _lib.ptr(self.theta), _lib.ptr(self.wplanes), _lib.ptr(self.in_mean) if use_stats else None,
                                       _lib.ptr(self.in_var) if use_stats else None,
                                       _lib.ptr(self.in_steps) if use_stats else None, _lib.ptr(self.grad), _lib.ptr(ws),
                                       _lib.ptr(loss_out), _lib.ptr(qv_out), _lib.stream_ptr()), "pqn_bigmlp_grad")
        return self.grad

    def apply(self):
        lib = _lib.load()
        _lib.check(lib.pqn_radam_clip_step(_lib.ptr(self.theta), _lib.ptr(self.grad), _lib.ptr(self.m), _lib.ptr(self.v),
                                           self.theta.numel(), _lib.ptr(self.count), self.lr, self.lr_end, self.lr_steps,
                                           self.max_norm, _lib.ptr(self.scratch), _lib.ptr(self.gnorm), _lib.stream_ptr()),
                   "pqn_radam_clip_step")
        self.refresh_planes()

    def theta_flax(self) -> torch.Tensor:
        return self.layout.to_flax(self.theta)


class BigMlpUpdateArgs(C.Structure):
    """pqn_bigmlp_update_args_t (include/pqn_hotpath.h)"""
    _fields_ = ([(n, C.c_int32) for n in ("env_id", "num_envs", "num_steps", "num_minibatches", "num_epochs",
                                           "metrics_capacity", "reset_ratio", "q_lambda", "done_weighted_info")] +
                [(n, C.c_float) for n in ("gamma", "lam", "rew_scale", "eps_start", "eps_finish",
                                          "lr_init", "lr_end", "max_grad_norm")] +
                [(n, C.c_double) for n in ("eps_decay_steps", "lr_steps")] +
                [(n, C.c_uint64) for n in ("key_roll", "key_shuf", "sort_temp_bytes")] +
                [("layout", BigMlpLayoutStruct)] +
                [(n, C.c_void_p) for n in ("clock", "sched_keys", "sched_eps", "state", "obs", "action", "reward",
                                           "done", "qmax", "discount", "rer", "rel", "ts", "target", "last_q",
                                           "sort_keys_in", "sort_keys_out", "sort_temp", "opt_scratch", "slot_scratch", "theta",
                                           "wplanes",
                                           "grad", "m", "v", "count", "in_mean", "in_var", "in_steps", "workspace",
                                           "radam_scratch", "loss_buf", "qv_buf", "metrics", "achievements", "ach_metrics")])


class BigMlpUpdateDriver:
    """Whole-update enqueue of the Craftax script's loop (pqn_bigmlp_update) with hipGraph replay: the wide-MLP twin of
    UpdateDriver.  `reset_ratio` > 0 = OptimisticResetVecEnvWrapper, 0 = BatchEnvWrapper; `q_lambda` False = the 1-step
    loss on concat(obs, next_obs); `done_weighted_info` = the Craftax script's info means (pqn_craftax.py:364-369)."""

    def __init__(self, env_id, n, t, mb, epochs, cfg, keys, trainer: "BigMlpTrainer", ro, words, num_updates, *,
                 reset_ratio: int, q_lambda: bool, done_weighted_info: bool, use_graph: bool = True,
                 log_achievements: bool = False):
        lib = _lib.load()
        dev = trainer.theta.device
        self.dev = dev
        tn = n * t
        b = tn // mb
        self.clock = torch.zeros(4, dtype=torch.int32, device=dev)
        self.sched_keys = torch.zeros(t + epochs, dtype=torch.int64, device=dev)
        self.sched_eps = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sk_in = torch.empty(tn, dtype=torch.int64, device=dev)
        self.sk_out = torch.empty(tn, dtype=torch.int64, device=dev)
        tb = int(lib.pqn_update_sort_temp_bytes(tn))
        if tb < 0:
            raise RuntimeError("pqn_update_sort_temp_bytes failed")
        self.sort_temp = torch.empty(max(tb, 16), dtype=torch.uint8, device=dev)
        self.opt_scratch = torch.empty(n, dtype=torch.int64, device=dev)
        self.slot_scratch = torch.empty(n, dtype=torch.int32, device=dev)
        self.loss_buf = torch.zeros(mb * epochs, dtype=torch.float32, device=dev)
        self.qv_buf = torch.zeros(mb * epochs, dtype=torch.float32, device=dev)
        self.metrics = torch.zeros((max(num_updates, 1), len(METRIC_NAMES)), dtype=torch.float64, device=dev)
        ls = C.byref(trainer.layout.struct)
        rows = b if q_lambda else 2 * b
        n_ws = max(int(lib.pqn_bigmlp_workspace_floats(ls, n, n)), int(lib.pqn_bigmlp_workspace_floats(ls, rows, b)))
        if n_ws <= 0:
            raise RuntimeError("pqn_bigmlp_workspace_floats failed")
        self.ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        a = BigMlpUpdateArgs()
        a.env_id, a.num_envs, a.num_steps, a.num_minibatches, a.num_epochs = env_id, n, t, mb, epochs
        a.metrics_capacity = self.metrics.shape[0]
        a.reset_ratio, a.q_lambda, a.done_weighted_info = int(reset_ratio), int(bool(q_lambda)), int(bool(done_weighted_info))
        a.gamma, a.lam, a.rew_scale = cfg["gamma"], cfg["lam"], cfg["rew_scale"]
        a.eps_start, a.eps_finish, a.eps_decay_steps = cfg["eps_start"], cfg["eps_finish"], cfg["eps_decay_steps"]
        a.lr_init, a.lr_end, a.lr_steps, a.max_grad_norm = trainer.lr, trainer.lr_end, trainer.lr_steps, trainer.max_norm
        a.key_roll, a.key_shuf = keys
        a.sort_temp_bytes = tb
        a.layout = trainer.layout.struct
        p = _lib.ptr
        a.clock, a.sched_keys, a.sched_eps = p(self.clock), p(self.sched_keys), p(self.sched_eps)
        a.state, a.obs = p(words), p(ro.obs)
        a.action, a.reward, a.done, a.qmax = p(ro.action), p(ro.reward), p(ro.done), p(ro.qmax)
        a.discount, a.rer, a.rel, a.ts = p(ro.discount), p(ro.rer), p(ro.rel), p(ro.ts)
        a.target, a.last_q = p(ro.target), p(ro.last_q)
        a.sort_keys_in, a.sort_keys_out, a.sort_temp = p(self.sk_in), p(self.sk_out), p(self.sort_temp)
        a.opt_scratch, a.slot_scratch = p(self.opt_scratch), p(self.slot_scratch)
        a.theta, a.wplanes, a.grad, a.m, a.v = p(trainer.theta), p(trainer.wplanes), p(trainer.grad), p(trainer.m), p(trainer.v)
        a.count = p(trainer.count)
        if trainer.layout.norm_input:
            a.in_mean, a.in_var, a.in_steps = p(trainer.in_mean), p(trainer.in_var), p(trainer.in_steps)
        a.workspace, a.radam_scratch = p(self.ws), p(trainer.scratch)
        a.loss_buf, a.qv_buf, a.metrics = p(self.loss_buf), p(self.qv_buf), p(self.metrics)
        self.ach_buf = self.ach_metrics = None
        if log_achievements:   # LOG_ACHIEVEMENTS: the 22 done-weighted Achievements/<name> columns, reduced on the device
            self.ach_buf = torch.zeros((t, n), dtype=torch.int32, device=dev)
            self.ach_metrics = torch.zeros((max(num_updates, 1), 32), dtype=torch.float64, device=dev)
            a.achievements, a.ach_metrics = p(self.ach_buf), p(self.ach_metrics)
        self.args = a
        self._keep = (trainer, ro, words)
        self.use_graph = use_graph
        self.graph = None
        self.graph_error = None
        self.calls = 0

    def _enqueue(self):
        _lib.check(_lib.load().pqn_bigmlp_update(C.byref(self.args), _lib.stream_ptr()), "pqn_bigmlp_update")

    update = UpdateDriver.update
