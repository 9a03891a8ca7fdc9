"""Thin torch-tensor wrappers over the C ABI (include/pqn_hotpath.h).

Each wrapper passes raw device pointers + torch's current stream; nothing here
computes on the host and nothing falls back to torch/CPU math.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib


def _req(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise ValueError(f"{name}: need contiguous CUDA tensor of {dtype}, got {t.dtype} "
                         f"contig={t.is_contiguous()} cuda={t.is_cuda}")


def eps_greedy(q: torch.Tensor, eps: float, key: int, action: Optional[torch.Tensor] = None,
               qmax: Optional[torch.Tensor] = None):
    """jax.vmap(eps_greedy_exploration)(keys, q, eps) (pqn_minatar.py:115-128,196) + max_a q."""
    lib = _lib.load()
    m, a = q.shape
    _req(q, torch.float32, "q")
    if action is None:
        action = torch.empty(m, dtype=torch.int32, device=q.device)
    if qmax is None:
        qmax = torch.empty(m, dtype=torch.float32, device=q.device)
    _lib.check(lib.pqn_eps_greedy(_lib.ptr(q), m, a, float(eps), key, _lib.ptr(action), _lib.ptr(qmax),
                                  _lib.stream_ptr()), "pqn_eps_greedy")
    return action, qmax


def q_lambda(reward: torch.Tensor, done: torch.Tensor, qmax: torch.Tensor, last_q: torch.Tensor, gamma: float,
             lam: float, quirk: bool = True, target: Optional[torch.Tensor] = None):
    """Q(lambda) targets, time-major [T, M] (pqn_minatar.py:237-260)."""
    lib = _lib.load()
    t_len, m = reward.shape
    _req(reward, torch.float32, "reward")
    _req(qmax, torch.float32, "qmax")
    _req(last_q, torch.float32, "last_q")
    if done.dtype == torch.bool:
        done = done.view(torch.uint8)
    _req(done, torch.uint8, "done")
    if target is None:
        target = torch.empty_like(reward)
    _lib.check(lib.pqn_q_lambda(_lib.ptr(reward), _lib.ptr(done), _lib.ptr(qmax), _lib.ptr(last_q), float(gamma),
                                float(lam), t_len, m, 1 if quirk else 0, _lib.ptr(target), _lib.stream_ptr()),
               "pqn_q_lambda")
    return target


def shuffle_permutation(key: int, n: int, device) -> torch.Tensor:
    """jax.random.permutation(key, n) stand-in (pqn_minatar.py:299-315): argsort of
    unique threefry sort keys.  torch.sort is plumbing (rocPRIM radix sort)."""
    lib = _lib.load()
    keys = torch.empty(n, dtype=torch.int64, device=device)
    _lib.check(lib.pqn_shuffle_keys(key, n, _lib.ptr(keys), _lib.stream_ptr()), "pqn_shuffle_keys")
    return torch.sort(keys).indices


class FlatRAdam:
    """optax.chain(clip_by_global_norm(max_norm), radam(lr)) on one flat f32 buffer
    (pqn_minatar.py:140-147,159-162).  The step counter lives on the device."""

    def __init__(self, params: torch.Tensor, lr: float, max_grad_norm: float, lr_decay_steps: float = 0.0,
                 lr_end: float = 1e-20):
        _req(params, torch.float32, "params")
        self.p = params
        self.m = torch.zeros_like(params)
        self.v = torch.zeros_like(params)
        self.count = torch.zeros(1, dtype=torch.int32, device=params.device)
        self.scratch = torch.zeros(1024, dtype=torch.float32, device=params.device)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=params.device)
        self.lr, self.lr_end, self.lr_steps = float(lr), float(lr_end), float(lr_decay_steps)
        self.max_norm = float(max_grad_norm)

    def step(self, grad: torch.Tensor):
        lib = _lib.load()
        _req(grad, torch.float32, "grad")
        _lib.check(lib.pqn_radam_clip_step(_lib.ptr(self.p), _lib.ptr(grad), _lib.ptr(self.m), _lib.ptr(self.v),
                                           self.p.numel(), _lib.ptr(self.count), self.lr, self.lr_end, self.lr_steps,
                                           self.max_norm, _lib.ptr(self.scratch), _lib.ptr(self.gnorm),
                                           _lib.stream_ptr()), "pqn_radam_clip_step")
