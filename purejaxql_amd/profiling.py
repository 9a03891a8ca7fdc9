"""Kernel timing helpers (HIP events on the stream the kernels are launched on)."""
from __future__ import annotations

import torch


def time_launches(fn, iters: int = 200, warmup: int = 20) -> float:
    """Average milliseconds per call of `fn` (which enqueues on torch's current stream)."""
    for _ in range(warmup):
        fn()
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) / iters


def time_env_step_kernel(num_envs: int, device, env_name: str = "Breakout-MinAtar", iters: int = 300) -> float:
    """ms per pqn_env_step launch (gymnax surface: f32 obs + info) on played-in states."""
    from .envs import LogWrapper, make
    env, params = make(env_name, device=device)
    env = LogWrapper(env)
    obs, state = env.reset(0, params, num_envs)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234)
    actions = torch.randint(0, env.num_actions, (64, num_envs), dtype=torch.int32, device=device, generator=gen)
    for t in range(200):  # play in
        obs, state, *_ = env.step(t, state, actions[t % 64], params, inplace=True)
    i = [0]

    def fn():
        i[0] += 1
        env.step(1000 + i[0], state, actions[i[0] % 64], params, inplace=True)

    return time_launches(fn, iters=iters)
