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


def env_step_hbm_roofline(num_envs: int, device, env_name: str = "Breakout-MinAtar", steps: int = 64, reps: int = 5):
    """GB/s of the gymnax-surface env.step kernel (f32 observation out): `steps` launches captured in one
    hipGraph (no host time between kernels), HIP events around the replays.  Algorithmic bytes per env-step
    = 1,926 (state r+w 300 at natural width, action 4, f32 obs 1600, reward 4, done 1, info 17)."""
    from .envs import LogWrapper, make
    env, params = make(env_name, device=device)
    env = LogWrapper(env)
    obs, state = env.reset(0, params, num_envs)
    gen = torch.Generator(device=device)
    gen.manual_seed(1234)
    actions = torch.randint(0, env.num_actions, (steps, num_envs), dtype=torch.int32, device=device, generator=gen)
    for t in range(200):  # played-in states
        env.step(t, state, actions[t % steps], params, inplace=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(steps):
            env.step(1000 + t, state, actions[t], params, inplace=True)
    g.replay()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        g.replay()
    end.record()
    end.synchronize()
    us = start.elapsed_time(end) * 1e3 / (reps * steps)
    gbs = 1926.0 * num_envs / (us * 1e-6) / 1e9
    ws_mb = 1926.0 * num_envs / 2 ** 20   # bytes one launch touches (the obs buffer is re-used by every captured launch)
    in_mall = ws_mb < 256.0
    return {"kernel": "minatar_kernel<Breakout> (gymnax surface: f32 obs + LogWrapper info)", "num_envs": num_envs,
            "avg_launch_us": us, "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
            "env_steps_per_s": num_envs / (us * 1e-6), "working_set_mib": ws_mb,
            "level": ("Infinity Cache (the launch's working set fits the 256 MiB MALL and is re-written every launch): an "
                      "on-chip rate, NOT an HBM figure") if in_mall else "HBM (working set > 256 MiB Infinity Cache)"}
