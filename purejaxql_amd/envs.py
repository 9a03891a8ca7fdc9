"""gymnax-shaped environment surface over the HIP env kernels.

Mirrors what make_train consumes from gymnax (reference
purejaxql/pqn_minatar.py:103-112,151,157,333; purejaxql/pqn_gymnax.py:92-104):

    env, env_params = make("Breakout-MinAtar")
    env = LogWrapper(env)                       # gymnax.wrappers.purerl.LogWrapper
    obs, state = env.reset(key, env_params, num_envs=N)
    obs, state, reward, done, info = env.step(key, state, action, env_params)
    env.action_space(env_params).n ; env.observation_space(env_params).shape
    env_params.max_steps_in_episode

Differences from gymnax, by design (MI355X-first):
  * calls are BATCHED (leading [N]) instead of vmapped per env; `key` is a
    uint64 threefry key, element e draws threefry(key, (e, stream));
  * `state` is an EnvState holding one [W, N] int32 tensor (SoA words in HBM);
    step() is functional (returns a new EnvState) unless inplace=True;
  * tensors live on the GPU; every call only enqueues kernels on torch's
    current stream (graph-capturable), nothing synchronises.
The LogWrapper record is fused into the step kernel (see csrc/pqn_env.hip).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib


@dataclass(frozen=True)
class EnvParams:
    max_steps_in_episode: int


@dataclass(frozen=True)
class Discrete:
    n: int


@dataclass(frozen=True)
class Box:
    shape: Tuple[int, ...]


@dataclass
class EnvState:
    """Opaque batched env state: words[w, e] (int32 bit patterns), see pqn_env_spec()."""
    words: torch.Tensor

    @property
    def num_envs(self) -> int:
        return self.words.shape[1]


CRAFTAX_CLASSIC_ACHIEVEMENTS = (
    "collect_coal", "collect_diamond", "collect_drink", "collect_iron", "collect_sapling", "collect_stone", "collect_wood",
    "defeat_skeleton", "defeat_zombie", "eat_cow", "eat_plant", "make_iron_pickaxe", "make_iron_sword", "make_stone_pickaxe",
    "make_stone_sword", "make_wood_pickaxe", "make_wood_sword", "place_furnace", "place_plant", "place_stone", "place_table",
    "wake_up")


class Environment:
    """Batched functional environment backed by pqn_env_reset / pqn_env_step."""

    def __init__(self, name: str, device: Optional[torch.device] = None):
        lib = _lib.load()
        env_id = lib.pqn_env_id(name.encode())
        if env_id < 0 and name.startswith("Craftax") and name != "Craftax-Classic-Symbolic-v1":
            # config/alg/pqn_craftax.yaml's default (full Craftax) is kept as the reference states it but is not built
            raise ValueError(f"make({name!r}): only the Craftax-Classic symbolic env is implemented -- pass "
                             "alg.ENV_NAME=Craftax-Classic-Symbolic-v1 (BASELINE.json configs[4])")
        _lib.check(min(env_id, 0), f"make({name!r})")
        self.name = name
        self.env_id = env_id
        spec = _lib.EnvSpec()
        _lib.check(lib.pqn_env_spec(env_id, C.byref(spec)), "pqn_env_spec")
        self.spec = spec
        d = tuple(int(x) for x in spec.obs_dim)
        self.obs_shape = d if d[1] > 0 else (d[0],)
        self.obs_size = int(spec.obs_size)
        self.obs_words = int(spec.obs_words)
        self.state_words = int(spec.state_words)
        self.num_actions = int(spec.num_actions)
        self.default_params = EnvParams(max_steps_in_episode=int(spec.max_steps))
        # bit k of the mask pqn_step_out_t.achievements = achievement k of this tuple: ALPHABETICAL order (this build's own
        # numbering, shared by the kernel, the oracle and the metric names -- NOT the package's Achievement enum values, which
        # this build never exchanges with anything)
        self.achievement_names = CRAFTAX_CLASSIC_ACHIEVEMENTS if name == "Craftax-Classic-Symbolic-v1" else ()
        self.device = torch.device(device) if device is not None else torch.device("cuda")

    # -- spaces ----------------------------------------------------------------
    def action_space(self, params=None) -> Discrete:
        return Discrete(self.num_actions)

    def observation_space(self, params=None) -> Box:
        return Box(self.obs_shape)

    # -- reset / step ------------------------------------------------------------
    def _check_params(self, params):
        """The episode limit is compiled into the transition kernels (Env::MAX_STEPS): a non-default EnvParams is
        rejected loudly instead of being silently ignored."""
        if params is not None and int(params.max_steps_in_episode) != self.default_params.max_steps_in_episode:
            raise ValueError(f"{self.name}: max_steps_in_episode={params.max_steps_in_episode} is not supported "
                             f"(the kernels are built for {self.default_params.max_steps_in_episode})")

    def _alloc_obs(self, n, want_obs, want_bits):
        obs = torch.empty((n, *self.obs_shape), dtype=torch.float32, device=self.device) if want_obs else None
        bits = None
        if want_bits:
            if self.obs_words == 0:
                raise ValueError(f"{self.name} has no packed observation")
            bits = torch.empty((n, self.obs_words), dtype=torch.int32, device=self.device)
        return obs, bits

    def reset(self, key: int, params: Optional[EnvParams] = None, num_envs: int = 1, *, want_obs: bool = True,
              want_bits: bool = False):
        lib = _lib.load()
        self._check_params(params)
        n = int(num_envs)
        words = torch.empty((self.state_words, n), dtype=torch.int32, device=self.device)
        obs, bits = self._alloc_obs(n, want_obs, want_bits)
        _lib.check(lib.pqn_env_reset(self.env_id, n, key, _lib.ptr(words), _lib.ptr(obs), _lib.ptr(bits),
                                     _lib.stream_ptr()), "pqn_env_reset")
        state = EnvState(words)
        if want_bits:
            return (obs, bits), state
        return obs, state

    def step(self, key: int, state: EnvState, action: torch.Tensor, params: Optional[EnvParams] = None, *,
             want_obs: bool = True, want_bits: bool = False, inplace: bool = False, log_info: bool = False):
        lib = _lib.load()
        self._check_params(params)
        n = state.num_envs
        if action.dtype != torch.int32:
            action = action.to(torch.int32)
        if action.numel() != n or not action.is_contiguous():
            raise ValueError(f"action must be a contiguous [{n}] tensor")
        dev = self.device
        src_words = state.words
        new_words = state.words if inplace else torch.empty_like(state.words)   # functional by default, like gymnax
        obs, bits = self._alloc_obs(n, want_obs, want_bits)
        reward = torch.empty(n, dtype=torch.float32, device=dev)
        done = torch.empty(n, dtype=torch.uint8, device=dev)
        discount = torch.empty(n, dtype=torch.float32, device=dev)
        out = _lib.StepOut(obs=_lib.ptr(obs), obs_bits=_lib.ptr(bits), reward=_lib.ptr(reward), done=_lib.ptr(done),
                           discount=_lib.ptr(discount))
        info = {"discount": discount}
        if log_info:
            rer = torch.empty(n, dtype=torch.float32, device=dev)
            rel = torch.empty(n, dtype=torch.int32, device=dev)
            ts = torch.empty(n, dtype=torch.int32, device=dev)
            out.returned_episode_returns = _lib.ptr(rer)
            out.returned_episode_lengths = _lib.ptr(rel)
            out.timestep = _lib.ptr(ts)
        ach = None
        if self.achievement_names:      # Craftax: the achievement mask of the episodes that end with this step
            ach = torch.empty(n, dtype=torch.int32, device=dev)
            out.achievements = _lib.ptr(ach)
        _lib.check(lib.pqn_env_step(self.env_id, n, key, _lib.ptr(src_words), _lib.ptr(new_words),
                                    _lib.ptr(action), C.byref(out), _lib.stream_ptr()), "pqn_env_step")
        done_b = done.view(torch.bool)
        if log_info:
            info["returned_episode_returns"] = rer
            info["returned_episode_lengths"] = rel
            info["timestep"] = ts
            info["returned_episode"] = done_b
        if ach is not None:
            info["achievements"] = ach
        new_state = EnvState(new_words)
        if want_bits:
            return (obs, bits), new_state, reward, done_b, info
        return obs, new_state, reward, done_b, info

    # -- canonical state (tests / checkpoints) -----------------------------------
    def export_state(self, state: EnvState):
        lib = _lib.load()
        n = state.num_envs
        si = torch.empty((n, int(self.spec.canon_si)), dtype=torch.int32, device=self.device)
        sf = torch.empty((n, max(int(self.spec.canon_sf), 1)), dtype=torch.float32, device=self.device)
        log = torch.empty((n, 5), dtype=torch.int32, device=self.device)
        _lib.check(lib.pqn_env_export_state(self.env_id, n, _lib.ptr(state.words), _lib.ptr(si), _lib.ptr(sf),
                                            _lib.ptr(log), _lib.stream_ptr()), "pqn_env_export_state")
        return si, sf[:, :int(self.spec.canon_sf)], log

    def import_state(self, si: torch.Tensor, sf: Optional[torch.Tensor] = None, log: Optional[torch.Tensor] = None):
        lib = _lib.load()
        n = si.shape[0]
        si = si.to(self.device, torch.int32).contiguous()
        sf = sf.to(self.device, torch.float32).contiguous() if sf is not None and sf.numel() else None
        log = log.to(self.device, torch.int32).contiguous() if log is not None else None
        words = torch.empty((self.state_words, n), dtype=torch.int32, device=self.device)
        _lib.check(lib.pqn_env_import_state(self.env_id, n, _lib.ptr(si), _lib.ptr(sf), _lib.ptr(log),
                                            _lib.ptr(words), _lib.stream_ptr()), "pqn_env_import_state")
        return EnvState(words)


class GymnaxWrapper:
    """Base wrapper: proxies attribute access (utils/craftax_wrappers.py:10-18)."""

    def __init__(self, env):
        self._env = env

    def __getattr__(self, name):
        return getattr(self._env, name)


class LogWrapper(GymnaxWrapper):
    """Episode return/length logging (utils/craftax_wrappers.py:161-200).  The
    record itself is maintained inside the step kernel; this wrapper turns on
    the info keys returned_episode_returns/_lengths, timestep, returned_episode."""

    def __init__(self, env):
        super().__init__(env)

    def reset(self, key, params=None, num_envs=1, **kw):
        return self._env.reset(key, params, num_envs, **kw)

    def step(self, key, state, action, params=None, **kw):
        return self._env.step(key, state, action, params, log_info=True, **kw)


class FlattenObservationWrapper(GymnaxWrapper):
    """obs.reshape(-1) per env (gymnax purerl wrapper; in-tree twin
    utils/brax_wrappers.py:40-75), applied at pqn_gymnax.py:93."""

    def observation_space(self, params=None) -> Box:
        shape = self._env.observation_space(params).shape
        size = 1
        for s in shape:
            size *= s
        return Box((size,))

    def reset(self, key, params=None, num_envs=1, **kw):
        obs, state = self._env.reset(key, params, num_envs, **kw)
        return obs.reshape(obs.shape[0], -1), state

    def step(self, key, state, action, params=None, **kw):
        obs, state, reward, done, info = self._env.step(key, state, action, params, **kw)
        return obs.reshape(obs.shape[0], -1), state, reward, done, info


class BatchEnvWrapper(GymnaxWrapper):
    """utils/craftax_wrappers.py:21-45: reset / step of `num_envs` envs with ONE key (the reference vmaps the wrapped
    env over split keys; here the env kernels are batched already and element e draws threefry(key, (e, stream)))."""

    def __init__(self, env, num_envs: int):
        super().__init__(env)
        self.num_envs = int(num_envs)

    def reset(self, rng, params=None, **kw):
        return self._env.reset(rng, params, self.num_envs, **kw)

    def step(self, rng, state, action, params=None, **kw):
        return self._env.step(rng, state, action, params, **kw)


class OptimisticResetVecEnvWrapper(GymnaxWrapper):
    """utils/craftax_wrappers.py:83-148: every env steps, only num_envs / reset_ratio fresh states are generated per
    step and handed to the envs that finished (pqn_env_step_optimistic).  Wrap LogWrapper(env), as pqn_craftax.py:99-108
    does: the LogWrapper record of a finished env then restarts from zero with the rest of its state."""

    def __init__(self, env, num_envs: int, reset_ratio: int):
        super().__init__(env)
        self.num_envs, self.reset_ratio = int(num_envs), int(reset_ratio)
        assert self.num_envs % self.reset_ratio == 0, "Reset ratio must perfectly divide num envs."   # (:96-98)
        self.num_resets = self.num_envs // self.reset_ratio
        self._scratch = {}   # sort-key scratch per HIP stream: seeds running as concurrent streams share this wrapper

    def reset(self, rng, params=None, **kw):
        return self._env.reset(rng, params, self.num_envs, **kw)

    def step(self, rng, state, action, params=None, *, want_obs: bool = True, want_bits: bool = False,
             inplace: bool = False, want_slots: bool = False):
        base = self._env
        log_info = flatten = False
        while not isinstance(base, Environment):
            log_info = log_info or isinstance(base, LogWrapper)
            flatten = flatten or isinstance(base, FlattenObservationWrapper)
            base = base._env
        lib = _lib.load()
        n, dev = state.num_envs, base.device
        if n != self.num_envs:
            raise ValueError(f"state holds {n} envs, the wrapper was built for {self.num_envs}")
        if action.dtype != torch.int32:
            action = action.to(torch.int32)
        sid = _lib.stream_ptr()
        scratch = self._scratch.get(sid)
        if scratch is None or scratch.device != state.words.device:
            scratch = self._scratch[sid] = torch.empty(n, dtype=torch.int64, device=dev)
        src_words = state.words
        new_words = state.words if inplace else torch.empty_like(state.words)
        obs, bits = base._alloc_obs(n, want_obs, want_bits)
        reward = torch.empty(n, dtype=torch.float32, device=dev)
        done = torch.empty(n, dtype=torch.uint8, device=dev)
        discount = torch.empty(n, dtype=torch.float32, device=dev)
        out = _lib.StepOut(obs=_lib.ptr(obs), obs_bits=_lib.ptr(bits), reward=_lib.ptr(reward), done=_lib.ptr(done),
                           discount=_lib.ptr(discount))
        info = {"discount": discount}
        if log_info:
            rer = torch.empty(n, dtype=torch.float32, device=dev)
            rel = torch.empty(n, dtype=torch.int32, device=dev)
            ts = torch.empty(n, dtype=torch.int32, device=dev)
            out.returned_episode_returns, out.returned_episode_lengths, out.timestep = _lib.ptr(rer), _lib.ptr(rel), _lib.ptr(ts)
        slots = torch.empty(n, dtype=torch.int32, device=dev) if want_slots else None
        if base.achievement_names:
            info["achievements"] = torch.empty(n, dtype=torch.int32, device=dev)
            out.achievements = _lib.ptr(info["achievements"])
        _lib.check(lib.pqn_env_step_optimistic(base.env_id, n, rng, self.reset_ratio, _lib.ptr(src_words),
                                               _lib.ptr(new_words), _lib.ptr(action), C.byref(out),
                                               _lib.ptr(scratch), _lib.ptr(slots), sid),
                   "pqn_env_step_optimistic")
        done_b = done.view(torch.bool)
        if log_info:
            info.update(returned_episode_returns=rer, returned_episode_lengths=rel, timestep=ts, returned_episode=done_b)
        if want_slots:
            info["reset_slot"] = slots
        new_state = EnvState(new_words)
        if flatten and obs is not None:
            obs = obs.reshape(n, -1)
        if want_bits:
            return (obs, bits), new_state, reward, done_b, info
        return obs, new_state, reward, done_b, info


def make(env_name: str, device=None, **env_kwargs):
    """gymnax.make(name) -> (env, env_params)  (pqn_minatar.py:103).  ENV_KWARGS other than {} are rejected: the
    reference passes them to the env constructor, none of the envs built here has a constructor option."""
    if env_kwargs:
        raise ValueError(f"make({env_name!r}): unsupported ENV_KWARGS {sorted(env_kwargs)}")
    env = Environment(env_name, device=device)
    return env, env.default_params
