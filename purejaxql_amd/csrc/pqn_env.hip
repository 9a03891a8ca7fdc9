// pqn_env.hip -- vectorised environment reset/step kernels for gfx950.
//
// Stands in for vmap_reset / vmap_step over gymnax envs + LogWrapper
// (reference purejaxql/pqn_minatar.py:103-112; auto-reset semantics
// utils/craftax_wrappers.py:59-80; LogWrapper :151-200).  Env rules follow
// gymnax==0.0.6 / MinAtar (third-party; see oracle/pqn_oracle.h on parity).
//
// Layout: env state is SoA of u32 words, state[w*n + e]; one lane owns one env
// for the (integer) transition rule.  MinAtar observations are produced as a
// bit-packed grid in LDS by the owning lane and then expanded to the f32
// [n,10,10,C] tensor by the whole workgroup with 16-B coalesced stores, so the
// 1.6 KB/env observation (83 % of the step's HBM bytes) streams at full width.
#include <string.h>

#include "pqn_common.h"

#include "pqn_env_rules.h"

// ===========================================================================
// Optimistic resets (OptimisticResetVecEnvWrapper.step, utils/craftax_wrappers.py:111-148, over LogWrapper(env):
// the wrapper order of pqn_craftax.py:99-108).  Pass 1 = the step kernels below with opt_keys: every env steps, no
// reset; a finished env publishes rand31(key_ch, e) << 32 | e, the others ~0.  Pass 2 = opt_reset_kernel: a finished
// env's rank among the finished ones is its position in `being_reset` (choice(p=done, replace=False), :125-131: a
// uniformly random ordered subset); rank r < num_resets takes reset r, a later one the default slot e / reset_ratio
// (:122,132).  Reset j is a pure function reset_env(key_re, j), so the env evaluates it in place -- the N/ratio
// reset states are never materialised or gathered (:118-120,134-135).  The whole LogEnvState of a finished env
// becomes LogWrapper.reset's zeros (:165-171); info was written by pass 1 from the stepped record (:184-199).
// ===========================================================================
PQN_D uint64_t opt_choice_key(uint64_t key, uint32_t e, int done) {
  if (!done) return ~(uint64_t)0;
  uint32_t o0, o1;
  pqn_bits(pqn_fold(key, 2u), e, 0u, o0, o1);
  return ((uint64_t)(o0 >> 1) << 32) | e;
}

template <class Env, bool MINATAR>
__global__ __launch_bounds__(256) void opt_reset_kernel(int n, uint64_t key, const uint64_t *__restrict__ key_dev, int reset_ratio,
                                                        const uint64_t *__restrict__ opt_keys, uint32_t *state,
                                                        pqn_step_out_t out, int32_t *__restrict__ slot_out) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  if (key_dev) key = *key_dev;
  const uint64_t mine = opt_keys[e];
  int slot = -1;
  if (mine != ~(uint64_t)0) {
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += opt_keys[j] < mine;   // wave-uniform address: one broadcast load per step
    const int num_resets = n / reset_ratio;
    slot = rank < num_resets ? rank : e / reset_ratio;
    Env env;
    env.reset(pqn_fold(key, 1u), (uint32_t)slot);
    LogRec log;
    log.zero();
    uint32_t w[Env::ENV_WORDS];
    env.pack(w);
#pragma unroll
    for (int i = 0; i < Env::ENV_WORDS; ++i) state[(size_t)i * n + e] = w[i];
    log.store(state, n, e, Env::ENV_WORDS);
    if constexpr (MINATAR) {
      uint32_t b[Env::OBS_WORDS];
      env.obs_bits(b);
      if (out.obs_bits)
        for (int i = 0; i < Env::OBS_WORDS; ++i) out.obs_bits[(size_t)e * Env::OBS_WORDS + i] = b[i];
      if (out.obs)
        for (int i = 0; i < Env::OBS_SIZE; ++i) out.obs[(size_t)e * Env::OBS_SIZE + i] = (float)((b[i >> 5] >> (i & 31)) & 1u);
    } else {
      if (out.obs) {
        float o[Env::OBS_SIZE];
        env.obs_f32(o);
        for (int i = 0; i < Env::OBS_SIZE; ++i) out.obs[(size_t)e * Env::OBS_SIZE + i] = o[i];
      }
    }
  }
  if (slot_out) slot_out[e] = slot;
}

// ===========================================================================
// MinAtar kernels: EPB envs per 256-thread workgroup.  Lanes [0,EPB) run the
// transition rule; then all 256 lanes expand the packed grid from LDS to f32.
// N=4096 -> 256 workgroups at EPB=16 (one per CU).
// ===========================================================================
template <class Env, int EPB, bool IS_RESET>
__global__ __launch_bounds__(256) void minatar_kernel(int n, uint64_t key, const uint64_t *__restrict__ key_dev,
                                                      float rscale, const uint32_t *state_in,
                                                      uint32_t *state_out,   // may alias (in-place step)
                                                      const int32_t *__restrict__ action, pqn_step_out_t out,
                                                      uint64_t *__restrict__ opt_keys) {
  // opt_keys != NULL: the step half of OptimisticResetVecEnvWrapper.step -- no reset here, finished envs publish the
  // sort key that ranks them for the reset pass (opt_reset_kernel)
  __shared__ __attribute__((aligned(16))) uint32_t s_bits[EPB * Env::OBS_WORDS];
  if (key_dev) key = *key_dev;  // graph-replayable launches read the step key from device memory
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * EPB;
  const int e = e0 + tid;
  if (tid < EPB && e < n) {
    Env env;
    LogRec log;
    if (IS_RESET) {
      env.reset(key, (uint32_t)e);
      log.zero();
    } else {
      uint32_t w[Env::ENV_WORDS];
#pragma unroll
      for (int i = 0; i < Env::ENV_WORDS; ++i) w[i] = state_in[(size_t)i * n + e];
      env.unpack(w);
      log.load(state_in, n, e, Env::ENV_WORDS);
      int done = 0;
      const float reward = env.step(action[e], key, (uint32_t)e, done);
      log.step(reward, done);
      if (opt_keys) opt_keys[e] = opt_choice_key(key, (uint32_t)e, done);
      else if (done) env.reset(key, (uint32_t)e);  // select(done, reset_env, step_env)
      out.reward[e] = reward * rscale;
      out.done[e] = (uint8_t)done;
      if (out.discount) out.discount[e] = done ? 0.0f : 1.0f;
      if (out.returned_episode_returns) out.returned_episode_returns[e] = log.ret_ret;
      if (out.returned_episode_lengths) out.returned_episode_lengths[e] = log.ret_len;
      if (out.timestep) out.timestep[e] = log.timestep;
    }
    uint32_t w[Env::ENV_WORDS];
    env.pack(w);
#pragma unroll
    for (int i = 0; i < Env::ENV_WORDS; ++i) state_out[(size_t)i * n + e] = w[i];
    log.store(state_out, n, e, Env::ENV_WORDS);
    env.obs_bits(&s_bits[tid * Env::OBS_WORDS]);   // the owning lane fills its LDS row
  }
  __syncthreads();
  const int nloc = min(EPB, n - e0);
  if (out.obs_bits) {
    for (int j = tid; j < nloc * Env::OBS_WORDS; j += 256) out.obs_bits[(size_t)e0 * Env::OBS_WORDS + j] = s_bits[j];
  }
  if (out.obs) {
    constexpr int Q = Env::OBS_SIZE / 4;  // float4 per env
    float4 *dst = reinterpret_cast<float4 *>(out.obs + (size_t)e0 * Env::OBS_SIZE);
    for (int j = tid; j < nloc * Q; j += 256) {
      const int le = j / Q;
      const int q = j - le * Q;
      const uint32_t wv = s_bits[le * Env::OBS_WORDS + (q >> 3)];
      const uint32_t nib = wv >> ((q & 7) * 4);
      dst[j] = make_float4((float)(nib & 1u), (float)((nib >> 1) & 1u), (float)((nib >> 2) & 1u),
                           (float)((nib >> 3) & 1u));
    }
  }
}

// ===========================================================================
// Flat-observation kernels (CartPole): one lane per env, float4 obs store.
// ===========================================================================
template <class Env, bool IS_RESET>
__global__ __launch_bounds__(256) void flat_kernel(int n, uint64_t key, const uint64_t *__restrict__ key_dev,
                                                   float rscale, const uint32_t *state_in,
                                                   uint32_t *state_out,   // may alias (in-place step)
                                                   const int32_t *__restrict__ action, pqn_step_out_t out,
                                                   int n_per_seed, int key_stride, uint64_t *__restrict__ opt_keys) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  // seed batching: env e of the stacked batch is env e_rng of seed e / n_per_seed, driven by that seed's key
  const int seed = n_per_seed > 0 ? e / n_per_seed : 0;
  const int e_rng = e - seed * n_per_seed;
  if (key_dev) key = key_dev[(size_t)seed * key_stride];
  Env env;
  LogRec log;
  if (IS_RESET) {
    env.reset(key, (uint32_t)e_rng);
    log.zero();
  } else {
    uint32_t w[Env::ENV_WORDS];
#pragma unroll
    for (int i = 0; i < Env::ENV_WORDS; ++i) w[i] = state_in[(size_t)i * n + e];
    env.unpack(w);
    log.load(state_in, n, e, Env::ENV_WORDS);
    int done = 0;
    const float reward = env.step(action[e], key, (uint32_t)e_rng, done);
    log.step(reward, done);
    if (opt_keys) opt_keys[e] = opt_choice_key(key, (uint32_t)e, done);
    else if (done) env.reset(key, (uint32_t)e_rng);
    out.reward[e] = reward * rscale;
    out.done[e] = (uint8_t)done;
    if (out.discount) out.discount[e] = done ? 0.0f : 1.0f;
    if (out.returned_episode_returns) out.returned_episode_returns[e] = log.ret_ret;
    if (out.returned_episode_lengths) out.returned_episode_lengths[e] = log.ret_len;
    if (out.timestep) out.timestep[e] = log.timestep;
  }
  uint32_t w[Env::ENV_WORDS];
  env.pack(w);
#pragma unroll
  for (int i = 0; i < Env::ENV_WORDS; ++i) state_out[(size_t)i * n + e] = w[i];
  log.store(state_out, n, e, Env::ENV_WORDS);
  if (out.obs) {
    float o[Env::OBS_SIZE];
    env.obs_f32(o);
    if constexpr (Env::OBS_SIZE == 4) reinterpret_cast<float4 *>(out.obs)[e] = make_float4(o[0], o[1], o[2], o[3]);
    else {
      static_assert(Env::OBS_SIZE % 2 == 0, "float2 stores");
#pragma unroll
      for (int i = 0; i < Env::OBS_SIZE / 2; ++i)
        reinterpret_cast<float2 *>(out.obs)[(size_t)e * (Env::OBS_SIZE / 2) + i] = make_float2(o[2 * i], o[2 * i + 1]);
    }
  }
}

// canonical export / import (tests, checkpoints) -- not on the hot path
template <class Env, bool EXPORT>
__global__ void canon_kernel(int n, uint32_t *state, int32_t *si, float *sf, uint32_t *log) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  Env env;
  uint32_t w[Env::ENV_WORDS];
  if (EXPORT) {
    for (int i = 0; i < Env::ENV_WORDS; ++i) w[i] = state[(size_t)i * n + e];
    env.unpack(w);
    env.to_canon(si + (size_t)e * Env::CANON_SI, sf ? sf + (size_t)e * Env::CANON_SF : nullptr);
    if (log)
      for (int i = 0; i < PQN_LOG_WORDS; ++i) log[(size_t)e * PQN_LOG_WORDS + i] = state[(size_t)(Env::ENV_WORDS + i) * n + e];
  } else {
    env.from_canon(si + (size_t)e * Env::CANON_SI, sf ? sf + (size_t)e * Env::CANON_SF : nullptr);
    env.pack(w);
    for (int i = 0; i < Env::ENV_WORDS; ++i) state[(size_t)i * n + e] = w[i];
    for (int i = 0; i < PQN_LOG_WORDS; ++i)
      state[(size_t)(Env::ENV_WORDS + i) * n + e] = log ? log[(size_t)e * PQN_LOG_WORDS + i] : 0u;
  }
}

// ===========================================================================
// host entry points
// ===========================================================================
template <class Env>
static void fill_spec(pqn_env_spec_t *s, int h, int w, int c) {
  s->obs_dim[0] = h; s->obs_dim[1] = w; s->obs_dim[2] = c;
  s->obs_size = Env::OBS_SIZE;
  s->num_actions = Env::NUM_ACTIONS;
  s->max_steps = Env::MAX_STEPS;
  s->state_words = Env::ENV_WORDS + PQN_LOG_WORDS;
  s->obs_words = Env::OBS_WORDS;
  s->canon_si = Env::CANON_SI;
  s->canon_sf = Env::CANON_SF;
}

extern "C" int pqn_env_id(const char *name) {
  if (!name) return PQN_E_INVALID;
  if (!strcmp(name, "Breakout-MinAtar")) return PQN_ENV_BREAKOUT;
  if (!strcmp(name, "CartPole-v1")) return PQN_ENV_CARTPOLE;
  if (!strcmp(name, "Asterix-MinAtar")) return PQN_ENV_ASTERIX;
  if (!strcmp(name, "Freeway-MinAtar")) return PQN_ENV_FREEWAY;
  if (!strcmp(name, "SpaceInvaders-MinAtar")) return PQN_ENV_SPACEINVADERS;
  if (!strcmp(name, "Craftax-Classic-Symbolic-v1")) return PQN_ENV_CRAFTAX_CLASSIC;
  if (!strcmp(name, "Acrobot-v1")) return PQN_ENV_ACROBOT;
  pqn_set_error("unknown or unsupported env name '%s'", name);
  return PQN_E_UNSUPPORTED;
}

extern "C" int pqn_env_spec(int env_id, pqn_env_spec_t *spec) {
  PQN_REQUIRE(spec, "pqn_env_spec: spec is NULL");
  switch (env_id) {
    case PQN_ENV_BREAKOUT: fill_spec<Breakout>(spec, 10, 10, 4); return PQN_OK;
    case PQN_ENV_CARTPOLE: fill_spec<CartPole>(spec, 4, 0, 0); return PQN_OK;
    case PQN_ENV_ASTERIX: fill_spec<Asterix>(spec, 10, 10, 4); return PQN_OK;
    case PQN_ENV_FREEWAY: fill_spec<Freeway>(spec, 10, 10, 7); return PQN_OK;
    case PQN_ENV_SPACEINVADERS: fill_spec<SpaceInvaders>(spec, 10, 10, 6); return PQN_OK;
    case PQN_ENV_CRAFTAX_CLASSIC: pqn_craftax_spec(spec); return PQN_OK;
    case PQN_ENV_ACROBOT: fill_spec<Acrobot>(spec, 6, 0, 0); return PQN_OK;
    default: pqn_set_error("pqn_env_spec: unsupported env id %d", env_id); return PQN_E_UNSUPPORTED;
  }
}

template <class Env, bool IS_RESET>
static void launch_minatar(int n, uint64_t key, const uint64_t *key_dev, float rscale, const uint32_t *si,
                           uint32_t *so, const int32_t *action, const pqn_step_out_t &out, hipStream_t st,
                           uint64_t *opt_keys = nullptr) {
  // EPB=16 keeps >=256 workgroups at N=4096; larger batches use 64 envs/WG.
  if (n <= 32768) {
    hipLaunchKernelGGL((minatar_kernel<Env, 16, IS_RESET>), dim3((n + 15) / 16), dim3(256), 0, st, n, key, key_dev, rscale,
                       si, so, action, out, opt_keys);
  } else {
    hipLaunchKernelGGL((minatar_kernel<Env, 64, IS_RESET>), dim3((n + 63) / 64), dim3(256), 0, st, n, key, key_dev, rscale,
                       si, so, action, out, opt_keys);
  }
}

// Craftax-Classic keeps its 64 x 64 map in the state (4.2 KB per env) and its kernels step it where it lies; a functional
// call (state_out != state_in) copies the state across first and steps the copy -- same results, one extra pass over the
// state, state_in left untouched.  The training loops step in place.
static int craftax_functional_copy(int n, const uint32_t *si, uint32_t *so, hipStream_t st) {
  if (si == so) return PQN_OK;
  pqn_env_spec_t sp = {};
  pqn_craftax_spec(&sp);
  if (hipMemcpyAsync(so, si, sizeof(uint32_t) * (size_t)sp.state_words * (size_t)n, hipMemcpyDeviceToDevice, st) != hipSuccess) {
    pqn_set_error("Craftax-Classic: copying the state for a functional step failed");
    return PQN_E_HIP;
  }
  return PQN_OK;
}

template <bool IS_RESET>
static int dispatch(int env_id, int n, uint64_t key, const uint64_t *key_dev, float rscale, const uint32_t *si,
                    uint32_t *so, const int32_t *action, const pqn_step_out_t &out, hipStream_t st, int n_per_seed = 0,
                    int key_stride = 0, uint64_t *opt_keys = nullptr, int32_t *cc_slots = nullptr) {
  if (n_per_seed > 0 && env_id != PQN_ENV_CARTPOLE && env_id != PQN_ENV_ACROBOT) {
    pqn_set_error("seed-batched env.step is implemented for the flat-observation envs (the MinAtar path batches seeds "
                  "inside pqn_cnn_rollout_seeds)");
    return PQN_E_UNSUPPORTED;
  }
  if (env_id == PQN_ENV_CRAFTAX_CLASSIC) {   // map-in-memory env: its own kernels, stepped in place
    PQN_REQUIRE(!opt_keys, "Craftax-Classic: use pqn_env_step_optimistic");
    if (!IS_RESET) { const int rc = craftax_functional_copy(n, si, so, st); if (rc != PQN_OK) return rc; }
    if (IS_RESET) return pqn_craftax_reset(n, key, so, out.obs, st);
    return pqn_craftax_step(n, key, key_dev, rscale, so, action, out, 0, nullptr, cc_slots, st);
  }
  switch (env_id) {
    case PQN_ENV_BREAKOUT: launch_minatar<Breakout, IS_RESET>(n, key, key_dev, rscale, si, so, action, out, st, opt_keys); break;
    case PQN_ENV_ASTERIX: launch_minatar<Asterix, IS_RESET>(n, key, key_dev, rscale, si, so, action, out, st, opt_keys); break;
    case PQN_ENV_FREEWAY: launch_minatar<Freeway, IS_RESET>(n, key, key_dev, rscale, si, so, action, out, st, opt_keys); break;
    case PQN_ENV_SPACEINVADERS:
      launch_minatar<SpaceInvaders, IS_RESET>(n, key, key_dev, rscale, si, so, action, out, st, opt_keys);
      break;
    case PQN_ENV_CARTPOLE:
      PQN_REQUIRE(out.obs_bits == nullptr, "CartPole-v1 has no packed observation");
      hipLaunchKernelGGL((flat_kernel<CartPole, IS_RESET>), dim3((n + 255) / 256), dim3(256), 0, st, n, key, key_dev, rscale,
                         si, so, action, out, n_per_seed, key_stride, opt_keys);
      break;
    case PQN_ENV_ACROBOT:
      PQN_REQUIRE(out.obs_bits == nullptr, "Acrobot-v1 has no packed observation");
      hipLaunchKernelGGL((flat_kernel<Acrobot, IS_RESET>), dim3((n + 255) / 256), dim3(256), 0, st, n, key, key_dev, rscale,
                         si, so, action, out, n_per_seed, key_stride, opt_keys);
      break;
    default: pqn_set_error("unsupported env id %d", env_id); return PQN_E_UNSUPPORTED;
  }
  return pqn_check_launch(IS_RESET ? "pqn_env_reset" : "pqn_env_step");
}

extern "C" int pqn_env_reset(int env_id, int32_t n, uint64_t key, uint32_t *state, float *obs, uint32_t *obs_bits,
                             void *stream) {
  PQN_REQUIRE(n > 0, "pqn_env_reset: n must be > 0 (got %d)", n);
  PQN_REQUIRE(state, "pqn_env_reset: state is NULL");
  pqn_step_out_t out = {};
  out.obs = obs;
  out.obs_bits = obs_bits;
  return dispatch<true>(env_id, n, key, nullptr, 1.0f, nullptr, state, nullptr, out, (hipStream_t)stream);
}

extern "C" int pqn_env_step(int env_id, int32_t n, uint64_t key, const uint32_t *state_in, uint32_t *state_out,
                            const int32_t *action, const pqn_step_out_t *out, void *stream) {
  PQN_REQUIRE(n > 0, "pqn_env_step: n must be > 0 (got %d)", n);
  PQN_REQUIRE(state_in && state_out && action && out, "pqn_env_step: NULL argument");
  PQN_REQUIRE(out->reward && out->done, "pqn_env_step: out->reward and out->done are required");
  return dispatch<false>(env_id, n, key, nullptr, 1.0f, state_in, state_out, action, *out, (hipStream_t)stream);
}

static int step_optimistic(int env_id, int n, uint64_t key, const uint64_t *key_dev, float rscale, int reset_ratio,
                           const uint32_t *state_in, uint32_t *state_out, const int32_t *action, const pqn_step_out_t &out,
                           uint64_t *scratch, int32_t *slot_out, hipStream_t st) {
  if (env_id == PQN_ENV_CRAFTAX_CLASSIC) {
    const int rc = craftax_functional_copy(n, state_in, state_out, st);
    if (rc != PQN_OK) return rc;
    return pqn_craftax_step(n, key, key_dev, rscale, state_out, action, out, reset_ratio, scratch, slot_out, st);
  }
  const int rc = dispatch<false>(env_id, n, key, key_dev, rscale, state_in, state_out, action, out, st, 0, 0, scratch);
  if (rc != PQN_OK) return rc;
  const dim3 g((n + 255) / 256), b(256);
#define OPT_RESET(ENV, MIN) hipLaunchKernelGGL((opt_reset_kernel<ENV, MIN>), g, b, 0, st, n, key, key_dev, reset_ratio, scratch, state_out, out, slot_out)
  switch (env_id) {
    case PQN_ENV_BREAKOUT: OPT_RESET(Breakout, true); break;
    case PQN_ENV_ASTERIX: OPT_RESET(Asterix, true); break;
    case PQN_ENV_FREEWAY: OPT_RESET(Freeway, true); break;
    case PQN_ENV_SPACEINVADERS: OPT_RESET(SpaceInvaders, true); break;
    case PQN_ENV_CARTPOLE: OPT_RESET(CartPole, false); break;
    case PQN_ENV_ACROBOT: OPT_RESET(Acrobot, false); break;
    default: pqn_set_error("unsupported env id %d", env_id); return PQN_E_UNSUPPORTED;
  }
#undef OPT_RESET
  return pqn_check_launch("pqn_env_step_optimistic");
}

extern "C" int pqn_env_step_optimistic(int env_id, int32_t n, uint64_t key, int32_t reset_ratio, const uint32_t *state_in,
                                       uint32_t *state_out, const int32_t *action, const pqn_step_out_t *out,
                                       uint64_t *scratch, int32_t *slot_out, void *stream) {
  PQN_REQUIRE(n > 0, "pqn_env_step_optimistic: n must be > 0 (got %d)", n);
  PQN_REQUIRE(state_in && state_out && action && out && scratch, "pqn_env_step_optimistic: NULL argument");
  PQN_REQUIRE(out->reward && out->done, "pqn_env_step_optimistic: out->reward and out->done are required");
  PQN_REQUIRE(reset_ratio > 0 && n % reset_ratio == 0,
              "pqn_env_step_optimistic: reset ratio %d must perfectly divide num envs %d", reset_ratio, n);   // :96-98
  return step_optimistic(env_id, n, key, nullptr, 1.0f, reset_ratio, state_in, state_out, action, *out, scratch, slot_out,
                         (hipStream_t)stream);
}

// internal (pqn_update.hip): the same with the step key read from device memory, stepped in place, reward scaled at the source
int pqn_env_step_optimistic_dyn(int env_id, int n, const uint64_t *key_dev, float rscale, int reset_ratio, uint32_t *state,
                                const int32_t *action, const pqn_step_out_t &out, uint64_t *scratch, int32_t *slot_scratch,
                                hipStream_t st) {
  return step_optimistic(env_id, n, 0, key_dev, rscale, reset_ratio, state, state, action, out, scratch, slot_scratch, st);
}

// internal (pqn_update.hip): step key read from device memory, reward scaled at the source
int pqn_env_step_dyn(int env_id, int n, const uint64_t *key_dev, float rscale, uint32_t *state, const int32_t *action,
                     const pqn_step_out_t &out, hipStream_t st, int n_per_seed, int key_stride, int32_t *slot_scratch) {
  return dispatch<false>(env_id, n, 0, key_dev, rscale, state, state, action, out, st, n_per_seed, key_stride, nullptr, slot_scratch);
}

template <bool EXPORT>
static int canon(int env_id, int n, uint32_t *state, int32_t *si, float *sf, uint32_t *log, hipStream_t st) {
  PQN_REQUIRE(n > 0 && state && si, "canonical state: NULL argument or n <= 0");
  if (env_id == PQN_ENV_CRAFTAX_CLASSIC) return pqn_craftax_canon(n, EXPORT ? 1 : 0, state, si, sf, log, st);
  const dim3 g((n + 127) / 128), b(128);
  switch (env_id) {
    case PQN_ENV_BREAKOUT: hipLaunchKernelGGL((canon_kernel<Breakout, EXPORT>), g, b, 0, st, n, state, si, sf, log); break;
    case PQN_ENV_ASTERIX: hipLaunchKernelGGL((canon_kernel<Asterix, EXPORT>), g, b, 0, st, n, state, si, sf, log); break;
    case PQN_ENV_FREEWAY: hipLaunchKernelGGL((canon_kernel<Freeway, EXPORT>), g, b, 0, st, n, state, si, sf, log); break;
    case PQN_ENV_SPACEINVADERS:
      hipLaunchKernelGGL((canon_kernel<SpaceInvaders, EXPORT>), g, b, 0, st, n, state, si, sf, log);
      break;
    case PQN_ENV_CARTPOLE:
      PQN_REQUIRE(sf, "CartPole-v1 canonical state needs sf");
      hipLaunchKernelGGL((canon_kernel<CartPole, EXPORT>), g, b, 0, st, n, state, si, sf, log);
      break;
    case PQN_ENV_ACROBOT:
      PQN_REQUIRE(sf, "Acrobot-v1 canonical state needs sf");
      hipLaunchKernelGGL((canon_kernel<Acrobot, EXPORT>), g, b, 0, st, n, state, si, sf, log);
      break;
    default: pqn_set_error("unsupported env id %d", env_id); return PQN_E_UNSUPPORTED;
  }
  return pqn_check_launch("pqn_env_canon");
}

extern "C" int pqn_env_export_state(int env_id, int32_t n, const uint32_t *state, int32_t *si, float *sf,
                                    uint32_t *log, void *stream) {
  return canon<true>(env_id, n, const_cast<uint32_t *>(state), si, sf, log, (hipStream_t)stream);
}

extern "C" int pqn_env_import_state(int env_id, int32_t n, const int32_t *si, const float *sf, const uint32_t *log,
                                    uint32_t *state, void *stream) {
  return canon<false>(env_id, n, state, const_cast<int32_t *>(si), const_cast<float *>(sf),
                      const_cast<uint32_t *>(log), (hipStream_t)stream);
}
