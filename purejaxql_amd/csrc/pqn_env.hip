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

// ===========================================================================
// Breakout-MinAtar.  2 state words:
//  w0 = bricks[30] (bit (y-1)*10+x, rows 1..3) | strike<<30 | terminal<<31
//  w1 = ball_x | ball_y<<4 | dir<<8 | pos<<10 | last_x<<14 | last_y<<18 | time<<22
// ===========================================================================
struct Breakout {
  static constexpr int ENV_WORDS = 2;
  static constexpr int OBS_C = 4;
  static constexpr int OBS_SIZE = 400;
  static constexpr int OBS_WORDS = 16;  // 13 used, padded to 16 B multiple
  static constexpr int NUM_ACTIONS = 3;
  static constexpr int MAX_STEPS = 1000;
  static constexpr int CANON_SI = 109;
  static constexpr int CANON_SF = 0;
  static constexpr uint32_t FULL = 0x3FFFFFFFu;

  uint32_t bricks;
  int ball_x, ball_y, dir, pos, last_x, last_y, time;
  int strike, terminal;

  PQN_D void unpack(const uint32_t *w) {
    bricks = w[0] & FULL;
    strike = (w[0] >> 30) & 1;
    terminal = (w[0] >> 31) & 1;
    ball_x = w[1] & 15;
    ball_y = (w[1] >> 4) & 15;
    dir = (w[1] >> 8) & 3;
    pos = (w[1] >> 10) & 15;
    last_x = (w[1] >> 14) & 15;
    last_y = (w[1] >> 18) & 15;
    time = (w[1] >> 22) & 1023;
  }
  PQN_D void pack(uint32_t *w) const {
    w[0] = bricks | ((uint32_t)strike << 30) | ((uint32_t)terminal << 31);
    w[1] = (uint32_t)ball_x | ((uint32_t)ball_y << 4) | ((uint32_t)dir << 8) | ((uint32_t)pos << 10) |
           ((uint32_t)last_x << 14) | ((uint32_t)last_y << 18) | ((uint32_t)time << 22);
  }
  PQN_D void reset(uint64_t key, uint32_t e) {
    uint32_t o0, o1;
    pqn_bits(key, e, PQN_STREAM_RESET, o0, o1);
    const int start = (int)(o0 & 1u);
    bricks = FULL;
    strike = 0;
    terminal = 0;
    ball_y = 3;
    ball_x = start ? 9 : 0;
    dir = start ? 3 : 2;
    pos = 4;
    last_y = 3;
    last_x = ball_x;
    time = 0;
  }
  PQN_D bool brick_at(int y, int x) const {
    const unsigned r = (unsigned)(y - 1);
    return r < 3u && ((bricks >> (r * 10 + x)) & 1u);
  }
  // step_env; returns reward, sets done.  key unused (deterministic rules).
  PQN_D float step(int action, uint64_t, uint32_t, int &done) {
    float reward = 0.0f;
    if (action == 1) pos = max(0, pos - 1);
    else if (action == 2) pos = min(9, pos + 1);
    const int ox = ball_x, oy = ball_y;
    int nx = ox + ((dir == 1 || dir == 2) ? 1 : -1);
    int ny = oy + ((dir >= 2) ? 1 : -1);
    if (nx < 0 || nx > 9) {
      nx = nx < 0 ? 0 : 9;
      dir ^= 1;  // [1,0,3,2]
    }
    int term = 0, toggle = 0;
    if (ny < 0) {
      ny = 0;
      dir ^= 3;  // [3,2,1,0]
    } else if (brick_at(ny, nx)) {
      toggle = 1;
      if (!strike) {
        reward = 1.0f;
        bricks &= ~(1u << ((ny - 1) * 10 + nx));
        ny = oy;
        dir ^= 3;
      }
    } else if (ny == 9) {
      if (bricks == 0u) bricks = FULL;
      if (ox == pos) {
        dir ^= 3;
        ny = oy;
      } else if (nx == pos) {
        dir ^= 2;  // [2,3,0,1]
        ny = oy;
      } else {
        term = 1;
      }
    }
    strike = toggle;
    last_x = ox;
    last_y = oy;
    ball_x = nx;
    ball_y = ny;
    time += 1;
    done = term | (time >= MAX_STEPS);
    terminal = done;
    return reward;
  }
  // bit (cell*4 + c): c0 paddle, c1 ball, c2 trail, c3 brick
  PQN_D void obs_bits(uint32_t *o) const {
#pragma unroll
    for (int i = 0; i < OBS_WORDS; ++i) o[i] = 0u;
    // bricks occupy cells 10..39 -> nibble bit 3 of words 1..4
#pragma unroll
    for (int w = 1; w <= 4; ++w) {
      const int c0 = w * 8 - 10;  // first brick index covered by this word (may be negative)
      uint32_t b = c0 >= 0 ? (bricks >> c0) : (bricks << (-c0));
      b &= 0xFFu;
      b = (b | (b << 12)) & 0x000F000Fu;
      b = (b | (b << 6)) & 0x03030303u;
      b = (b | (b << 3)) & 0x11111111u;
      o[w] = b << 3;
    }
    set(o, (90 + pos) * 4 + 0);
    set(o, (ball_y * 10 + ball_x) * 4 + 1);
    set(o, (last_y * 10 + last_x) * 4 + 2);
  }
  static PQN_D void set(uint32_t *o, int bit) {
    // dynamic index into a register array -> select chain (keeps o[] in VGPRs)
#pragma unroll
    for (int i = 0; i < 13; ++i)
      if ((bit >> 5) == i) o[i] |= 1u << (bit & 31);
  }
  PQN_D void to_canon(int32_t *si, float *) const {
    si[0] = ball_y; si[1] = ball_x; si[2] = dir; si[3] = pos; si[4] = strike;
    si[5] = last_y; si[6] = last_x; si[7] = time; si[8] = terminal;
    for (int c = 0; c < 100; ++c) si[9 + c] = brick_at(c / 10, c % 10) ? 1 : 0;
  }
  PQN_D void from_canon(const int32_t *si, const float *) {
    ball_y = si[0]; ball_x = si[1]; dir = si[2]; pos = si[3]; strike = si[4];
    last_y = si[5]; last_x = si[6]; time = si[7]; terminal = si[8];
    bricks = 0u;
    for (int c = 10; c < 40; ++c)
      if (si[9 + c]) bricks |= 1u << (c - 10);
  }
};

// ===========================================================================
// CartPole-v1.  5 state words: x, x_dot, theta, theta_dot (f32 bits), time.
// ===========================================================================
struct CartPole {
  static constexpr int ENV_WORDS = 5;
  static constexpr int OBS_SIZE = 4;
  static constexpr int OBS_WORDS = 0;
  static constexpr int NUM_ACTIONS = 2;
  static constexpr int MAX_STEPS = 500;
  static constexpr int CANON_SI = 1;
  static constexpr int CANON_SF = 4;

  float x, x_dot, theta, theta_dot;
  int time;

  PQN_D void unpack(const uint32_t *w) {
    x = __uint_as_float(w[0]); x_dot = __uint_as_float(w[1]);
    theta = __uint_as_float(w[2]); theta_dot = __uint_as_float(w[3]);
    time = (int)w[4];
  }
  PQN_D void pack(uint32_t *w) const {
    w[0] = __float_as_uint(x); w[1] = __float_as_uint(x_dot);
    w[2] = __float_as_uint(theta); w[3] = __float_as_uint(theta_dot);
    w[4] = (uint32_t)time;
  }
  PQN_D void reset(uint64_t key, uint32_t e) {
    uint32_t a0, a1, b0, b1;
    pqn_bits(key, e, PQN_STREAM_RESET, a0, a1);
    pqn_bits(key, e, PQN_STREAM_RESET2, b0, b1);
    x = pqn_uniform(a0) * 0.1f - 0.05f;
    x_dot = pqn_uniform(a1) * 0.1f - 0.05f;
    theta = pqn_uniform(b0) * 0.1f - 0.05f;
    theta_dot = pqn_uniform(b1) * 0.1f - 0.05f;
    time = 0;
  }
  PQN_D int is_terminal() const {
    const float x_thr = 2.4f;
    const float th_thr = (float)(12.0 * 2.0 * 3.14159265358979323846 / 360.0);
    return (x < -x_thr) | (x > x_thr) | (theta < -th_thr) | (theta > th_thr) | (time >= MAX_STEPS);
  }
  PQN_D float step(int action, uint64_t, uint32_t, int &done) {
    const float gravity = 9.8f, masspole = 0.1f, total_mass = 1.1f, length = 0.5f;
    const float polemass_length = 0.05f, force_mag = 10.0f, tau = 0.02f;
    const int prev_terminal = is_terminal();
    const float force = force_mag * (float)action - force_mag * (float)(1 - action);
    const float costheta = cosf(theta);
    const float sintheta = sinf(theta);
    const float temp = (force + polemass_length * (theta_dot * theta_dot) * sintheta) / total_mass;
    const float thetaacc = (gravity * sintheta - costheta * temp) /
                           (length * (4.0f / 3.0f - masspole * (costheta * costheta) / total_mass));
    const float xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    const float nx = x + tau * x_dot;
    const float nxd = x_dot + tau * xacc;
    const float nt = theta + tau * theta_dot;
    const float ntd = theta_dot + tau * thetaacc;
    x = nx; x_dot = nxd; theta = nt; theta_dot = ntd;
    time += 1;
    done = is_terminal();
    return 1.0f - (float)prev_terminal;
  }
  PQN_D void obs_f32(float *o) const { o[0] = x; o[1] = x_dot; o[2] = theta; o[3] = theta_dot; }
  PQN_D void to_canon(int32_t *si, float *sf) const {
    si[0] = time; sf[0] = x; sf[1] = x_dot; sf[2] = theta; sf[3] = theta_dot;
  }
  PQN_D void from_canon(const int32_t *si, const float *sf) {
    time = si[0]; x = sf[0]; x_dot = sf[1]; theta = sf[2]; theta_dot = sf[3];
  }
};

// ===========================================================================
// LogWrapper record (utils/craftax_wrappers.py:151-200), fused into the step.
// ===========================================================================
struct LogRec {
  float ep_ret;
  int ep_len;
  float ret_ret;
  int ret_len;
  int timestep;
  PQN_D void load(const uint32_t *st, int n, int e, int base) {
    ep_ret = __uint_as_float(st[(size_t)(base + 0) * n + e]);
    ep_len = (int)st[(size_t)(base + 1) * n + e];
    ret_ret = __uint_as_float(st[(size_t)(base + 2) * n + e]);
    ret_len = (int)st[(size_t)(base + 3) * n + e];
    timestep = (int)st[(size_t)(base + 4) * n + e];
  }
  PQN_D void store(uint32_t *st, int n, int e, int base) const {
    st[(size_t)(base + 0) * n + e] = __float_as_uint(ep_ret);
    st[(size_t)(base + 1) * n + e] = (uint32_t)ep_len;
    st[(size_t)(base + 2) * n + e] = __float_as_uint(ret_ret);
    st[(size_t)(base + 3) * n + e] = (uint32_t)ret_len;
    st[(size_t)(base + 4) * n + e] = (uint32_t)timestep;
  }
  PQN_D void zero() { ep_ret = 0.f; ep_len = 0; ret_ret = 0.f; ret_len = 0; timestep = 0; }
  PQN_D void step(float reward, int done) {
    const float new_ret = ep_ret + reward;
    const int new_len = ep_len + 1;
    ep_ret = done ? 0.0f : new_ret;
    ep_len = done ? 0 : new_len;
    ret_ret = done ? new_ret : ret_ret;
    ret_len = done ? new_len : ret_len;
    timestep += 1;
  }
};

// ===========================================================================
// MinAtar kernels: EPB envs per 256-thread workgroup.  Lanes [0,EPB) run the
// transition rule; then all 256 lanes expand the packed grid from LDS to f32.
// N=4096 -> 256 workgroups at EPB=16 (one per CU).
// ===========================================================================
template <class Env, int EPB, bool IS_RESET>
__global__ __launch_bounds__(256) void minatar_kernel(int n, uint64_t key, const uint64_t *__restrict__ key_dev,
                                                      float rscale, const uint32_t *__restrict__ state_in,
                                                      uint32_t *__restrict__ state_out,
                                                      const int32_t *__restrict__ action, pqn_step_out_t out) {
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
      if (done) env.reset(key, (uint32_t)e);  // select(done, reset_env, step_env)
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
    uint32_t ob[Env::OBS_WORDS];
    env.obs_bits(ob);
#pragma unroll
    for (int i = 0; i < Env::OBS_WORDS; i += 4)
      *reinterpret_cast<uint4 *>(&s_bits[tid * Env::OBS_WORDS + i]) = make_uint4(ob[i], ob[i + 1], ob[i + 2], ob[i + 3]);
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
                                                   float rscale, const uint32_t *__restrict__ state_in,
                                                   uint32_t *__restrict__ state_out,
                                                   const int32_t *__restrict__ action, pqn_step_out_t out) {
  if (key_dev) key = *key_dev;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
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
    if (done) env.reset(key, (uint32_t)e);
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
    static_assert(Env::OBS_SIZE == 4, "float4 store assumes 4 floats");
    reinterpret_cast<float4 *>(out.obs)[e] = make_float4(o[0], o[1], o[2], o[3]);
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
  pqn_set_error("unknown or unsupported env name '%s'", name);
  return PQN_E_UNSUPPORTED;
}

extern "C" int pqn_env_spec(int env_id, pqn_env_spec_t *spec) {
  PQN_REQUIRE(spec, "pqn_env_spec: spec is NULL");
  switch (env_id) {
    case PQN_ENV_BREAKOUT: fill_spec<Breakout>(spec, 10, 10, 4); return PQN_OK;
    case PQN_ENV_CARTPOLE: fill_spec<CartPole>(spec, 4, 0, 0); return PQN_OK;
    default: pqn_set_error("pqn_env_spec: unsupported env id %d", env_id); return PQN_E_UNSUPPORTED;
  }
}

template <class Env, bool IS_RESET>
static void launch_minatar(int n, uint64_t key, const uint64_t *key_dev, float rscale, const uint32_t *si,
                           uint32_t *so, const int32_t *action, const pqn_step_out_t &out, hipStream_t st) {
  // EPB=16 keeps >=256 workgroups at N=4096; larger batches use 64 envs/WG.
  if (n <= 32768) {
    hipLaunchKernelGGL((minatar_kernel<Env, 16, IS_RESET>), dim3((n + 15) / 16), dim3(256), 0, st, n, key, key_dev, rscale,
                       si, so, action, out);
  } else {
    hipLaunchKernelGGL((minatar_kernel<Env, 64, IS_RESET>), dim3((n + 63) / 64), dim3(256), 0, st, n, key, key_dev, rscale,
                       si, so, action, out);
  }
}

template <bool IS_RESET>
static int dispatch(int env_id, int n, uint64_t key, const uint64_t *key_dev, float rscale, const uint32_t *si,
                    uint32_t *so, const int32_t *action, const pqn_step_out_t &out, hipStream_t st) {
  switch (env_id) {
    case PQN_ENV_BREAKOUT: launch_minatar<Breakout, IS_RESET>(n, key, key_dev, rscale, si, so, action, out, st); break;
    case PQN_ENV_CARTPOLE:
      PQN_REQUIRE(out.obs_bits == nullptr, "CartPole-v1 has no packed observation");
      hipLaunchKernelGGL((flat_kernel<CartPole, IS_RESET>), dim3((n + 255) / 256), dim3(256), 0, st, n, key, key_dev, rscale,
                         si, so, action, out);
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

// internal (pqn_update.hip): step key read from device memory, reward scaled at the source
int pqn_env_step_dyn(int env_id, int n, const uint64_t *key_dev, float rscale, uint32_t *state, const int32_t *action,
                     const pqn_step_out_t &out, hipStream_t st) {
  return dispatch<false>(env_id, n, 0, key_dev, rscale, state, state, action, out, st);
}

template <bool EXPORT>
static int canon(int env_id, int n, uint32_t *state, int32_t *si, float *sf, uint32_t *log, hipStream_t st) {
  PQN_REQUIRE(n > 0 && state && si, "canonical state: NULL argument or n <= 0");
  const dim3 g((n + 127) / 128), b(128);
  switch (env_id) {
    case PQN_ENV_BREAKOUT: hipLaunchKernelGGL((canon_kernel<Breakout, EXPORT>), g, b, 0, st, n, state, si, sf, log); break;
    case PQN_ENV_CARTPOLE:
      PQN_REQUIRE(sf, "CartPole-v1 canonical state needs sf");
      hipLaunchKernelGGL((canon_kernel<CartPole, EXPORT>), g, b, 0, st, n, state, si, sf, log);
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
