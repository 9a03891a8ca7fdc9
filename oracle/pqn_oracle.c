/*
 * pqn_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never on the
 * product path).  See pqn_oracle.h for scope and parity status.
 *
 * Every function cites the reference line range it restates
 * (paths relative to the reference tree, mttga/purejaxql @ 2025-11-14).
 * Env dynamics follow gymnax==0.0.6 (third-party, absent from the reference
 * tree; pinned at reference pyproject.toml:51) -> "parity unpinned".
 */
#include "pqn_oracle.h"
#include <stdlib.h>

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* PRNG: threefry2x32, 20 rounds (Salmon et al. 2011, Random123).  jax.random
 * (reference pqn_minatar.py:116-125,183,194,213,303) is built on the same
 * block function; the key-derivation convention below is this build's own.  */
/* ------------------------------------------------------------------------ */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

void pqn_oracle_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) {
  static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
  uint32_t ks[3] = {key[0], key[1], key[0] ^ key[1] ^ 0x1BD11BDAu};
  uint32_t x0 = ctr[0] + ks[0];
  uint32_t x1 = ctr[1] + ks[1];
  for (int g = 0; g < 5; ++g) {
    const int *rot = R[g & 1];
    for (int i = 0; i < 4; ++i) {
      x0 += x1;
      x1 = rotl32(x1, rot[i]);
      x1 ^= x0;
    }
    x0 += ks[(g + 1) % 3];
    x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
  }
  out[0] = x0;
  out[1] = x1;
}

/* key64 = (k0 << 32) | k1.  PRNGKey(seed) = (seed >> 32, seed & 0xffffffff). */
uint64_t pqn_oracle_prng_key(uint64_t seed) { return seed; }

/* fold_in(key, d) = threefry(key, (0, d)) -- the counter convention of
 * jax.random.fold_in.  Used where the reference calls jax.random.split. */
uint64_t pqn_oracle_fold_in(uint64_t key, uint32_t data) {
  uint32_t k[2] = {(uint32_t)(key >> 32), (uint32_t)key};
  uint32_t c[2] = {0u, data};
  uint32_t o[2];
  pqn_oracle_threefry2x32(k, c, o);
  return ((uint64_t)o[0] << 32) | o[1];
}

/* Per-element randomness: element `index` of a vmapped call draws
 * threefry(key, (index, stream)).  Replaces split(rng, n)[index]. */
void pqn_oracle_env_bits(uint64_t key, uint32_t index, uint32_t stream, uint32_t out[2]) {
  uint32_t k[2] = {(uint32_t)(key >> 32), (uint32_t)key};
  uint32_t c[2] = {index, stream};
  pqn_oracle_threefry2x32(k, c, out);
}

/* jax.random.uniform's bit trick: 23 mantissa bits, value in [0,1). */
float pqn_oracle_bits_to_uniform(uint32_t bits) {
  union { uint32_t u; float f; } v;
  v.u = (bits >> 9) | 0x3f800000u;
  return v.f - 1.0f;
}

/* Sort keys for the minibatch shuffle (pqn_minatar.py:299-315:
 * jax.random.permutation is sort-based).  key_i = (bits_i >> 1) << 32 | i is
 * unique, so argsort(key) is a well-defined permutation on every backend. */
void pqn_oracle_sort_keys(uint64_t key, int32_t n, int64_t *out) {
  for (int32_t i = 0; i < n; ++i) {
    uint32_t o[2];
    pqn_oracle_env_bits(key, (uint32_t)i, 0u, o);
    out[i] = (int64_t)(((uint64_t)(o[0] >> 1) << 32) | (uint32_t)i);
  }
}

/* ------------------------------------------------------------------------ */
/* Breakout-MinAtar (gymnax 0.0.6 environments/minatar/breakout.py; original
 * rules: MinAtar breakout.py).  Canonical int state, 109 words per env:
 *   [0] ball_y [1] ball_x [2] ball_dir [3] pos [4] strike [5] last_y
 *   [6] last_x [7] time [8] terminal [9..108] brick_map[y*10+x]           */
/* ------------------------------------------------------------------------ */
#define BO_SI 109
enum { BO_BALL_Y, BO_BALL_X, BO_DIR, BO_POS, BO_STRIKE, BO_LAST_Y, BO_LAST_X, BO_TIME, BO_TERM, BO_MAP };

static void breakout_reset_one(uint32_t bits, int32_t *s) {
  /* reset_env: ball_start = choice([0,1]); ball_x=[0,9][start]; dir=[2,3][start] */
  int start = (int)(bits & 1u);
  memset(s, 0, sizeof(int32_t) * BO_SI);
  s[BO_BALL_Y] = 3;
  s[BO_BALL_X] = start ? 9 : 0;
  s[BO_DIR] = start ? 3 : 2;
  s[BO_POS] = 4;
  s[BO_STRIKE] = 0;
  s[BO_LAST_Y] = 3;
  s[BO_LAST_X] = s[BO_BALL_X];
  s[BO_TIME] = 0;
  s[BO_TERM] = 0;
  for (int y = 1; y < 4; ++y)
    for (int x = 0; x < 10; ++x) s[BO_MAP + y * 10 + x] = 1;
}

static void breakout_obs_one(const int32_t *s, float *obs) {
  /* get_obs: ch0 paddle [9,pos]; ch1 ball; ch2 trail; ch3 brick_map. (10,10,4) */
  memset(obs, 0, sizeof(float) * 400);
  obs[(9 * 10 + s[BO_POS]) * 4 + 0] = 1.0f;
  obs[(s[BO_BALL_Y] * 10 + s[BO_BALL_X]) * 4 + 1] = 1.0f;
  obs[(s[BO_LAST_Y] * 10 + s[BO_LAST_X]) * 4 + 2] = 1.0f;
  for (int i = 0; i < 100; ++i)
    if (s[BO_MAP + i]) obs[i * 4 + 3] = 1.0f;
}

/* step_env: minimal action set [n,l,r] -> full-set codes [0,1,3]. */
static void breakout_step_one(int32_t *s, int32_t action, int32_t max_steps, float *reward, int *done) {
  static const int ACT[3] = {0, 1, 3};
  static const int FLIP_X[4] = {1, 0, 3, 2};
  static const int FLIP_Y[4] = {3, 2, 1, 0};
  static const int FLIP_XY[4] = {2, 3, 0, 1};
  int a = ACT[action];
  float r = 0.0f;
  /* step_agent */
  int pos = s[BO_POS];
  if (a == 1) pos = pos - 1 < 0 ? 0 : pos - 1;
  else if (a == 3) pos = pos + 1 > 9 ? 9 : pos + 1;
  int last_x = s[BO_BALL_X], last_y = s[BO_BALL_Y];
  int dir = s[BO_DIR];
  int new_x, new_y;
  switch (dir) { /* 0 up-left, 1 up-right, 2 down-right, 3 down-left */
    case 0: new_x = last_x - 1; new_y = last_y - 1; break;
    case 1: new_x = last_x + 1; new_y = last_y - 1; break;
    case 2: new_x = last_x + 1; new_y = last_y + 1; break;
    default: new_x = last_x - 1; new_y = last_y + 1; break;
  }
  if (new_x < 0 || new_x > 9) {
    new_x = new_x < 0 ? 0 : 9;
    dir = FLIP_X[dir];
  }
  /* step_ball_brick */
  int terminal = 0;
  int strike_toggle = 0;
  if (new_y < 0) {
    new_y = 0;
    dir = FLIP_Y[dir];
  } else if (new_y <= 9 && s[BO_MAP + new_y * 10 + new_x] == 1) { /* jnp index clamps; rows >3 hold no bricks */
    strike_toggle = 1;
    if (!s[BO_STRIKE]) {
      r += 1.0f;
      s[BO_MAP + new_y * 10 + new_x] = 0;
      new_y = last_y;
      dir = FLIP_Y[dir];
    }
  } else if (new_y == 9) {
    int any = 0;
    for (int i = 0; i < 100; ++i) any |= s[BO_MAP + i];
    if (!any)
      for (int y = 1; y < 4; ++y)
        for (int x = 0; x < 10; ++x) s[BO_MAP + y * 10 + x] = 1;
    if (last_x == pos) { /* old ball_x vs NEW paddle pos */
      dir = FLIP_Y[dir];
      new_y = last_y;
    } else if (new_x == pos) {
      dir = FLIP_XY[dir];
      new_y = last_y;
    } else {
      terminal = 1;
    }
  }
  s[BO_STRIKE] = strike_toggle;
  s[BO_POS] = pos;
  s[BO_LAST_X] = last_x;
  s[BO_LAST_Y] = last_y;
  s[BO_DIR] = dir;
  s[BO_BALL_X] = new_x;
  s[BO_BALL_Y] = new_y;
  s[BO_TIME] += 1;
  int d = terminal || (s[BO_TIME] >= max_steps);
  s[BO_TERM] = d;
  *reward = r;
  *done = d;
}

/* ------------------------------------------------------------------------ */
/* CartPole-v1 (gymnax 0.0.6 environments/classic_control/cartpole.py).
 * Canonical state: sf = [x, x_dot, theta, theta_dot], si = [time].        */
/* ------------------------------------------------------------------------ */
static void cartpole_reset_one(uint64_t key, uint32_t e, int32_t *si, float *sf) {
  /* uniform(minval=-0.05, maxval=0.05, shape=(4,)): 4 draws = streams 1,2 x 2 words */
  uint32_t o[2];
  pqn_oracle_env_bits(key, e, 1u, o);
  sf[0] = pqn_oracle_bits_to_uniform(o[0]) * 0.1f - 0.05f;
  sf[1] = pqn_oracle_bits_to_uniform(o[1]) * 0.1f - 0.05f;
  pqn_oracle_env_bits(key, e, 2u, o);
  sf[2] = pqn_oracle_bits_to_uniform(o[0]) * 0.1f - 0.05f;
  sf[3] = pqn_oracle_bits_to_uniform(o[1]) * 0.1f - 0.05f;
  si[0] = 0;
}

/* sin / cos as explicit f32 arithmetic (Cody-Waite reduction by pi/2 + Cephes sinf / cosf polynomials): libm's
 * sinf / cosf and the GPU's differ in the last ulp; a fixed sequence of IEEE operations (compiled with
 * -ffp-contract=off) is reproducible bit for bit by any implementation, which turns the CartPole comparison into an
 * exact trajectory test.  Within 1 ulp of libm on the range the dynamics visit (|theta| < 0.25 rad). */
static void oracle_sincos_f32(float x, float *s, float *c) {
  const float k = rintf(x * 0.636619772367581343f);
  float r = x - k * 1.5703125f;
  r = r - k * 4.837512969970703125e-4f;
  r = r - k * 7.54978995489188216e-8f;
  const float z = r * r;
  const float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  const float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  const int q = ((int)k) & 3;
  *s = (q == 0) ? sp : (q == 1) ? cp : (q == 2) ? -sp : -cp;
  *c = (q == 0) ? cp : (q == 1) ? -sp : (q == 2) ? -cp : sp;
}

static int cartpole_terminal(const int32_t *si, const float *sf, int32_t max_steps) {
  const float x_thr = 2.4f;
  const float th_thr = (float)(12.0 * 2.0 * 3.14159265358979323846 / 360.0);
  int d1 = (sf[0] < -x_thr) || (sf[0] > x_thr);
  int d2 = (sf[2] < -th_thr) || (sf[2] > th_thr);
  return d1 || d2 || (si[0] >= max_steps);
}

static void cartpole_step_one(int32_t *si, float *sf, int32_t action, int32_t max_steps, float *reward,
                              int *done) {
  const float gravity = 9.8f, masspole = 0.1f, total_mass = 1.1f, length = 0.5f;
  const float polemass_length = 0.05f, force_mag = 10.0f, tau = 0.02f;
  int prev_terminal = cartpole_terminal(si, sf, max_steps);
  float force = force_mag * (float)action - force_mag * (float)(1 - action);
  float costheta, sintheta;
  oracle_sincos_f32(sf[2], &sintheta, &costheta);
  float temp = (force + polemass_length * (sf[3] * sf[3]) * sintheta) / total_mass;
  float thetaacc = (gravity * sintheta - costheta * temp) /
                   (length * (4.0f / 3.0f - masspole * (costheta * costheta) / total_mass));
  float xacc = temp - polemass_length * thetaacc * costheta / total_mass;
  float x = sf[0] + tau * sf[1];
  float x_dot = sf[1] + tau * xacc;
  float theta = sf[2] + tau * sf[3];
  float theta_dot = sf[3] + tau * thetaacc;
  sf[0] = x; sf[1] = x_dot; sf[2] = theta; sf[3] = theta_dot;
  si[0] += 1;
  *reward = 1.0f - (float)prev_terminal;
  *done = cartpole_terminal(si, sf, max_steps);
}


/* ------------------------------------------------------------------------ */
/* Acrobot-v1 (gymnax 0.0.6 environments/classic_control/acrobot.py [3P-RECALL]: the "book" dynamics of Sutton &
 * Barto integrated with one RK4 step of dt = 0.2, torque in {-1, 0, +1}, no torque noise, angles wrapped to
 * [-pi, pi), velocities clipped to 4 pi / 9 pi, reward -1 until -cos(t1) - cos(t1 + t2) > 1, 500 steps).
 * Canonical state: sf = [theta1, theta2, dtheta1, dtheta2], si = [time].  Every sin / cos is oracle_sincos_f32 and
 * every expression is spelled out operation by operation (the HIP rule repeats them verbatim; both are compiled with
 * -ffp-contract=off), so the trajectories are bit-exact across the two implementations. */
/* ------------------------------------------------------------------------ */
static void acrobot_reset_one(uint64_t key, uint32_t e, int32_t *si, float *sf) {
  /* uniform(minval=-0.1, maxval=0.1, shape=(4,)) */
  uint32_t o[2];
  pqn_oracle_env_bits(key, e, 1u, o);
  sf[0] = pqn_oracle_bits_to_uniform(o[0]) * 0.2f - 0.1f;
  sf[1] = pqn_oracle_bits_to_uniform(o[1]) * 0.2f - 0.1f;
  pqn_oracle_env_bits(key, e, 2u, o);
  sf[2] = pqn_oracle_bits_to_uniform(o[0]) * 0.2f - 0.1f;
  sf[3] = pqn_oracle_bits_to_uniform(o[1]) * 0.2f - 0.1f;
  si[0] = 0;
}

static void acrobot_dsdt(const float *y, float a, float *dy) {
  /* m1 = m2 = 1, l1 = 1, lc1 = lc2 = 0.5, I1 = I2 = 1, g = 9.8 */
  const float half_pi = 1.57079632679489661923f;
  const float t1 = y[0], t2 = y[1], w1 = y[2], w2 = y[3];
  float s2, c2, su, c12, c1;
  oracle_sincos_f32(t2, &s2, &c2);
  oracle_sincos_f32((t1 + t2) - half_pi, &su, &c12);
  oracle_sincos_f32(t1 - half_pi, &su, &c1);
  const float d1 = (0.25f + (1.25f + c2)) + 2.0f;          /* m1 lc1^2 + m2 (l1^2 + lc2^2 + 2 l1 lc2 cos t2) + I1 + I2 */
  const float d2 = (0.25f + 0.5f * c2) + 1.0f;             /* m2 (lc2^2 + l1 lc2 cos t2) + I2 */
  const float phi2 = 4.9f * c12;                           /* m2 lc2 g cos(t1 + t2 - pi/2) */
  const float phi1 = (((-0.5f * (w2 * w2)) * s2 - ((1.0f * w2) * w1) * s2) + 14.7f * c1) + phi2;
  const float dd2 = (((a + (d2 / d1) * phi1) - (0.5f * (w1 * w1)) * s2) - phi2) / (1.25f - (d2 * d2) / d1);
  const float dd1 = -((d2 * dd2 + phi1) / d1);
  dy[0] = w1; dy[1] = w2; dy[2] = dd1; dy[3] = dd2;
}

static float acrobot_wrap(float x) {   /* gymnax wrap(x, -pi, pi) */
  const float m = -3.14159265358979323846f, M = 3.14159265358979323846f, diff = M - m;
  const int up = x < m, down = x >= M;
  const float how = (float)up * ceilf((m - x) / diff) + (float)down * floorf((x - M) / diff + 1.0f);
  return (x - (how * diff) * (float)down) + (how * diff) * (float)up;
}

static int acrobot_done_angle(const float *sf) {
  float s, c1, c12;
  oracle_sincos_f32(sf[0], &s, &c1);
  oracle_sincos_f32(sf[1] + sf[0], &s, &c12);
  return (-c1 - c12) > 1.0f;
}

static void acrobot_step_one(int32_t *si, float *sf, int32_t action, int32_t max_steps, float *reward, int *done) {
  const float dt = 0.2f, a = (float)(action - 1);
  float k1[4], k2[4], k3[4], k4[4], y[4];
  int i;
  acrobot_dsdt(sf, a, k1);
  for (i = 0; i < 4; ++i) y[i] = sf[i] + (dt * 0.5f) * k1[i];
  acrobot_dsdt(y, a, k2);
  for (i = 0; i < 4; ++i) y[i] = sf[i] + (dt * 0.5f) * k2[i];
  acrobot_dsdt(y, a, k3);
  for (i = 0; i < 4; ++i) y[i] = sf[i] + dt * k3[i];
  acrobot_dsdt(y, a, k4);
  for (i = 0; i < 4; ++i) y[i] = sf[i] + (dt / 6.0f) * (((k1[i] + 2.0f * k2[i]) + 2.0f * k3[i]) + k4[i]);
  sf[0] = acrobot_wrap(y[0]);
  sf[1] = acrobot_wrap(y[1]);
  sf[2] = fminf(fmaxf(y[2], -12.566370614359172f), 12.566370614359172f);    /* 4 pi */
  sf[3] = fminf(fmaxf(y[3], -28.274333882308138f), 28.274333882308138f);    /* 9 pi */
  const int da = acrobot_done_angle(sf);
  *reward = -1.0f * (float)(1 - da);
  si[0] += 1;
  *done = da || (si[0] >= max_steps);
}

static void acrobot_obs_one(const float *sf, float *obs) {
  float s, c;
  oracle_sincos_f32(sf[0], &s, &c);
  obs[0] = c; obs[1] = s;
  oracle_sincos_f32(sf[1], &s, &c);
  obs[2] = c; obs[3] = s;
  obs[4] = sf[2]; obs[5] = sf[3];
}


/* ------------------------------------------------------------------------ */
/* Asterix-MinAtar (rules: MinAtar asterix.py, Young & Tian 2019; gymnax 0.0.6 mirrors them with
 * selects).  Canonical int state, 43 words: [0] player_x [1] player_y [2] shot_timer [3] spawn_speed
 * [4] spawn_timer [5] move_speed [6] move_timer [7] ramp_timer [8] ramp_index [9] time [10] terminal
 * [11+4i..] entity i (row i+1): x, present, moves_right, is_gold.  parity unpinned (third-party). */
/* ------------------------------------------------------------------------ */
#define AX_SI 43
static void asterix_reset_one(int32_t *s) {
  memset(s, 0, sizeof(int32_t) * AX_SI);
  s[0] = 5; s[1] = 5; s[3] = 10; s[4] = 10; s[5] = 5; s[6] = 5; s[7] = 100;
}
static void asterix_obs_one(const int32_t *s, float *obs) {
  memset(obs, 0, sizeof(float) * 400);
  obs[(s[1] * 10 + s[0]) * 4 + 0] = 1.0f;
  for (int i = 0; i < 8; ++i) {
    const int32_t *en = s + 11 + 4 * i;
    if (!en[1]) continue;
    obs[((i + 1) * 10 + en[0]) * 4 + (en[3] ? 3 : 1)] = 1.0f;
    int back = en[2] ? en[0] - 1 : en[0] + 1;
    if (back >= 0 && back <= 9) obs[((i + 1) * 10 + back) * 4 + 2] = 1.0f;
  }
}
static void asterix_step_one(int32_t *s, int32_t action, uint64_t key, uint32_t e, int32_t max_steps, float *reward, int *done) {
  float r = 0.0f;
  int terminal = 0;
  if (s[4] == 0) { /* _spawn_entity */
    uint32_t o[2], p[2];
    pqn_oracle_env_bits(key, e, 3u, o);
    pqn_oracle_env_bits(key, e, 4u, p);
    int lr = (int)(o[0] & 1u);
    int gold = pqn_oracle_bits_to_uniform(o[1]) < (1.0f / 3.0f);
    int nfree = 0;
    for (int i = 0; i < 8; ++i) nfree += !s[11 + 4 * i + 1];
    if (nfree > 0) {
      int k = (int)(((uint64_t)p[0] * (uint64_t)nfree) >> 32);
      for (int i = 0; i < 8; ++i) {
        if (s[11 + 4 * i + 1]) continue;
        if (k-- == 0) { s[11 + 4 * i] = lr ? 0 : 9; s[11 + 4 * i + 1] = 1; s[11 + 4 * i + 2] = lr; s[11 + 4 * i + 3] = gold; break; }
      }
    }
    s[4] = s[3];
  }
  if (action == 1) s[0] = s[0] - 1 < 0 ? 0 : s[0] - 1;
  else if (action == 3) s[0] = s[0] + 1 > 9 ? 9 : s[0] + 1;
  else if (action == 2) s[1] = s[1] - 1 < 1 ? 1 : s[1] - 1;
  else if (action == 4) s[1] = s[1] + 1 > 8 ? 8 : s[1] + 1;
  for (int i = 0; i < 8; ++i) {
    int32_t *en = s + 11 + 4 * i;
    if (en[1] && en[0] == s[0] && i + 1 == s[1]) {
      if (en[3]) { en[1] = 0; r += 1.0f; } else terminal = 1;
    }
  }
  if (s[6] == 0) {
    s[6] = s[5];
    for (int i = 0; i < 8; ++i) {
      int32_t *en = s + 11 + 4 * i;
      if (!en[1]) continue;
      en[0] += en[2] ? 1 : -1;
      if (en[0] < 0 || en[0] > 9) { en[1] = 0; continue; }
      if (en[0] == s[0] && i + 1 == s[1]) {
        if (en[3]) { en[1] = 0; r += 1.0f; } else terminal = 1;
      }
    }
  }
  s[4] -= 1;
  s[6] -= 1;
  if (s[3] > 1 || s[5] > 1) { /* ramping */
    if (s[7] >= 0) s[7] -= 1;
    else {
      if (s[5] > 1 && (s[8] % 2)) s[5] -= 1;
      if (s[3] > 1) s[3] -= 1;
      s[8] += 1;
      s[7] = 100;
    }
  }
  for (int i = 0; i < 8; ++i)
    if (!s[11 + 4 * i + 1]) { s[11 + 4 * i] = 0; s[11 + 4 * i + 2] = 0; s[11 + 4 * i + 3] = 0; } /* canonical empty slot */
  s[9] += 1;
  int d = terminal || (s[9] >= max_steps);
  s[10] = d;
  *reward = r;
  *done = d;
}

/* ------------------------------------------------------------------------ */
/* Freeway-MinAtar (MinAtar freeway.py).  Canonical, 29 words: [0] pos [1] move_timer
 * [2] terminate_timer [3] time [4] terminal [5+3i..] car i (row i+1): x, timer, signed speed. */
/* ------------------------------------------------------------------------ */
#define FW_SI 29
static void freeway_randomize(int32_t *s, uint64_t key, uint32_t e, uint32_t stream0, int initialize) {
  for (int i = 0; i < 8; ++i) {
    uint32_t o[2];
    pqn_oracle_env_bits(key, e, stream0 + (uint32_t)i, o);
    int speed = 1 + (int)(((uint64_t)o[0] * 5u) >> 32);
    int dir = (o[1] & 1u) ? 1 : -1;
    if (initialize) s[5 + 3 * i] = 0;
    s[5 + 3 * i + 1] = speed;
    s[5 + 3 * i + 2] = speed * dir;
  }
}
static void freeway_reset_one(int32_t *s, uint64_t key, uint32_t e) {
  memset(s, 0, sizeof(int32_t) * FW_SI);
  freeway_randomize(s, key, e, 16u, 1);
  s[0] = 9; s[1] = 3; s[2] = 2500;
}
static void freeway_obs_one(const int32_t *s, float *obs) {
  memset(obs, 0, sizeof(float) * 700);
  obs[(s[0] * 10 + 4) * 7 + 0] = 1.0f;
  for (int i = 0; i < 8; ++i) {
    const int32_t *car = s + 5 + 3 * i;
    obs[((i + 1) * 10 + car[0]) * 7 + 1] = 1.0f;
    int back = car[2] > 0 ? car[0] - 1 : car[0] + 1;
    if (back < 0) back = 9; else if (back > 9) back = 0;
    int sp = car[2] < 0 ? -car[2] : car[2];
    obs[((i + 1) * 10 + back) * 7 + 1 + sp] = 1.0f;
  }
}
static void freeway_step_one(int32_t *s, int32_t action, uint64_t key, uint32_t e, int32_t max_steps, float *reward, int *done) {
  float r = 0.0f;
  if (action == 1 && s[1] == 0) { s[1] = 3; s[0] = s[0] - 1 < 0 ? 0 : s[0] - 1; }
  else if (action == 2 && s[1] == 0) { s[1] = 3; s[0] = s[0] + 1 > 9 ? 9 : s[0] + 1; }
  if (s[0] == 0) { r += 1.0f; freeway_randomize(s, key, e, 32u, 0); s[0] = 9; }
  for (int i = 0; i < 8; ++i) {
    int32_t *car = s + 5 + 3 * i;
    if (car[0] == 4 && i + 1 == s[0]) s[0] = 9;
    if (car[1] == 0) {
      car[1] = car[2] < 0 ? -car[2] : car[2];
      car[0] += car[2] > 0 ? 1 : -1;
      if (car[0] < 0) car[0] = 9; else if (car[0] > 9) car[0] = 0;
      if (car[0] == 4 && i + 1 == s[0]) s[0] = 9;
    } else car[1] -= 1;
  }
  s[1] -= s[1] > 0;
  s[2] -= 1;
  int terminal = s[2] < 0;
  s[3] += 1;
  int d = terminal || (s[3] >= max_steps);
  s[4] = d;
  *reward = r;
  *done = d;
}

/* ------------------------------------------------------------------------ */
/* SpaceInvaders-MinAtar (MinAtar space_invaders.py).  Canonical, 309 words: [0] pos [1] alien_dir
 * [2] enemy_move_interval [3] alien_move_timer [4] alien_shot_timer [5] shot_timer [6] ramp_index
 * [7] time [8] terminal [9..] alien_map[100] [109..] f_bullet_map[100] [209..] e_bullet_map[100]. */
/* ------------------------------------------------------------------------ */
#define SI_SI 309
static void si_reset_one(int32_t *s) {
  memset(s, 0, sizeof(int32_t) * SI_SI);
  s[0] = 5; s[1] = -1; s[2] = 12; s[3] = 12; s[4] = 10;
  for (int y = 0; y < 4; ++y)
    for (int x = 2; x < 8; ++x) s[9 + y * 10 + x] = 1;
}
static void si_obs_one(const int32_t *s, float *obs) {
  memset(obs, 0, sizeof(float) * 600);
  obs[(9 * 10 + s[0]) * 6 + 0] = 1.0f;
  for (int c = 0; c < 100; ++c) {
    if (s[9 + c]) { obs[c * 6 + 1] = 1.0f; obs[c * 6 + (s[1] < 0 ? 2 : 3)] = 1.0f; }
    if (s[109 + c]) obs[c * 6 + 4] = 1.0f;
    if (s[209 + c]) obs[c * 6 + 5] = 1.0f;
  }
}
static int si_count(const int32_t *m) { int n = 0; for (int c = 0; c < 100; ++c) n += m[c] != 0; return n; }
static void si_step_one(int32_t *s, int32_t action, int32_t max_steps, float *reward, int *done) {
  int32_t *al = s + 9, *fb = s + 109, *eb = s + 209;
  float r = 0.0f;
  int terminal = 0;
  if (action == 3 && s[5] == 0) { fb[90 + s[0]] = 1; s[5] = 5; }
  else if (action == 1) s[0] = s[0] - 1 < 0 ? 0 : s[0] - 1;
  else if (action == 2) s[0] = s[0] + 1 > 9 ? 9 : s[0] + 1;
  for (int y = 0; y < 9; ++y) for (int x = 0; x < 10; ++x) fb[y * 10 + x] = fb[(y + 1) * 10 + x]; /* roll up */
  for (int x = 0; x < 10; ++x) fb[90 + x] = 0;
  for (int y = 9; y > 0; --y) for (int x = 0; x < 10; ++x) eb[y * 10 + x] = eb[(y - 1) * 10 + x]; /* roll down */
  for (int x = 0; x < 10; ++x) eb[x] = 0;
  if (eb[90 + s[0]]) terminal = 1;
  if (al[90 + s[0]]) terminal = 1;
  if (s[3] == 0) {
    int cnt = si_count(al);
    s[3] = cnt < s[2] ? cnt : s[2];
    int left = 0, right = 0, bottom = 0;
    for (int y = 0; y < 10; ++y) { left += al[y * 10]; right += al[y * 10 + 9]; }
    for (int x = 0; x < 10; ++x) bottom += al[90 + x];
    if ((left > 0 && s[1] < 0) || (right > 0 && s[1] > 0)) {
      s[1] = -s[1];
      if (bottom > 0) terminal = 1;
      int32_t last[10];
      for (int x = 0; x < 10; ++x) last[x] = al[90 + x];
      for (int y = 9; y > 0; --y) for (int x = 0; x < 10; ++x) al[y * 10 + x] = al[(y - 1) * 10 + x];
      for (int x = 0; x < 10; ++x) al[x] = last[x]; /* np.roll wraps */
    } else {
      for (int y = 0; y < 10; ++y) {
        int32_t row[10];
        for (int x = 0; x < 10; ++x) row[x] = al[y * 10 + x];
        for (int x = 0; x < 10; ++x) al[y * 10 + ((x + s[1] + 10) % 10)] = row[x];
      }
    }
    if (al[90 + s[0]]) terminal = 1;
  }
  if (s[4] == 0) {
    s[4] = 10;
    /* _nearest_alien: columns by |x - pos| (ties: smaller x first), lowest alien in that column */
    int found = 0;
    for (int d = 0; d < 10 && !found; ++d) {
      for (int sgn = -1; sgn <= 1 && !found; sgn += 2) {
        int x = s[0] + sgn * d;
        if (d == 0 && sgn == 1) continue;
        if (x < 0 || x > 9) continue;
        int ymax = -1;
        for (int y = 0; y < 10; ++y) if (al[y * 10 + x]) ymax = y;
        if (ymax >= 0) { eb[ymax * 10 + x] = 1; found = 1; }
      }
    }
  }
  for (int c = 0; c < 100; ++c)
    if (al[c] && fb[c]) { r += 1.0f; al[c] = 0; fb[c] = 0; }
  s[5] -= s[5] > 0;
  s[3] -= 1;
  s[4] -= 1;
  int cnt = si_count(al);
  if (s[2] > 6 && cnt == 0) { s[2] -= 1; s[6] += 1; }
  if (cnt == 0)
    for (int y = 0; y < 4; ++y) for (int x = 2; x < 8; ++x) al[y * 10 + x] = 1;
  s[7] += 1;
  int dn = terminal || (s[7] >= max_steps);
  s[8] = dn;
  *reward = r;
  *done = dn;
}

/* ------------------------------------------------------------------------ */
int pqn_oracle_env_spec(int env_id, pqn_oracle_spec_t *spec) {
  memset(spec, 0, sizeof(*spec));
  switch (env_id) {
    case PQN_ORACLE_ENV_BREAKOUT:
      spec->obs_dim[0] = 10; spec->obs_dim[1] = 10; spec->obs_dim[2] = 4;
      spec->obs_size = 400; spec->num_actions = 3; spec->max_steps = 1000;
      spec->si = BO_SI; spec->sf = 0;
      return 0;
    case PQN_ORACLE_ENV_CARTPOLE:
      spec->obs_dim[0] = 4; spec->obs_dim[1] = 0; spec->obs_dim[2] = 0;
      spec->obs_size = 4; spec->num_actions = 2; spec->max_steps = 500;
      spec->si = 1; spec->sf = 4;
      return 0;
    case PQN_ORACLE_ENV_ACROBOT:
      spec->obs_dim[0] = 6; spec->obs_dim[1] = 0; spec->obs_dim[2] = 0;
      spec->obs_size = 6; spec->num_actions = 3; spec->max_steps = 500;
      spec->si = 1; spec->sf = 4;
      return 0;
    case PQN_ORACLE_ENV_ASTERIX:
      spec->obs_dim[0] = 10; spec->obs_dim[1] = 10; spec->obs_dim[2] = 4;
      spec->obs_size = 400; spec->num_actions = 5; spec->max_steps = 1000; spec->si = AX_SI; spec->sf = 0;
      return 0;
    case PQN_ORACLE_ENV_FREEWAY:
      spec->obs_dim[0] = 10; spec->obs_dim[1] = 10; spec->obs_dim[2] = 7;
      spec->obs_size = 700; spec->num_actions = 3; spec->max_steps = 2500; spec->si = FW_SI; spec->sf = 0;
      return 0;
    case PQN_ORACLE_ENV_SPACEINVADERS:
      spec->obs_dim[0] = 10; spec->obs_dim[1] = 10; spec->obs_dim[2] = 6;
      spec->obs_size = 600; spec->num_actions = 4; spec->max_steps = 1000; spec->si = SI_SI; spec->sf = 0;
      return 0;
    case PQN_ORACLE_ENV_CRAFTAX_CLASSIC:   /* flat symbolic observation (craftax_classic.c) */
      spec->obs_dim[0] = 1345; spec->obs_dim[1] = 0; spec->obs_dim[2] = 0;
      spec->obs_size = 1345; spec->num_actions = 17; spec->max_steps = 10000; spec->si = 4096 + 111; spec->sf = 4;
      return 0;
    default:
      return -1;
  }
}

/* Craftax-Classic (craftax_classic.c) */
void cc_reset_one(uint64_t key, uint32_t e, int32_t *si, float *sf);
void cc_obs_one(const int32_t *si, float *obs);
float cc_step_env(int32_t *si, float *sf, int32_t action, uint64_t key, uint32_t e, int *done);

static void obs_one(int env_id, const int32_t *si, const float *sf, float *obs) {
  if (env_id == PQN_ORACLE_ENV_BREAKOUT) breakout_obs_one(si, obs);
  else if (env_id == PQN_ORACLE_ENV_CARTPOLE) memcpy(obs, sf, 4 * sizeof(float));
  else if (env_id == PQN_ORACLE_ENV_ACROBOT) acrobot_obs_one(sf, obs);
  else if (env_id == PQN_ORACLE_ENV_ASTERIX) asterix_obs_one(si, obs);
  else if (env_id == PQN_ORACLE_ENV_FREEWAY) freeway_obs_one(si, obs);
  else if (env_id == PQN_ORACLE_ENV_SPACEINVADERS) si_obs_one(si, obs);
  else if (env_id == PQN_ORACLE_ENV_CRAFTAX_CLASSIC) cc_obs_one(si, obs);
}

static void reset_one(int env_id, uint64_t key, uint32_t e, int32_t *si, float *sf) {
  if (env_id == PQN_ORACLE_ENV_BREAKOUT) {
    uint32_t o[2];
    pqn_oracle_env_bits(key, e, 1u, o);
    breakout_reset_one(o[0], si);
  } else if (env_id == PQN_ORACLE_ENV_CARTPOLE) {
    cartpole_reset_one(key, e, si, sf);
  } else if (env_id == PQN_ORACLE_ENV_ACROBOT) {
    acrobot_reset_one(key, e, si, sf);
  } else if (env_id == PQN_ORACLE_ENV_ASTERIX) {
    asterix_reset_one(si);
  } else if (env_id == PQN_ORACLE_ENV_FREEWAY) {
    freeway_reset_one(si, key, e);
  } else if (env_id == PQN_ORACLE_ENV_SPACEINVADERS) {
    si_reset_one(si);
  } else if (env_id == PQN_ORACLE_ENV_CRAFTAX_CLASSIC) {
    cc_reset_one(key, e, si, sf);
  }
}

int pqn_oracle_env_obs(int env_id, int32_t n, const int32_t *si, const float *sf, float *obs) {
  pqn_oracle_spec_t sp;
  if (pqn_oracle_env_spec(env_id, &sp)) return -1;
  for (int32_t e = 0; e < n; ++e)
    obs_one(env_id, si + (size_t)e * sp.si, sf ? sf + (size_t)e * sp.sf : 0, obs + (size_t)e * sp.obs_size);
  return 0;
}

/* vmap_reset (pqn_minatar.py:107-109): per-env reset_env with per-env keys. */
int pqn_oracle_env_reset(int env_id, int32_t n, uint64_t key, int32_t *si, float *sf, float *obs) {
  pqn_oracle_spec_t sp;
  if (pqn_oracle_env_spec(env_id, &sp)) return -1;
#pragma omp parallel for schedule(static)
  for (int32_t e = 0; e < n; ++e) {
    int32_t *s = si + (size_t)e * sp.si;
    float *f = sf ? sf + (size_t)e * sp.sf : 0;
    reset_one(env_id, key, (uint32_t)e, s, f);
    if (obs) obs_one(env_id, s, f, obs + (size_t)e * sp.obs_size);
  }
  return 0;
}

/* vmap_step (pqn_minatar.py:110-112) over gymnax Environment.step, whose
 * auto-reset semantics are stated in-tree by utils/craftax_wrappers.py:59-80:
 * step_env, reset_env, then select(done, reset, stepped) for obs and state. */
int pqn_oracle_env_step(int env_id, int32_t n, uint64_t key, int32_t *si, float *sf,
                        const int32_t *action, int autoreset, float *obs, float *reward,
                        uint8_t *done, float *discount) {
  pqn_oracle_spec_t sp;
  if (pqn_oracle_env_spec(env_id, &sp)) return -1;
#pragma omp parallel for schedule(static)
  for (int32_t e = 0; e < n; ++e) {
    int32_t *s = si + (size_t)e * sp.si;
    float *f = sf ? sf + (size_t)e * sp.sf : 0;
    float r = 0.0f;
    int d = 0;
    if (env_id == PQN_ORACLE_ENV_BREAKOUT) breakout_step_one(s, action[e], sp.max_steps, &r, &d);
    else if (env_id == PQN_ORACLE_ENV_CARTPOLE) cartpole_step_one(s, f, action[e], sp.max_steps, &r, &d);
    else if (env_id == PQN_ORACLE_ENV_ACROBOT) acrobot_step_one(s, f, action[e], sp.max_steps, &r, &d);
    else if (env_id == PQN_ORACLE_ENV_ASTERIX) asterix_step_one(s, action[e], key, (uint32_t)e, sp.max_steps, &r, &d);
    else if (env_id == PQN_ORACLE_ENV_FREEWAY) freeway_step_one(s, action[e], key, (uint32_t)e, sp.max_steps, &r, &d);
    else if (env_id == PQN_ORACLE_ENV_CRAFTAX_CLASSIC) r = cc_step_env(s, f, action[e], key, (uint32_t)e, &d);
    else si_step_one(s, action[e], sp.max_steps, &r, &d);
    if (d && autoreset) reset_one(env_id, key, (uint32_t)e, s, f);
    if (obs) obs_one(env_id, s, f, obs + (size_t)e * sp.obs_size);
    reward[e] = r;
    done[e] = (uint8_t)d;
    if (discount) discount[e] = d ? 0.0f : 1.0f; /* info["discount"] */
  }
  return 0;
}

/* OptimisticResetVecEnvWrapper.step (utils/craftax_wrappers.py:111-148) over LogWrapper(env) -- the wrapper order
 * of pqn_craftax.py:99-108, so the WHOLE LogEnvState of a finished env is replaced by LogWrapper.reset's zeros
 * (:165-171), returned_* and timestep included; info is the stepped LogWrapper's (:184-199).
 *   1. step every env, no reset                                            (:114-116; LogWrapper :173-199)
 *   2. num_resets = n / reset_ratio fresh states: reset j = reset_env(key_re, j)          (:118-120)
 *   3. being_reset = choice(arange(n), (num_resets,), p=done, replace=False) (:125-131): a uniformly random ordered
 *      subset of the finished envs.  This build's own draw (jax streams cannot be reproduced): finished env e gets the
 *      sort key rand31(key_ch, e) << 32 | e; its rank r among the finished envs is its position in being_reset
 *   4. finished env e takes reset r if r < num_resets (a reset of its own), else reset e / reset_ratio (the default
 *      slot, possibly shared with other envs)                               (:122,132,134-135)
 *   5. select(done, reset, stepped) for state and obs                     (:137-146)
 * key_re = fold_in(key, 1), key_ch = fold_in(key, 2); the env's own step draws use `key` as in pqn_oracle_env_step.
 * Log arrays (ep_ret ... timestep) may be NULL (no LogWrapper). */
int pqn_oracle_env_step_optimistic(int env_id, int32_t n, uint64_t key, int32_t reset_ratio, int32_t *si, float *sf,
                                   const int32_t *action, float *obs, float *reward, uint8_t *done, float *discount,
                                   float *ep_ret, int32_t *ep_len, float *ret_ret, int32_t *ret_len, int32_t *timestep,
                                   float *info_ret_ret, int32_t *info_ret_len, int32_t *info_timestep, int32_t *slot_out) {
  pqn_oracle_spec_t sp;
  if (pqn_oracle_env_spec(env_id, &sp)) return -1;
  if (reset_ratio <= 0 || n % reset_ratio) return -1;            /* :96-98 */
  const int32_t num_resets = n / reset_ratio;
  if (pqn_oracle_env_step(env_id, n, key, si, sf, action, 0, obs, reward, done, discount)) return -1;
  if (ep_ret) {
    pqn_oracle_log_step(n, reward, done, ep_ret, ep_len, ret_ret, ret_len, timestep);
    for (int32_t e = 0; e < n; ++e) {
      if (info_ret_ret) info_ret_ret[e] = ret_ret[e];
      if (info_ret_len) info_ret_len[e] = ret_len[e];
      if (info_timestep) info_timestep[e] = timestep[e];
    }
  }
  const uint64_t key_re = pqn_oracle_fold_in(key, 1u), key_ch = pqn_oracle_fold_in(key, 2u);
  uint64_t *ck = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
  if (!ck) return -1;
  for (int32_t e = 0; e < n; ++e) {
    uint32_t o[2];
    pqn_oracle_env_bits(key_ch, (uint32_t)e, 0u, o);
    ck[e] = done[e] ? (((uint64_t)(o[0] >> 1) << 32) | (uint32_t)e) : ~(uint64_t)0;
  }
  for (int32_t e = 0; e < n; ++e) {
    int32_t slot = -1;
    if (done[e]) {
      int32_t rank = 0;
      for (int32_t j = 0; j < n; ++j) rank += ck[j] < ck[e];
      slot = rank < num_resets ? rank : e / reset_ratio;
      int32_t *s = si + (size_t)e * sp.si;
      float *f = sf ? sf + (size_t)e * sp.sf : 0;
      reset_one(env_id, key_re, (uint32_t)slot, s, f);
      if (obs) obs_one(env_id, s, f, obs + (size_t)e * sp.obs_size);
      if (ep_ret) { ep_ret[e] = 0.0f; ep_len[e] = 0; ret_ret[e] = 0.0f; ret_len[e] = 0; timestep[e] = 0; }
    }
    if (slot_out) slot_out[e] = slot;
  }
  free(ck);
  return 0;
}

/* LogWrapper.step, utils/craftax_wrappers.py:173-200. */
void pqn_oracle_log_step(int32_t n, const float *reward, const uint8_t *done, float *ep_ret,
                         int32_t *ep_len, float *ret_ret, int32_t *ret_len, int32_t *timestep) {
  for (int32_t e = 0; e < n; ++e) {
    float new_ret = ep_ret[e] + reward[e];
    int32_t new_len = ep_len[e] + 1;
    int d = done[e] ? 1 : 0;
    ep_ret[e] = new_ret * (float)(1 - d);
    ep_len[e] = new_len * (1 - d);
    ret_ret[e] = ret_ret[e] * (float)(1 - d) + new_ret * (float)d;
    ret_len[e] = ret_len[e] * (1 - d) + new_len * d;
    timestep[e] += 1;
  }
}

/* eps_greedy_exploration, pqn_minatar.py:115-128 (vmapped at :196).
 * argmax = first maximal index (jnp.argmax).  Also emits max_a q. */
void pqn_oracle_eps_greedy(const float *q, int32_t m, int32_t a, float eps, uint64_t key,
                           int32_t *action, float *qmax) {
  for (int32_t i = 0; i < m; ++i) {
    const float *qi = q + (size_t)i * a;
    int best = 0;
    float bv = qi[0];
    for (int j = 1; j < a; ++j)
      if (qi[j] > bv) { bv = qi[j]; best = j; }
    uint32_t o[2];
    pqn_oracle_env_bits(key, (uint32_t)i, 0u, o);
    float u = pqn_oracle_bits_to_uniform(o[0]);            /* rng_e */
    int rnd = (int)(((uint64_t)o[1] * (uint64_t)a) >> 32); /* rng_a: randint(0, A) */
    action[i] = (u < eps) ? rnd : best;
    if (qmax) qmax[i] = bv;
  }
}

/* Q(lambda) targets.  quirk=1: pqn_minatar.py:237-260 (initial next_q carry is
 * last_q*(1-done[T-1])).  quirk=0: pqn_atari.py:280-302 (initial
 * next_q = max_a q_val[T-1]).  Layout [T][M], f32 arithmetic as in A.8. */
void pqn_oracle_q_lambda(const float *reward, const uint8_t *done, const float *qmax,
                         const float *last_q, float gamma, float lambda, int32_t t_len,
                         int32_t m, int32_t quirk, float *target) {
  for (int32_t e = 0; e < m; ++e) {
    size_t last = (size_t)(t_len - 1) * m + e;
    float lq = last_q[e] * (float)(1 - (int)done[last]);
    float lr = reward[last] + gamma * lq;
    float nq = quirk ? lq : qmax[last];
    target[last] = lr;
    for (int32_t t = t_len - 2; t >= 0; --t) {
      size_t i = (size_t)t * m + e;
      float d = (float)done[i];
      float tb = reward[i] + gamma * (1.0f - d) * nq;
      float delta = lr - nq;
      lr = tb + gamma * lambda * delta;
      lr = (1.0f - d) * lr + d * reward[i];
      nq = qmax[i];
      target[i] = lr;
    }
  }
}

/* optax.linear_schedule (A.5), used at pqn_minatar.py:134-147. */
double pqn_oracle_linear_schedule(double init, double end, double transition_steps, double count) {
  if (transition_steps <= 0) return init;  /* optax: non-positive transition_steps = constant init_value */
  double c = count < 0 ? 0 : (count > transition_steps ? transition_steps : count);
  double frac = 1.0 - c / transition_steps;
  return (init - end) * frac + end;
}

/* optax.chain(clip_by_global_norm(max_norm), radam(lr)) -- pqn_minatar.py:159-162
 * -- one optimizer step on a flat parameter vector.  `count` = number of
 * previous steps (0-based); lr is already evaluated at `count` by the caller.
 * Returns the pre-clip global norm. */
float pqn_oracle_radam_clip_step(float *p, float *g, float *m, float *v, int64_t n,
                                 int64_t count, float lr, float max_norm) {
  const double b1 = 0.9, b2 = 0.999, eps = 1e-8, threshold = 5.0;
  double ss = 0.0;
  for (int64_t i = 0; i < n; ++i) ss += (double)g[i] * (double)g[i];
  float gnorm = (float)sqrt(ss);
  int clip = !(gnorm < max_norm);
  double t = (double)(count + 1);
  double b1t = pow(b1, t), b2t = pow(b2, t);
  double ro_inf = 2.0 / (1.0 - b2) - 1.0;
  double ro = ro_inf - 2.0 * t * b2t / (1.0 - b2t);
  float bc1 = (float)(1.0 - b1t); /* bias_correction: m / (1 - decay**count) */
  float bc2 = (float)(1.0 - b2t);
  int rect = ro >= threshold;
  float r = rect ? (float)sqrt((ro - 4.0) * (ro - 2.0) * ro_inf / ((ro_inf - 4.0) * (ro_inf - 2.0) * ro)) : 0.0f;
  for (int64_t i = 0; i < n; ++i) {
    float gi = g[i];
    if (clip) gi = (gi / gnorm) * max_norm;
    float mi = (float)(1.0 - b1) * gi + (float)b1 * m[i];
    float vi = (float)(1.0 - b2) * (gi * gi) + (float)b2 * v[i];
    m[i] = mi;
    v[i] = vi;
    float mh = mi / bc1;
    float vh = vi / bc2;
    float u = rect ? r * mh / (sqrtf(vh) + (float)eps) : mh;
    p[i] = p[i] - lr * u;
  }
  return gnorm;
}
