/*
 * pqn_oracle.h -- CPU ORACLE for the PQN hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithm of the reference hot path
 * (mttga/purejaxql, purejaxql/pqn_minatar.py:89-431 and its MLP twin
 * purejaxql/pqn_gymnax.py:78-424).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, link or call it.  The product
 * (purejaxql_amd/) never includes this header and never links this library.
 *
 * PARITY STATUS
 *   - Q(lambda), eps-greedy, LogWrapper, auto-reset, schedules, RAdam: restated
 *     from in-tree reference lines (cited per function) and pinned by the
 *     hand-derived known-answer vectors of SURVEY.md 8(c) (KA1..KA8); the
 *     reference ships no tests and cannot be imported here (no jax).
 *   - Environment dynamics (Breakout-MinAtar, CartPole-v1, ...) live in the
 *     un-vendored third-party dependency gymnax==0.0.6 (reference
 *     pyproject.toml:51; call sites pqn_minatar.py:103,107-112).  They are
 *     restated from the published MinAtar rules (Young & Tian 2019) / gymnax's
 *     published algorithm.  **parity unpinned** for env dynamics: nothing under
 *     /root/reference holds a golden vector for them.
 *   - PRNG: threefry2x32-20 (the generator jax uses), pinned by the Random123
 *     known-answer vectors; key derivation is this build's own, so streams are
 *     not bit-compatible with jax.random.
 */
#ifndef PQN_ORACLE_H
#define PQN_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  PQN_ORACLE_ENV_BREAKOUT = 0,      /* Breakout-MinAtar      */
  PQN_ORACLE_ENV_CARTPOLE = 1,      /* CartPole-v1           */
  PQN_ORACLE_ENV_ASTERIX = 2,       /* Asterix-MinAtar       */
  PQN_ORACLE_ENV_FREEWAY = 3,       /* Freeway-MinAtar       */
  PQN_ORACLE_ENV_SPACEINVADERS = 4, /* SpaceInvaders-MinAtar */
  PQN_ORACLE_ENV_CRAFTAX_CLASSIC = 5, /* Craftax-Classic-Symbolic-v1 (craftax_classic.c; third-party rules, parity unpinned) */
  PQN_ORACLE_ENV_ACROBOT = 6,       /* Acrobot-v1 (the alternative env of config/alg/pqn_cartpole.yaml:24) */
};

typedef struct {
  int32_t obs_dim[3];  /* H,W,C (C==0: flat obs of H floats) */
  int32_t obs_size;    /* floats per observation */
  int32_t num_actions;
  int32_t max_steps;   /* params.max_steps_in_episode */
  int32_t si;          /* int32 state words per env (canonical layout) */
  int32_t sf;          /* float state words per env */
} pqn_oracle_spec_t;

/* ---- PRNG -------------------------------------------------------------- */
void pqn_oracle_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]);
uint64_t pqn_oracle_prng_key(uint64_t seed);
uint64_t pqn_oracle_fold_in(uint64_t key, uint32_t data);
void pqn_oracle_env_bits(uint64_t key, uint32_t index, uint32_t stream, uint32_t out[2]);
float pqn_oracle_bits_to_uniform(uint32_t bits);
void pqn_oracle_sort_keys(uint64_t key, int32_t n, int64_t *out);

/* ---- environments ------------------------------------------------------ */
int pqn_oracle_env_spec(int env_id, pqn_oracle_spec_t *spec);
/* reset_env for n envs: env e draws its randomness from env_bits(key, e, 1). */
int pqn_oracle_env_reset(int env_id, int32_t n, uint64_t key, int32_t *si, float *sf, float *obs);
/* gymnax Environment.step = step_env + reset_env + select(done, re, st). */
int pqn_oracle_env_step(int env_id, int32_t n, uint64_t key, int32_t *si, float *sf,
                        const int32_t *action, int autoreset, float *obs, float *reward,
                        uint8_t *done, float *discount);
int pqn_oracle_env_obs(int env_id, int32_t n, const int32_t *si, const float *sf, float *obs);
/* OptimisticResetVecEnvWrapper(LogWrapper(env)).step (utils/craftax_wrappers.py:83-148, order of pqn_craftax.py:99-108) */
int pqn_oracle_env_step_optimistic(int env_id, int32_t n, uint64_t key, int32_t reset_ratio, int32_t *si, float *sf,
                                   const int32_t *action, float *obs, float *reward, uint8_t *done, float *discount,
                                   float *ep_ret, int32_t *ep_len, float *ret_ret, int32_t *ret_len, int32_t *timestep,
                                   float *info_ret_ret, int32_t *info_ret_len, int32_t *info_timestep, int32_t *slot_out);

/* LogWrapper (utils/craftax_wrappers.py:151-200) on arrays */
void pqn_oracle_log_step(int32_t n, const float *reward, const uint8_t *done, float *ep_ret,
                         int32_t *ep_len, float *ret_ret, int32_t *ret_len, int32_t *timestep);

/* ---- algorithm pieces -------------------------------------------------- */
void pqn_oracle_eps_greedy(const float *q, int32_t m, int32_t a, float eps, uint64_t key,
                           int32_t *action, float *qmax);
void pqn_oracle_q_lambda(const float *reward, const uint8_t *done, const float *qmax,
                         const float *last_q, float gamma, float lambda, int32_t t_len,
                         int32_t m, int32_t quirk, float *target);
double pqn_oracle_linear_schedule(double init, double end, double transition_steps, double count);
float pqn_oracle_radam_clip_step(float *p, float *g, float *m, float *v, int64_t n,
                                 int64_t count, float lr, float max_norm);

#ifdef __cplusplus
}
#endif
#endif
