/*
 * pqn_hotpath.h -- C ABI of the MI355X-native PQN hot path (libpqn_hip.so).
 *
 * The reference (mttga/purejaxql) has no FFI: its boundary is the Python-level
 * gymnax functional env API consumed by make_train
 * (purejaxql/pqn_minatar.py:103-112,151,157) plus the algorithm closures of
 * make_train itself.  Each entry point below replaces one of those call sites;
 * the reference line it stands in for is cited next to it.  Python binds this
 * with ctypes (purejaxql_amd/_lib.py); see INTEGRATION.md for the stub.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer (HBM) owned by the caller unless marked
 *    "host".  The library never allocates, frees or synchronises on the hot
 *    path: calls only enqueue kernels on `stream` (a hipStream_t passed as
 *    void*; NULL = the default stream), so they are hipGraph-capturable.
 *  - Every function returns 0 on success or a negative PQN_E_* code; the
 *    message is available from pqn_last_error() (thread-local).  No C++
 *    exception crosses the ABI.
 *  - Keys: a PRNG key is a uint64 (k0<<32|k1) for threefry2x32-20.  Element i
 *    of a batched call draws threefry(key, (i, stream_id)); there is no RNG
 *    state in memory.  pqn_fold_in(key, d) = threefry(key, (0, d)) replaces
 *    jax.random.split.
 *  - Env state is struct-of-arrays of 32-bit words: state[w * n + e] is word w
 *    of env e (coalesced across envs).  Word meaning per env: pqn_env_spec().
 *    The last PQN_LOG_WORDS words are the LogWrapper record
 *    (utils/craftax_wrappers.py:151-158).
 *  - Packed observations ("obs_bits"): MinAtar observations are {0,1} grids;
 *    bit (y*10+x)*C+c of env e lives in obs_bits[e*obs_words + b/32] bit b%32
 *    (flatten order h,w,c of pqn_minatar.py:47).  obs_words is padded to a
 *    multiple of 4 words (16-B loads).
 */
#ifndef PQN_HOTPATH_H
#define PQN_HOTPATH_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQN_OK 0
#define PQN_E_INVALID (-1)   /* bad argument (null pointer, bad size, unknown env) */
#define PQN_E_HIP (-2)       /* HIP runtime error at launch */
#define PQN_E_UNSUPPORTED (-3)

#define PQN_LOG_WORDS 5 /* episode_returns f32, episode_lengths i32, returned_episode_returns f32,
                           returned_episode_lengths i32, timestep i32 */

enum {
  PQN_ENV_BREAKOUT = 0,      /* "Breakout-MinAtar"      */
  PQN_ENV_CARTPOLE = 1,      /* "CartPole-v1"           */
  PQN_ENV_ASTERIX = 2,       /* "Asterix-MinAtar"       */
  PQN_ENV_FREEWAY = 3,       /* "Freeway-MinAtar"       */
  PQN_ENV_SPACEINVADERS = 4, /* "SpaceInvaders-MinAtar" */
  PQN_ENV_CRAFTAX_CLASSIC = 5, /* "Craftax-Classic-Symbolic-v1": flat symbolic observation f32[1345], 17 actions
                                  (make_craftax_env_from_name, pqn_craftax.py:96-98; rules restated, csrc/pqn_craftax.hip) */
  PQN_ENV_ACROBOT = 6,       /* "Acrobot-v1": the alternative env named in config/alg/pqn_cartpole.yaml:24; f32[6], 3 actions */
};

/* What gymnax.make(name) -> (env, env_params) exposes to make_train
 * (pqn_minatar.py:103-105,151,157,333). */
typedef struct {
  int32_t obs_dim[3];   /* (H, W, C); flat envs: (D, 0, 0) */
  int32_t obs_size;     /* floats per observation */
  int32_t num_actions;  /* env.action_space(params).n */
  int32_t max_steps;    /* params.max_steps_in_episode */
  int32_t state_words;  /* u32 words per env in the SoA state, INCLUDING PQN_LOG_WORDS */
  int32_t obs_words;    /* u32 words per env of packed obs (0 if the env has none) */
  int32_t canon_si;     /* int32 words per env in the canonical (export) layout */
  int32_t canon_sf;     /* f32 words per env in the canonical (export) layout */
} pqn_env_spec_t;

/* Outputs of one batched env.step (all device pointers, any may be NULL except
 * reward/done).  Matches the gymnax step tuple + LogWrapper info keys
 * (utils/craftax_wrappers.py:194-199) + gymnax's info["discount"]. */
typedef struct {
  float *obs;                        /* [n, obs_size] f32 */
  uint32_t *obs_bits;                /* [n, obs_words] packed (MinAtar only) */
  float *reward;                     /* [n] */
  uint8_t *done;                     /* [n] */
  float *discount;                   /* [n] info["discount"] */
  float *returned_episode_returns;   /* [n] */
  int32_t *returned_episode_lengths; /* [n] */
  int32_t *timestep;                 /* [n] */
  uint32_t *achievements;            /* [n] Craftax only (other envs ignore it): bit k = achievement k unlocked in the episode
                                        that ENDED with this step, 0 for envs that did not finish -- the source of the
                                        info["Achievements/<name>"] = done * unlocked * 100 keys the Craftax script logs
                                        (pqn_craftax.py:364-369,384-387 with LOG_ACHIEVEMENTS) */
} pqn_step_out_t;

const char *pqn_last_error(void);
int pqn_version(void);

/* ---- PRNG (host helpers; same functions the kernels evaluate) ---------- */
void pqn_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]); /* host */
uint64_t pqn_fold_in(uint64_t key, uint32_t data);                                    /* host */
/* out[i] = fold_in(key, first + i), i < count, written on the device: the per-step keys of a rollout
 * (jax.random.split(rng, num_steps) feeding a lax.scan, pqn_minatar.py:181-183,399-401). */
int pqn_fold_in_range(uint64_t key, uint32_t first, int32_t count, uint64_t *out /* device [count] */, void *stream);

/* ---- environment: replaces gymnax.make / vmap_reset / vmap_step -------- */
int pqn_env_id(const char *name); /* "Breakout-MinAtar" -> PQN_ENV_BREAKOUT, <0 if unknown */
int pqn_env_spec(int env_id, pqn_env_spec_t *spec /* host */);

/* vmap_reset(n)(key) -- pqn_minatar.py:107-109,397,419.  Also zeroes the
 * LogWrapper record (LogWrapper.reset, craftax_wrappers.py:168-171). */
int pqn_env_reset(int env_id, int32_t n, uint64_t key, uint32_t *state, float *obs,
                  uint32_t *obs_bits, void *stream);

/* vmap_step(n)(key, state, action) with LogWrapper -- pqn_minatar.py:110-112,
 * 198-200,391-393.  gymnax auto-reset semantics (craftax_wrappers.py:59-80).
 * state_in may equal state_out (in-place). */
int pqn_env_step(int env_id, int32_t n, uint64_t key, const uint32_t *state_in,
                 uint32_t *state_out, const int32_t *action, const pqn_step_out_t *out /* host */,
                 void *stream);

/* Canonical <-> packed state (tests, checkpoints).  Canonical layout per env:
 * si int32[canon_si], sf f32[canon_sf] (env-major), log f32/i32 [n,5]. */
int pqn_env_export_state(int env_id, int32_t n, const uint32_t *state, int32_t *si, float *sf,
                         uint32_t *log, void *stream);
int pqn_env_import_state(int env_id, int32_t n, const int32_t *si, const float *sf,
                         const uint32_t *log, uint32_t *state, void *stream);

/* OptimisticResetVecEnvWrapper(LogWrapper(env), num_envs = n, reset_ratio).step (utils/craftax_wrappers.py:83-148;
 * the wrapper order of pqn_craftax.py:99-108): every env steps; only n / reset_ratio fresh states exist, reset j =
 * reset_env(fold_in(key, 1), j); the finished envs are ranked by a per-env random sort key (fold_in(key, 2)) -- a
 * uniformly random ordered subset, the reference's choice(p=done, replace=False) -- the first n / reset_ratio of
 * them take a reset of their own (rank r -> reset r), later ones share the default slot e / reset_ratio; a finished
 * env's whole LogWrapper record restarts from zero (LogWrapper sits INSIDE this wrapper), `out` info arrays hold the
 * stepped record.  scratch: u64[n].  slot_out (nullable): i32[n], the reset an env took, -1 if it did not finish.
 * reset_ratio must divide n (:96-98).  In-place stepping (state_in == state_out) is supported. */
int pqn_env_step_optimistic(int env_id, int32_t n, uint64_t key, int32_t reset_ratio, const uint32_t *state_in,
                            uint32_t *state_out, const int32_t *action, const pqn_step_out_t *out /* host */,
                            uint64_t *scratch, int32_t *slot_out, void *stream);

/* ---- algorithm pieces of make_train ------------------------------------ */
/* jax.vmap(eps_greedy_exploration) -- pqn_minatar.py:115-128,194-196.  Also
 * emits max_a q (the only use of Transition.q_val, :249). qmax may be NULL. */
int pqn_eps_greedy(const float *q, int32_t m, int32_t a, float eps, uint64_t key,
                   int32_t *action, float *qmax, void *stream);

/* Q(lambda) targets -- pqn_minatar.py:237-260 (quirk=1) or the
 * pqn_atari.py:280-302 form (quirk=0).  All arrays time-major [T, m]. */
int pqn_q_lambda(const float *reward, const uint8_t *done, const float *qmax, const float *last_q,
                 float gamma, float lambda, int32_t t_len, int32_t m, int32_t quirk,
                 float *target, void *stream);

/* Sort keys for the per-epoch shuffle -- pqn_minatar.py:299-315. */
int pqn_shuffle_keys(uint64_t key, int32_t n, int64_t *keys, void *stream);

/* optax.chain(clip_by_global_norm, radam) on a flat f32 buffer --
 * pqn_minatar.py:159-162,292.  `count` (device int32[1]) is the number of
 * previous steps; it is read, used for the LR schedule and bias corrections,
 * then incremented on the device.  lr(count) = linear(lr_init -> lr_end over
 * lr_steps) if lr_steps > 0 else lr_init (:140-147).  scratch: >= 1024 floats.
 * gnorm_out (nullable): pre-clip global norm. */
int pqn_radam_clip_step(float *p, const float *g, float *m, float *v, int64_t n, int32_t *count,
                        float lr_init, float lr_end, double lr_steps, float max_norm,
                        float *scratch, float *gnorm_out, void *stream);

/* ---- fused MinAtar CNN Q-network (QNetwork/CNN, pqn_minatar.py:24-69) -------------- */
/* "Kernel layout" of the CNN parameters in one flat f32 buffer: flax order
 * (BatchNorm_0 dummy, Conv_0 kernel HWIO + bias, LayerNorm_0, Dense_0, LayerNorm_1,
 * Dense_0 head), segment starts padded to 16 B, and the fc1 kernel W1[i][o]
 * (1024 x 128) permuted into MFMA fragment order
 *   idx(i,o) = ((i/16*8 + o/16)*64 + ((i%16)/4)*16 + o%16)*4 + i%4.
 * Gradients and RAdam moments use the same layout (the update is elementwise). */
typedef struct {
  int32_t c, a;
  int32_t off_bn, off_wc, off_bc, off_ln0s, off_ln0b, off_w1, off_b1, off_ln1s, off_ln1b, off_w2, off_b2;
  int32_t total; /* floats that are parameters (>= flax parameter count; pads are zero): the span of grad / m / v */
  /* Matmul operand precision of the two big layers.  0: f32 MFMA everywhere (the reference's arithmetic type).
   * 1: the fc1 products (forward, input gradient, weight gradient: 78 % of the FLOPs) take fp16 operands with
   * f32 accumulation (v_mfma_f32_16x16x16_f16) -- the 10-bit mantissa of the TF32 mode JAX uses for f32
   * matmuls on the reference's A40; master weights, optimizer, LayerNorm, conv, head and loss stay f32.
   * The theta buffer then carries two fp16 fragment-order copies of the fc1 kernel behind the parameters
   * (kept in step by pqn_qnet_cnn_apply / pqn_qnet_cnn_pack_w1b): allocate `alloc` floats for theta. */
  int32_t matmul_f16;
  int32_t off_w1h; /* float offset of the fp16 forward copy (131072 halves), the dgrad copy follows it */
  int32_t alloc;   /* floats to allocate for theta (== total when matmul_f16 == 0) */
  /* pqn_cnn_layout_ex mode 3 ("f16x2"): matmul_f16 = 2 (every kernel form other than the position-parallel one runs bf16x3 as in mode
   * 2) and pos_f16x2 = 1: the position-parallel kernels (forward, backward, rollout) take their fc1 / conv operands as TWO fp16
   * pieces (22 significand bits, exact power-of-two range scaling, 3 matrix instructions per product instead of 6).  theta then
   * also carries four fp16 planes of the fc1 kernel behind the six bf16 planes (alloc covers them). */
  int32_t pos_f16x2;
} pqn_cnn_layout_t;

int pqn_cnn_layout(int32_t c, int32_t a, pqn_cnn_layout_t *layout /* host */);           /* matmul_f16 = 0 */
int pqn_cnn_layout_ex(int32_t c, int32_t a, int32_t matmul_f16, pqn_cnn_layout_t *layout /* host */);

/* network.apply(params, obs, train=False) on packed observations, fused with the
 * eps-greedy draw of pqn_minatar.py:184-196 (and :227-235, :380-390).  Outputs
 * (each nullable): q [n,A]; action [n] (eps-greedy with `key`, element e draws
 * threefry(key,(e,0)) exactly as pqn_eps_greedy); qmax [n] = max_a q. */
int pqn_qnet_cnn_forward(const pqn_cnn_layout_t *layout /* host */, int32_t n, const uint32_t *obs_bits,
                         const float *theta, float *q, int32_t *action, float *qmax, float eps, uint64_t key,
                         void *stream);

/* One optimizer step of _learn_phase (pqn_minatar.py:266-297) on the fused CNN, in two halves so a
 * multi-GPU gradient all-reduce can sit between them (env-sharded mode):
 *   pqn_qnet_cnn_grad : value_and_grad(_loss_fn) (:271-291) for the minibatch {idx[j]}: sample j reads
 *       obs_bits[idx[j]], action[idx[j]], target[idx[j]] (the shuffle of :299-315 is never
 *       materialised).  Writes the flat gradient (kernel layout), loss = 0.5*mean((q_a-target)^2) and
 *       mean(q_a) (metrics td_loss/qvals, :334-335).  nb must be a multiple of 16.
 *   pqn_qnet_cnn_apply: optax clip_by_global_norm + radam (:159-162,292) on theta/m/v, plus the refresh
 *       of w1b, the dgrad-fragment copy of the fc1 kernel that the backward GEMM streams.
 *       recompute_norm=0 reuses the sums of squares pqn_qnet_cnn_grad left in workspace; pass 1 after
 *       the gradient was modified (all-reduce).
 * workspace: pqn_qnet_cnn_workspace_floats(layout, nb) floats, caller-owned. */
int64_t pqn_qnet_cnn_workspace_floats(const pqn_cnn_layout_t *layout /* host */, int32_t nb);
int pqn_qnet_cnn_grad(const pqn_cnn_layout_t *layout /* host */, int32_t nb, const int64_t *idx,
                      const uint32_t *obs_bits, const int32_t *action, const float *target, const float *theta,
                      const float *w1b, float *grad, const int32_t *count, float *workspace, float *loss_out,
                      float *qv_out, void *stream);
/* jax.vmap over seeds (pqn_minatar.py:459-461) of value_and_grad(_loss_fn) (:271-291): num_seeds independent
 * minibatches of nb samples in the SAME launches (grid.y = seed).  Every per-seed buffer is slice `s` of a stacked
 * allocation: idx[s*idx_stride + j], theta/grad[s*theta_stride ..], w1b[s*131072 ..], count[s],
 * workspace[s*ws_stride ..] (ws_stride >= pqn_qnet_cnn_workspace_floats), loss_out[s], qv_out[s].  Transition
 * idx value j of seed s reads row (j / n_env) * n_env_total + s * n_env + j % n_env of obs_bits / action / target
 * -- the stacked [T][S*N] rollout record; with n_env_total == n_env all seeds index one shared pool directly.
 * This is the launch shape of the training kernels inside pqn_cnn_update_seeds (16 seeds x 4096 samples = the
 * bench's headline launch: pair kernels with the XCD-aware (seed, pair) mapping). */
int pqn_qnet_cnn_grad_seeds(const pqn_cnn_layout_t *layout /* host */, int32_t num_seeds, int32_t nb, const int64_t *idx,
                            int64_t idx_stride, int32_t n_env, int32_t n_env_total, const uint32_t *obs_bits,
                            const int32_t *action, const float *target, const float *theta, int64_t theta_stride,
                            const float *w1b, float *grad, const int32_t *count, float *workspace, int64_t ws_stride,
                            float *loss_out, float *qv_out, void *stream);
int pqn_qnet_cnn_apply(const pqn_cnn_layout_t *layout /* host */, float *theta, float *w1b, const float *grad,
                       float *m, float *v, int32_t *count, float lr_init, float lr_end, double lr_steps,
                       float max_norm, float *workspace, float *gnorm_out, int32_t recompute_norm, void *stream);
/* (re)derive the copies of the fc1 kernel from theta: w1b (f32 dgrad fragments, 131072 floats; NULL = skip) and,
 * for a matmul_f16 layout, the two fp16 copies in theta's own tail.  Call after writing parameters by hand. */
int pqn_qnet_cnn_pack_w1b(const pqn_cnn_layout_t *layout /* host */, float *theta, float *w1b, void *stream);

/* Kernel timer for bench.py's roofline lines: on = 1, pqn_qnet_cnn_grad brackets its dominant kernel with HIP
 * events on the launch stream (the position-parallel form: cnn_pos_bwd_kernel; the other forms: qnet_cnn_train_kernel);
 * on = 2, every launch of the wide-MLP GEMM kernel (pqn_bigmlp_forward / _grad / _gemm) is bracketed instead;
 * on = 3 / 4 (position-parallel form only): cnn_pos_fwd_kernel / (per-minibatch gather +) forward + backward together; on = 5: the
 * once-per-epoch gather of the position-parallel form (pqn_cnn_update*); 0 = off.  pqn_prof_read synchronises on the
 * events and returns the number of timed launches and their summed duration, then resets.  Not capturable in a
 * hipGraph: time eager enqueues. */
int pqn_prof_enable(int32_t on);
int pqn_prof_read(int32_t *count /* host */, float *total_ms /* host */);

/* ---- one whole PQN update enqueued from C++ ------------------------------------------------------ */
/* `_update_step` (pqn_minatar.py:176-369) for the MinAtar CNN: NUM_STEPS x (forward + eps-greedy, env
 * step), bootstrap forward, Q(lambda), NUM_EPOCHS x (shuffle, NUM_MINIBATCHES x (grad, clip + RAdam)),
 * metrics.  Everything that varies between updates is derived on the device from clock[0] (the update
 * index u): step keys fold_in(key_roll, u*T+t), epoch keys fold_in(key_shuf, u*EPOCHS+ep), eps(u), the
 * metrics row.  The call only enqueues (~340 kernels), so it can be captured once in a hipGraph and
 * replayed.  All buffers are caller-owned device memory. */
#define PQN_NUM_METRICS 11 /* env_step, update_steps, env_frame, grad_steps, td_loss, qvals, discount,
                              returned_episode_returns, returned_episode_lengths, timestep, returned_episode */
typedef struct {
  int32_t env_id, num_envs, num_steps, num_minibatches, num_epochs, obs_words, metrics_capacity;
  int32_t reserved; /* flags.  bit 0: ignored (rounds 1-4: a one-kernel fold + clip + RAdam with a grid-wide barrier,
                       measured slower; removed in round 5).
                       bit 1: choose the form of the training kernel from the minibatch size alone, never from the number
                       of seeds batched into the launch (pqn_cnn_update_seeds): every seed then takes the kernels of its
                       solo run and is bit-identical to it in the one regime where the default is not -- f32 operand mode,
                       minibatches <= 256 samples, more than t1_ksplit_tiles tiles x seeds per launch -- at the cost of the
                       slower K-split form there. */
  float gamma, lambda, rew_scale;                 /* GAMMA, LAMBDA, REW_SCALE */
  float eps_start, eps_finish;                    /* linear_schedule over updates (:134-138) */
  float lr_init, lr_end, max_grad_norm;           /* :140-147,159-162 */
  double eps_decay_steps, lr_steps;               /* schedule lengths, in f64 like the reference's Python floats
                                                     (EPS_DECAY*NUM_UPDATES_DECAY may be fractional, e.g. 244.1) */
  uint64_t key_roll, key_shuf;
  uint64_t sort_temp_bytes;
  pqn_cnn_layout_t layout;
  int32_t *clock;        /* [4]  clock[0] = update index u, advanced by the call */
  uint64_t *sched_keys;  /* [num_steps + num_epochs] scratch */
  float *sched_eps;      /* [1] scratch */
  uint32_t *state;       /* [state_words][N] env state, stepped in place */
  uint32_t *bits;        /* [T+1][N][obs_words] packed observations; slot 0 = current obs on entry and exit */
  int32_t *action;       /* [T][N] */
  float *reward;         /* [T][N] (already scaled by rew_scale) */
  uint8_t *done;         /* [T][N] */
  float *qmax;           /* [T][N] max_a q(s_t) */
  float *discount, *rer; /* [T][N] info["discount"], info["returned_episode_returns"] */
  int32_t *rel, *ts;     /* [T][N] info["returned_episode_lengths"], info["timestep"] */
  float *target;         /* [T][N] Q(lambda) targets */
  float *last_q;         /* [N] */
  int64_t *sort_keys_in, *sort_keys_out; /* [T*N] */
  void *sort_temp;       /* pqn_update_sort_temp_bytes(T*N) bytes */
  float *theta, *w1b, *grad, *m, *v; /* kernel-layout parameter / optimizer buffers (pqn_cnn_layout) */
  int32_t *count;        /* [1] optimizer step counter */
  float *workspace;      /* pqn_qnet_cnn_workspace_floats(layout, T*N/num_minibatches) */
  float *loss_buf, *qv_buf; /* [num_minibatches*num_epochs] per-step td loss / mean chosen q */
  double *metrics;       /* [metrics_capacity][PQN_NUM_METRICS], row u written by update u */
} pqn_update_args_t;

int64_t pqn_update_sort_temp_bytes(int32_t n);
/* Workspace floats (per seed) of a whole update: pqn_qnet_cnn_workspace_floats(layout, T*N/num_minibatches) plus, in the bf16x3
 * mode at minibatch sizes the position-parallel kernels take, the EPOCH region -- the gathered rows / bit-transposes / actions /
 * targets of all num_minibatches minibatches of an epoch, written by one launch per epoch instead of one per optimizer step
 * (the shuffled batch of pqn_minatar.py:299-315 cut into its minibatches, :316-320).  A caller whose workspace (stride) is only
 * pqn_qnet_cnn_workspace_floats keeps the per-minibatch gather: same results, 64 small launches more per update. */
int64_t pqn_cnn_update_workspace_floats(const pqn_cnn_layout_t *layout /* host */, int32_t num_envs, int32_t num_steps,
                                        int32_t num_minibatches);
int pqn_cnn_update(const pqn_update_args_t *args /* host */, void *stream);

/* The same update enqueued one PHASE at a time, so that the caller can put a collective between the gradient
 * and the optimizer step of every minibatch: the envs of ONE seed sharded over ranks (SURVEY 8(e);
 * pqn_minatar.py:159-162,285-292 -- clip + RAdam must see the gradient averaged over the global minibatch).
 *   BEGIN            step keys / eps from the clock, rollout scan + bootstrap forward, Q(lambda) targets (:181-260)
 *   SHUFFLE index=ep epoch permutation (:299-315)
 *   GRAD    index=i  i = ep*num_minibatches + mb: forward + backward -> args->grad (flat, kernel layout) (:271-291)
 *   APPLY   index=i  global norm of args->grad (recomputed: the caller may have all-reduced it), clip + RAdam (:292)
 *   END              carry last_obs, metrics row, clock tick (:329-338)
 * pqn_cnn_update(args) == BEGIN, then per epoch SHUFFLE and per minibatch GRAD, APPLY, then END (up to the
 * summation order of the global norm).  Single seed; every phase only enqueues (hipGraph-capturable). */
enum { PQN_PHASE_BEGIN = 0, PQN_PHASE_SHUFFLE = 1, PQN_PHASE_GRAD = 2, PQN_PHASE_APPLY = 3, PQN_PHASE_END = 4 };
int pqn_cnn_update_phase(const pqn_update_args_t *args /* host */, int32_t phase, int32_t index, void *stream);

/* jax.vmap(make_train(config))(rngs) (pqn_minatar.py:459-461) inside the launches: num_seeds independent seeds
 * advance by one update in the SAME kernels (grid.y = seed), one enqueue / one hipGraph for all of them.  Every
 * buffer of `args` is the stacked allocation of all seeds:
 *   env-indexed arrays     state u32[W][S*N], bits [T+1][S*N][OW], records / target [T][S*N], last_q [S*N]
 *                          (seed s owns envs s*N .. s*N+N-1; num_envs stays N, the per-seed count)
 *   parameter-like arrays  theta, grad, m, v [S][theta_stride]; w1b [S][1024*128]; count i32[S];
 *                          workspace [S][workspace_stride]; loss_buf, qv_buf [S][MB*EP]
 *   sched_keys u64[S][T+EP]; sort_keys_in/out i64[S*T*N]; sort_temp: pqn_update_sort_temp_bytes(S*T*N)
 *   metrics f64[S][metrics_capacity][PQN_NUM_METRICS]; clock, sched_eps shared (same update index and eps)
 * key_roll_dev / key_shuf_dev: device u64[S] (args->key_roll / key_shuf are ignored when num_seeds > 1).
 * Results per seed are bit-identical to num_seeds single-seed pqn_cnn_update calls whenever both take the same form of the
 * training kernel (the f32-mode K-split form for minibatches <= 256 samples is chosen by tiles x seeds of the launch, option
 * t1_ksplit_tiles; across that threshold the results agree to f32 summation order).  Needs NUM_ENVS % 16 == 0,
 * T*N <= 2^25, num_seeds <= 128; strides in floats, multiples of 4. */
int pqn_cnn_update_seeds(const pqn_update_args_t *args /* host */, int32_t num_seeds, const uint64_t *key_roll_dev,
                         const uint64_t *key_shuf_dev, int64_t theta_stride, int64_t workspace_stride, void *stream);

/* The same update for num_groups GROUPS of seeds (each group = its own pqn_cnn_update_seeds argument set: own stacked
 * buffers, clock, workspace), software-pipelined over two streams: one optimizer step of a group is a compute-bound
 * training kernel followed by an HBM-bound tail (fc1 weight gradient, fold of the partials, clip + RAdam); the seeds of
 * jax.vmap(make_train) (pqn_minatar.py:459-461) are independent, so every group's tail is enqueued on `tail_stream` and
 * runs UNDER the next group's training kernel on `stream` (edges training kernel(g,i) -> tail(g,i) -> training
 * kernel(g,i+1) as events).  tail_stream is forked from and joined back into `stream` inside the call: capturing `stream`
 * yields one hipGraph with two branches.  Per seed bit-identical to pqn_cnn_update_seeds on the same group.  All groups
 * must share NUM_MINIBATCHES / NUM_EPOCHS; array arguments are host arrays of num_groups entries.  The dependency events are
 * kept per device (the device current at the call).  An error return can leave tail_stream forked but not joined: a caller
 * that is capturing `stream` must end the capture and discard the graph (pqn_bigmlp_update with option upd_overlap: the same). */
int pqn_cnn_update_seed_groups(int32_t num_groups, const pqn_update_args_t *const *args /* host */,
                               const int32_t *num_seeds /* host */, const uint64_t *const *key_roll_dev,
                               const uint64_t *const *key_shuf_dev, const int64_t *theta_stride /* host */,
                               const int64_t *workspace_stride /* host */, void *stream, void *tail_stream);
/* A HIP stream restricted to the compute units whose bit is set in cu_mask (mask_words 32-bit words; 0 words = no mask,
 * then high_priority != 0 asks for the device's highest stream priority).  Measurement aid for the eager form of
 * pqn_cnn_update_seed_groups (a replayed hipGraph does not carry stream attributes); no reference counterpart. */
int pqn_stream_create_masked(const uint32_t *cu_mask /* host */, int32_t mask_words, int32_t high_priority,
                             void **stream_out /* host */);
int pqn_stream_destroy(void *stream);

/* The rollout scan alone, as ONE persistent launch: num_steps x (Q-network forward, eps-greedy, env.step with
 * auto-reset + LogWrapper) for every env, then the bootstrap forward of the last observation.  Replaces
 * jax.lax.scan(_step_env) + the last_q forward (pqn_minatar.py:181-235) and, with eps = EPS_TEST and
 * store_obs = 0, the evaluation scan of get_test_metrics (:380-401).  Step t uses keys_dev[t] for both the
 * eps-greedy draw and env.step, exactly as pqn_qnet_cnn_forward + pqn_env_step with that key would.
 *   state     u32[state_words][n]   in/out (in place)
 *   obs_bits  store_obs != 0: u32[num_steps+1][n][obs_words], slot 0 = current observation on entry, slot t+1 =
 *             observation after step t;  store_obs == 0: u32[1][n][obs_words], current observation in / out
 *   rec       device arrays [num_steps][n] (reward is multiplied by rew_scale; LogWrapper sees the raw reward);
 *             any pointer may be NULL; rec->obs / rec->obs_bits must be NULL
 *   action i32[num_steps][n], qmax f32[num_steps][n] (max_a Q(obs_t)), last_q f32[n] (max_a Q(obs_T)): nullable
 *   eps_dev   device f32[1];  keys_dev device u64[num_steps] (see pqn_fold_in_range) */
int pqn_cnn_rollout(int env_id, const pqn_cnn_layout_t *layout, int32_t num_envs, int32_t num_steps, uint32_t *state,
                    uint32_t *obs_bits, int32_t store_obs, const float *theta, const pqn_step_out_t *rec /* host */,
                    int32_t *action, float *qmax, float *last_q, const float *eps_dev, const uint64_t *keys_dev,
                    float rew_scale, void *stream);
/* The same scan for num_seeds independent seeds in one launch: env e of seed s is column s*envs_per_seed + e of
 * every array, draws its randomness as env e of a single-seed call with keys_dev[s*keys_stride + t], and is driven by
 * the parameters theta + s*theta_stride (envs_per_seed % 16 == 0).  Bit-identical to num_seeds pqn_cnn_rollout calls whenever
 * both take the same rollout kernel; the kernel follows the LAUNCH size by default (bf16x3 mode: the position-structure
 * kernel from 40,960 envs per launch on, option rollout_pos), so a seed batch that fills the chip and its solo runs can differ
 * in the f32 summation order of q (near-tied actions may then differ).  Option pin_form = 1 (pqn_set_option; config
 * SEED_BATCH_BIT_IDENTICAL sets it) takes the kernel from envs_per_seed alone in both calls: bit-identical again. */
int pqn_cnn_rollout_seeds(int env_id, const pqn_cnn_layout_t *layout, int32_t num_seeds, int32_t envs_per_seed,
                          int32_t num_steps, uint32_t *state, uint32_t *obs_bits, int32_t store_obs, const float *theta,
                          int64_t theta_stride, const pqn_step_out_t *rec /* host */, int32_t *action, float *qmax,
                          float *last_q, const float *eps_dev, const uint64_t *keys_dev, int32_t keys_stride,
                          float rew_scale, void *stream);

/* Profiling aid (PQN_T1_STAMPS=1): s_memtime stamps at the phase boundaries of qnet_cnn_train_kernel for
 * workgroups 0..3; 16 slots per workgroup.  Not part of the hot path. */
int pqn_debug_t1_stamps(unsigned long long *out /* host, 64 entries */);
/* The same for the bf16x3 fc1 weight-gradient kernel (workgroup 0): [0] start, [1] operands resident, [2..9] the eight
 * steps of the first row block, [10..25] the row blocks. */
int pqn_debug_t2_stamps(unsigned long long *out /* host, 32 entries */);
/* The same for the position-parallel kernels of a library built with -DPOS_STAMPS (tools/build_pos_variant.sh; the default
 * build compiles the stamp stores out, because vector-memory stores inside the loops would drain the LDS-DMA ring):
 * [0..7] one super-tile of the backward's loop, [16..27] one K step of the forward's loop and its tail phases. */
int pqn_debug_pos_stamps(unsigned long long *out /* host, 32 entries */);
/* Seeds covered by ONE launch of the training kernel (and therefore by one pqn_prof_read sample) when `nseeds` seeds
 * are batched: nseeds unless the profiling override PQN_SEED_GROUP cuts the launches into T1 -> T2 pairs per group. */
int pqn_cnn_seed_group(int matmul_mode, int nseeds);

/* Run-time switches of the kernel selection (profiling, A/B runs, tests; no reference counterpart -- XLA picks its
 * fusions itself).  Names: "t1_pair", "rollout_pair" (pair form of the bf16x3 training / rollout kernel: 0 never,
 * 1 when its grid fills the chip (default), 2 whenever the shape allows), "bwd_pos", "rollout_pos" (position-parallel form of
 * the bf16x3 training step / rollout, pqn_qnet_pos.hip: 0 never, 1 (default) when the launch fills the chip -- minibatch tiles of 256 x seeds >= 160 --
 * or, under SEED_BATCH_BIT_IDENTICAL, from the per-seed size alone; 2 whenever the shape allows), "seed_group", "ablate_train", "ablate", "bm_tile", "bm_split" (wide-MLP GEMM tile height 64 / 128 -- a value above 128 = "128-row tiles from that many tiles
 * up" -- / K splits), "bm_overlap" (parameter-gradient side of the wide-MLP backward layer by layer on a second stream instead of batched behind
 * the input-gradient chain; default 0), "upd_overlap" (pqn_bigmlp_update: first permutation and last gradient-copy plane refresh on a side stream;
 * default 0, measured slower), "peer_timeout_s" (seconds pqn_peer_allreduce_mean waits for a peer; default 60), "t2_acc" (bf16x3 fc1 weight
 * gradient without split-K partials: 0 never, 1 (default) when row blocks x seeds of a launch fill the chip, 2 always), "t1_ksplit" (K-split form of the f32-mode training kernel for minibatches of
 * at most 256 samples: a tile's work cut along the conv positions over many workgroups; 1 (default) = forward partial +
 * head-and-backward as two launches, 2 / 3 / 4 = three launches with 4 / 8 / 16 positions per workgroup, 0 = the
 * single-tile kernel), "t1_ksplit_tiles" (the K-split form is taken while tiles x seeds of a launch stay at or below
 * this; default 48), "fold_apply" (pqn_cnn_update / pqn_cnn_update_seeds: the fold of the gradient partials, clip_by_global_norm and
 * RAdam of a minibatch in ONE launch: 0 never, 1 (default) for launches of one or two seeds, 2 always; bit-identical to the two
 * launches; args->grad then holds the LAST minibatch's gradient after the update -- the only one a caller can observe -- instead of
 * being rewritten by every optimizer step), "sort_impl" (the epoch permutation's sort: 0 rocPRIM's radix sort, 1 (default) the library's bucket sort above
 * 4096 keys per seed, 2 at every size; the same permutation), "gather_group".  Each starts from its PQN_<NAME> environment
 * variable.  Not thread-safe against concurrent launches; results never depend on them beyond f32 rounding. */
int pqn_set_option(const char *name, int32_t value);
int pqn_get_option(const char *name, int32_t *value /* host */);
/* Options are read when a launch is ENQUEUED: a captured hipGraph keeps replaying the kernels chosen at capture time.  The
 * epoch counts the pqn_set_option calls that changed a value; a caller that replays captured updates compares it with the
 * epoch at capture and re-captures when it moved (purejaxql_amd/qnet.py drivers do). */
int pqn_options_epoch(void);
/* Which form the LAST enqueued training (pqn_qnet_cnn_grad / pqn_cnn_update*) and rollout (pqn_cnn_rollout*) launch
 * used: 0 none yet, 1 single-tile kernels, 2 pair kernels, 5 K-split kernels (small minibatches, f32 mode), 6 the position-parallel
 * kernels of pqn_qnet_pos.hip (3 and 4 were round-2 variants, retired).  Lets a test assert in-process that the configuration it means to cover is the one that ran. */
int pqn_cnn_last_kernel_form(int32_t *train_form /* host, nullable */, int32_t *rollout_form /* host, nullable */);

/* ---- fused MLP Q-network (QNetwork of pqn_gymnax.py:29-58, layer_norm, NORM_INPUT=False) ----------- */
/* Parameter buffer in flax order with 16-B aligned segments and natural (in,out) kernels:
 * BatchNorm_0 dummy [2*D] | per hidden layer l: Dense_l kernel [in][H], bias [H], LayerNorm_l scale [H],
 * bias [H] | Dense_L kernel [H][A], bias [A]. */
#define PQN_MLP_MAX_LAYERS 4
typedef struct {
  int32_t d, h, layers, a;
  int32_t off_bn;
  int32_t off_w[PQN_MLP_MAX_LAYERS], off_b[PQN_MLP_MAX_LAYERS], off_lns[PQN_MLP_MAX_LAYERS], off_lnb[PQN_MLP_MAX_LAYERS];
  int32_t off_wout, off_bout;
  int32_t total;
} pqn_mlp_layout_t;

int pqn_mlp_layout(int32_t d, int32_t h, int32_t layers, int32_t a, pqn_mlp_layout_t *layout /* host */);
/* network.apply(train=False) + eps-greedy (pqn_gymnax.py:178-190); obs f32 [n, D]; outputs nullable. */
int pqn_mlp_forward(const pqn_mlp_layout_t *layout /* host */, int32_t n, const float *obs, const float *theta,
                    float *q, int32_t *action, float *qmax, float eps, uint64_t key, void *stream);
/* _learn_phase (pqn_gymnax.py:257-288) in two halves, as for the CNN.  `wt`: (layers-1)*H*H floats, the
 * transposed hidden kernels of layers >= 1, kept in step by pqn_mlp_apply / pqn_mlp_refresh_transposed. */
int64_t pqn_mlp_workspace_floats(const pqn_mlp_layout_t *layout /* host */, int32_t nb);
int pqn_mlp_grad(const pqn_mlp_layout_t *layout /* host */, int32_t nb, const int64_t *idx, const float *obs,
                 const int32_t *action, const float *target, const float *theta, const float *wt, float *grad,
                 const int32_t *count, float *workspace, float *loss_out, float *qv_out, void *stream);
int pqn_mlp_apply(const pqn_mlp_layout_t *layout /* host */, float *theta, float *wt, const float *grad, float *m,
                  float *v, int32_t *count, float lr_init, float lr_end, double lr_steps, float max_norm,
                  float *workspace, float *gnorm_out, int32_t recompute_norm, void *stream);
int pqn_mlp_refresh_transposed(const pqn_mlp_layout_t *layout /* host */, const float *theta, float *wt, void *stream);

/* ONE whole update of the gymnax-classic loop (`_update_step`, pqn_gymnax.py:167-360) enqueued from C++ -- the MLP
 * twin of pqn_cnn_update, same device-side clock / key schedule / metrics row (env_frame is written as env_step),
 * hipGraph-capturable.  Observations are f32: obs[T+1][N][D], slot 0 = current observation on entry and exit. */
typedef struct {
  int32_t env_id, num_envs, num_steps, num_minibatches, num_epochs, metrics_capacity;
  float gamma, lambda, rew_scale;
  float eps_start, eps_finish;
  float lr_init, lr_end, max_grad_norm;
  double eps_decay_steps, lr_steps;
  uint64_t key_roll, key_shuf;
  uint64_t sort_temp_bytes;
  pqn_mlp_layout_t layout;
  int32_t *clock;        /* [4] */
  uint64_t *sched_keys;  /* [num_steps + num_epochs] scratch */
  float *sched_eps;      /* [1] scratch */
  uint32_t *state;       /* [state_words][N] */
  float *obs;            /* [T+1][N][D] */
  int32_t *action;       /* [T][N] */
  float *reward;         /* [T][N] (scaled by rew_scale) */
  uint8_t *done;         /* [T][N] */
  float *qmax;           /* [T][N] */
  float *discount, *rer; /* [T][N] */
  int32_t *rel, *ts;     /* [T][N] */
  float *target;         /* [T][N] */
  float *last_q;         /* [N] */
  int64_t *sort_keys_in, *sort_keys_out; /* [T*N] */
  void *sort_temp;       /* pqn_update_sort_temp_bytes(T*N) bytes */
  float *theta, *wt, *grad, *m, *v; /* kernel-layout buffers (pqn_mlp_layout); wt as for pqn_mlp_grad */
  int32_t *count;        /* [1] */
  float *workspace;      /* pqn_mlp_workspace_floats(layout, T*N/num_minibatches) */
  float *loss_buf, *qv_buf; /* [num_minibatches*num_epochs] */
  double *metrics;       /* [metrics_capacity][PQN_NUM_METRICS] */
} pqn_mlp_update_args_t;

int pqn_mlp_update(const pqn_mlp_update_args_t *args /* host */, void *stream);
/* num_seeds independent seeds in the same launches, buffers stacked exactly as for pqn_cnn_update_seeds (obs
 * [T+1][S*N][D]; wt [S][wt_stride]); per-seed results bit-identical to single-seed calls. */
int pqn_mlp_update_seeds(const pqn_mlp_update_args_t *args /* host */, int32_t num_seeds, const uint64_t *key_roll_dev,
                         const uint64_t *key_shuf_dev, int64_t theta_stride, int64_t workspace_stride, int64_t wt_stride,
                         void *stream);

/* ---- env-sharded mode: one-shot all-reduce of the flat gradient over peer-mapped buffers (csrc/pqn_peer.hip) ------- */
/* The collective `clip_by_global_norm` / `radam` need when the envs of ONE seed are sharded over the GPUs of a node
 * (pqn_minatar.py:159-162,285-292: the gradient of the GLOBAL minibatch), without the host between the gradient and the
 * optimizer kernels: every rank allocates a staging region (pqn_peer_alloc: fine-grained device memory + a 64-byte hipIpc
 * handle), the ranks exchange the handles out of band (torch.distributed all_gather in purejaxql_amd/dist.py), map each
 * other's regions (pqn_peer_open) and fill a pqn_peers_t; pqn_peer_allreduce_mean then only enqueues two small kernels
 * (publish / wait + sum in rank order + scale by 1 / world), so it can sit inside a captured update.  All ranks obtain
 * bit-identical results.  A peer that never arrives within the WALL-CLOCK time-out (option "peer_timeout_s", default 60 s;
 * the device's constant 100 MHz clock) ends in a sticky error word (pqn_peer_status: 0 = fine, r + 1 = rank r never
 * published), not in a hung device; the bucket of that call -- and of every later one, which fails fast -- is overwritten
 * with NaN rather than with a mean of stale staging buffers, so that the optimizer state of the detecting rank (and, through
 * the NaN bucket it publishes at the next step, of every rank still alive) cannot pass for a synchronised result.  Callers
 * poll the word once per update (purejaxql_amd/dist.py) and stop. */
#define PQN_PEER_MAX 8
typedef struct {
  int32_t rank, world;
  int64_t n;                    /* floats in the bucket */
  void *region[PQN_PEER_MAX];   /* region[r]: rank r's staging region as mapped into THIS process (own: the allocation itself) */
  uint32_t *local_state;        /* device u32[4], zero-initialised, private to this rank */
} pqn_peers_t;
int64_t pqn_peer_region_bytes(int64_t n);
int pqn_peer_alloc(int64_t bytes, void **ptr /* host out */, uint8_t *handle64 /* host out [64] */);
int pqn_peer_open(const uint8_t *handle64 /* host */, void **ptr /* host out */);
int pqn_peer_close(void *ptr);
int pqn_peer_free(void *ptr);
int pqn_peer_allreduce_mean(const pqn_peers_t *peers /* host */, float *grad /* device, 16-B aligned, in place */, void *stream);
int pqn_peer_status(const pqn_peers_t *peers /* host */, int32_t *error_out /* host; synchronises */);

/* ---- wide MLP Q-network of the Craftax script (QNetwork, pqn_craftax.py:33-62, NORM_TYPE = layer_norm) ---------- */
/* [BatchRenorm | BatchNorm | nothing](x) -> `layers` x (Dense(h) -> LayerNorm -> relu) -> Dense(a): the shape of
 * config/alg/pqn_craftax.yaml (1345 -> 4 x 1024 -> 17, NORM_INPUT with BatchRenorm).  Every Dense layer -- forward, input
 * gradient, weight gradient -- is a tiled bf16x3 MFMA GEMM (csrc/pqn_bigmlp.hip).  Parameter buffer: flax order
 * ({BatchRenorm,BatchNorm}_0 scale, bias | per layer Dense kernel (in,out) row-major, Dense bias, LayerNorm scale,
 * LayerNorm bias | output Dense kernel, bias), every segment start padded to 16 B; the gradient and the RAdam moments
 * share the layout, so the optimizer is pqn_radam_clip_step on the flat buffer. */
#define PQN_BIGMLP_MAX_LAYERS 8
typedef struct {
  int32_t d, h, layers, a;
  int32_t norm_input;                 /* 0: none (the dummy input normalisation never trains), 1: nn.BatchNorm, 2: BatchRenorm */
  int32_t off_in_scale, off_in_bias;
  int32_t off_w[PQN_BIGMLP_MAX_LAYERS + 1], off_b[PQN_BIGMLP_MAX_LAYERS + 1]; /* index `layers` = the output layer */
  int32_t off_lns[PQN_BIGMLP_MAX_LAYERS], off_lnb[PQN_BIGMLP_MAX_LAYERS];
  int32_t total;
} pqn_bigmlp_layout_t;
int pqn_bigmlp_layout(int32_t d, int32_t h, int32_t layers, int32_t a, int32_t norm_input,
                      pqn_bigmlp_layout_t *layout /* host */);
/* workspace floats for `rows` forward rows of which the first `nb` carry gradient (pqn_bigmlp_forward: rows = nb = n;
 * pqn_bigmlp_grad: rows = 2 nb with the 1-step loss, nb with the Q(lambda) loss) */
int64_t pqn_bigmlp_workspace_floats(const pqn_bigmlp_layout_t *layout /* host */, int32_t rows, int32_t nb);
/* The GEMM operands of the Dense kernels: every kernel split exactly into three bf16 planes (x = hi + mid + lo), once in
 * (in, out) order (input gradient) and once transposed (forward).  `wplanes`: pqn_bigmlp_weight_plane_floats floats,
 * caller-owned; call pqn_bigmlp_refresh_planes whenever theta has changed (after every optimizer step). */
int64_t pqn_bigmlp_weight_plane_floats(const pqn_bigmlp_layout_t *layout /* host */);
int pqn_bigmlp_refresh_planes(const pqn_bigmlp_layout_t *layout /* host */, const float *theta, float *wplanes, void *stream);
/* The same with the two copies on two streams: the transposed copy (what the next forward pass reads) on `stream`, the
 * (in, out)-order copy (what the next backward pass reads) on `gradient_stream`; the caller orders gradient_stream against
 * theta's producer and the next pqn_bigmlp_grad (events).  pqn_bigmlp_update uses it after the last optimizer step of an
 * update so that the second copy runs beside the update's closing bookkeeping kernels. */
int pqn_bigmlp_refresh_planes_streams(const pqn_bigmlp_layout_t *layout /* host */, const float *theta, float *wplanes, void *stream,
                                      void *gradient_stream);
/* network.apply(params, obs, train=False) (pqn_craftax.py:184-197,226-237,403-413) + the eps-greedy draw: obs [n][d]
 * contiguous; in_mean / in_var [d] = the running moments of the input normalisation (batch_stats; NULL if norm_input = 0).
 * Outputs (each nullable): q [n][a], action [n] (element e draws threefry(key, (e, 0)) as pqn_eps_greedy), qmax [n].
 * eps_dev / key_dev (nullable): take eps / key from device memory instead (hipGraph-capturable callers). */
int pqn_bigmlp_forward(const pqn_bigmlp_layout_t *layout /* host */, int32_t n, const float *obs, const float *theta,
                       const float *wplanes, float *in_mean, float *in_var, float *workspace, float *q, int32_t *action,
                       float *qmax, float eps, uint64_t key, const float *eps_dev, const uint64_t *key_dev, void *stream);
/* value_and_grad(_loss_fn) of pqn_craftax.py:277-312, train = True with mutable batch_stats.  Minibatch row r reads
 * transition idx[r] of the flat record: obs row idx[r], action / target / reward / done [idx[r]].
 *   next_offset > 0: the `Q_LAMBDA: False` branch -- next_obs of transition j is obs row j + next_offset; obs and
 *       next_obs go through the network as ONE batch of 2 nb rows (batch statistics over both halves), q_next carries no
 *       gradient, target = reward + (1 - done) gamma max_a q_next (:296-306).  `target` unused.
 *   next_offset = 0: the Q(lambda) branch, `target` given (:280-286); reward / done unused.
 * in_mean / in_var [d], in_steps int32[2]: running statistics of the input normalisation, updated in place (BatchRenorm:
 * utils/batch_renorm.py:95-116; in_steps[0] = its train-call counter, in_steps[1] = scratch, zero between calls).
 * Writes the flat gradient, loss and mean(q_a). */
int pqn_bigmlp_grad(const pqn_bigmlp_layout_t *layout /* host */, int32_t nb, const int64_t *idx, const float *obs,
                    int64_t next_offset, const int32_t *action, const float *target, const float *reward,
                    const uint8_t *done, float gamma, const float *theta, const float *wplanes, float *in_mean,
                    float *in_var, int32_t *in_steps, float *grad, float *workspace, float *loss_out, float *qv_out,
                    void *stream);
/* ONE whole update of the Craftax script's loop (`_update_step`, pqn_craftax.py:176-399) enqueued from C++ -- the wide-MLP
 * twin of pqn_mlp_update: same device-side clock / key schedule / metrics row, hipGraph-capturable.  What differs from the
 * gymnax script: the env is batched by a wrapper (reset_ratio > 0: OptimisticResetVecEnvWrapper(LogWrapper(env)), :99-108;
 * 0: BatchEnvWrapper = auto-reset, :109-114); `q_lambda` = 0 selects the 1-step loss on concat(obs, next_obs) (:287-304,
 * no bootstrap forward, no Q(lambda) scan: nothing consumes them); `done_weighted_info` = 1 writes the info columns of the
 * metrics row as (x * returned_episode).sum() / returned_episode.sum() (:364-369; NaN when no episode finished).
 * obs f32 [T+1][N][d], slot 0 = current observation on entry and exit. */
typedef struct {
  int32_t env_id, num_envs, num_steps, num_minibatches, num_epochs, metrics_capacity;
  int32_t reset_ratio, q_lambda, done_weighted_info;
  float gamma, lambda, rew_scale;
  float eps_start, eps_finish;
  float lr_init, lr_end, max_grad_norm;
  double eps_decay_steps, lr_steps;
  uint64_t key_roll, key_shuf;
  uint64_t sort_temp_bytes;
  pqn_bigmlp_layout_t layout;
  int32_t *clock;        /* [4] */
  uint64_t *sched_keys;  /* [num_steps + num_epochs] scratch */
  float *sched_eps;      /* [1] scratch */
  uint32_t *state;       /* [state_words][N], stepped in place */
  float *obs;            /* [T+1][N][d] */
  int32_t *action;       /* [T][N] */
  float *reward;         /* [T][N] (scaled by rew_scale) */
  uint8_t *done;         /* [T][N] */
  float *qmax;           /* [T][N] */
  float *discount, *rer; /* [T][N] */
  int32_t *rel, *ts;     /* [T][N] */
  float *target;         /* [T][N] (q_lambda = 1) */
  float *last_q;         /* [N]    (q_lambda = 1) */
  int64_t *sort_keys_in, *sort_keys_out; /* [T*N]; sort_keys_out ends up as the epoch's permutation */
  void *sort_temp;       /* pqn_update_sort_temp_bytes(T*N) bytes */
  uint64_t *opt_scratch; /* [N] sort keys of the optimistic resets (reset_ratio > 0) */
  int32_t *slot_scratch; /* [N] reset-slot table of the env kernels */
  float *theta, *wplanes, *grad, *m, *v; /* pqn_bigmlp_layout buffers; wplanes as for pqn_bigmlp_forward */
  int32_t *count;        /* [1] optimizer step counter */
  float *in_mean, *in_var; /* [d] running statistics of the input normalisation (norm_input != 0) */
  int32_t *in_steps;     /* [2] */
  float *workspace;      /* max over pqn_bigmlp_workspace_floats(layout, N, N) and (layout, rows, B): B = T*N/num_minibatches,
                            rows = 2 B for the 1-step loss, B for Q(lambda) */
  float *radam_scratch;  /* [1024] */
  float *loss_buf, *qv_buf; /* [num_minibatches*num_epochs] */
  double *metrics;       /* [metrics_capacity][PQN_NUM_METRICS] (env_frame is written as env_step) */
  /* LOG_ACHIEVEMENTS (pqn_craftax.py:364-369,384-387), both nullable together: achievements u32[T][N] = scratch for the
   * achievement mask of the episodes that end with each step (pqn_step_out_t.achievements); ach_metrics
   * f64[metrics_capacity][32]: column k of row u = (100 * unlocked_k * returned_episode).sum() / returned_episode.sum() over
   * update u's [T][N] steps (NaN when no episode finished), k = bit k of the mask */
  uint32_t *achievements;
  double *ach_metrics;
} pqn_bigmlp_update_args_t;
int pqn_bigmlp_update(const pqn_bigmlp_update_args_t *args /* host */, void *stream);

/* Where a forward intermediate of the last pqn_bigmlp_forward / pqn_bigmlp_grad call lives inside `workspace`, for
 * tests (e.g. to compare relu decisions with a reference forward).  f32 tensors -- what 1 = z_layer [rows][h], 3 = (mean,
 * rstd) of z_layer, 4 = q: *offset in floats.  bf16 plane triples -- what 0 = the normalised input, 2 = h_layer =
 * relu(LN(z_layer)): *offset in bf16 elements from the start of the workspace, planes pad16(rows) * ld elements apart,
 * value = hi + mid + lo.  A plane is stored fragment-major (round 5): one 1 KB block per (16 rows, 32 columns), the blocks of
 * a row block consecutive along the columns; inside a block the 16-B slot of (row r, columns 8 kb .. 8 kb + 7) is number
 * 4 r + (kb ^ (-(r >> 2) & 3)) (purejaxql_amd/qnet.py BigMlpTrainer.intermediate undoes it). */
int pqn_bigmlp_workspace_view(const pqn_bigmlp_layout_t *layout /* host */, int32_t rows, int32_t nb, int32_t what,
                              int32_t layer, int64_t *offset /* host */, int64_t *ld /* host */);
/* The GEMM behind every Dense layer of the wide MLP, exposed for tests: C[m][n] = op(A) op(B) (+ bias[n]) with f32-grade
 * bf16x3 products, from f32 matrices (split into planes first, transposed where K is not the contiguous dimension).
 * trans_a: A stored [k][m] (else [m][k]); trans_b: B stored [k][n] (else [n][k]); row-major with leading dimensions
 * lda / ldb / ldc.  nsplit (1..4) K splits leave nsplit partial outputs split_stride floats apart (the caller sums them; no
 * bias then); tile_rows = 64 | 128; scratch: pqn_bigmlp_gemm_scratch_floats(m, n, k) floats. */
/* Profiling aid (PQN_BM_STAMPS=1): cycle stamps of workgroup 0 / wave 0 of the last pqn_bigmlp_gemm launch -- [0] start,
 * [1] first loads issued, [2] K loop done, [8 + 8 s ..]: K step s at LDS stores / stores issued / barrier passed / next
 * loads issued / fragments in registers / MFMAs issued. */
int pqn_debug_bm_stamps(unsigned long long *out /* host, 128 entries */);
int64_t pqn_bigmlp_gemm_scratch_floats(int32_t m, int32_t n, int32_t k);
int pqn_bigmlp_gemm(int32_t m, int32_t n, int32_t k, const float *a, int64_t lda, int32_t trans_a, const float *b,
                    int64_t ldb, int32_t trans_b, const float *bias, float *c, int64_t ldc, int32_t nsplit,
                    int64_t split_stride, int32_t tile_rows, float *scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif
