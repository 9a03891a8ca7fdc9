// pqn_common.h -- shared device/host helpers for libpqn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pqn_hotpath.h"

#define PQN_HD __host__ __device__ __forceinline__
#define PQN_D __device__ __forceinline__

// "do this once per device" flag for hipFuncSetAttribute(MaxDynamicSharedMemorySize): the attribute belongs to the (function, device)
// pair, and a process may drive several devices (a process-wide bool skipped it on the second device)
#define PQN_MAX_DEVICES 16
inline bool pqn_not_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
}
struct pqn_once_per_device {
  bool done[PQN_MAX_DEVICES] = {};
  bool first() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= PQN_MAX_DEVICES) return true;   // unknown device: set the attribute again (idempotent)
    if (done[d]) return false;
    done[d] = true;
    return true;
  }
};

// ---------------------------------------------------------------------------
// threefry2x32-20.  Counter-based: no RNG state in HBM, every lane evaluates
// its own block function from (key, (index, stream)).
// ---------------------------------------------------------------------------
PQN_HD uint32_t pqn_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

PQN_HD void pqn_tf2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t &o0, uint32_t &o1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  uint32_t x0 = c0 + k0, x1 = c1 + k1;
#define PQN_R(r) x0 += x1; x1 = pqn_rotl(x1, r); x1 ^= x0;
  PQN_R(13) PQN_R(15) PQN_R(26) PQN_R(6)
  x0 += k1; x1 += k2 + 1u;
  PQN_R(17) PQN_R(29) PQN_R(16) PQN_R(24)
  x0 += k2; x1 += k0 + 2u;
  PQN_R(13) PQN_R(15) PQN_R(26) PQN_R(6)
  x0 += k0; x1 += k1 + 3u;
  PQN_R(17) PQN_R(29) PQN_R(16) PQN_R(24)
  x0 += k1; x1 += k2 + 4u;
  PQN_R(13) PQN_R(15) PQN_R(26) PQN_R(6)
  x0 += k2; x1 += k0 + 5u;
#undef PQN_R
  o0 = x0;
  o1 = x1;
}

PQN_HD void pqn_bits(uint64_t key, uint32_t index, uint32_t stream, uint32_t &o0, uint32_t &o1) {
  pqn_tf2x32((uint32_t)(key >> 32), (uint32_t)key, index, stream, o0, o1);
}

PQN_HD uint64_t pqn_fold(uint64_t key, uint32_t data) {
  uint32_t o0, o1;
  pqn_tf2x32((uint32_t)(key >> 32), (uint32_t)key, 0u, data, o0, o1);
  return ((uint64_t)o0 << 32) | o1;
}

// 23 mantissa bits -> [0,1)
PQN_HD float pqn_uniform(uint32_t bits) {
  union { uint32_t u; float f; } v;
  v.u = (bits >> 9) | 0x3f800000u;
  return v.f - 1.0f;
}

PQN_HD uint32_t pqn_randint(uint32_t bits, uint32_t n) { return (uint32_t)(((uint64_t)bits * (uint64_t)n) >> 32); }

// RNG stream ids (second counter word)
enum { PQN_STREAM_ACT = 0, PQN_STREAM_RESET = 1, PQN_STREAM_RESET2 = 2, PQN_STREAM_ENV = 3 };

// ---------------------------------------------------------------------------
// error channel
// ---------------------------------------------------------------------------
void pqn_set_error(const char *fmt, ...);
int pqn_check_launch(const char *what);

#define PQN_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      pqn_set_error(__VA_ARGS__);     \
      return PQN_E_INVALID;           \
    }                                 \
  } while (0)

// ---------------------------------------------------------------------------
// run-time switches of the kernel selection (profiling, A/B runs, tests): pqn_set_option / pqn_get_option in
// include/pqn_hotpath.h.  Each starts from its PQN_* environment variable (read once) or its default.
// ---------------------------------------------------------------------------
enum {
  PQN_OPT_T1_PAIR = 0,    // PQN_T1_PAIR: pair form of the bf16x3 training kernel 0 never / 1 when its grid fills the chip / 2 always
  PQN_OPT_ROLLOUT_PAIR,   // PQN_ROLLOUT_PAIR: pair form of the bf16x3 rollout kernel, same meaning
  PQN_OPT_BWD_POS,        // PQN_BWD_POS: position-parallel form of the bf16x3 training step 0 off / 1 when the launch fills the chip (default) / 2 whenever the shape allows (pqn_qnet.hip launch_train)
  PQN_OPT_SEED_GROUP,     // PQN_SEED_GROUP: seeds per T1 -> T2 launch pair (0 = all)
  PQN_OPT_ABLATE_TRAIN,   // PQN_ABLATE_TRAIN: phase ablation mask of the training kernels (profiling)
  PQN_OPT_ABLATE,         // PQN_ABLATE: phase ablation of the forward kernel (profiling)
  PQN_OPT_BM_TILE,        // PQN_BM_TILE: tile height of the wide-MLP GEMMs (0 auto, 64, 128; > 128: 128-row tiles from that many tiles up)
  PQN_OPT_BM_SPLIT,       // PQN_BM_SPLIT: K splits of the wide-MLP GEMMs (0 auto, 1 .. 4)
  PQN_OPT_T1_KSPLIT,      // PQN_T1_KSPLIT: K-split form of the f32-mode training kernel for minibatches <= 256 samples (default 1)
  PQN_OPT_T1_KSPLIT_TILES, // PQN_T1_KSPLIT_TILES: the K-split form is taken while tiles x seeds of the launch stay at or below this (default 48)
  PQN_OPT_BM_OVERLAP,     // PQN_BM_OVERLAP: parameter-gradient side of the wide-MLP backward on a second stream (default 0: measured no gain)
  PQN_OPT_PEER_TIMEOUT_S, // PQN_PEER_TIMEOUT_S: wall-clock seconds the in-graph peer all-reduce waits for a peer's gradient (default 60)
  PQN_OPT_T2_ACC,         // PQN_T2_ACC: bf16x3 fc1 weight gradient without split-K partials 0 never / 1 when row blocks x seeds fill the chip / 2 always
  PQN_OPT_UPD_OVERLAP,    // PQN_UPD_OVERLAP: pqn_bigmlp_update puts the first epoch's permutation and the last gradient-copy plane refresh on a side stream (bit 0 / bit 1; default 0: a fork / join pair in the graph costs ~30 us)
  PQN_OPT_ROLLOUT_POS,    // PQN_ROLLOUT_POS: position-structure rollout kernel (256 envs per workgroup, pqn_qnet_pos.hip) 0 never / 1 when the launch fills the chip (default) / 2 whenever the shape allows
  PQN_OPT_PIN_FORM,       // PQN_PIN_FORM: pqn_cnn_rollout / pqn_cnn_rollout_seeds choose the rollout kernel from the envs PER SEED alone, never from the number of seeds in the launch (default 0)
  PQN_OPT_POS_WAVES,      // PQN_POS_WAVES: waves per workgroup of the position-parallel forward / rollout kernels, f16x2 layouts: 0 from the launch (default) / 8 / 4 / 2
  PQN_OPT_POS_CHUNKS,     // PQN_POS_CHUNKS: sample chunks of the position-parallel backward, f16x2 layouts: 0 from the launch (default) / 1 / 2 / 4 / 8
  PQN_OPT_FOLD_APPLY,     // PQN_FOLD_APPLY: fold of the gradient partials + clip + RAdam of a fused CNN update in ONE launch (radam_apply_kernel<true>, pqn_fold.h): 0 never / 1 (default) for launches of one or two seeds (seeds x blocks <= 400) / 2 always; bit-identical either way
  PQN_OPT_GATHER_GROUP,   // PQN_GATHER_GROUP: super-tiles per workgroup of the position-parallel form's gather 0 (default) from the launch (4 while >= 2048 workgroups remain) / 1 / 4; same bytes either way
  PQN_OPT_SORT_IMPL,      // PQN_SORT_IMPL: the epoch shuffle's sort 1 (default) = the two-level bucket sort of pqn_update.hip above 4096 keys per seed, rocPRIM's radix sort below / 0 = rocPRIM always / 2 = the library's sort at every size; the same permutation
  PQN_OPT_SORT_CAP,       // PQN_SORT_CAP: tests only -- keys a bucket may hold before its workgroup takes the rank-sort path (0 = 4096, the LDS capacity)
  PQN_OPT_COUNT
};
int pqn_opt(int id);
// which form of the training / rollout kernels the last launch used (pqn_cnn_last_kernel_form)
// (3 and 4 named round-2 opt-in variants that are gone; the numbers stay retired so that old logs keep their meaning)
enum { PQN_FORM_NONE = 0, PQN_FORM_SINGLE = 1, PQN_FORM_PAIR = 2, PQN_FORM_KSPLIT = 5, PQN_FORM_POS = 6 };
void pqn_note_kernel_form(int which /* 0 = training, 1 = rollout */, int form);

// bf16x3 weight planes (pqn_qnet.hip): x = hi + mid + lo exactly, each a bf16 (round to nearest even); element (i, o)
// of the fc1 kernel goes to the forward-order and dgrad-order slot of every plane.  Plane = 131072 bf16; the six
// planes are [3 forward | 3 dgrad].
PQN_HD unsigned short pqn_bf16_rne(float x) {
  union { float f; uint32_t u; } v;
  v.f = x;
  return (unsigned short)((v.u + 0x7FFFu + ((v.u >> 16) & 1u)) >> 16);
}
PQN_HD float pqn_bf16_to_f32(unsigned short b) {
  union { float f; uint32_t u; } v;
  v.u = (uint32_t)b << 16;
  return v.f;
}
PQN_HD void pqn_x3_store_planes(unsigned short *planes, int i, int o, float w) {
  const unsigned short h = pqn_bf16_rne(w);
  const float r1 = w - pqn_bf16_to_f32(h);
  const unsigned short m = pqn_bf16_rne(r1);
  const float r2 = r1 - pqn_bf16_to_f32(m);
  const unsigned short l = pqn_bf16_rne(r2);
  const int s = i >> 5, hh = (i >> 4) & 1, kk = (i >> 2) & 3, sx = i & 3;
  const int jf = ((((s * 8 + (o >> 4)) * 64) + kk * 16 + (o & 15)) << 3) + 4 * hh + sx;
  const int sK = o >> 5, hd = (o >> 4) & 1, kd = (o >> 2) & 3, sd = o & 3;
  const int jd = (((((i >> 4) * 4 + sK) * 64) + kd * 16 + (i & 15)) << 3) + 4 * hd + sd;
  const int P = 1024 * 128;
  planes[jf] = h; planes[P + jf] = m; planes[2 * P + jf] = l;
  planes[3 * P + jd] = h; planes[4 * P + jd] = m; planes[5 * P + jd] = l;
}

// f16x2 planes of the fc1 kernel (position-parallel kernels, pqn_cnn_layout_t.pos_f16x2; pqn_qnet_x3.h): four planes of 131072 halves
// behind the six bf16 planes -- forward order hi, lo, then dgrad order hi, lo (the fragment orders of the bf16 planes), holding
// 128 w = hi + lo with hi = f16(128 w), lo = f16(128 w - hi)
#define H2_W_SCALE 128.0f
#define H2_W_SHIFT 7
#define H2_PLANES_OFF (3 * 1024 * 128)            // floats from off_w1h to the f16 planes
PQN_HD void pqn_h2_split1(float w, _Float16 &h, _Float16 &l) {
  const float x = w * H2_W_SCALE;
  h = (_Float16)x;
  l = (_Float16)(x - (float)h);
}
PQN_HD void pqn_h2_store_planes(_Float16 *planes, int i, int o, float w) {
  _Float16 h, l;
  pqn_h2_split1(w, h, l);
  const int s = i >> 5, hh = (i >> 4) & 1, kk = (i >> 2) & 3, sx = i & 3;
  const int jf = ((((s * 8 + (o >> 4)) * 64) + kk * 16 + (o & 15)) << 3) + 4 * hh + sx;
  const int sK = o >> 5, hd = (o >> 4) & 1, kd = (o >> 2) & 3, sd = o & 3;
  const int jd = (((((i >> 4) * 4 + sK) * 64) + kd * 16 + (i & 15)) << 3) + 4 * hd + sd;
  const int P = 1024 * 128;
  planes[jf] = h; planes[P + jf] = l;
  planes[2 * P + jd] = h; planes[3 * P + jd] = l;
}

// b^t for integer t >= 1 in f64 (<= 2 ulp from pow(); only its f32 cast is used)
PQN_HD double pqn_powi(double b, int t) {
  double r = 1.0;
  while (t > 0) {
    if (t & 1) r *= b;
    b *= b;
    t >>= 1;
  }
  return r;
}

// shared between pqn_algo.hip and pqn_qnet.hip
inline int pqn_radam_blocks(int64_t n) {
  int64_t b = (n + 1023) / 1024;  // 4 elements per lane
  return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}
// Seed batching (jax.vmap over seeds, pqn_minatar.py:459-461): S independent seeds in ONE launch, grid.y (T2: grid.z)
// = seed.  Every per-seed buffer is a slice of a stacked allocation; these are the slice strides in elements.
// All-zero strides + nseeds = 1 is the single-seed case (the public single-seed entry points).
struct pqn_seeds_t {
  int nseeds;
  int n_env, n_env_total;        // envs per seed / of all seeds: transition j = t*n_env + e of seed s lives at
                                 // t*n_env_total + s*n_env + e of the stacked [T][S*N] rollout record
  long long idx_stride;          // sorted shuffle keys of one seed (T*N)
  long long theta_stride;        // parameters / gradient / moments
  long long w1b_stride;          // dgrad-fragment copy of the fc1 kernel
  long long ws_stride;           // optimizer + training workspace
  long long lq_stride;           // loss_buf / qv_buf
  long long idx_mask;            // low bits of a sorted key that hold the local transition index
  int seed_base;                 // first seed of this launch (seed = blockIdx.y + seed_base): launches over seed groups
  int pin_form;                  // != 0: choose the training-kernel form from the minibatch size alone (never from how many
                                 // seeds share the launch), so a seed inside a batch takes the form of its solo run
};
inline pqn_seeds_t pqn_one_seed() {
  pqn_seeds_t s = {};
  s.nseeds = 1;
  s.idx_mask = 0xFFFFFFFFll;
  return s;
}

int pqn_launch_radam(float *p, const float *g, float *m, float *v, int64_t n, int32_t *count, float lr_init,
                     float lr_end, double lr_steps, float max_norm, float *scratch, float *gnorm_out, int w1_off,
                     float *w1b, int norm_pass, int nparts, hipStream_t st, int nseeds = 1, long long pstride = 0,
                     long long sstride = 0, long long w1bstride = 0, int half_off = 0, int copy_mode = 1);
// half_off: float offset of the operand-copy region behind the parameters (pqn_cnn_layout_t.off_w1h; 0 = none);
// copy_mode: 1 = two fp16 copies of the fc1 kernel (matmul_f16), 2 = six bf16 planes (bf16x3), 3 = those + the four f16x2 planes,
// 4 = the f16x2 planes ONLY (the bf16 planes go stale: pqn_update.hip, for optimizer steps whose successor is a position-parallel step)

// kernel timer of pqn_prof_enable(mode) for kernels outside pqn_qnet.hip (mode 2 = the wide-MLP GEMM kernel)
bool pqn_prof_begin(int mode, hipStream_t st);
void pqn_prof_end(hipStream_t st);

// Craftax-Classic (pqn_craftax.hip); reset_ratio 0 = gymnax auto-reset, > 0 = optimistic resets (scratch u64[n])
void pqn_craftax_spec(pqn_env_spec_t *s);
int pqn_craftax_reset(int n, uint64_t key, uint32_t *state, float *obs, hipStream_t st);
int pqn_craftax_step(int n, uint64_t key, const uint64_t *key_dev, float rscale, uint32_t *state, const int32_t *action,
                     const pqn_step_out_t &out, int reset_ratio, uint64_t *scratch, int32_t *slot_out, hipStream_t st);
int pqn_craftax_canon(int n, int do_export, uint32_t *state, int32_t *si, float *sf, uint32_t *log, hipStream_t st);

// internal launchers with device-resident keys / eps (used by the whole-update driver, pqn_update.hip)
int pqn_env_step_dyn(int env_id, int n, const uint64_t *key_dev, float rscale, uint32_t *state, const int32_t *action,
                     const pqn_step_out_t &out, hipStream_t st, int n_per_seed = 0, int key_stride = 0,
                     int32_t *slot_scratch = nullptr);
// OptimisticResetVecEnvWrapper.step (pqn_env_step_optimistic) with the step key in device memory, stepped in place.
// slot_scratch i32[n] (both): Craftax-Classic's reset-slot table; without it the env allocates one per stream on first
// use, which a hipGraph capture does not allow
int pqn_env_step_optimistic_dyn(int env_id, int n, const uint64_t *key_dev, float rscale, int reset_ratio, uint32_t *state,
                                const int32_t *action, const pqn_step_out_t &out, uint64_t *scratch, int32_t *slot_scratch,
                                hipStream_t st);
// shuffle keys of the whole-update driver: keys[i] = rand31(i; *key_dev) << ib | i with ib = pqn_index_bits(n) --
// the same order as the public pqn_shuffle_keys (rand31 << 32 | i), packed so the sort visits 31 + ib bits only
int pqn_index_bits(int n);
int pqn_shuffle_keys_dyn(const uint64_t *key_dev, int n, int64_t *keys, hipStream_t st);
// seed-batched variant: keys[s*n + i] = (s << (31+ib)) | (rand31(i; key_dev[s*key_stride]) << ib) | i  (n <= 2^25,
// S <= 128): one global radix sort orders every seed's segment exactly as its single-seed keys would be
int pqn_shuffle_keys_seeds(const uint64_t *key_dev, int key_stride, int nseeds, int n, int64_t *keys, hipStream_t st);
struct pqn_fold_args_t;
int pqn_cnn_grad_reduce_blocks(int total);   // number of sum-of-squares partials pqn_qnet_cnn_grad leaves in the scratch
int pqn_qnet_cnn_grad_seeds_dyn(const pqn_cnn_layout_t &L, int nb, const int64_t *idx, const uint32_t *obs_bits,
                            const int32_t *action, const float *target, const float *theta, const float *w1b, float *grad,
                            const int32_t *count, float *workspace, float *loss_out, float *qv_out,
                            const pqn_seeds_t &sd, hipStream_t st, bool with_reduce = true, int part = 0, int epoch_mb = -1,
                            int epoch_nmb = 0, struct pqn_fold_args_t *defer = nullptr);
// defer: the fold of the partials is not launched but described in *defer (pqn_fold.h) for pqn_launch_radam_fold -- fold + clip + RAdam in one launch
// the position-parallel form's gather once per epoch (pqn_qnet.hip); _applies: pure predicate of shape, options and workspace stride
bool pqn_qnet_cnn_epoch_applies(const pqn_cnn_layout_t &L, int nb, int nmb, const pqn_seeds_t &sd);
bool pqn_qnet_cnn_pos_form_taken(const pqn_cnn_layout_t &L, int nb, const pqn_seeds_t &sd);   // the training step of this shape takes the position-parallel form
int pqn_qnet_cnn_epoch_gather(const pqn_cnn_layout_t &L, int nb, int nmb, const int64_t *idx_epoch, const uint32_t *obs_bits,
                              const int32_t *action, const float *target, float *workspace, const pqn_seeds_t &sd, hipStream_t st);
long long pqn_qnet_cnn_epoch_floats(const pqn_cnn_layout_t &L, int nb, int nmb);
// part: 0 = whole gradient, 1 = the training kernel(s) only, 2 = fc1 weight gradient + fold of the partials only
int pqn_qnet_cnn_forward_dyn(const pqn_cnn_layout_t &L, int n, const uint32_t *obs_bits, const float *theta, float *q,
                             int32_t *action, float *qmax, float eps, uint64_t key, const float *eps_dev,
                             const uint64_t *key_dev, hipStream_t st);
int pqn_mlp_forward_dyn(const pqn_mlp_layout_t &L, int n, const float *obs, const float *theta, float *q, int32_t *action,
                        float *qmax, float eps, uint64_t key, const float *eps_dev, const uint64_t *key_dev, hipStream_t st,
                        int n_per_seed = 0, long long theta_stride = 0, int key_stride = 0);
int pqn_mlp_grad_seeds(const pqn_mlp_layout_t &L, int nb, const int64_t *idx, const float *obs, const int32_t *action,
                       const float *target, const float *theta, const float *wt, float *grad, const int32_t *count,
                       float *workspace, float *loss_out, float *qv_out, const pqn_seeds_t &sd, long long wt_stride,
                       hipStream_t st);
int pqn_mlp_refresh_transposed_seeds(const pqn_mlp_layout_t &L, const float *theta, float *wt, int nseeds,
                                     long long theta_stride, long long wt_stride, hipStream_t st);
int pqn_qnet_cnn_rollout(int env_id, const pqn_cnn_layout_t &L, int n, int t_len, uint32_t *state, uint32_t *bits,
                         const float *theta, const pqn_step_out_t &rec, int32_t *action, float *qmax, float *last_q,
                         const float *eps_dev, const uint64_t *keys, float rscale, int store_obs, hipStream_t st,
                         int n_per_seed = 0, long long theta_stride = 0, int keys_stride = 0, int pin_form = 0);
