// pqn_algo.hip -- the non-network pieces of make_train as gfx950 kernels:
// eps-greedy action selection, Q(lambda) targets, shuffle keys, and the fused
// clip_by_global_norm + RAdam step.  Reference lines are cited per entry point
// in include/pqn_hotpath.h.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "pqn_common.h"
#include "pqn_fold.h"

// ---------------------------------------------------------------------------
// error channel + host PRNG helpers
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void pqn_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int pqn_check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pqn_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return PQN_E_HIP;
  }
  return PQN_OK;
}

extern "C" const char *pqn_last_error(void) { return g_err; }
extern "C" int pqn_version(void) { return 1; }

// run-time switches (pqn_common.h): value = environment variable at first use, else the default; pqn_set_option overrides
static struct {
  const char *name, *env;
  int def, value;
  bool init;
} g_opts[PQN_OPT_COUNT] = {
    {"t1_pair", "PQN_T1_PAIR", 1, 0, false},       {"rollout_pair", "PQN_ROLLOUT_PAIR", 1, 0, false},
    {"bwd_pos", "PQN_BWD_POS", 1, 0, false},       {"seed_group", "PQN_SEED_GROUP", 0, 0, false},
    {"ablate_train", "PQN_ABLATE_TRAIN", 0, 0, false}, {"ablate", "PQN_ABLATE", 0, 0, false},
    {"bm_tile", "PQN_BM_TILE", 0, 0, false},       {"bm_split", "PQN_BM_SPLIT", 0, 0, false},
    {"t1_ksplit", "PQN_T1_KSPLIT", 1, 0, false},   {"t1_ksplit_tiles", "PQN_T1_KSPLIT_TILES", 48, 0, false},
    {"bm_overlap", "PQN_BM_OVERLAP", 0, 0, false}, {"peer_timeout_s", "PQN_PEER_TIMEOUT_S", 60, 0, false},
    {"t2_acc", "PQN_T2_ACC", 1, 0, false},        {"upd_overlap", "PQN_UPD_OVERLAP", 0, 0, false},
    {"rollout_pos", "PQN_ROLLOUT_POS", 1, 0, false}, {"pin_form", "PQN_PIN_FORM", 0, 0, false},
    {"pos_waves", "PQN_POS_WAVES", 0, 0, false},   {"pos_chunks", "PQN_POS_CHUNKS", 0, 0, false},
    {"fold_apply", "PQN_FOLD_APPLY", 1, 0, false}, {"gather_group", "PQN_GATHER_GROUP", 0, 0, false},
    {"sort_impl", "PQN_SORT_IMPL", 1, 0, false},   {"sort_cap", "PQN_SORT_CAP", 0, 0, false},
};
static int g_forms[2] = {PQN_FORM_NONE, PQN_FORM_NONE};

int pqn_opt(int id) {
  if (id < 0 || id >= PQN_OPT_COUNT) return 0;
  if (!g_opts[id].init) {
    const char *e = getenv(g_opts[id].env);
    g_opts[id].value = e ? atoi(e) : g_opts[id].def;
    g_opts[id].init = true;
  }
  return g_opts[id].value;
}
void pqn_note_kernel_form(int which, int form) { g_forms[which & 1] = form; }

static int opt_index(const char *name) {
  for (int i = 0; name && i < PQN_OPT_COUNT; ++i)
    if (!strcmp(name, g_opts[i].name)) return i;
  return -1;
}
static int g_opts_epoch = 0;   // bumped by every pqn_set_option that changes a value: see pqn_options_epoch
extern "C" int pqn_set_option(const char *name, int32_t value) {
  const int i = opt_index(name);
  PQN_REQUIRE(i >= 0, "pqn_set_option: unknown option '%s'", name ? name : "(null)");
  if (pqn_opt(i) != value) ++g_opts_epoch;
  g_opts[i].value = value;
  g_opts[i].init = true;
  return PQN_OK;
}
extern "C" int pqn_options_epoch(void) { return g_opts_epoch; }
extern "C" int pqn_get_option(const char *name, int32_t *value) {
  const int i = opt_index(name);
  PQN_REQUIRE(i >= 0 && value, "pqn_get_option: unknown option '%s' or NULL output", name ? name : "(null)");
  *value = pqn_opt(i);
  return PQN_OK;
}
extern "C" int pqn_cnn_last_kernel_form(int32_t *train_form, int32_t *rollout_form) {
  if (train_form) *train_form = g_forms[0];
  if (rollout_form) *rollout_form = g_forms[1];
  return PQN_OK;
}

extern "C" void pqn_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) {
  pqn_tf2x32(key[0], key[1], ctr[0], ctr[1], out[0], out[1]);
}
extern "C" uint64_t pqn_fold_in(uint64_t key, uint32_t data) { return pqn_fold(key, data); }

__global__ void fold_in_range_kernel(uint64_t key, uint32_t first, int count, uint64_t *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = pqn_fold(key, first + (uint32_t)i);
}

extern "C" int pqn_fold_in_range(uint64_t key, uint32_t first, int32_t count, uint64_t *out, void *stream) {
  PQN_REQUIRE(out && count > 0, "pqn_fold_in_range: bad arguments (count=%d)", count);
  hipLaunchKernelGGL(fold_in_range_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, key, first, count,
                     out);
  return pqn_check_launch("pqn_fold_in_range");
}

// ---------------------------------------------------------------------------
// eps-greedy: one lane per row of q[m, a].  First-max tie rule (jnp.argmax).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eps_greedy_kernel(const float *__restrict__ q, int m, int a, float eps,
                                                         uint64_t key, int32_t *__restrict__ action,
                                                         float *__restrict__ qmax) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const float *qi = q + (size_t)i * a;
  int best = 0;
  float bv = qi[0];
  for (int j = 1; j < a; ++j) {
    const float v = qi[j];
    if (v > bv) { bv = v; best = j; }
  }
  uint32_t o0, o1;
  pqn_bits(key, (uint32_t)i, PQN_STREAM_ACT, o0, o1);
  const float u = pqn_uniform(o0);
  const int rnd = (int)pqn_randint(o1, (uint32_t)a);
  action[i] = (u < eps) ? rnd : best;
  if (qmax) qmax[i] = bv;
}

extern "C" int pqn_eps_greedy(const float *q, int32_t m, int32_t a, float eps, uint64_t key, int32_t *action,
                              float *qmax, void *stream) {
  PQN_REQUIRE(q && action, "pqn_eps_greedy: NULL argument");
  PQN_REQUIRE(m > 0 && a > 0, "pqn_eps_greedy: m and a must be > 0 (m=%d a=%d)", m, a);
  hipLaunchKernelGGL(eps_greedy_kernel, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, q, m, a, eps, key,
                     action, qmax);
  return pqn_check_launch("pqn_eps_greedy");
}

// ---------------------------------------------------------------------------
// Q(lambda): one lane per env, serial over T in reverse; [T, m] time-major so
// every load/store is a coalesced dword across the wave.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void q_lambda_kernel(const float *__restrict__ reward,
                                                       const uint8_t *__restrict__ done,
                                                       const float *__restrict__ qmax,
                                                       const float *__restrict__ last_q, float gamma, float lambda,
                                                       int t_len, int m, int quirk, float *__restrict__ target) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= m) return;
  size_t i = (size_t)(t_len - 1) * m + e;
  const float lq = last_q[e] * (float)(1 - (int)done[i]);
  float lr = reward[i] + gamma * lq;
  float nq = quirk ? lq : qmax[i];
  target[i] = lr;
  for (int t = t_len - 2; t >= 0; --t) {
    i -= m;
    const float d = (float)done[i];
    const float r = reward[i];
    const float tb = r + gamma * (1.0f - d) * nq;
    const float delta = lr - nq;
    lr = tb + gamma * lambda * delta;
    lr = (1.0f - d) * lr + d * r;
    nq = qmax[i];
    target[i] = lr;
  }
}

extern "C" int pqn_q_lambda(const float *reward, const uint8_t *done, const float *qmax, const float *last_q,
                            float gamma, float lambda, int32_t t_len, int32_t m, int32_t quirk, float *target,
                            void *stream) {
  PQN_REQUIRE(reward && done && qmax && last_q && target, "pqn_q_lambda: NULL argument");
  PQN_REQUIRE(t_len > 0 && m > 0, "pqn_q_lambda: T and m must be > 0 (T=%d m=%d)", t_len, m);
  hipLaunchKernelGGL(q_lambda_kernel, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, reward, done, qmax,
                     last_q, gamma, lambda, t_len, m, quirk, target);
  return pqn_check_launch("pqn_q_lambda");
}

// ---------------------------------------------------------------------------
// shuffle keys: key_i = rand31_i << ib | i  (unique -> argsort is a well-defined permutation: transitions
// ordered by (rand31, i)).  The public entry point uses ib = 32; the whole-update driver packs the index into
// ib = ceil(log2 n) bits so its radix sort only has to visit the 31 + ib significant bits.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shuffle_keys_kernel(uint64_t key, const uint64_t *__restrict__ key_dev, int n,
                                                           int ib, int64_t *__restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (key_dev) key = *key_dev;
  uint32_t o0, o1;
  pqn_bits(key, (uint32_t)i, 0u, o0, o1);
  keys[i] = (int64_t)(((uint64_t)(o0 >> 1) << ib) | (uint32_t)i);
}

extern "C" int pqn_shuffle_keys(uint64_t key, int32_t n, int64_t *keys, void *stream) {
  PQN_REQUIRE(keys && n > 0, "pqn_shuffle_keys: NULL keys or n <= 0");
  hipLaunchKernelGGL(shuffle_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, key, nullptr, n, 32,
                     keys);
  return pqn_check_launch("pqn_shuffle_keys");
}

int pqn_index_bits(int n) {
  int ib = 1;
  while (ib < 31 && (1ll << ib) < (long long)n) ++ib;
  return ib;
}

int pqn_shuffle_keys_dyn(const uint64_t *key_dev, int n, int64_t *keys, hipStream_t st) {
  hipLaunchKernelGGL(shuffle_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, st, 0, key_dev, n, pqn_index_bits(n), keys);
  return pqn_check_launch("pqn_shuffle_keys");
}

__global__ __launch_bounds__(256) void shuffle_keys_seeds_kernel(const uint64_t *__restrict__ key_dev, int key_stride, int n,
                                                                 int ib, int64_t *__restrict__ keys) {
  const int i = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
  if (i >= n) return;
  uint32_t o0, o1;
  pqn_bits(key_dev[(size_t)s * key_stride], (uint32_t)i, 0u, o0, o1);   // the same draw as shuffle_keys_kernel
  keys[(size_t)s * n + i] = (int64_t)(((uint64_t)s << (31 + ib)) | ((uint64_t)(o0 >> 1) << ib) | (uint64_t)i);   // 7 | 31 | ib bits
}

int pqn_shuffle_keys_seeds(const uint64_t *key_dev, int key_stride, int nseeds, int n, int64_t *keys, hipStream_t st) {
  PQN_REQUIRE(key_dev && keys && n > 0 && n <= (1 << 25) && nseeds >= 1 && nseeds <= 128,
              "pqn_shuffle_keys_seeds: bad arguments (n=%d, seeds=%d)", n, nseeds);
  hipLaunchKernelGGL(shuffle_keys_seeds_kernel, dim3((n + 255) / 256, nseeds), dim3(256), 0, st, key_dev, key_stride, n,
                     pqn_index_bits(n), keys);
  return pqn_check_launch("pqn_shuffle_keys_seeds");
}

// ---------------------------------------------------------------------------
// clip_by_global_norm + RAdam on a flat buffer.  Pass 1: per-block sum of
// squares -> scratch[block]; block 0 snapshots *count into scratch[1023].
// Pass 2: every block folds the partials (<=1022), lane 0 derives the step
// scalars in f64, all lanes apply the update; block 0 bumps *count.
// ---------------------------------------------------------------------------
#define PQN_RADAM_MAXB 1022

__global__ __launch_bounds__(256) void radam_norm_kernel(const float *__restrict__ g, int64_t n,
                                                         const int32_t *__restrict__ count,
                                                         float *__restrict__ scratch) {
  __shared__ float s_part[4];
  float acc = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n; i += 8 * stride) {   // 8 loads in flight, the additions in the order of the plain loop
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = g[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = fmaf(t[u], t[u], acc);
  }
  for (; i < n; i += stride) {
    const float v = g[i];
    acc = fmaf(v, v, acc);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    scratch[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    if (blockIdx.x == 0) reinterpret_cast<int32_t *>(scratch)[1023] = *count;
  }
}

// FOLD (round 6, pqn_fold.h): the launch also folds the training step's partials -- block b of a seed folds what block b of
// qnet_grad_reduce_kernel folds, keeps the result in registers, publishes its sum of squares in a tagged 8-byte slot of the seed's
// scratch and reads the other blocks' slots until every tag is this launch's (agent-scope loads and stores: no cache-wide fence, no
// read-modify-write; a seed's blocks are consecutive in dispatch order and fewer than the chip holds, so the wait cannot starve
// them).  Then it applies the step to exactly those elements.  One launch and one round trip of the gradient less per optimizer step;
// the same additions in the same order as the two-launch form.
template <bool FOLD>
__global__ __launch_bounds__(256) void radam_apply_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                          float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                          int32_t *__restrict__ count, float lr_init, float lr_end,
                                                          double lr_steps, float max_norm, int nparts,
                                                          float *__restrict__ scratch,
                                                          float *__restrict__ gnorm_out, int w1_off,
                                                          float *__restrict__ w1b, long long pstride, long long sstride,
                                                          long long w1bstride, int half_off, int copy_mode,
                                                          typename std::conditional<FOLD, pqn_fold_args_t, int>::type fa) {
  __shared__ float s_part[4];
  __shared__ float s_sc[8];
  {  // seed slice (grid.y; all strides 0 for a single seed)
    const long long sd = blockIdx.y;
    p += sd * pstride; m += sd * pstride; v += sd * pstride;
    if (g) g += sd * pstride;
    scratch += sd * sstride;
    count += sd;
    if (w1b) w1b += sd * w1bstride;
  }
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t n4 = ((n & 3) == 0 && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0)) ? n / 4 : 0;
  const bool x3_w1 = w1b && half_off && copy_mode >= 2 && n4 > 0 && (w1_off & 3) == 0;   // fragment-aligned walk over the fc1 kernel (below)
  // round 6: the fc1 walk's operands are requested BEFORE the norm partials are folded -- the clip decision then arrives while they
  // are in flight instead of ahead of a second memory round trip (every element is read and written by one thread only)
  f4 pf_g = {0.f, 0.f, 0.f, 0.f}, pf_m = pf_g, pf_v = pf_g, pf_p = pf_g;
  int32_t c_snap = 0;
  float part[4] = {0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] float fold_g = 0.0f, fold_m = 0.0f, fold_v = 0.0f, fold_p = 0.0f;   // FOLD: the element of a non-fc1 block's lane
  [[maybe_unused]] int fold_i = -1;
  [[maybe_unused]] unsigned fold_tag = 0;
  __shared__ int32_t s_cnt;
  if constexpr (FOLD) {
    __shared__ float s_red[4][64];
    __shared__ float s_fpart[4];
    const bool fc1_blk = blockIdx.x < QR_W1_BLOCKS;
    const int64_t i4 = (int64_t)(w1_off >> 2) + (int)blockIdx.x * 256 + threadIdx.x;
    if (fc1_blk) {   // this lane's float4 of the fc1 kernel: moments and parameters requested beside the slab loads of the fold
      pf_m = reinterpret_cast<f4 *>(m)[i4];
      pf_v = reinterpret_cast<f4 *>(v)[i4];
      pf_p = reinterpret_cast<f4 *>(p)[i4];
    }
    // this launch's tag: the seed's launch serial + 1 (block 0 stores it back at the end; any initial value will do)
    const unsigned tag = __hip_atomic_load(reinterpret_cast<unsigned *>(scratch) + PQN_FOLD_SERIAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (threadIdx.x == 192) s_cnt = *count;   // read before this block publishes: block 0 bumps it only after every block has
    float ss;
    pqn_fold_block(fa, (int)blockIdx.x, (long long)blockIdx.y, const_cast<float *>(g), s_red, pf_g, fold_g, fold_i, ss);
    if (fold_i >= 0) { fold_m = m[fold_i]; fold_v = v[fold_i]; fold_p = p[fold_i]; }
    if ((threadIdx.x & 63) == 0) s_fpart[threadIdx.x >> 6] = ss;
    __syncthreads();
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(scratch + PQN_FOLD_SLOT0);
    if (threadIdx.x == 0) {
      const float tot = (s_fpart[0] + s_fpart[1]) + (s_fpart[2] + s_fpart[3]);
      __hip_atomic_store(slots + blockIdx.x, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(tot), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 192) c_snap = s_cnt;
    if ((int)threadIdx.x < nparts) {   // nparts <= PQN_FOLD_MAX_BLOCKS < 256: one slot per lane
      unsigned long long sv;
      int spins = 0;
      for (;;) {
        sv = __hip_atomic_load(slots + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(sv >> 32) == tag) break;
        if (++spins > (1 << 24)) { sv = 0x7FC00000ull; break; }   // never seen; a NaN norm is loud, a hung queue is not
        __builtin_amdgcn_s_sleep(2);
      }
      part[0] = __uint_as_float((unsigned)sv);
    }
    fold_tag = tag;
  } else {
  if (x3_w1) {
    const int q0 = (int)blockIdx.x * 256 + threadIdx.x;
    if (q0 < 1024 * 128 / 4) {
      const int64_t i4 = (int64_t)(w1_off >> 2) + q0;
      pf_g = reinterpret_cast<const f4 *>(g)[i4];
      pf_m = reinterpret_cast<f4 *>(m)[i4];
      pf_v = reinterpret_cast<f4 *>(v)[i4];
      pf_p = reinterpret_cast<f4 *>(p)[i4];
    }
  }
  // the step-count-dependent scalars (f64: two integer powers, a square root, the schedule) are derived by lane 0 of
  // wave 3 while the norm partials are in flight: the f64 chain is off the critical path
  if (threadIdx.x == 192) c_snap = reinterpret_cast<const int32_t *>(scratch)[1023];
#pragma unroll
  for (int q = 0; q < 4; ++q) {   // nparts <= 1022: all loads of a lane issued before any is consumed
    const int i = threadIdx.x + 256 * q;
    if (i < nparts) part[q] = scratch[i];
  }
  }
  if (threadIdx.x == 192) {
    const int32_t c = c_snap;
    const double b1 = 0.9, b2 = 0.999, thr = 5.0;
    const double t = (double)c + 1.0;
    const double b1t = pqn_powi(b1, c + 1), b2t = pqn_powi(b2, c + 1);   // square-and-multiply: ~40 f64 mults, no libm pow
    const double ro_inf = 2.0 / (1.0 - b2) - 1.0;
    const double ro = ro_inf - 2.0 * t * b2t / (1.0 - b2t);
    const int rect = ro >= thr;
    const float r = rect ? (float)sqrt((ro - 4.0) * (ro - 2.0) * ro_inf / ((ro_inf - 4.0) * (ro_inf - 2.0) * ro)) : 0.0f;
    float lr = lr_init;
    if (lr_steps > 0.0) {  // optax.linear_schedule evaluated at the pre-increment count
      double cc = (double)c;
      if (cc > lr_steps) cc = lr_steps;
      lr = (float)(((double)lr_init - (double)lr_end) * (1.0 - cc / lr_steps) + (double)lr_end);
    }
    s_sc[2] = (float)(1.0 - b1t);
    s_sc[3] = (float)(1.0 - b2t);
    s_sc[4] = rect ? 1.0f : 0.0f;
    s_sc[5] = r;
    s_sc[6] = lr;
    if (!FOLD && blockIdx.x == 0) *count = c + 1;
  }
  float acc = ((part[0] + part[1]) + part[2]) + part[3];   // == the strided loop's order
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float gnorm = sqrtf((s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
    s_sc[0] = gnorm;
    s_sc[1] = (gnorm < max_norm) ? 0.0f : 1.0f;
    if (blockIdx.x == 0 && gnorm_out) *gnorm_out = gnorm;
    if constexpr (FOLD) {
      // every lane of this block is past its wait: every block of the seed has published, hence has read the serial and the count
      if (blockIdx.x == 0) {
        reinterpret_cast<unsigned *>(scratch)[PQN_FOLD_SERIAL] = fold_tag;
        *count = s_cnt + 1;
      }
    }
  }
  __syncthreads();
  const float gnorm = s_sc[0];
  const bool clip = s_sc[1] != 0.0f;
  const float bc1 = s_sc[2], bc2 = s_sc[3];
  const bool rect = s_sc[4] != 0.0f;
  const float r = s_sc[5], lr = s_sc[6];
  const float c1 = (float)(1.0 - 0.9), d1 = (float)0.9, c2 = (float)(1.0 - 0.999), d2 = (float)0.999;
  auto step1 = [&](float gi, float &mi, float &vi, float pi) -> float {
    if (clip) gi = (gi / gnorm) * max_norm;
    mi = c1 * gi + d1 * mi;
    vi = c2 * (gi * gi) + d2 * vi;
    const float mh = mi / bc1;
    const float vh = vi / bc2;
    const float u = rect ? r * mh / (sqrtf(vh) + 1e-8f) : mh;
    return pi - lr * u;
  };
  auto mirror = [&](int64_t i, float pn) {  // keep the dgrad-fragment copy of the CNN fc1 kernel in step
    const int64_t j = i - w1_off;
    if (j >= 0 && j < 1024 * 128) {
      const int frag = (int)(j >> 8), ln = (int)(j >> 2) & 63, sx = (int)j & 3;
      const int gi = frag >> 3, cb = frag & 7, kk = ln >> 4, jj = ln & 15;
      const int jb = (((cb * 64 + gi) * 64 + (jj >> 2) * 16 + 4 * kk + sx) << 2) + (jj & 3);
      if (half_off && copy_mode >= 2) {   // bf16x3 layout: the six bf16 planes in the tail of the parameter buffer (3: + four f16x2 planes; 4: only those)
        if (copy_mode != 4) pqn_x3_store_planes(reinterpret_cast<unsigned short *>(p + half_off), 16 * gi + 4 * kk + sx, 16 * cb + jj, pn);
        if (copy_mode >= 3) pqn_h2_store_planes(reinterpret_cast<_Float16 *>(p + half_off + H2_PLANES_OFF), 16 * gi + 4 * kk + sx, 16 * cb + jj, pn);
        return;
      }
      w1b[jb] = pn;
      if (half_off) {   // fp16 operand copies of a matmul_f16 layout, in the tail of the parameter buffer itself
        _Float16 *w1h = reinterpret_cast<_Float16 *>(p + half_off);
        w1h[j] = (_Float16)pn;
        w1h[1024 * 128 + jb] = (_Float16)pn;
      }
    }
  };
  // bf16x3 layout: the fc1 kernel is walked fragment-aligned (one 256-element MFMA fragment per wave) so that the six
  // bf16 planes are written with 8-B stores -- four K-consecutive values per lane directly for the forward-order
  // planes, and after a 4 x 4 exchange inside each lane quad (through LDS) for the dgrad-order planes -- instead of
  // 24 two-byte stores per thread (measured: the optimizer kernel took 33 us per 16-seed launch against 18 without
  // the planes).  The generic loops below then skip that range.
  if (x3_w1) {
    __shared__ float s_tr[4][256];
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned short *planes = reinterpret_cast<unsigned short *>(p + half_off);
    unsigned short *hplanes = reinterpret_cast<unsigned short *>(p + half_off + H2_PLANES_OFF);   // copy_mode 3: f16x2 planes (hi, lo per order)
    constexpr int NQ = 1024 * 128 / 4, P = 1024 * 128;
    const int iters = (NQ + (int)gridDim.x * 256 - 1) / ((int)gridDim.x * 256);
    for (int it = 0; it < iters; ++it) {
      const int q = ((int)blockIdx.x + it * (int)gridDim.x) * 256 + threadIdx.x;   // float4 index inside the fc1 kernel
      const bool on = q < NQ;
      float pa[4] = {0.f, 0.f, 0.f, 0.f};
      if (on) {
        const int64_t i4 = (int64_t)(w1_off >> 2) + q;
        f4 g4 = pf_g, m4 = pf_m, v4 = pf_v, p4 = pf_p;    // first pass: requested at the top of the kernel
        if (it > 0) {
          g4 = reinterpret_cast<const f4 *>(g)[i4];
          m4 = reinterpret_cast<f4 *>(m)[i4]; v4 = reinterpret_cast<f4 *>(v)[i4]; p4 = reinterpret_cast<f4 *>(p)[i4];
        }
        const float ga[4] = {g4.x, g4.y, g4.z, g4.w};
        float ma[4] = {m4.x, m4.y, m4.z, m4.w}, va[4] = {v4.x, v4.y, v4.z, v4.w};
        pa[0] = p4.x; pa[1] = p4.y; pa[2] = p4.z; pa[3] = p4.w;
#pragma unroll
        for (int c = 0; c < 4; ++c) pa[c] = step1(ga[c], ma[c], va[c], pa[c]);
        reinterpret_cast<f4 *>(m)[i4] = f4{ma[0], ma[1], ma[2], ma[3]};
        reinterpret_cast<f4 *>(v)[i4] = f4{va[0], va[1], va[2], va[3]};
        reinterpret_cast<f4 *>(p)[i4] = f4{pa[0], pa[1], pa[2], pa[3]};
      }
      // element (i = 16 gi + 4 kk + sx, o = 16 cb + jj), sx = 0..3 in this lane
      const int frag = q >> 6, gi = frag >> 3, cb = frag & 7, kk = lane >> 4, jj = lane & 15;
      auto split4 = [&](const float (&x)[4], u2 &H, u2 &M, u2 &Lo) {
        unsigned short h[4], mm[4], l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          h[c] = pqn_bf16_rne(x[c]);
          const float r1 = x[c] - pqn_bf16_to_f32(h[c]);
          mm[c] = pqn_bf16_rne(r1);
          l[c] = pqn_bf16_rne(r1 - pqn_bf16_to_f32(mm[c]));
        }
        H = u2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
        M = u2{(unsigned)mm[0] | ((unsigned)mm[1] << 16), (unsigned)mm[2] | ((unsigned)mm[3] << 16)};
        Lo = u2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
      };
      auto split4h = [&](const float (&x)[4], u2 &H, u2 &Lo) {
        unsigned short h[4], l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          _Float16 hh, ll;
          pqn_h2_split1(x[c], hh, ll);
          h[c] = __builtin_bit_cast(unsigned short, hh);
          l[c] = __builtin_bit_cast(unsigned short, ll);
        }
        H = u2{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
        Lo = u2{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
      };
      if (on) {   // forward-order planes: the lane's four values are K-consecutive (same index math as pqn_x3_store_planes)
        const int i0 = 16 * gi + 4 * kk, o = 16 * cb + jj;
        const int jf = ((((i0 >> 5) * 8 + (o >> 4)) * 64 + ((i0 >> 2) & 3) * 16 + (o & 15)) << 3) + 4 * ((i0 >> 4) & 1);
        u2 H, M, Lo;
        if (copy_mode != 4) {
          split4(pa, H, M, Lo);
          *reinterpret_cast<u2 *>(planes + jf) = H;
          *reinterpret_cast<u2 *>(planes + P + jf) = M;
          *reinterpret_cast<u2 *>(planes + 2 * P + jf) = Lo;
        }
        if (copy_mode >= 3) {
          split4h(pa, H, Lo);
          *reinterpret_cast<u2 *>(hplanes + jf) = H;
          *reinterpret_cast<u2 *>(hplanes + P + jf) = Lo;
        }
      }
      // dgrad-order planes want four consecutive o for one i: exchange inside the quad of lanes jj = 4a .. 4a+3
      __syncthreads();
      *reinterpret_cast<f4 *>(&s_tr[wave][lane * 4]) = f4{pa[0], pa[1], pa[2], pa[3]};
      __syncthreads();
      if (on) {
        const int r = lane & 3, lq = lane & ~3;
        float tv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) tv[c] = s_tr[wave][(lq + c) * 4 + r];   // value (i = i0 + r, o = o0 + c)
        const int i = 16 * gi + 4 * kk + r, o0 = 16 * cb + (jj & ~3);
        const int jd = (((((i >> 4) * 4 + (o0 >> 5)) * 64) + ((o0 >> 2) & 3) * 16 + (i & 15)) << 3) + 4 * ((o0 >> 4) & 1);
        u2 H, M, Lo;
        if (copy_mode != 4) {
          split4(tv, H, M, Lo);
          *reinterpret_cast<u2 *>(planes + 3 * P + jd) = H;
          *reinterpret_cast<u2 *>(planes + 4 * P + jd) = M;
          *reinterpret_cast<u2 *>(planes + 5 * P + jd) = Lo;
        }
        if (copy_mode >= 3) {
          split4h(tv, H, Lo);
          *reinterpret_cast<u2 *>(hplanes + 2 * P + jd) = H;
          *reinterpret_cast<u2 *>(hplanes + 3 * P + jd) = Lo;
        }
      }
    }
  }
  if constexpr (FOLD) {   // the elements this block folded; the generic walks below belong to the two-launch form
    if (!x3_w1 && blockIdx.x < QR_W1_BLOCKS) {   // layouts without operand planes: the lane's float4 of the fc1 kernel (+ its mirrors)
      const int64_t i4 = (int64_t)(w1_off >> 2) + (int)blockIdx.x * 256 + threadIdx.x;
      const float ga[4] = {pf_g.x, pf_g.y, pf_g.z, pf_g.w};
      float ma[4] = {pf_m.x, pf_m.y, pf_m.z, pf_m.w}, va[4] = {pf_v.x, pf_v.y, pf_v.z, pf_v.w}, pa[4] = {pf_p.x, pf_p.y, pf_p.z, pf_p.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) pa[c] = step1(ga[c], ma[c], va[c], pa[c]);
      reinterpret_cast<f4 *>(m)[i4] = f4{ma[0], ma[1], ma[2], ma[3]};
      reinterpret_cast<f4 *>(v)[i4] = f4{va[0], va[1], va[2], va[3]};
      reinterpret_cast<f4 *>(p)[i4] = f4{pa[0], pa[1], pa[2], pa[3]};
      if (w1b) {
#pragma unroll
        for (int c = 0; c < 4; ++c) mirror(4 * i4 + c, pa[c]);
      }
    }
    if (fold_i >= 0) {   // wave 0 of a non-fc1 block: lane = element
      float mi = fold_m, vi = fold_v;
      const float pn = step1(fold_g, mi, vi, fold_p);
      m[fold_i] = mi;
      v[fold_i] = vi;
      p[fold_i] = pn;
    }
    return;
  }
  const int64_t w1_lo4 = x3_w1 ? (w1_off >> 2) : 0, w1_hi4 = x3_w1 ? (w1_off >> 2) + 1024 * 128 / 4 : 0;
  for (int64_t i4 = (int64_t)blockIdx.x * 256 + threadIdx.x; i4 < n4; i4 += (int64_t)gridDim.x * 256) {
    if (i4 >= w1_lo4 && i4 < w1_hi4) continue;   // done above
    const f4 g4 = reinterpret_cast<const f4 *>(g)[i4];
    const f4 m4 = reinterpret_cast<f4 *>(m)[i4], v4 = reinterpret_cast<f4 *>(v)[i4], p4 = reinterpret_cast<f4 *>(p)[i4];
    const float ga[4] = {g4.x, g4.y, g4.z, g4.w};
    float ma[4] = {m4.x, m4.y, m4.z, m4.w}, va[4] = {v4.x, v4.y, v4.z, v4.w}, pa[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) pa[c] = step1(ga[c], ma[c], va[c], pa[c]);
    reinterpret_cast<f4 *>(m)[i4] = f4{ma[0], ma[1], ma[2], ma[3]};
    reinterpret_cast<f4 *>(v)[i4] = f4{va[0], va[1], va[2], va[3]};
    reinterpret_cast<f4 *>(p)[i4] = f4{pa[0], pa[1], pa[2], pa[3]};
    if (w1b) {
#pragma unroll
      for (int c = 0; c < 4; ++c) mirror(4 * i4 + c, pa[c]);
    }
  }
  for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float mi = m[i], vi = v[i];
    const float pn = step1(g[i], mi, vi, p[i]);
    m[i] = mi;
    v[i] = vi;
    p[i] = pn;
    if (w1b) mirror(i, pn);
  }
}

int pqn_launch_radam(float *p, const float *g, float *m, float *v, int64_t n, int32_t *count, float lr_init,
                     float lr_end, double lr_steps, float max_norm, float *scratch, float *gnorm_out, int w1_off,
                     float *w1b, int norm_pass, int nparts, hipStream_t st, int nseeds, long long pstride,
                     long long sstride, long long w1bstride, int half_off, int copy_mode) {
  // nparts: number of sum-of-squares partials already in scratch when norm_pass == 0
  const int blocks = pqn_radam_blocks(n);
  if (norm_pass) {
    if (nseeds != 1) {
      pqn_set_error("radam: the separate norm pass is single-seed only");
      return PQN_E_INVALID;
    }
    hipLaunchKernelGGL(radam_norm_kernel, dim3(blocks), dim3(256), 0, st, g, n, count, scratch);
    nparts = blocks;
  }
  hipLaunchKernelGGL(radam_apply_kernel<false>, dim3(blocks, nseeds), dim3(256), 0, st, p, g, m, v, n, count, lr_init, lr_end,
                     lr_steps, max_norm, nparts, scratch, gnorm_out, w1_off, w1b, pstride, sstride, w1bstride, half_off, copy_mode, 0);
  return pqn_check_launch("radam");
}

// Fold + clip + RAdam of a CNN training step in one launch (radam_apply_kernel<true>; pqn_fold.h).  g: where the folded gradient is
// ALSO stored (the flat gradient buffer), or NULL.  Option fold_apply = 0 (or a shape it does not cover) -> PQN_E_UNSUPPORTED: the
// caller then takes the two launches.
int pqn_launch_radam_fold(const pqn_fold_args_t &fa, float *p, float *g, float *m, float *v, int32_t *count, float lr_init, float lr_end,
                          double lr_steps, float max_norm, float *scratch, float *w1b, hipStream_t st, int nseeds, long long pstride,
                          long long sstride, long long w1bstride, int half_off, int copy_mode) {
  const int blocks = grad_reduce_blocks(fa.L.total);
  if (!fa.valid || blocks > PQN_FOLD_MAX_BLOCKS || (fa.L.total & 3) || (fa.L.off_w1 & 3) ||
      ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) || (pstride & 3) || (sstride & 1))
    return PQN_E_UNSUPPORTED;
  hipLaunchKernelGGL(radam_apply_kernel<true>, dim3(blocks, nseeds), dim3(256), 0, st, p, g, m, v, (int64_t)fa.L.total, count, lr_init,
                     lr_end, lr_steps, max_norm, blocks, scratch, (float *)nullptr, fa.L.off_w1, w1b, pstride, sstride, w1bstride, half_off,
                     copy_mode, fa);
  return pqn_check_launch("radam (fold)");
}

extern "C" int pqn_radam_clip_step(float *p, const float *g, float *m, float *v, int64_t n, int32_t *count,
                                   float lr_init, float lr_end, double lr_steps, float max_norm, float *scratch,
                                   float *gnorm_out, void *stream) {
  PQN_REQUIRE(p && g && m && v && count && scratch, "pqn_radam_clip_step: NULL argument");
  PQN_REQUIRE(n > 0, "pqn_radam_clip_step: n must be > 0");
  return pqn_launch_radam(p, g, m, v, n, count, lr_init, lr_end, lr_steps, max_norm, scratch, gnorm_out, 0, nullptr, 1, 0,
                          (hipStream_t)stream);
}
