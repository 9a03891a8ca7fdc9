// pqn_update.hip -- ONE whole PQN update (`_update_step`, purejaxql/pqn_minatar.py:176-369) enqueued from
// C++: T x (Q-net forward + eps-greedy, env step), bootstrap forward, Q(lambda), NUM_EPOCHS x (shuffle,
// NUM_MINIBATCHES x (grad, clip+RAdam)), metric reductions.  ~340 kernel launches from one call, no host
// synchronisation and no host-side state: everything that changes from update to update (step keys,
// epoch keys, eps, the metrics row) is derived on the DEVICE from a clock word, so the same enqueue is
// valid for every update and can be captured once in a hipGraph and replayed.
//
// Key schedule (identical to purejaxql_amd/pqn.py and the oracle):
//   step key  (u,t)  = fold_in(key_roll, u*T + t)      epoch key (u,ep) = fold_in(key_shuf, u*EPOCHS + ep)
//   eps(u) = optax.linear_schedule(eps_start, eps_finish, eps_decay_steps)(u)   (pqn_minatar.py:134-138,195)
#include <rocprim/device/device_radix_sort.hpp>

#include "pqn_common.h"
#include "pqn_fold.h"

// scalar + mean metrics of one update, in the order of pqn_minatar.py:330-338
enum { M_ENV_STEP, M_UPDATE_STEPS, M_ENV_FRAME, M_GRAD_STEPS, M_TD_LOSS, M_QVALS, M_DISCOUNT, M_RET_RETURNS,
       M_RET_LENGTHS, M_TIMESTEP, M_RET_EPISODE, M_COUNT };
static_assert(M_COUNT == PQN_NUM_METRICS, "metrics row layout");

// grid = seeds: block s derives the keys of seed s (key_*_dev[s] when given) into keys[s*(t_len+epochs) ...];
// block 0 also latches the update index into clock[1] for the tick kernel and writes eps
__global__ void update_sched_kernel(int32_t *__restrict__ clock, uint64_t key_roll, uint64_t key_shuf,
                                    const uint64_t *__restrict__ key_roll_dev, const uint64_t *__restrict__ key_shuf_dev,
                                    int t_len, int epochs, float eps_start, float eps_finish, double eps_decay_steps,
                                    uint64_t *__restrict__ keys, float *__restrict__ eps, float *__restrict__ workspace,
                                    long long ws_stride) {
  const int u = clock[0];
  const int i = threadIdx.x;
  // ticket counter of the fused optimizer kernel's grid barrier (scratch word 1022 of this seed's workspace)
  if (workspace && i == 0) reinterpret_cast<unsigned *>(workspace + blockIdx.x * ws_stride)[1022] = 0u;
  if (key_roll_dev) {
    key_roll = key_roll_dev[blockIdx.x];
    key_shuf = key_shuf_dev[blockIdx.x];
  }
  keys += (size_t)blockIdx.x * (t_len + epochs);
  if (blockIdx.x == 0 && i == 0) clock[1] = u;
  if (i < t_len) keys[i] = pqn_fold(key_roll, (uint32_t)(u * t_len + i));
  else if (i < t_len + epochs) keys[i] = pqn_fold(key_shuf, (uint32_t)(u * epochs + (i - t_len)));
  if (blockIdx.x == 0 && i == 0) {
    double e = eps_start;   // optax.linear_schedule with transition_steps <= 0: a constant schedule at init_value
    if (eps_decay_steps > 0.0) {
      double c = (double)u;
      if (c > eps_decay_steps) c = eps_decay_steps;
      e = ((double)eps_start - (double)eps_finish) * (1.0 - c / eps_decay_steps) + (double)eps_finish;
    }
    *eps = (float)e;
  }
}

// grid (5 arrays, MEANS_CHUNKS chunks): fixed-order partial sums of the [T*N] info arrays; the tick
// kernel folds the chunk partials.  (info means of pqn_minatar.py:338)
#define MEANS_CHUNKS 64

// grid (5, MEANS_CHUNKS, seeds): seed s reads its columns [s*n_env, (s+1)*n_env) of the stacked [T][n_env_total] arrays
__global__ __launch_bounds__(256) void update_means_kernel(int count, int n_env, int n_env_total,
                                                           const float *__restrict__ discount,
                                                           const float *__restrict__ rer, const int32_t *__restrict__ rel,
                                                           const int32_t *__restrict__ ts,
                                                           const uint8_t *__restrict__ done, double *__restrict__ partial,
                                                           long long partial_stride, int weighted = 0) {
  __shared__ double s_part[4];
  const int which = blockIdx.x, chunk = blockIdx.y, seed = blockIdx.z;
  partial += seed * partial_stride;
  const int per = (count + MEANS_CHUNKS - 1) / MEANS_CHUNKS;
  const int lo = chunk * per, hi = min(count, lo + per);
  double acc = 0.0;
  for (int j = lo + threadIdx.x; j < hi; j += 256) {
    size_t i = (size_t)j;
    if (n_env_total != n_env) {
      const int t = j / n_env;
      i = (size_t)t * n_env_total + (size_t)seed * n_env + (j - t * n_env);
    }
    float v;
    switch (which) {
      case 0: v = discount[i]; break;
      case 1: v = rer[i]; break;
      case 2: v = (float)rel[i]; break;
      case 3: v = (float)ts[i]; break;
      default: v = (float)done[i]; break;
    }
    // weighted: (x * returned_episode).sum() of pqn_craftax.py:364-369 (column 4 then holds returned_episode.sum())
    acc += (weighted && !done[i]) ? 0.0 : (double)v;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[which * MEANS_CHUNKS + chunk] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// grid = seeds; the update index was latched into clock[1] by the sched kernel, block 0 advances clock[0]
__global__ void update_tick_kernel(int32_t *__restrict__ clock, int t_len, int n, int channels, int n_mb_total,
                                   const float *__restrict__ loss_buf, const float *__restrict__ qv_buf,
                                   const double *__restrict__ partial, double *__restrict__ metrics, int capacity,
                                   long long partial_stride, int weighted = 0) {
  const int u = clock[1];
  loss_buf += (size_t)blockIdx.x * n_mb_total;
  qv_buf += (size_t)blockIdx.x * n_mb_total;
  partial += blockIdx.x * partial_stride;
  metrics += (size_t)blockIdx.x * capacity * M_COUNT;
  if (u < capacity) {
    double *row = metrics + (size_t)u * M_COUNT;
    if (threadIdx.x < 5) {
      double s = 0.0;
      for (int c = 0; c < MEANS_CHUNKS; ++c) s += partial[threadIdx.x * MEANS_CHUNKS + c];
      double den = (double)t_len * (double)n;
      if (weighted) {   // / returned_episode.sum(): 0 / 0 = NaN when no episode finished, as in the reference
        den = 0.0;
        for (int c = 0; c < MEANS_CHUNKS; ++c) den += partial[4 * MEANS_CHUNKS + c];
      }
      row[M_DISCOUNT + threadIdx.x] = s / den;
    }
    if (threadIdx.x == 5) {
      double l = 0.0, qv = 0.0;
      for (int i = 0; i < n_mb_total; ++i) { l += (double)loss_buf[i]; qv += (double)qv_buf[i]; }
      const double steps = (double)(u + 1) * (double)t_len * (double)n;
      row[M_ENV_STEP] = steps;
      row[M_UPDATE_STEPS] = (double)(u + 1);
      row[M_ENV_FRAME] = steps * (double)channels;
      row[M_GRAD_STEPS] = (double)(u + 1) * (double)n_mb_total;
      row[M_TD_LOSS] = l / (double)n_mb_total;
      row[M_QVALS] = qv / (double)n_mb_total;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) clock[0] = u + 1;
}

// ---------------------------------------------------------------------------------------------------------
// The epoch shuffle's sort (jax.random.permutation's stand-in, pqn_minatar.py:299-315): every seed's T*N keys
// `rand31 << ib | index` in ascending order.  Round 6: a sort written for exactly these keys -- uniformly random leading bits,
// unique, a segment per seed:
//   n <= 4096 keys per seed: ONE launch, a workgroup per seed sorts its segment in LDS (bitonic network on the full keys);
//   above: the leading bits of rand31 cut a seed's keys into B = 2^b buckets of ~1024 (b = ceil(log2(n / 1024))) --
//     count (LDS histogram -> one global atomic per workgroup and bucket), scatter into the bucket ranges of `out` (claimed the same way;
//     the order inside a bucket is whatever the atomics give), then a workgroup per bucket sorts its <= 4096 keys in LDS, in place.
//     A bucket holds 1024 +- 32 keys (binomial); one above 4096 is not a practical event, and is still sorted (rank sort through the
//     dead input buffer).  4 launches and 80 MB of traffic at 16 seeds x 131,072 keys where rocPRIM's onesweep took 5 passes + 11 fills.
// The keys are unique, so the result is THE sorted order whatever the algorithm: tests/test_fullsize_gpu.py compares it with the
// full-width sort of the same keys at every shape.  Option sort_impl = 0 keeps rocPRIM's radix sort (restricted to the random + seed
// bits: the keys enter in index order and an LSD radix sort is stable, so equal draws keep their index order -- 4 passes, not 6).
// ---------------------------------------------------------------------------------------------------------
#define SRT_CAP 4096          // keys a workgroup sorts in LDS (32 KB)
#define SRT_THREADS 256           // (1024: 98 us per 16-seed bucket launch -- barriers of 16 waves, two workgroups per CU; 256: see profiles/r06_v13_sort.txt)
typedef unsigned long long srt_key_t;

// bitonic network over s[0..P), P a power of two <= SRT_CAP, ascending; every thread of the workgroup calls it
__device__ inline void srt_bitonic(srt_key_t *s, int P, int tid, int nt) {
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (P >> 1); t += nt) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
        const srt_key_t x = s[i], y = s[l];
        if ((x > y) == ((i & k) == 0)) { s[i] = y; s[l] = x; }
      }
      __syncthreads();
    }
}

// one workgroup per (bucket, seed): keys [base, base + cnt) of the seed's segment.  whole != 0: the segment itself (from `in`).
__global__ __launch_bounds__(SRT_THREADS) void srt_bucket_sort_kernel(const srt_key_t *__restrict__ in, srt_key_t *__restrict__ out,
                                                                      srt_key_t *__restrict__ spare, int n, int B,
                                                                      const unsigned *__restrict__ counts,
                                                                      const unsigned *__restrict__ bases, int whole, int cap, int sub_shift) {
  constexpr int SB = 1024;          // sub-buckets of the two-level paths
  __shared__ srt_key_t s[SRT_CAP];
  __shared__ unsigned s_c[SB], s_off[SB + 1], s_sc[SRT_THREADS];
  const int seed = blockIdx.y, bucket = blockIdx.x, tid = threadIdx.x;
  unsigned base = 0, cnt = (unsigned)n;
  if (!whole) {
    base = bases[(size_t)seed * B + bucket];
    cnt = counts[(size_t)seed * B + bucket];
  }
  const srt_key_t *src = (whole ? in : out) + (size_t)seed * n + base;
  srt_key_t *dst = out + (size_t)seed * n + base;
  if (cnt > (unsigned)cap) {   // cap = SRT_CAP; not a practical event (see above; tests lower cap to walk this path): rank sort, the (dead) input buffer as the stable copy
    srt_key_t *cp = spare + (size_t)seed * n + base;
    for (unsigned i = tid; i < cnt; i += SRT_THREADS) cp[i] = src[i];
    __threadfence_block();
    __syncthreads();
    for (unsigned i = tid; i < cnt; i += SRT_THREADS) {
      const srt_key_t k = cp[i];
      unsigned r = 0;
      for (unsigned q = 0; q < cnt; ++q) r += cp[q] < k;
      dst[r] = k;
    }
    return;
  }
  if (whole && cnt <= SRT_CAP && sub_shift >= 0) {
    // a whole segment of at most 4096 keys (one workgroup per seed): the same two-level idea with the keys in registers -- the leading ten
    // random bits cut the segment into 1024 sub-buckets of <= ~4 keys (a bitonic network here took ~45 us, rocPRIM's block / merge sort
    // kernels 25-35 us over six launches)
    constexpr int NK = SRT_CAP / SRT_THREADS;
    unsigned *w_c = s_c, *w_off = s_off, *w_sc = s_sc;
    srt_key_t k[NK];
#pragma unroll
    for (int u = 0; u < NK; ++u) { const int i = u * SRT_THREADS + tid; k[u] = i < (int)cnt ? src[i] : 0; }
    for (int q = tid; q < SB; q += SRT_THREADS) w_c[q] = 0u;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NK; ++u)
      if (u * SRT_THREADS + tid < (int)cnt) atomicAdd(&w_c[(unsigned)(k[u] >> sub_shift) & (SB - 1)], 1u);
    __syncthreads();
    {
      constexpr int per = SB / SRT_THREADS;
      unsigned local = 0;
#pragma unroll
      for (int u = 0; u < per; ++u) local += w_c[tid * per + u];
      w_sc[tid] = local;
      __syncthreads();
      for (int off = 1; off < SRT_THREADS; off <<= 1) {
        const unsigned v = tid >= off ? w_sc[tid - off] : 0u;
        __syncthreads();
        w_sc[tid] += v;
        __syncthreads();
      }
      unsigned run = w_sc[tid] - local;
#pragma unroll
      for (int u = 0; u < per; ++u) {
        const unsigned c = w_c[tid * per + u];
        w_off[tid * per + u] = run;
        w_c[tid * per + u] = run;
        run += c;
      }
      if (tid == SRT_THREADS - 1) w_off[SB] = run;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NK; ++u)
      if (u * SRT_THREADS + tid < (int)cnt) s[atomicAdd(&w_c[(unsigned)(k[u] >> sub_shift) & (SB - 1)], 1u)] = k[u];
    __syncthreads();
    for (int q = tid; q < SB; q += SRT_THREADS) {
      const int lo = (int)w_off[q], hi = (int)w_off[q + 1];
      for (int i = lo + 1; i < hi; ++i) {
        const srt_key_t kk = s[i];
        int j = i - 1;
        while (j >= lo && s[j] > kk) { s[j + 1] = s[j]; --j; }
        s[j + 1] = kk;
      }
    }
    __syncthreads();
    for (int i = tid; i < (int)cnt; i += SRT_THREADS) dst[i] = s[i];
    return;
  }
  if (!whole && cnt <= SRT_CAP / 2 && sub_shift >= 0) {
    // the usual case: ~1024 keys.  A bitonic network over them is bound by LDS bandwidth (66 stages x 48 B per comparator: 90 us per
    // 16-seed launch).  Instead: the NEXT ten random bits cut the bucket into 1024 sub-buckets of ~1 key -- count (LDS atomics), scan,
    // place (LDS atomics: unordered inside a sub-bucket), then one thread per sub-bucket insertion-sorts its handful of keys on the
    // full key.  Three passes over the keys; correct for any key distribution (a crowded sub-bucket only costs time).
    constexpr int H = SRT_CAP / 2;
    srt_key_t *s_in = s, *s_out = s + H;
    for (int q = tid; q < SB; q += SRT_THREADS) s_c[q] = 0u;
    for (int i = tid; i < (int)cnt; i += SRT_THREADS) s_in[i] = src[i];
    __syncthreads();
    for (int i = tid; i < (int)cnt; i += SRT_THREADS) atomicAdd(&s_c[(unsigned)(s_in[i] >> sub_shift) & (SB - 1)], 1u);
    __syncthreads();
    {
      constexpr int per = SB / SRT_THREADS;
      unsigned local = 0;
#pragma unroll
      for (int u = 0; u < per; ++u) local += s_c[tid * per + u];
      s_sc[tid] = local;
      __syncthreads();
      for (int off = 1; off < SRT_THREADS; off <<= 1) {
        const unsigned v = tid >= off ? s_sc[tid - off] : 0u;
        __syncthreads();
        s_sc[tid] += v;
        __syncthreads();
      }
      unsigned run = s_sc[tid] - local;
#pragma unroll
      for (int u = 0; u < per; ++u) {
        const unsigned c = s_c[tid * per + u];
        s_off[tid * per + u] = run;
        s_c[tid * per + u] = run;     // the placing pass's cursor
        run += c;
      }
      if (tid == SRT_THREADS - 1) s_off[SB] = run;
    }
    __syncthreads();
    for (int i = tid; i < (int)cnt; i += SRT_THREADS) {
      const srt_key_t k = s_in[i];
      s_out[atomicAdd(&s_c[(unsigned)(k >> sub_shift) & (SB - 1)], 1u)] = k;
    }
    __syncthreads();
    for (int q = tid; q < SB; q += SRT_THREADS) {
      const int lo = (int)s_off[q], hi = (int)s_off[q + 1];
      for (int i = lo + 1; i < hi; ++i) {
        const srt_key_t k = s_out[i];
        int j = i - 1;
        while (j >= lo && s_out[j] > k) { s_out[j + 1] = s_out[j]; --j; }
        s_out[j + 1] = k;
      }
    }
    __syncthreads();
    for (int i = tid; i < (int)cnt; i += SRT_THREADS) dst[i] = s_out[i];
    return;
  }
  int P = 2;
  while (P < (int)cnt) P <<= 1;
  for (int i = tid; i < P; i += SRT_THREADS) s[i] = i < (int)cnt ? src[i] : ~(srt_key_t)0;
  __syncthreads();
  srt_bitonic(s, P, tid, SRT_THREADS);
  for (int i = tid; i < (int)cnt; i += SRT_THREADS) dst[i] = s[i];
}

__global__ __launch_bounds__(256) void srt_zero_kernel(unsigned *__restrict__ c, int count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) c[i] = 0u;
}

#define SRT_ITEMS 8           // keys per thread of the count / scatter kernels
// bucket of a key: the leading b bits of its rand31 (bits [ib + 31 - b, ib + 31))
__global__ __launch_bounds__(256) void srt_count_kernel(const srt_key_t *__restrict__ in, int n, int shift, int B,
                                                        unsigned *__restrict__ counts) {
  extern __shared__ unsigned s_cnt[];
  const int seed = blockIdx.y, tid = threadIdx.x;
  for (int q = tid; q < B; q += 256) s_cnt[q] = 0u;
  __syncthreads();
  const srt_key_t *src = in + (size_t)seed * n;
  const int i0 = blockIdx.x * (256 * SRT_ITEMS);
#pragma unroll
  for (int u = 0; u < SRT_ITEMS; ++u) {
    const int i = i0 + u * 256 + tid;
    if (i < n) atomicAdd(&s_cnt[(unsigned)(src[i] >> shift) & (unsigned)(B - 1)], 1u);
  }
  __syncthreads();
  for (int q = tid; q < B; q += 256)
    if (s_cnt[q]) atomicAdd(&counts[(size_t)seed * B + q], s_cnt[q]);
}

__global__ __launch_bounds__(256) void srt_scatter_kernel(const srt_key_t *__restrict__ in, srt_key_t *__restrict__ out, int n, int shift,
                                                          int B, const unsigned *__restrict__ counts, unsigned *__restrict__ cursors,
                                                          unsigned *__restrict__ bases) {
  extern __shared__ unsigned s_mem[];   // [B] local counts -> this workgroup's first slot in every bucket, [B] bucket bases
  unsigned *s_cnt = s_mem, *s_base = s_mem + B;
  __shared__ unsigned s_scan[256];
  const int seed = blockIdx.y, tid = threadIdx.x;
  for (int q = tid; q < B; q += 256) { s_cnt[q] = 0u; s_base[q] = counts[(size_t)seed * B + q]; }
  __syncthreads();
  {   // exclusive scan of the seed's bucket counts: thread = a run of `per` consecutive buckets, Hillis-Steele over the 256 run sums
    const int per = (B + 255) / 256;
    unsigned local = 0;
    for (int u = 0; u < per; ++u) { const int q = tid * per + u; if (q < B) local += s_base[q]; }
    s_scan[tid] = local;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      const unsigned v = tid >= off ? s_scan[tid - off] : 0u;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    unsigned run = s_scan[tid] - local;
    for (int u = 0; u < per; ++u) {
      const int q = tid * per + u;
      if (q < B) { const unsigned c = s_base[q]; s_base[q] = run; run += c; }
    }
  }
  __syncthreads();
  if (blockIdx.x == 0) for (int q = tid; q < B; q += 256) bases[(size_t)seed * B + q] = s_base[q];   // for the bucket sorts
  const srt_key_t *src = in + (size_t)seed * n;
  const int i0 = blockIdx.x * (256 * SRT_ITEMS);
  srt_key_t k[SRT_ITEMS];
  unsigned pos[SRT_ITEMS];
#pragma unroll
  for (int u = 0; u < SRT_ITEMS; ++u) {
    const int i = i0 + u * 256 + tid;
    k[u] = i < n ? src[i] : 0;
  }
#pragma unroll
  for (int u = 0; u < SRT_ITEMS; ++u) {
    const int i = i0 + u * 256 + tid;
    pos[u] = i < n ? atomicAdd(&s_cnt[(unsigned)(k[u] >> shift) & (unsigned)(B - 1)], 1u) : 0u;
  }
  __syncthreads();
  for (int q = tid; q < B; q += 256) {
    const unsigned c = s_cnt[q];
    s_cnt[q] = c ? s_base[q] + atomicAdd(&cursors[(size_t)seed * B + q], c) : 0u;
  }
  __syncthreads();
  srt_key_t *dst = out + (size_t)seed * n;
#pragma unroll
  for (int u = 0; u < SRT_ITEMS; ++u) {
    const int i = i0 + u * 256 + tid;
    if (i < n) dst[s_cnt[(unsigned)(k[u] >> shift) & (unsigned)(B - 1)] + pos[u]] = k[u];
  }
}

static size_t srt_counter_bytes(long long n_total) { return (size_t)(n_total / 32 + 8192); }   // 3 x u32 per bucket (count, cursor, base), buckets <= 2 n / 1024 + seeds

extern "C" int64_t pqn_update_sort_temp_bytes(int32_t n) {
  size_t bytes = 0;
  if (n <= 0) return -1;
  if (rocprim::radix_sort_keys(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                               (unsigned int)n, 0u, 64u, (hipStream_t) nullptr) != hipSuccess)
    return -1;
  return (int64_t)(bytes + srt_counter_bytes(n));
}

// n = keys of ALL seeds, n_per_seed = T*N of one seed.  `in` is left intact.
static int pqn_sort_keys(void *temp, size_t temp_bytes, const int64_t *in, int64_t *out, int n, int nseeds, int n_per_seed,
                         hipStream_t st) {
  const int ib = pqn_index_bits(n_per_seed);
  const int impl = pqn_opt(PQN_OPT_SORT_IMPL);
  // (segments of <= 4096 keys -- one workgroup per seed -- are no faster than rocPRIM's block / merge sort kernels: yaml default 2.303 vs 2.289 ms per
  //  update, C5 equal; sort_impl = 2 takes the library's sort at every size: tests)
  if (impl != 0 && (n_per_seed > SRT_CAP || impl == 2) && n_per_seed <= (1 << 22) && temp_bytes >= srt_counter_bytes(n)) {
    const srt_key_t *kin = reinterpret_cast<const srt_key_t *>(in);
    srt_key_t *kout = reinterpret_cast<srt_key_t *>(out);
    if (n_per_seed <= SRT_CAP) {
      hipLaunchKernelGGL(srt_bucket_sort_kernel, dim3(1, nseeds), dim3(SRT_THREADS), 0, st, kin, kout, (srt_key_t *)nullptr, n_per_seed, 1,
                         (const unsigned *)nullptr, (const unsigned *)nullptr, 1, SRT_CAP,
                         pqn_opt(PQN_OPT_SORT_CAP) > 0 ? -1 : ib + 31 - 10);   // (sort_cap set: the bitonic network, for the tests)
      return pqn_check_launch("pqn_sort_keys");
    }
    const int cap_opt = pqn_opt(PQN_OPT_SORT_CAP);   // tests only: a lower in-LDS capacity sends every bucket down the rank-sort path
    int b = 1;
    while ((1 << b) * 1024 < n_per_seed) ++b;
    const int B = 1 << b, shift = ib + 31 - b;
    unsigned *counts = reinterpret_cast<unsigned *>(temp), *cursors = counts + (size_t)nseeds * B, *bases = cursors + (size_t)nseeds * B;
    const int ctr = 2 * nseeds * B, chunks = (n_per_seed + 256 * SRT_ITEMS - 1) / (256 * SRT_ITEMS);
    hipLaunchKernelGGL(srt_zero_kernel, dim3((ctr + 255) / 256), dim3(256), 0, st, counts, ctr);
    hipLaunchKernelGGL(srt_count_kernel, dim3(chunks, nseeds), dim3(256), B * sizeof(unsigned), st, kin, n_per_seed, shift, B, counts);
    hipLaunchKernelGGL(srt_scatter_kernel, dim3(chunks, nseeds), dim3(256), 2 * B * sizeof(unsigned), st, kin, kout, n_per_seed, shift, B,
                       counts, cursors, bases);
    // (the rank-sort path of an oversize bucket copies through `in`: const_cast, see srt_bucket_sort_kernel -- never taken in practice)
    hipLaunchKernelGGL(srt_bucket_sort_kernel, dim3(B, nseeds), dim3(SRT_THREADS), 0, st, kin, kout, const_cast<srt_key_t *>(kin), n_per_seed,
                       B, counts, bases, 0, (cap_opt > 0 && cap_opt < SRT_CAP) ? cap_opt : SRT_CAP, shift - 10);
    return pqn_check_launch("pqn_sort_keys");
  }
  const unsigned begin_bit = (unsigned)ib;
  const unsigned end_bit = (unsigned)(31 + ib + (nseeds > 1 ? 7 : 0));
  size_t need = 0;
  if (rocprim::radix_sort_keys(nullptr, need, (const unsigned long long *)in, (unsigned long long *)out, (unsigned int)n, begin_bit,
                               end_bit, st) != hipSuccess || need > temp_bytes) {
    pqn_set_error("radix sort: %llu temp bytes provided, %llu needed", (unsigned long long)temp_bytes, (unsigned long long)need);
    return PQN_E_INVALID;
  }
  if (rocprim::radix_sort_keys(temp, need, (const unsigned long long *)in, (unsigned long long *)out, (unsigned int)n, begin_bit, end_bit,
                               st) != hipSuccess) {
    pqn_set_error("radix sort failed (temp bytes %llu)", (unsigned long long)temp_bytes);
    return PQN_E_HIP;
  }
  return PQN_OK;
}

#define UPD_CHECK(call)          \
  do {                           \
    const int rc_ = (call);      \
    if (rc_ != PQN_OK) return rc_; \
  } while (0)

// S seeds per launch (S = 1: the single-seed entry point).  With S > 1 every buffer in `a` is the stacked
// allocation of all seeds: env-indexed arrays are [..][S*N] (seed s owns envs s*N .. s*N+N-1), parameter-like
// buffers are [S][stride]; key_roll_dev / key_shuf_dev hold the per-seed keys.
//
// The update is enqueued in PHASES (pqn_cnn_update = all of them in order; pqn_cnn_update_phase = one at a time, so
// that a caller can put a collective between the gradient and the optimizer step of every minibatch -- the envs of
// one seed sharded over ranks):
//   BEGIN    clock -> step keys / eps; the rollout scan + bootstrap forward; Q(lambda) targets
//   SHUFFLE  (ep)    epoch permutation (threefry sort keys + radix sort)
//   GRAD     (i_mb)  forward + backward of minibatch i_mb -> flat gradient a->grad (+ its sum-of-squares partials)
//   APPLY    (i_mb)  clip_by_global_norm + RAdam from a->grad (norm recomputed when the caller changed the gradient)
//   END      carry last_obs, metrics row, clock tick
struct UpdCtx {
  int N, T, MB, EP, B, S, SN, TN, OW;
  pqn_seeds_t sd;
};

static int upd_ctx(const pqn_update_args_t *a, int S, const uint64_t *key_roll_dev, const uint64_t *key_shuf_dev,
                   long long theta_stride, long long ws_stride, UpdCtx &c) {
  PQN_REQUIRE(a, "pqn_cnn_update: args is NULL");
  PQN_REQUIRE(a->clock && a->sched_keys && a->sched_eps && a->state && a->bits && a->action && a->reward && a->done &&
                  a->qmax && a->discount && a->rer && a->rel && a->ts && a->target && a->last_q && a->sort_keys_in &&
                  a->sort_keys_out && a->sort_temp && a->theta && a->w1b && a->grad && a->m && a->v && a->count &&
                  a->workspace && a->loss_buf && a->qv_buf && a->metrics,
              "pqn_cnn_update: NULL buffer in args");
  c.N = a->num_envs; c.T = a->num_steps; c.MB = a->num_minibatches; c.EP = a->num_epochs;
  PQN_REQUIRE(c.N > 0 && c.T > 0 && c.MB > 0 && c.EP > 0 && c.T + c.EP <= 1024, "pqn_cnn_update: bad shape N=%d T=%d MB=%d EP=%d",
              c.N, c.T, c.MB, c.EP);
  PQN_REQUIRE(((int64_t)c.N * c.T) % c.MB == 0, "NUM_MINIBATCHES must divide NUM_STEPS*NUM_ENVS");
  c.B = (int)(((int64_t)c.N * c.T) / c.MB);
  PQN_REQUIRE(c.B % 16 == 0, "pqn_cnn_update: minibatch size %d must be a multiple of 16", c.B);
  PQN_REQUIRE(S >= 1 && S <= 128 && (S == 1 || (key_roll_dev && key_shuf_dev && c.N % 16 == 0 && (int64_t)c.N * c.T <= (1 << 25))),
              "pqn_cnn_update: seed batching needs 1 <= seeds <= 128, device key arrays, NUM_ENVS %% 16 == 0, T*N <= 2^25");
  c.S = S;
  c.OW = a->obs_words;
  c.SN = S * c.N;
  c.TN = c.T * c.N;
  c.sd = pqn_one_seed();
  if (S > 1) {
    c.sd.nseeds = S;
    c.sd.seed_base = 0;
    c.sd.n_env = c.N;
    c.sd.n_env_total = c.SN;
    c.sd.idx_stride = c.TN;
    c.sd.theta_stride = theta_stride;
    c.sd.w1b_stride = 1024 * 128;
    c.sd.ws_stride = ws_stride;
    c.sd.lq_stride = (long long)c.MB * c.EP;
  }
  c.sd.idx_mask = (1ll << pqn_index_bits(c.TN)) - 1;   // low bits of a sorted shuffle key = the transition index
  // (reserved bit 0 selected a one-kernel fold + clip + RAdam with a grid-wide barrier in rounds 1-4: measured slower in round 1
  // -- 5.47 vs 4.92 ms per update, the barrier costs more than the launch it saves -- and its summation order had drifted
  // from qnet_grad_reduce_kernel's by round 5; removed, the bit is ignored.  Round 6's one-launch form -- no barrier object, no
  // fence: tagged slots, pqn_fold.h -- shares the fold's code with the two-launch form and is chosen per launch in cnn_update_impl)
  // reserved bit 1: kernel form of the training launches from the minibatch size alone (pqn_seeds_t.pin_form)
  c.sd.pin_form = (a->reserved & 2) != 0;
  return PQN_OK;
}

static int upd_begin(const pqn_update_args_t *a, const UpdCtx &c, const uint64_t *key_roll_dev, const uint64_t *key_shuf_dev,
                     hipStream_t st) {
  const pqn_cnn_layout_t &L = a->layout;
  hipLaunchKernelGGL(update_sched_kernel, dim3(c.S), dim3(1024), 0, st, a->clock, a->key_roll, a->key_shuf, key_roll_dev,
                     key_shuf_dev, c.T, c.EP, a->eps_start, a->eps_finish, a->eps_decay_steps, a->sched_keys, a->sched_eps,
                     a->workspace, c.sd.ws_stride);
  // SAMPLE PHASE (_step_env scan, :181-220) + bootstrap forward (:227-235): one persistent launch over all S*N envs
  pqn_step_out_t rec = {};
  rec.reward = a->reward;
  rec.done = a->done;
  rec.discount = a->discount;
  rec.returned_episode_returns = a->rer;
  rec.returned_episode_lengths = a->rel;
  rec.timestep = a->ts;
  UPD_CHECK(pqn_qnet_cnn_rollout(a->env_id, L, c.SN, c.T, a->state, a->bits, a->theta, rec, a->action, a->qmax, a->last_q,
                                 a->sched_eps, a->sched_keys, a->rew_scale, 1, st, c.S > 1 ? c.N : 0, c.sd.theta_stride,
                                 c.T + c.EP, c.sd.pin_form));
  // Q(lambda) TARGETS (:237-260): lane per env over the stacked [T][S*N] record
  return pqn_q_lambda(a->reward, a->done, a->qmax, a->last_q, a->gamma, a->lambda, c.T, c.SN, 1, a->target, st);
}

// one shared permutation per epoch (:299-315), consumed as a gather index
static int upd_shuffle(const pqn_update_args_t *a, const UpdCtx &c, int ep, hipStream_t st) {
  size_t tb = (size_t)a->sort_temp_bytes;
  if (c.S == 1) {
    UPD_CHECK(pqn_shuffle_keys_dyn(a->sched_keys + c.T + ep, c.TN, a->sort_keys_in, st));
  } else {   // one global sort: the seed id in the top bits keeps every seed's segment in place
    UPD_CHECK(pqn_shuffle_keys_seeds(a->sched_keys + c.T + ep, c.T + c.EP, c.S, c.TN, a->sort_keys_in, st));
  }
  UPD_CHECK(pqn_sort_keys(a->sort_temp, tb, a->sort_keys_in, a->sort_keys_out, c.S * c.TN, c.S, c.TN, st));
  // position-parallel form: the rows / bit-transposes / actions / targets of all MB minibatches of this epoch in one launch
  if (pqn_qnet_cnn_epoch_applies(a->layout, c.B, c.MB, c.sd))
    return pqn_qnet_cnn_epoch_gather(a->layout, c.B, c.MB, a->sort_keys_out, a->bits, a->action, a->target, a->workspace, c.sd, st);
  return PQN_OK;
}

static int upd_grad(const pqn_update_args_t *a, const UpdCtx &c, int i_mb, bool with_reduce, hipStream_t st, int part = 0,
                    pqn_fold_args_t *defer = nullptr) {
  const int mb = i_mb % c.MB;
  // low bits of a sorted shuffle key = the transition index inside the seed (kernels mask with sd.idx_mask)
  const bool epoch = with_reduce && pqn_qnet_cnn_epoch_applies(a->layout, c.B, c.MB, c.sd);   // gathered by upd_shuffle
  return pqn_qnet_cnn_grad_seeds_dyn(a->layout, c.B, a->sort_keys_out + (size_t)mb * c.B, a->bits, a->action, a->target, a->theta,
                                 a->w1b, a->grad, a->count, a->workspace, a->loss_buf + i_mb, a->qv_buf + i_mb, c.sd, st,
                                 with_reduce, part, epoch ? mb : -1, c.MB, defer);
}

static int upd_apply(const pqn_update_args_t *a, const UpdCtx &c, int i_mb, bool norm_pass, hipStream_t st,
                     const pqn_fold_args_t *fold = nullptr) {
  const pqn_cnn_layout_t &L = a->layout;
  // f16x2 layouts carry two plane sets of the fc1 kernel: fp16 for the position-parallel kernels, bf16 for every other form.  Inside
  // an update whose optimizer steps take the position-parallel form nothing reads the bf16 set until the update is over (the next
  // consumer is the next step's forward; rollouts, evaluation and whoever reads theta run between updates), so only the LAST step of
  // the update writes it: 24 MB less store traffic per 16-seed launch, 25 -> 20 us per optimizer kernel.
  int copy_mode = L.matmul_f16 + L.pos_f16x2;
  if (L.pos_f16x2 && i_mb + 1 < c.MB * c.EP && pqn_qnet_cnn_pos_form_taken(L, c.B, c.sd)) copy_mode = 4;
  if (fold && fold->valid) {
    // the fold was deferred to the optimizer kernel (pqn_fold.h): one launch folds, clips and applies; a->grad is written by the LAST
    // step of the update only (it then holds the last minibatch's gradient, as it does after the two-launch form)
    const int rc = pqn_launch_radam_fold(*fold, a->theta, i_mb + 1 == c.MB * c.EP ? a->grad : nullptr, a->m, a->v, a->count, a->lr_init,
                                         a->lr_end, a->lr_steps, a->max_grad_norm, a->workspace, a->w1b, st, c.S, c.sd.theta_stride,
                                         c.sd.ws_stride, c.sd.w1b_stride, L.matmul_f16 != 0 ? L.off_w1h : 0, copy_mode);
    if (rc != PQN_E_UNSUPPORTED) return rc;
    UPD_CHECK(pqn_cnn_fold_launch(*fold, a->grad, a->count, a->workspace, c.S, c.sd.theta_stride, st));
  }
  return pqn_launch_radam(a->theta, a->grad, a->m, a->v, L.total, a->count, a->lr_init, a->lr_end, a->lr_steps,
                          a->max_grad_norm, a->workspace, nullptr, L.off_w1, a->w1b, norm_pass ? 1 : 0,
                          pqn_cnn_grad_reduce_blocks(L.total), st, c.S, c.sd.theta_stride, c.sd.ws_stride, c.sd.w1b_stride,
                          L.matmul_f16 != 0 ? L.off_w1h : 0, copy_mode);
}

// carry last_obs into the next update; metrics (:329-338); advance the clock
static int upd_end(const pqn_update_args_t *a, const UpdCtx &c, hipStream_t st) {
  const size_t bits_stride = (size_t)c.SN * c.OW;
  if (hipMemcpyAsync(a->bits, a->bits + (size_t)c.T * bits_stride, bits_stride * sizeof(uint32_t), hipMemcpyDeviceToDevice,
                     st) != hipSuccess) {
    pqn_set_error("pqn_cnn_update: hipMemcpyAsync failed");
    return PQN_E_HIP;
  }
  double *partial = reinterpret_cast<double *>(a->workspace);  // first 1024 floats: optimizer scratch, idle here
  hipLaunchKernelGGL(update_means_kernel, dim3(5, MEANS_CHUNKS, c.S), dim3(256), 0, st, c.TN, c.N, c.SN, a->discount, a->rer,
                     a->rel, a->ts, a->done, partial, c.sd.ws_stride / 2);
  hipLaunchKernelGGL(update_tick_kernel, dim3(c.S), dim3(64), 0, st, a->clock, c.T, c.N, a->layout.c, c.MB * c.EP, a->loss_buf,
                     a->qv_buf, partial, a->metrics, a->metrics_capacity, c.sd.ws_stride / 2);
  return pqn_check_launch("pqn_cnn_update");
}

static int cnn_update_impl(const pqn_update_args_t *a, int S, const uint64_t *key_roll_dev, const uint64_t *key_shuf_dev,
                           long long theta_stride, long long ws_stride, hipStream_t st) {
  UpdCtx c;
  UPD_CHECK(upd_ctx(a, S, key_roll_dev, key_shuf_dev, theta_stride, ws_stride, c));
  UPD_CHECK(upd_begin(a, c, key_roll_dev, key_shuf_dev, st));
  // NETWORKS UPDATE (:263-327)
  int i_mb = 0;
  for (int ep = 0; ep < c.EP; ++ep) {
    UPD_CHECK(upd_shuffle(a, c, ep, st));
    for (int mb = 0; mb < c.MB; ++mb, ++i_mb) {
      // fold + clip + RAdam in one launch (pqn_fold.h) for launches of one or two seeds (seeds x blocks <= 400): measured +2-3 % for
      // one seed (128 / 1024 / 4096 envs), +0.5-1.2 % for two, -0.5-1 % at four, -4 % at 8 and 16 seeds x 4096 envs and -8 % at 16 x 512,
      // where blocks wait on their seed's norm in front of blocks that could be streaming (profiles/r06_v8_fold_apply_ab.txt);
      // option fold_apply: 0 never, 1 by that rule, 2 always
      pqn_fold_args_t fold = {};
      const int fold_opt = pqn_opt(PQN_OPT_FOLD_APPLY);
      const bool fold_on = fold_opt == 2 || (fold_opt == 1 && (long long)c.S * pqn_cnn_grad_reduce_blocks(a->layout.total) <= 400);
      UPD_CHECK(upd_grad(a, c, i_mb, true, st, 0, fold_on ? &fold : nullptr));
      UPD_CHECK(upd_apply(a, c, i_mb, false, st, &fold));
    }
  }
  return upd_end(a, c, st);
}

extern "C" int pqn_cnn_update(const pqn_update_args_t *a, void *stream) {
  return cnn_update_impl(a, 1, nullptr, nullptr, 0, 0, (hipStream_t)stream);
}

extern "C" int pqn_cnn_update_phase(const pqn_update_args_t *a, int32_t phase, int32_t index, void *stream) {
  UpdCtx c;
  UPD_CHECK(upd_ctx(a, 1, nullptr, nullptr, 0, 0, c));
  hipStream_t st = (hipStream_t)stream;
  switch (phase) {
    case PQN_PHASE_BEGIN: return upd_begin(a, c, nullptr, nullptr, st);
    case PQN_PHASE_SHUFFLE:
      PQN_REQUIRE(index >= 0 && index < c.EP, "pqn_cnn_update_phase: epoch %d out of range [0,%d)", index, c.EP);
      return upd_shuffle(a, c, index, st);
    case PQN_PHASE_GRAD:
      PQN_REQUIRE(index >= 0 && index < c.MB * c.EP, "pqn_cnn_update_phase: minibatch step %d out of range [0,%d)", index, c.MB * c.EP);
      return upd_grad(a, c, index, true, st);
    case PQN_PHASE_APPLY:
      PQN_REQUIRE(index >= 0 && index < c.MB * c.EP, "pqn_cnn_update_phase: minibatch step %d out of range [0,%d)", index, c.MB * c.EP);
      return upd_apply(a, c, index, true, st);   // the caller may have all-reduced a->grad: recompute its norm
    case PQN_PHASE_END: return upd_end(a, c, st);
    default: pqn_set_error("pqn_cnn_update_phase: unknown phase %d", phase); return PQN_E_INVALID;
  }
}

extern "C" int pqn_cnn_update_seeds(const pqn_update_args_t *a, int32_t num_seeds, const uint64_t *key_roll_dev,
                                    const uint64_t *key_shuf_dev, int64_t theta_stride, int64_t workspace_stride,
                                    void *stream) {
  PQN_REQUIRE(num_seeds == 1 || (theta_stride > 0 && workspace_stride > 0 && theta_stride % 4 == 0 && workspace_stride % 4 == 0),
              "pqn_cnn_update_seeds: strides must be positive multiples of 4 floats");
  return cnn_update_impl(a, num_seeds, key_roll_dev, key_shuf_dev, theta_stride, workspace_stride, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------
// G groups of seeds, software-pipelined over TWO streams (round 4).  One optimizer step of a seed group is a
// compute-bound training kernel (T1: ~75 % of the step, ~1 TB/s of HBM traffic, every CU's LDS and registers) followed by
// an HBM-bound tail (fc1 weight gradient, fold of the partials, clip + RAdam: ~25 % of the step at 4.5-5.4 TB/s with the
// matrix cores nearly idle).  Seeds are independent (jax.vmap axis, pqn_minatar.py:459-461), so the launches of group g+1
// do not depend on group g: the training kernels of all groups run back to back on `stream`, and every group's tail runs
// on `tail_stream` UNDER the next group's training kernel:
//     stream       T1(g0,i)  T1(g1,i)  T1(g0,i+1)  T1(g1,i+1) ...
//     tail_stream            tail(g0,i) tail(g1,i)  tail(g0,i+1) ...
// with the edges T1(g,i) -> tail(g,i) -> T1(g,i+1) as events.  Only enqueues; capturable (the second stream is forked
// from and joined back into `stream`, so a capture of `stream` records one graph with two branches).  Each group's
// buffers are its own pqn_cnn_update_seeds arguments; results per seed are bit-identical to pqn_cnn_update_seeds (same
// kernels, same launch shapes per group, only their order in time changes).
// ---------------------------------------------------------------------------------------------------------
#define PQN_MAX_SEED_GROUPS 8

// dependency markers, one set per DEVICE (an event belongs to the device that was current when it was created; a process
// that drives several GPUs gets a set for each)
struct SeedGroupEvents {
  bool created = false;
  hipEvent_t fork, join, t1[PQN_MAX_SEED_GROUPS], tail[PQN_MAX_SEED_GROUPS];
};
static SeedGroupEvents g_sg_ev_dev[PQN_MAX_DEVICES];

// the enclosing function names itself in `pqn_fn_`.  An error return between a fork and its join leaves the side stream
// un-joined: a caller that is capturing must end and discard the capture (the qnet.py drivers do -- torch.cuda.graph's exit
// ends the capture, the exception drops the graph); include/pqn_hotpath.h says so at both entry points.
#define HIP_OK(call, what)                                  \
  do {                                                      \
    if ((call) != hipSuccess) {                             \
      (void)hipGetLastError();                              \
      pqn_set_error("%s: %s failed", pqn_fn_, what);        \
      return PQN_E_HIP;                                     \
    }                                                       \
  } while (0)

extern "C" int pqn_cnn_update_seed_groups(int32_t num_groups, const pqn_update_args_t *const *args, const int32_t *num_seeds,
                                          const uint64_t *const *key_roll_dev, const uint64_t *const *key_shuf_dev,
                                          const int64_t *theta_stride, const int64_t *workspace_stride, void *stream,
                                          void *tail_stream) {
  PQN_REQUIRE(num_groups >= 1 && num_groups <= PQN_MAX_SEED_GROUPS && args && num_seeds && key_roll_dev && key_shuf_dev &&
                  theta_stride && workspace_stride,
              "pqn_cnn_update_seed_groups: 1 <= num_groups <= %d and non-NULL argument arrays", PQN_MAX_SEED_GROUPS);
  PQN_REQUIRE(tail_stream && tail_stream != stream, "pqn_cnn_update_seed_groups: tail_stream must be a second, non-NULL stream");
  static const char *const pqn_fn_ = "pqn_cnn_update_seed_groups";
  hipStream_t sc = (hipStream_t)stream, sm = (hipStream_t)tail_stream;
  const int G = num_groups;
  int dev = 0;
  HIP_OK(hipGetDevice(&dev), "hipGetDevice");
  PQN_REQUIRE(dev >= 0 && dev < PQN_MAX_DEVICES, "pqn_cnn_update_seed_groups: device ordinal %d out of range", dev);
  SeedGroupEvents &g_sg_ev = g_sg_ev_dev[dev];
  UpdCtx c[PQN_MAX_SEED_GROUPS];
  for (int g = 0; g < G; ++g) {
    PQN_REQUIRE(num_seeds[g] == 1 || (theta_stride[g] > 0 && workspace_stride[g] > 0 && theta_stride[g] % 4 == 0 && workspace_stride[g] % 4 == 0),
                "pqn_cnn_update_seed_groups: strides must be positive multiples of 4 floats");
    UPD_CHECK(upd_ctx(args[g], num_seeds[g], key_roll_dev[g], key_shuf_dev[g], theta_stride[g], workspace_stride[g], c[g]));
    PQN_REQUIRE(c[g].MB == c[0].MB && c[g].EP == c[0].EP, "pqn_cnn_update_seed_groups: every group must have the same NUM_MINIBATCHES / NUM_EPOCHS");
    for (int h = 0; h < g; ++h)
      PQN_REQUIRE(args[h]->workspace != args[g]->workspace && args[h]->theta != args[g]->theta && args[h]->clock != args[g]->clock,
                  "pqn_cnn_update_seed_groups: groups %d and %d share buffers", h, g);
  }
  if (!g_sg_ev.created) {   // events are plain dependency markers: no timing, reusable across steps and captures
    HIP_OK(hipEventCreateWithFlags(&g_sg_ev.fork, hipEventDisableTiming), "hipEventCreate");
    HIP_OK(hipEventCreateWithFlags(&g_sg_ev.join, hipEventDisableTiming), "hipEventCreate");
    for (int g = 0; g < PQN_MAX_SEED_GROUPS; ++g) {
      HIP_OK(hipEventCreateWithFlags(&g_sg_ev.t1[g], hipEventDisableTiming), "hipEventCreate");
      HIP_OK(hipEventCreateWithFlags(&g_sg_ev.tail[g], hipEventDisableTiming), "hipEventCreate");
    }
    g_sg_ev.created = true;
  }
  HIP_OK(hipEventRecord(g_sg_ev.fork, sc), "hipEventRecord");
  HIP_OK(hipStreamWaitEvent(sm, g_sg_ev.fork, 0), "hipStreamWaitEvent");
  // rollouts + targets: compute-bound persistent kernels, one group after the other on the compute stream
  for (int g = 0; g < G; ++g) UPD_CHECK(upd_begin(args[g], c[g], key_roll_dev[g], key_shuf_dev[g], sc));
  int i_mb = 0;
  for (int ep = 0; ep < c[0].EP; ++ep) {
    // the epoch's permutation is read by the training kernels only (all on `sc`, in order): no extra edge needed
    for (int g = 0; g < G; ++g) UPD_CHECK(upd_shuffle(args[g], c[g], ep, sc));
    for (int mb = 0; mb < c[0].MB; ++mb, ++i_mb) {
      for (int g = 0; g < G; ++g) {
        if (i_mb > 0) HIP_OK(hipStreamWaitEvent(sc, g_sg_ev.tail[g], 0), "hipStreamWaitEvent");   // parameters of step i_mb - 1
        UPD_CHECK(upd_grad(args[g], c[g], i_mb, true, sc, 1));
        HIP_OK(hipEventRecord(g_sg_ev.t1[g], sc), "hipEventRecord");
        HIP_OK(hipStreamWaitEvent(sm, g_sg_ev.t1[g], 0), "hipStreamWaitEvent");
        UPD_CHECK(upd_grad(args[g], c[g], i_mb, true, sm, 2));
        UPD_CHECK(upd_apply(args[g], c[g], i_mb, false, sm));
        HIP_OK(hipEventRecord(g_sg_ev.tail[g], sm), "hipEventRecord");
      }
    }
  }
  for (int g = 0; g < G; ++g) {
    HIP_OK(hipStreamWaitEvent(sc, g_sg_ev.tail[g], 0), "hipStreamWaitEvent");
    UPD_CHECK(upd_end(args[g], c[g], sc));
  }
  HIP_OK(hipEventRecord(g_sg_ev.join, sm), "hipEventRecord");
  HIP_OK(hipStreamWaitEvent(sc, g_sg_ev.join, 0), "hipStreamWaitEvent");
  return PQN_OK;
}

// A stream restricted to a subset of the compute units (bit i of cu_mask = CU i may run its workgroups), optionally with
// the highest priority the device offers: lets a caller give the HBM-bound tail stream of pqn_cnn_update_seed_groups its
// own CUs in the EAGER enqueue (a hipGraph replay does not carry stream attributes).  Measurement aid; the default
// path does not need it.
extern "C" int pqn_stream_create_masked(const uint32_t *cu_mask /* host */, int32_t mask_words, int32_t high_priority,
                                        void **stream_out) {
  PQN_REQUIRE(stream_out && (mask_words == 0 || cu_mask) && mask_words >= 0 && mask_words <= 32,
              "pqn_stream_create_masked: bad arguments");
  hipStream_t s = nullptr;
  hipError_t e;
  if (mask_words > 0) e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask_words, cu_mask);
  else {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // hi = numerically lowest = highest priority
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, high_priority ? hi : lo);
  }
  if (e != hipSuccess) {
    pqn_set_error("pqn_stream_create_masked: %s", hipGetErrorString(e));
    return PQN_E_HIP;
  }
  *stream_out = (void *)s;
  return PQN_OK;
}
extern "C" int pqn_stream_destroy(void *stream) {
  if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
    pqn_set_error("pqn_stream_destroy failed");
    return PQN_E_HIP;
  }
  return PQN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// The gymnax-classic twin (pqn_gymnax.py:167-360): f32 observations, MLP Q-network kernels (pqn_mlp.hip).
// Per rollout step two launches (forward + eps-greedy, env.step); everything else as above.
// ---------------------------------------------------------------------------------------------------------
static int mlp_update_impl(const pqn_mlp_update_args_t *a, int S, const uint64_t *key_roll_dev, const uint64_t *key_shuf_dev,
                           long long theta_stride, long long ws_stride, long long wt_stride, hipStream_t st) {
  PQN_REQUIRE(a, "pqn_mlp_update: args is NULL");
  PQN_REQUIRE(a->clock && a->sched_keys && a->sched_eps && a->state && a->obs && a->action && a->reward && a->done &&
                  a->qmax && a->discount && a->rer && a->rel && a->ts && a->target && a->last_q && a->sort_keys_in &&
                  a->sort_keys_out && a->sort_temp && a->theta && a->grad && a->m && a->v && a->count && a->workspace &&
                  a->loss_buf && a->qv_buf && a->metrics,
              "pqn_mlp_update: NULL buffer in args");
  const int N = a->num_envs, T = a->num_steps, MB = a->num_minibatches, EP = a->num_epochs;
  PQN_REQUIRE(N > 0 && T > 0 && MB > 0 && EP > 0 && T + EP <= 1024, "pqn_mlp_update: bad shape N=%d T=%d MB=%d EP=%d", N, T,
              MB, EP);
  PQN_REQUIRE(((int64_t)N * T) % MB == 0, "NUM_MINIBATCHES must divide NUM_STEPS*NUM_ENVS");
  PQN_REQUIRE(S >= 1 && S <= 128 && (S == 1 || (key_roll_dev && key_shuf_dev && N % 16 == 0 && (int64_t)N * T <= (1 << 25))),
              "pqn_mlp_update: seed batching needs 1 <= seeds <= 128, device key arrays, NUM_ENVS %% 16 == 0, T*N <= 2^25");
  const int B = (int)(((int64_t)N * T) / MB);
  const pqn_mlp_layout_t &L = a->layout;
  PQN_REQUIRE(L.layers == 1 || a->wt, "pqn_mlp_update: transposed hidden kernels (wt) required for NUM_LAYERS > 1");
  const int SN = S * N, TN = T * N;
  const size_t ostride = (size_t)SN * L.d;
  pqn_seeds_t sd = pqn_one_seed();
  if (S > 1) {
    sd.nseeds = S;
    sd.n_env = N;
    sd.n_env_total = SN;
    sd.idx_stride = TN;
    sd.theta_stride = theta_stride;
    sd.ws_stride = ws_stride;
    sd.lq_stride = (long long)MB * EP;
  }
  sd.idx_mask = (1ll << pqn_index_bits(TN)) - 1;
  const int nps = S > 1 ? N : 0;   // envs per seed for the seed-aware kernels (0 = single seed)

  hipLaunchKernelGGL(update_sched_kernel, dim3(S), dim3(1024), 0, st, a->clock, a->key_roll, a->key_shuf, key_roll_dev,
                     key_shuf_dev, T, EP, a->eps_start, a->eps_finish, a->eps_decay_steps, a->sched_keys, a->sched_eps,
                     (float *)nullptr, 0ll);
  // SAMPLE PHASE (_step_env scan, pqn_gymnax.py:172-211) over all S*N envs
  for (int t = 0; t < T; ++t) {
    const size_t o = (size_t)t * SN;
    UPD_CHECK(pqn_mlp_forward_dyn(L, SN, a->obs + t * ostride, a->theta, nullptr, a->action + o, a->qmax + o, 0.0f, 0,
                                  a->sched_eps, a->sched_keys + t, st, nps, sd.theta_stride, T + EP));
    pqn_step_out_t out = {};
    out.obs = a->obs + (t + 1) * ostride;
    out.reward = a->reward + o;
    out.done = a->done + o;
    out.discount = a->discount + o;
    out.returned_episode_returns = a->rer + o;
    out.returned_episode_lengths = a->rel + o;
    out.timestep = a->ts + o;
    UPD_CHECK(pqn_env_step_dyn(a->env_id, SN, a->sched_keys + t, a->rew_scale, a->state, a->action + o, out, st, nps, T + EP));
  }
  // bootstrap value of the last observation (:218-226) and Q(lambda) targets (:228-251)
  UPD_CHECK(pqn_mlp_forward_dyn(L, SN, a->obs + T * ostride, a->theta, nullptr, nullptr, a->last_q, 0.0f, 0, nullptr, nullptr,
                                st, nps, sd.theta_stride, 0));
  UPD_CHECK(pqn_q_lambda(a->reward, a->done, a->qmax, a->last_q, a->gamma, a->lambda, T, SN, 1, a->target, st));
  // NETWORKS UPDATE (:254-318)
  int i_mb = 0;
  for (int ep = 0; ep < EP; ++ep) {
    if (S == 1) {
      UPD_CHECK(pqn_shuffle_keys_dyn(a->sched_keys + T + ep, TN, a->sort_keys_in, st));
    } else {
      UPD_CHECK(pqn_shuffle_keys_seeds(a->sched_keys + T + ep, T + EP, S, TN, a->sort_keys_in, st));
    }
    UPD_CHECK(pqn_sort_keys(a->sort_temp, (size_t)a->sort_temp_bytes, a->sort_keys_in, a->sort_keys_out, S * TN, S, TN, st));
    for (int mb = 0; mb < MB; ++mb, ++i_mb) {
      UPD_CHECK(pqn_mlp_grad_seeds(L, B, a->sort_keys_out + (size_t)mb * B, a->obs, a->action, a->target, a->theta, a->wt,
                                   a->grad, a->count, a->workspace, a->loss_buf + i_mb, a->qv_buf + i_mb, sd, wt_stride, st));
      UPD_CHECK(pqn_launch_radam(a->theta, a->grad, a->m, a->v, L.total, a->count, a->lr_init, a->lr_end, a->lr_steps,
                                 a->max_grad_norm, a->workspace, nullptr, 0, nullptr, 0, pqn_radam_blocks(L.total), st, S,
                                 sd.theta_stride, sd.ws_stride, 0));
      UPD_CHECK(pqn_mlp_refresh_transposed_seeds(L, a->theta, a->wt, S, sd.theta_stride, wt_stride, st));
    }
  }
  if (hipMemcpyAsync(a->obs, a->obs + (size_t)T * ostride, ostride * sizeof(float), hipMemcpyDeviceToDevice, st) !=
      hipSuccess) {
    pqn_set_error("pqn_mlp_update: hipMemcpyAsync failed");
    return PQN_E_HIP;
  }
  double *partial = reinterpret_cast<double *>(a->workspace);  // first 1024 floats: optimizer scratch, idle here
  hipLaunchKernelGGL(update_means_kernel, dim3(5, MEANS_CHUNKS, S), dim3(256), 0, st, TN, N, SN, a->discount, a->rer, a->rel,
                     a->ts, a->done, partial, sd.ws_stride / 2);
  hipLaunchKernelGGL(update_tick_kernel, dim3(S), dim3(64), 0, st, a->clock, T, N, 1, MB * EP, a->loss_buf, a->qv_buf,
                     partial, a->metrics, a->metrics_capacity, sd.ws_stride / 2);
  return pqn_check_launch("pqn_mlp_update");
}

extern "C" int pqn_mlp_update(const pqn_mlp_update_args_t *a, void *stream) {
  return mlp_update_impl(a, 1, nullptr, nullptr, 0, 0, 0, (hipStream_t)stream);
}

extern "C" int pqn_mlp_update_seeds(const pqn_mlp_update_args_t *a, int32_t num_seeds, const uint64_t *key_roll_dev,
                                    const uint64_t *key_shuf_dev, int64_t theta_stride, int64_t workspace_stride,
                                    int64_t wt_stride, void *stream) {
  PQN_REQUIRE(num_seeds == 1 || (theta_stride > 0 && workspace_stride > 0 && theta_stride % 4 == 0 && workspace_stride % 4 == 0),
              "pqn_mlp_update_seeds: strides must be positive multiples of 4 floats");
  return mlp_update_impl(a, num_seeds, key_roll_dev, key_shuf_dev, theta_stride, workspace_stride, wt_stride,
                         (hipStream_t)stream);
}

// LOG_ACHIEVEMENTS (pqn_craftax.py:364-369,384-387): block k = achievement k: (100 * unlocked_k * done).sum() / done.sum() over the
// update's [T][N] steps, in fixed order (f64), into row clock[1] of ach_metrics
__global__ __launch_bounds__(256) void update_ach_means_kernel(const int32_t *__restrict__ clock, int count,
                                                               const uint32_t *__restrict__ ach, const uint8_t *__restrict__ done,
                                                               double *__restrict__ ach_metrics, int capacity) {
  __shared__ double s_num[4], s_den[4];
  const int k = blockIdx.x, u = clock[1];
  double num = 0.0, den = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) {
    const double d = done[i] ? 1.0 : 0.0;
    den += d;
    num += d * (double)((ach[i] >> k) & 1u) * 100.0;
  }
  for (int off = 32; off > 0; off >>= 1) { num += __shfl_down(num, off, 64); den += __shfl_down(den, off, 64); }
  if ((threadIdx.x & 63) == 0) { s_num[threadIdx.x >> 6] = num; s_den[threadIdx.x >> 6] = den; }
  __syncthreads();
  if (threadIdx.x == 0 && u < capacity)
    ach_metrics[(size_t)u * 32 + k] = ((s_num[0] + s_num[1]) + (s_num[2] + s_num[3])) / ((s_den[0] + s_den[1]) + (s_den[2] + s_den[3]));
}

// ---------------------------------------------------------------------------------------------------------
// The Craftax script's twin (pqn_craftax.py:176-399): wrapper-batched env (optimistic resets or auto-reset), the wide
// LayerNorm MLP through the tiled GEMM kernels of pqn_bigmlp.hip, Q(lambda) or 1-step loss, done-weighted info means.
// ---------------------------------------------------------------------------------------------------------
__global__ void keys_to_index_kernel(int64_t *__restrict__ keys, int n, long long mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) keys[i] &= mask;
}

// side stream + fork / join events of pqn_bigmlp_update (one set per device, created on first use there; NULL = run
// everything on the caller's stream: option upd_overlap = 0, or the stream / events could not be created)
struct UpdSide {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool tried = false, ok = false;
};
static UpdSide g_upd_side[PQN_MAX_DEVICES];
static UpdSide *upd_side() {
  if (pqn_opt(PQN_OPT_UPD_OVERLAP) <= 0) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PQN_MAX_DEVICES) {
    (void)hipGetLastError();
    return nullptr;
  }
  UpdSide &u = g_upd_side[dev];
  if (!u.tried) {
    u.tried = true;
    u.ok = hipStreamCreateWithFlags(&u.s, hipStreamNonBlocking) == hipSuccess &&
           hipEventCreateWithFlags(&u.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&u.join, hipEventDisableTiming) == hipSuccess;
    if (!u.ok) (void)hipGetLastError();
  }
  return u.ok ? &u : nullptr;
}

extern "C" int pqn_bigmlp_update(const pqn_bigmlp_update_args_t *a, void *stream) {
  static const char *const pqn_fn_ = "pqn_bigmlp_update";
  hipStream_t st = (hipStream_t)stream;
  PQN_REQUIRE(a, "pqn_bigmlp_update: args is NULL");
  PQN_REQUIRE(a->clock && a->sched_keys && a->sched_eps && a->state && a->obs && a->action && a->reward && a->done &&
                  a->qmax && a->discount && a->rer && a->rel && a->ts && a->sort_keys_in && a->sort_keys_out && a->sort_temp &&
                  a->theta && a->wplanes && a->grad && a->m && a->v && a->count && a->workspace && a->radam_scratch && a->slot_scratch &&
                  a->loss_buf && a->qv_buf && a->metrics,
              "pqn_bigmlp_update: NULL buffer in args");
  const int N = a->num_envs, T = a->num_steps, MB = a->num_minibatches, EP = a->num_epochs;
  PQN_REQUIRE(N > 0 && T > 0 && MB > 0 && EP > 0 && T + EP <= 1024, "pqn_bigmlp_update: bad shape N=%d T=%d MB=%d EP=%d", N, T,
              MB, EP);
  PQN_REQUIRE(((int64_t)N * T) % MB == 0, "NUM_MINIBATCHES must divide NUM_STEPS*NUM_ENVS");
  PQN_REQUIRE(a->reset_ratio == 0 || (a->reset_ratio > 0 && N % a->reset_ratio == 0 && a->opt_scratch),
              "pqn_bigmlp_update: reset ratio %d must perfectly divide num envs %d (and opt_scratch be given)", a->reset_ratio, N);
  PQN_REQUIRE(!a->q_lambda || (a->target && a->last_q), "pqn_bigmlp_update: the Q(lambda) branch needs target and last_q");
  PQN_REQUIRE((a->achievements == nullptr) == (a->ach_metrics == nullptr), "pqn_bigmlp_update: achievements and ach_metrics go together");
  const pqn_bigmlp_layout_t &L = a->layout;
  PQN_REQUIRE(L.norm_input == 0 || (a->in_mean && a->in_var && (L.norm_input == 1 || a->in_steps)),
              "pqn_bigmlp_update: the input normalisation needs its running statistics");
  const int B = (int)(((int64_t)N * T) / MB), TN = T * N;
  const size_t ostride = (size_t)N * L.d;

  hipLaunchKernelGGL(update_sched_kernel, dim3(1), dim3(1024), 0, st, a->clock, a->key_roll, a->key_shuf, (const uint64_t *)nullptr,
                     (const uint64_t *)nullptr, T, EP, a->eps_start, a->eps_finish, a->eps_decay_steps, a->sched_keys, a->sched_eps,
                     (float *)nullptr, 0ll);
  // Option upd_overlap = 1 (default 0): the first epoch's permutation (shuffle keys -> rocPRIM sort -> index mask: three
  // launch-bound kernels, 20 us at C5's 1024 transitions) depends on the key schedule only and goes to a side stream beside
  // the rollout, joining before the first minibatch; the input-gradient plane copy of the last optimizer step runs beside
  // the closing bookkeeping kernels.  Event record / wait pairs are capturable (the side stream joins a capture at its
  // first wait).  Bit 0 of the option = the permutation, bit 1 = the plane copy.  Measured on C5 inside one call
  // (profiles/r04_v7_c5_batched_backward.txt), ms per update: 0.781 without, 0.809 permutation only, 0.819 plane copy only,
  // 0.816 both -- ONE fork / join pair in the replayed graph costs ~30 us, more than the 20 us of kernels it hides (the
  // branches of a hipGraph run on separate queues and meet through barrier packets).  Same results either way.
  const long long mask = (1ll << pqn_index_bits(TN)) - 1;
  auto permutation = [&](int ep, hipStream_t s) -> int {
    UPD_CHECK(pqn_shuffle_keys_dyn(a->sched_keys + T + ep, TN, a->sort_keys_in, s));
    UPD_CHECK(pqn_sort_keys(a->sort_temp, (size_t)a->sort_temp_bytes, a->sort_keys_in, a->sort_keys_out, TN, 1, TN, s));
    hipLaunchKernelGGL(keys_to_index_kernel, dim3((TN + 255) / 256), dim3(256), 0, s, a->sort_keys_out, TN, mask);
    return PQN_OK;
  };
  UpdSide *side = upd_side();
  const int side_mask = pqn_opt(PQN_OPT_UPD_OVERLAP);   // bit 0: the permutation, bit 1: the plane copy
  bool side_perm = false;
  if (side && (side_mask & 1)) {
    if (hipEventRecord(side->fork, st) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess) {
      (void)hipGetLastError();
      side = nullptr;
    } else {
      UPD_CHECK(permutation(0, side->s));
      HIP_OK(hipEventRecord(side->join, side->s), "hipEventRecord");
      side_perm = true;
    }
  }
  // SAMPLE PHASE (_step_env scan, pqn_craftax.py:181-224)
  for (int t = 0; t < T; ++t) {
    const size_t o = (size_t)t * N;
    UPD_CHECK(pqn_bigmlp_forward(&L, N, a->obs + t * ostride, a->theta, a->wplanes, a->in_mean, a->in_var, a->workspace, nullptr,
                                 a->action + o, a->qmax + o, 0.0f, 0, a->sched_eps, a->sched_keys + t, stream));
    pqn_step_out_t out = {};
    out.obs = a->obs + (t + 1) * ostride;
    out.reward = a->reward + o;
    out.done = a->done + o;
    out.discount = a->discount + o;
    out.returned_episode_returns = a->rer + o;
    out.returned_episode_lengths = a->rel + o;
    out.timestep = a->ts + o;
    if (a->achievements) out.achievements = a->achievements + o;
    if (a->reset_ratio > 0) {
      UPD_CHECK(pqn_env_step_optimistic_dyn(a->env_id, N, a->sched_keys + t, a->rew_scale, a->reset_ratio, a->state, a->action + o,
                                            out, a->opt_scratch, a->slot_scratch, st));
    } else {
      UPD_CHECK(pqn_env_step_dyn(a->env_id, N, a->sched_keys + t, a->rew_scale, a->state, a->action + o, out, st, 0, 0,
                                 a->slot_scratch));
    }
  }
  if (a->q_lambda) {   // bootstrap value + Q(lambda) targets (:226-261); dead code of the reference's graph with Q_LAMBDA: False
    UPD_CHECK(pqn_bigmlp_forward(&L, N, a->obs + T * ostride, a->theta, a->wplanes, a->in_mean, a->in_var, a->workspace, nullptr,
                                 nullptr, a->last_q, 0.0f, 0, nullptr, nullptr, stream));
    UPD_CHECK(pqn_q_lambda(a->reward, a->done, a->qmax, a->last_q, a->gamma, a->lambda, T, N, 1, a->target, st));
  }
  // NETWORKS UPDATE (:263-357)
  int i_mb = 0;
  bool side_open = false;
  for (int ep = 0; ep < EP; ++ep) {
    if (ep == 0 && side_perm) HIP_OK(hipStreamWaitEvent(st, side->join, 0), "hipStreamWaitEvent");
    else UPD_CHECK(permutation(ep, st));
    for (int mb = 0; mb < MB; ++mb, ++i_mb) {
      UPD_CHECK(pqn_bigmlp_grad(&L, B, a->sort_keys_out + (size_t)mb * B, a->obs, a->q_lambda ? 0 : N, a->action,
                                a->q_lambda ? a->target : nullptr, a->reward, a->done, a->gamma, a->theta, a->wplanes, a->in_mean,
                                a->in_var, a->in_steps, a->grad, a->workspace, a->loss_buf + i_mb, a->qv_buf + i_mb, stream));
      UPD_CHECK(pqn_launch_radam(a->theta, a->grad, a->m, a->v, L.total, a->count, a->lr_init, a->lr_end, a->lr_steps,
                                 a->max_grad_norm, a->radam_scratch, nullptr, 0, nullptr, 1, 0, st));
      if (side && (side_mask & 2) && i_mb + 1 == MB * EP) {   // last optimizer step: the input-gradient copy beside the closing bookkeeping below
        HIP_OK(hipEventRecord(side->fork, st), "hipEventRecord");
        HIP_OK(hipStreamWaitEvent(side->s, side->fork, 0), "hipStreamWaitEvent");
        UPD_CHECK(pqn_bigmlp_refresh_planes_streams(&L, a->theta, a->wplanes, stream, side->s));
        HIP_OK(hipEventRecord(side->join, side->s), "hipEventRecord");
        side_open = true;
      } else {
        UPD_CHECK(pqn_bigmlp_refresh_planes(&L, a->theta, a->wplanes, stream));
      }
    }
  }
  if (hipMemcpyAsync(a->obs, a->obs + (size_t)T * ostride, ostride * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess) {
    pqn_set_error("pqn_bigmlp_update: hipMemcpyAsync failed");
    return PQN_E_HIP;
  }
  double *partial = reinterpret_cast<double *>(a->workspace);   // idle between the last optimizer step and the next forward
  hipLaunchKernelGGL(update_means_kernel, dim3(5, MEANS_CHUNKS, 1), dim3(256), 0, st, TN, N, N, a->discount, a->rer, a->rel, a->ts,
                     a->done, partial, 0ll, a->done_weighted_info);
  if (a->achievements)   // before the tick: it reads the update index the sched kernel latched into clock[1]
    hipLaunchKernelGGL(update_ach_means_kernel, dim3(32), dim3(256), 0, st, a->clock, TN, a->achievements, a->done, a->ach_metrics,
                       a->metrics_capacity);
  hipLaunchKernelGGL(update_tick_kernel, dim3(1), dim3(64), 0, st, a->clock, T, N, 1, MB * EP, a->loss_buf, a->qv_buf, partial,
                     a->metrics, a->metrics_capacity, 0ll, a->done_weighted_info);
  if (side_open) HIP_OK(hipStreamWaitEvent(st, side->join, 0), "hipStreamWaitEvent");
  return pqn_check_launch("pqn_bigmlp_update");
}
