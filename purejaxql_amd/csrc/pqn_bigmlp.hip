// pqn_bigmlp.hip -- the wide MLP Q-network of the Craftax script as gfx950 kernels.
//
// Network = QNetwork of the reference's purejaxql/pqn_craftax.py:33-62 with NORM_TYPE = layer_norm (the value of
// config/alg/pqn_craftax.yaml:12): [BatchRenorm | BatchNorm | nothing](x) -> NUM_LAYERS x (Dense(H) -> LayerNorm -> relu)
// -> Dense(A); C5 = 1345 -> 4 x 1024 -> 17 on 2 x 1024 rows per update (obs and next_obs as ONE batch, :287-304).
// These are GEMM-sized layers (2048 x 1345 x 1024, 2048 x 1024 x 1024), unlike the 16-sample tiles of pqn_mlp.hip whose
// activations live in LDS: here every Dense layer -- forward, input gradient, weight gradient -- is ONE tiled MFMA GEMM
// kernel, and everything elementwise sits in a handful of row / column kernels around it.
//
//   bm_gemm_kernel   C[M,N] = op(A)[M,K] x op(B)[K,N] on the bf16 matrix core with f32-grade "bf16x3" products (see
//                    pqn_qnet.hip: x = hi + mid + lo exactly, 6 x v_mfma_f32_16x16x32_bf16 per 32-wide K step, f32
//                    accumulate).  Operands come straight from f32 global memory: a loader thread fetches 8 K-values of
//                    one tile row, splits them into the three bf16 planes and writes one 16-B MFMA fragment slot per
//                    plane into LDS (double-buffered, one barrier per K step; the next step's global loads are in flight
//                    during the MFMAs).  Either operand may be read transposed (source rows = K), so forward (H W),
//                    input gradient (dZ W^T) and weight gradient (H^T dZ) are the same kernel.  Tile BM x 64
//                    (BM = 64 | 128), 4 waves as 2 x 2; epilogue = bias + store, or the column sums that are the
//                    input-normalisation parameter gradients.
//   bm_innorm_apply  gathers the minibatch rows out of the rollout record and applies the input normalisation
//   bm_colstats*     batch moments of the gathered input rows + BatchRenorm / BatchNorm bookkeeping
//                    (utils/batch_renorm.py:95-116) -> the coefficient vectors bm_innorm_apply uses
//   bm_ln_relu       LayerNorm (flax: var = E[x^2] - E[x]^2 clamped, eps 1e-6) + relu of a pre-activation matrix
//   bm_loss          TD loss of both branches of _loss_fn (pqn_craftax.py:277-312), dQ, d b_out, metrics
//   bm_ln_bwd        relu mask + LayerNorm backward in place, column partial sums for d scale / d bias / d dense-bias
//   bm_colreduce     fixed-order fold of per-workgroup column partials
// Everything is deterministic (fixed summation orders, no atomics).
#include <stdlib.h>

#include "pqn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

#define BM_THREADS 256
#define BM_BN 64
#define BM_KS 32
#define BM_LN_EPS 1e-6f

// exact 3-way bf16 split of two f32 values (same arithmetic as x3_split2 in pqn_qnet.hip)
PQN_D void bm_split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  const f32x2 x = {x0, x1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
  const f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xFFFF0000u)};
  const f32x2 r1 = x - hf;
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
  const f32x2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xFFFF0000u)};
  const f32x2 r2 = r1 - mf;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
}
// tied-accumulator MFMA as volatile inline asm (see x3_mfma_tied in pqn_qnet.hip for why not the builtin)
PQN_D f32x4 bm_mfma(const u32x4 &a, const u32x4 &b, f32x4 c) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
PQN_D void bm_drain(f32x4 &a, f32x4 &b) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b)); }

enum { BM_EPI_STORE = 0, BM_EPI_INNORM = 1 };

// One GEMM operand: a row-major f32 source matrix S[rows][cols] with leading dimension ld.
//   not TRANS: tile row t <-> S row, K index <-> S column;   TRANS: tile row t <-> S column, K index <-> S row.
struct BmOperand {
  const float *p;
  long long ld;
  int rows, cols;            // valid extent of S
};

struct BmEpilogue {
  float *out;                // BM_EPI_STORE: C, row-major, leading dimension ldc;  BM_EPI_INNORM: partials [K split][m tile][2][N]
  long long ldc;
  const float *bias;         // BM_EPI_STORE: optional per-column bias (only without split-K)
  long long split_stride;    // BM_EPI_STORE: elements between the partial outputs of consecutive K splits
  // BM_EPI_INNORM (input-normalisation parameter gradients): d scale_c = sum_r C[r][c] xhat[r][c], d bias_c = sum_r C[r][c]
  const float *xhat;         // [M][ldx]: (x - m_c) k_c of the gradient rows
  long long ldx;
};

// split a pack into three planes and write the 16-B fragment slots: slot (block = t >> 4, lane = kb * 16 + (t & 15))
PQN_D void bm_store_pack(u32x4 *planes, int nblk, int t_local, int kb, const float (&v)[8]) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bm_split2(v[2 * q], v[2 * q + 1], h[q], m[q], l[q]);
  const int slot = (t_local >> 4) * 64 + kb * 16 + (t_local & 15);
  planes[slot] = u32x4{h[0], h[1], h[2], h[3]};
  planes[nblk * 64 + slot] = u32x4{m[0], m[1], m[2], m[3]};
  planes[2 * nblk * 64 + slot] = u32x4{l[0], l[1], l[2], l[3]};
}

// Operand loaders.  A "pack" = the 8 K-values k .. k + 7 of one tile row = one lane's share of an MFMA fragment.  Values
// outside the matrix read as zero; addresses are clamped into the matrix so that every load stays unconditional
// (compiler finding 1) and validity is applied to the values.
//   BM_LM_ROWVEC  untransposed source, rows 16-B aligned: two aligned quads per pack
//   BM_LM_ROWSCL  untransposed source, any alignment: eight dword loads per pack
//   BM_LM_COLVEC  transposed source (K runs down the source rows), rows 16-B aligned: a unit = 8 K-rows x 4 adjacent tile
//                 rows = eight quad loads that yield FOUR packs; a BT-row tile has BT units, served by BT threads
//   BM_LM_COLSCL  transposed source, any alignment: eight dword loads per pack (adjacent lanes = adjacent tile rows)
enum { BM_LM_ROWVEC = 0, BM_LM_ROWSCL = 1, BM_LM_COLVEC = 2, BM_LM_COLSCL = 3 };

template <int LM, int BT, int TBASE>
struct BmLoader {
  static constexpr bool COL = LM >= BM_LM_COLVEC;
  static constexpr int NP = LM == BM_LM_COLVEC ? 4 : BT * 4 / BM_THREADS;   // packs per thread and K step
  float r[NP][8];
  int t[NP], kb[NP];
  bool active;
  PQN_D void init(int tid) {
    if (LM == BM_LM_COLVEC) {
      const int u = tid - TBASE;            // TBASE and BT are multiples of 64: whole waves are active or not
      active = u >= 0 && u < BT;
      const int uu = active ? u : 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) { t[i] = 4 * (uu % (BT / 4)) + i; kb[i] = uu / (BT / 4); }
    } else {
      active = true;
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int p = tid + BM_THREADS * q;
        t[q] = COL ? p % BT : p >> 2;
        kb[q] = COL ? p / BT : p & 3;
      }
    }
  }
  PQN_D void load(const BmOperand &o, int t0, int k0) {
    if (LM == BM_LM_COLVEC) {
      if (!active) return;
      const int k = k0 + 8 * kb[0], c = t0 + t[0];
      const int cc = min(c, (int)o.ld - 4);          // quads are aligned; a clamped quad lies wholly beyond the valid columns
      f32x4 u[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) u[j] = *reinterpret_cast<const f32x4 *>(o.p + (long long)min(k + j, o.rows - 1) * o.ld + cc);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool kok = k + j < o.rows;
        r[0][j] = (kok && c < o.cols) ? u[j].x : 0.0f;
        r[1][j] = (kok && c + 1 < o.cols) ? u[j].y : 0.0f;
        r[2][j] = (kok && c + 2 < o.cols) ? u[j].z : 0.0f;
        r[3][j] = (kok && c + 3 < o.cols) ? u[j].w : 0.0f;
      }
    } else {
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const int tt = t0 + t[q], k = k0 + 8 * kb[q];
        float (&v)[8] = r[q];
        if (!COL) {
          const float *row = o.p + (long long)min(tt, o.rows - 1) * o.ld;
          if (LM == BM_LM_ROWVEC) {   // a clamped quad lies wholly beyond the valid columns (k % 8 == 0, ld % 4 == 0)
            const int lim4 = (int)o.ld - 4;
            const f32x4 a = *reinterpret_cast<const f32x4 *>(row + min(k, lim4)), b = *reinterpret_cast<const f32x4 *>(row + min(k + 4, lim4));
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = row[min(k + j, o.cols - 1)];
          }
          const bool tok = tt < o.rows;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (tok && k + j < o.cols) ? v[j] : 0.0f;
        } else {
          const int c = min(tt, o.cols - 1);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = o.p[(long long)min(k + j, o.rows - 1) * o.ld + c];
          const bool tok = tt < o.cols;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = (tok && k + j < o.rows) ? v[j] : 0.0f;
        }
      }
    }
  }
  PQN_D void store(u32x4 *planes) const {
    if (LM == BM_LM_COLVEC && !active) return;
#pragma unroll
    for (int q = 0; q < NP; ++q) bm_store_pack(planes, BT / 16, t[q], kb[q], r[q]);
  }
};

template <int BM>
constexpr int bm_lds_bytes() { return 2 * 3 * (BM / 16 + BM_BN / 16) * 64 * 16; }

template <int BM, int LMA, int LMB, int EPI>
__global__ __launch_bounds__(BM_THREADS) void bm_gemm_kernel(int M, int N, int K, int klen, BmOperand A, BmOperand B, BmEpilogue E) {
  constexpr int NBA = BM / 16, NBB = BM_BN / 16;
  constexpr int MI = BM / 32;                                           // 16-row blocks per wave (2 x 2 waves)
  extern __shared__ __attribute__((aligned(16))) char bm_smem[];
  u32x4 *sA = reinterpret_cast<u32x4 *>(bm_smem);                         // [2][3][NBA][64]
  u32x4 *sB = sA + 2 * 3 * NBA * 64;                                      // [2][3][NBB][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BM_BN;
  // split-K: workgroup z owns K indices [z klen, (z + 1) klen) (klen % 32 == 0) and writes its own partial output
  const int kbeg = blockIdx.z * klen, kend = min(K, kbeg + klen);
  BmLoader<LMA, BM, 0> la;                         // transposed-vector units of A: threads 0 .. BM - 1
  BmLoader<LMB, BM_BN, BM_THREADS - BM_BN> lb;     // ... of B: the last wave
  la.init(tid);
  lb.init(tid);
  auto gload = [&](int k0) { la.load(A, m0, k0); lb.load(B, n0, k0); };
  auto lstore = [&](int buf) { la.store(sA + buf * 3 * NBA * 64); lb.store(sB + buf * 3 * NBB * 64); };
  f32x4 accb[MI][2], accs[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) { accb[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f}; accs[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const int nk = (kend - kbeg + BM_KS - 1) / BM_KS;
  gload(kbeg);
#pragma unroll 1
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    lstore(buf);
    __syncthreads();
    if (ks + 1 < nk) gload(kbeg + (ks + 1) * BM_KS);   // in flight during the MFMAs below
    const u32x4 *pa = sA + buf * 3 * NBA * 64 + (wm * MI) * 64 + lane;
    const u32x4 *pb = sB + buf * 3 * NBB * 64 + (wn * 2) * 64 + lane;
    u32x4 a[MI][3], b[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi][pl] = pa[(pl * NBA + mi) * 64];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[ni][pl] = pb[(pl * NBB + ni) * 64];
    }
    // products (l,h) (m,h) (h,l) (h,m) (m,m) (h,h): the three small terms meet in accs, the three leading ones in accb
#define BM_PROD(PA_, PB_, ACC)                                                     \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) ACC[mi][ni] = bm_mfma(a[mi][PA_], b[ni][PB_], ACC[mi][ni]);
    BM_PROD(2, 0, accs)
    BM_PROD(1, 0, accb)
    BM_PROD(0, 2, accs)
    BM_PROD(0, 1, accb)
    BM_PROD(1, 1, accs)
    BM_PROD(0, 0, accb)
#undef BM_PROD
  }
  // ---- epilogue ----
  const int col_l = lane & 15, rq = lane >> 4;
  if (EPI == BM_EPI_STORE) {
    float *outp = E.out + (long long)blockIdx.z * E.split_stride;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        bm_drain(accb[mi][ni], accs[mi][ni]);
        const f32x4 t = accb[mi][ni] + accs[mi][ni];
        const float tv[4] = {t.x, t.y, t.z, t.w};
        const int col = n0 + wn * 32 + ni * 16 + col_l;
        const float bias = E.bias ? E.bias[min(col, N - 1)] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + (wm * MI + mi) * 16 + 4 * rq + r;
          if (row < M && col < N) outp[(long long)row * E.ldc + col] = tv[r] + bias;
        }
      }
  } else {
    // input-normalisation parameter gradients: column sums over this tile's rows, folded lane -> row groups -> the two
    // wave rows through LDS in a fixed order; one partial record per m tile
    __syncthreads();   // the operand buffers are dead: reuse them
    float *red = reinterpret_cast<float *>(bm_smem);   // [2 wm][64 cols][2]
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = n0 + wn * 32 + ni * 16 + col_l;
      const int cc = min(col, N - 1);
      float ss = 0.f, sb = 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        bm_drain(accb[mi][ni], accs[mi][ni]);
        const f32x4 t = accb[mi][ni] + accs[mi][ni];
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + (wm * MI + mi) * 16 + 4 * rq + r;
          const float xh = E.xhat[(long long)min(row, M - 1) * E.ldx + cc];
          const float g = (row < M && col < N) ? tv[r] : 0.0f;
          ss += g * xh;
          sb += g;
        }
      }
      ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
      sb += __shfl_xor(sb, 16, 64); sb += __shfl_xor(sb, 32, 64);
      if (rq == 0) {
        red[(wm * 64 + wn * 32 + ni * 16 + col_l) * 2] = ss;
        red[(wm * 64 + wn * 32 + ni * 16 + col_l) * 2 + 1] = sb;
      }
    }
    __syncthreads();
    if (tid < 64 && n0 + tid < N) {
      float *po = E.out + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * 2 * N;
      po[n0 + tid] = red[tid * 2] + red[(64 + tid) * 2];
      po[N + n0 + tid] = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
    }
  }
}

// all-reduce over the 64 lanes of a wave, fixed butterfly order
PQN_D float bm_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// z = sum of the K-split partials of the layer's GEMM + bias;  h = relu(LayerNorm(z)) row by row (one wave per row);
// z and stat[r] = (mean, rstd) are kept for the backward pass.  n % 256 == 0, n <= 4096.
#define BM_MAXQ 4
__global__ __launch_bounds__(256) void bm_ln_relu_kernel(const float *__restrict__ zpart, int nsplit, long long pstride, int m,
                                                         int n, const float *__restrict__ bias, const float *__restrict__ g,
                                                         const float *__restrict__ beta, float *__restrict__ z,
                                                         float *__restrict__ h, float *__restrict__ stat) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const int nq = n / 256;
  f32x4 v[BM_MAXQ];
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int k = 0; k < BM_MAXQ; ++k)
    if (k < nq) {
      const int c = lane * 4 + 256 * k;
      v[k] = *reinterpret_cast<const f32x4 *>(zpart + (long long)row * n + c);
      for (int sp = 1; sp < nsplit; ++sp) v[k] += *reinterpret_cast<const f32x4 *>(zpart + sp * pstride + (long long)row * n + c);
      v[k] += *reinterpret_cast<const f32x4 *>(bias + c);
      *reinterpret_cast<f32x4 *>(z + (long long)row * n + c) = v[k];
      s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
      q += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
    }
  s = bm_wave_sum(s);
  q = bm_wave_sum(q);
  const float mean = s / (float)n;
  const float var = fmaxf(q / (float)n - mean * mean, 0.0f);
  const float rstd = 1.0f / sqrtf(var + BM_LN_EPS);
  if (lane == 0) { stat[2 * row] = mean; stat[2 * row + 1] = rstd; }
  float *hr = h + (long long)row * n;
#pragma unroll
  for (int k = 0; k < BM_MAXQ; ++k)
    if (k < nq) {
      const int c = lane * 4 + 256 * k;
      const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + c), bv = *reinterpret_cast<const f32x4 *>(beta + c);
      const f32x4 xh = (v[k] - mean) * rstd;
      const f32x4 y = xh * gv + bv;
      *reinterpret_cast<f32x4 *>(hr + c) = f32x4{fmaxf(y.x, 0.f), fmaxf(y.y, 0.f), fmaxf(y.z, 0.f), fmaxf(y.w, 0.f)};
    }
}

// source row of minibatch row r: idx == NULL: r itself;  else idx[r mod nb] (+ next_off for the next_obs half r >= nb)
PQN_D long long bm_src_row(const int64_t *idx, int nb, long long next_off, int r) {
  if (!idx) return r;
  return idx[r >= nb ? r - nb : r] + (r >= nb ? next_off : 0);
}

// batch moments of the gathered input rows, stage 1: partial (sum, sum of squares) per column over a chunk of rows.
// Accumulated in f64: flax's fast variance E[x^2] - E[x]^2 cancels catastrophically for columns whose spread is small
// against their mean, and an f32 running sum would put its own rounding (~1e-5 relative) straight into that difference.
#define BM_CS_ROWS 64
__global__ __launch_bounds__(256) void bm_colstats_kernel(const float *__restrict__ x, long long ldx, const int64_t *idx, int nb,
                                                          long long next_off, int m, int d, double *__restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * BM_CS_ROWS, r1 = min(r0 + BM_CS_ROWS, m);
  const int cc = min(c, d - 1);
  double s = 0.0, q = 0.0;
  for (int r = r0; r < r1; ++r) {
    const double v = (double)x[bm_src_row(idx, nb, next_off, r) * ldx + cc];
    s += v;
    q += v * v;
  }
  if (c < d) {
    part[((long long)blockIdx.y * 2) * d + c] = s;
    part[((long long)blockIdx.y * 2 + 1) * d + c] = q;
  }
}

// stage 2 + the bookkeeping of utils/batch_renorm.py:95-116 (renorm = 1) / flax nn.BatchNorm (renorm = 0), thread per
// column.  coef = [4][d]: m_c (mean used), a_c = k_c * scale_c, b_c = bias_c, k_c = 1 / sqrt(var used + eps).
// train = 0: coefficients from the running moments only (use_running_average).
__global__ __launch_bounds__(256) void bm_instat_finish_kernel(const double *__restrict__ part, int nparts, int m, int d,
                                                               const float *__restrict__ scale, const float *__restrict__ bias,
                                                               float *__restrict__ ra_mean, float *__restrict__ ra_var,
                                                               const int32_t *__restrict__ steps, int train, int renorm, float eps,
                                                               float momentum, float *__restrict__ coef) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= d) return;
  float mean = ra_mean[c], var = ra_var[c];
  if (train) {
    double s = 0.0, q = 0.0;
    for (int p = 0; p < nparts; ++p) {
      s += part[((long long)p * 2) * d + c];
      q += part[((long long)p * 2 + 1) * d + c];
    }
    const float bmean = (float)(s / (double)m);
    const float bvar = fmaxf((float)(q / (double)m) - bmean * bmean, 0.0f);
    mean = bmean;
    var = bvar;
    if (renorm && *steps >= 1000) {
      const float ra_std = sqrtf(ra_var[c] + eps);
      const float r = fminf(fmaxf(sqrtf(bvar + eps) / ra_std, 1.0f / 3.0f), 3.0f);
      const float dd = fminf(fmaxf((bmean - ra_mean[c]) / ra_std, -5.0f), 5.0f);
      var = bvar / (r * r);
      mean = bmean - dd * sqrtf(bvar) / r;
    }
    ra_mean[c] = momentum * ra_mean[c] + (1.0f - momentum) * bmean;
    ra_var[c] = momentum * ra_var[c] + (1.0f - momentum) * bvar;
  }
  const float k = 1.0f / sqrtf(var + eps);
  coef[c] = mean;
  coef[d + c] = k * scale[c];
  coef[2 * d + c] = bias[c];
  coef[3 * d + c] = k;
}
__global__ void bm_steps_inc_kernel(int32_t *steps) { *steps += 1; }

// xn[r][c] = (x[src(r)][c] - m_c) a_c + b_c (coef != NULL) or the gathered x itself; xhat[r][c] = (x - m_c) k_c for the
// gradient rows r < nb (xhat != NULL).  Columns [d, ldo) are zeroed.  One thread per (row, 4 columns).
__global__ __launch_bounds__(256) void bm_innorm_apply_kernel(const float *__restrict__ x, long long ldx, const int64_t *idx,
                                                              int nb, long long next_off, int m, int d, int ldo,
                                                              const float *__restrict__ coef, float *__restrict__ xn,
                                                              float *__restrict__ xhat) {
  const int qpr = ldo / 4;   // quads per row
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)m * qpr) return;
  const int r = (int)(e / qpr), c0 = (int)(e % qpr) * 4;
  const float *xr = x + bm_src_row(idx, nb, next_off, r) * ldx;
  float o[4], oh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + j, cc = min(c, d - 1);
    const float v = xr[cc];
    float y = v, yh = 0.0f;
    if (coef) {
      const float xc = v - coef[cc];
      y = xc * coef[d + cc] + coef[2 * d + cc];
      yh = xc * coef[3 * d + cc];
    }
    o[j] = c < d ? y : 0.0f;
    oh[j] = c < d ? yh : 0.0f;
  }
  *reinterpret_cast<f32x4 *>(xn + (long long)r * ldo + c0) = f32x4{o[0], o[1], o[2], o[3]};
  if (xhat && r < nb) *reinterpret_cast<f32x4 *>(xhat + (long long)r * ldo + c0) = f32x4{oh[0], oh[1], oh[2], oh[3]};
}

// TD loss (single workgroup: fixed-order sums).  Rows r < b of Q are the q-values of transition idx[r]; with
// next_rows the rows b + r hold Q(next_obs) and target = reward + (1 - done) gamma max_a Q_next (pqn_craftax.py:300-306),
// otherwise `target` is the given Q(lambda) target.  dQ[r][a] = [a == action] (q_a - target) / b; d b_out = column sums.
__global__ __launch_bounds__(1024) void bm_loss_kernel(const float *__restrict__ q, int ldq, int b, int a,
                                                       const int64_t *__restrict__ idx, const int32_t *__restrict__ action,
                                                       const float *__restrict__ target, const float *__restrict__ reward,
                                                       const uint8_t *__restrict__ done, float gamma, int next_rows,
                                                       float *__restrict__ dq, float *__restrict__ dbias,
                                                       float *__restrict__ loss_out, float *__restrict__ qv_out) {
  __shared__ float s_b[16][64];    // per-wave partial column sums of dQ (a <= 64)
  __shared__ float s_l[16], s_q[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float lsum = 0.f, qsum = 0.f, bacc = 0.f;   // bacc: lane j of a wave accumulates that wave's sum for action j
  const float inv_b = 1.0f / (float)b;
  for (int r0 = 0; r0 < b; r0 += 1024) {
    const int r = r0 + tid;
    float g = 0.f;
    int act = -1;
    if (r < b) {
      const long long j = idx ? idx[r] : r;
      act = action[j];
      const float qa = q[(long long)r * ldq + act];
      float tgt;
      if (next_rows) {
        const float *qn = q + (long long)(b + r) * ldq;
        float mx = qn[0];
        for (int k = 1; k < a; ++k) mx = fmaxf(mx, qn[k]);
        tgt = reward[j] + (1.0f - (float)done[j]) * gamma * mx;
      } else {
        tgt = target[j];
      }
      const float diff = qa - tgt;
      g = diff * inv_b;
      lsum += 0.5f * diff * diff;
      qsum += qa;
      for (int k = 0; k < ldq; ++k) dq[(long long)r * ldq + k] = (k == act) ? g : 0.0f;
    }
    for (int k = 0; k < a; ++k) {   // fixed-order butterfly per action; every lane gets the wave total
      const float t = bm_wave_sum(act == k ? g : 0.0f);
      bacc += (lane == k) ? t : 0.0f;
    }
  }
  s_b[wave][lane] = bacc;
  lsum = bm_wave_sum(lsum);
  qsum = bm_wave_sum(qsum);
  if (lane == 0) { s_l[wave] = lsum; s_q[wave] = qsum; }
  __syncthreads();
  if (tid < a) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += s_b[w][tid];
    dbias[tid] = t;
  }
  if (tid == 0) {
    float l = 0.f, qq = 0.f;
    for (int w = 0; w < 16; ++w) { l += s_l[w]; qq += s_q[w]; }
    if (loss_out) *loss_out = l * inv_b;
    if (qv_out) *qv_out = qq * inv_b;
  }
}

// relu mask + LayerNorm backward: d (rows x n) <- f(sum of dpart), one wave per row, BM_LB_ROWS rows per workgroup.
//   y = xhat g + beta;  dy = d [y > 0];  dxh = dy g;  dz = rstd (dxh - mean(dxh) - xhat mean(dxh xhat))
// part[wg][0] = sum_rows dy xhat (d scale), [1] = sum_rows dy (d LN bias), [2] = sum_rows dz (d dense bias).  n <= 4096, n % 256 == 0.
#define BM_LB_ROWS 8
__global__ __launch_bounds__(256) void bm_ln_bwd_kernel(const float *__restrict__ dpart, int nsplit, long long pstride,
                                                        float *__restrict__ d, const float *__restrict__ z,
                                                        const float *__restrict__ stat, const float *__restrict__ g,
                                                        const float *__restrict__ beta, int rows, int n,
                                                        float *__restrict__ part) {
  __shared__ float s_red[4][3 * 1024];   // 1024 columns per pass of the cross-wave fold
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = n / 256;                // float4 per lane and row
  f32x4 a0[BM_MAXQ], a1[BM_MAXQ], a2[BM_MAXQ];
#pragma unroll
  for (int q = 0; q < BM_MAXQ; ++q) { a0[q] = f32x4{0.f, 0.f, 0.f, 0.f}; a1[q] = a0[q]; a2[q] = a0[q]; }
  for (int rr = wave; rr < BM_LB_ROWS; rr += 4) {
    const int row = blockIdx.x * BM_LB_ROWS + rr;
    if (row >= rows) break;
    const float mean = stat[2 * row], rstd = stat[2 * row + 1];
    float *dr = d + (long long)row * n;
    const float *zr = z + (long long)row * n;
    f32x4 xh[BM_MAXQ], dxh[BM_MAXQ], dy[BM_MAXQ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q = 0; q < BM_MAXQ; ++q) {
      if (q < nq) {
        const int c = lane * 4 + 256 * q;
        const f32x4 zv = *reinterpret_cast<const f32x4 *>(zr + c);
        f32x4 dv = *reinterpret_cast<const f32x4 *>(dpart + (long long)row * n + c);   // d loss / d h: sum of the K-split partials
        for (int sp = 1; sp < nsplit; ++sp) dv += *reinterpret_cast<const f32x4 *>(dpart + sp * pstride + (long long)row * n + c);
        const f32x4 gv = *reinterpret_cast<const f32x4 *>(g + c), bv = *reinterpret_cast<const f32x4 *>(beta + c);
        xh[q] = (zv - mean) * rstd;
        const f32x4 y = xh[q] * gv + bv;
        dy[q] = f32x4{y.x > 0.f ? dv.x : 0.f, y.y > 0.f ? dv.y : 0.f, y.z > 0.f ? dv.z : 0.f, y.w > 0.f ? dv.w : 0.f};
        dxh[q] = dy[q] * gv;
        s1 += (dxh[q].x + dxh[q].y) + (dxh[q].z + dxh[q].w);
        const f32x4 t = dxh[q] * xh[q];
        s2 += (t.x + t.y) + (t.z + t.w);
      }
    }
    s1 = bm_wave_sum(s1) / (float)n;
    s2 = bm_wave_sum(s2) / (float)n;
#pragma unroll
    for (int q = 0; q < BM_MAXQ; ++q) {
      if (q < nq) {
        const int c = lane * 4 + 256 * q;
        const f32x4 dz = (dxh[q] - s1 - xh[q] * s2) * rstd;
        *reinterpret_cast<f32x4 *>(dr + c) = dz;
        a0[q] += dy[q] * xh[q];
        a1[q] += dy[q];
        a2[q] += dz;
      }
    }
  }
  // fold the four waves (fixed order) and write this workgroup's record
  for (int base = 0; base < n; base += 1024) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < BM_MAXQ; ++q) {
      const int c = lane * 4 + 256 * q;
      if (q < nq && c >= base && c < base + 1024) {
        *reinterpret_cast<f32x4 *>(&s_red[wave][c - base]) = a0[q];
        *reinterpret_cast<f32x4 *>(&s_red[wave][1024 + c - base]) = a1[q];
        *reinterpret_cast<f32x4 *>(&s_red[wave][2048 + c - base]) = a2[q];
      }
    }
    __syncthreads();
    const int lim = min(1024, n - base);
    for (int e = tid; e < 3 * lim; e += 256) {
      const int k = e / lim, c = e - k * lim;
      const float v = (s_red[0][k * 1024 + c] + s_red[1][k * 1024 + c]) + (s_red[2][k * 1024 + c] + s_red[3][k * 1024 + c]);
      part[((long long)blockIdx.x * 3 + k) * n + base + c] = v;
    }
  }
}

// out_k[c] = sum_p part[(p * nseg + k) * n + c], k < nseg.  Workgroup = 64 (k, c) elements x 16 partial groups: group j
// sums partials j, j + 16, ... (fixed order), the 16 group sums are folded in a fixed tree through LDS.
__global__ __launch_bounds__(1024) void bm_colreduce_kernel(const float *__restrict__ part, int nparts, int nseg, int n,
                                                            float *__restrict__ out0, float *__restrict__ out1,
                                                            float *__restrict__ out2) {
  __shared__ float s_r[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const bool ok = e < nseg * n;
  const int ee = ok ? e : 0;
  const int k = ee / n, c = ee - k * n;
  float s = 0.f;
  for (int p = grp; p < nparts; p += 16) s += part[((long long)p * nseg + k) * n + c];
  s_r[grp][lane] = s;
  __syncthreads();
  if (grp == 0 && ok) {
    float t[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) t[j] = s_r[j][lane];
    const float r = (((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]))) +
                    (((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15])));
    float *o = k == 0 ? out0 : (k == 1 ? out1 : out2);
    if (o) o[c] = r;
  }
}

// out[i] = sum of the nsplit K-split partials of a weight-gradient GEMM (i < n, n % 4 == 0, everything 16-B aligned)
__global__ __launch_bounds__(256) void bm_sum_partials_kernel(const float *__restrict__ part, int nsplit, long long pstride,
                                                              long long n, float *__restrict__ out) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 v = *reinterpret_cast<const f32x4 *>(part + i);
  for (int sp = 1; sp < nsplit; ++sp) v += *reinterpret_cast<const f32x4 *>(part + sp * pstride + i);
  *reinterpret_cast<f32x4 *>(out + i) = v;
}

// eps-greedy over q rows with stride ldq (first-max argmax; element e draws threefry(key, (e, PQN_STREAM_ACT)))
__global__ __launch_bounds__(256) void bm_epsgreedy_kernel(const float *__restrict__ q, int ldq, int m, int a, float eps,
                                                           uint64_t key, const float *__restrict__ eps_dev,
                                                           const uint64_t *__restrict__ key_dev, int32_t *__restrict__ action,
                                                           float *__restrict__ qmax, float *__restrict__ q_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  if (eps_dev) eps = *eps_dev;
  if (key_dev) key = *key_dev;
  const float *qi = q + (long long)i * ldq;
  int best = 0;
  float bv = qi[0];
  for (int j = 1; j < a; ++j) {
    const float v = qi[j];
    if (v > bv) { bv = v; best = j; }
  }
  if (q_out)
    for (int j = 0; j < a; ++j) q_out[(long long)i * a + j] = qi[j];
  if (action) {
    uint32_t o0, o1;
    pqn_bits(key, (uint32_t)i, PQN_STREAM_ACT, o0, o1);
    const float u = pqn_uniform(o0);
    const int rnd = (int)pqn_randint(o1, (uint32_t)a);
    action[i] = (u < eps) ? rnd : best;
  }
  if (qmax) qmax[i] = bv;
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
bool bm_vec_ok(const BmOperand &o) { return (o.ld % 4) == 0 && o.ld >= 4 && (reinterpret_cast<uintptr_t>(o.p) & 15) == 0; }

#define BM_MAX_SPLIT 4

// One GEMM launch.  nsplit K splits write nsplit partial outputs E.split_stride apart (STORE) / nsplit x m-tile partial
// records (INNORM); the consumer folds them.  Loader modes are picked from the operands' alignment.
template <int BM, bool TA, bool TB, int EPI>
int bm_launch(int M, int N, int K, int nsplit, const BmOperand &A, const BmOperand &B, const BmEpilogue &E, hipStream_t st) {
  const int klen = ((K + nsplit - 1) / nsplit + BM_KS - 1) / BM_KS * BM_KS;
  const int nz = (K + klen - 1) / klen;
  const dim3 grid((N + BM_BN - 1) / BM_BN, (M + BM - 1) / BM, nz);
  const int lds = bm_lds_bytes<BM>();
  const bool va = bm_vec_ok(A), vb = bm_vec_ok(B);
  constexpr int A_V = TA ? BM_LM_COLVEC : BM_LM_ROWVEC, A_S = TA ? BM_LM_COLSCL : BM_LM_ROWSCL;
  constexpr int B_V = TB ? BM_LM_COLVEC : BM_LM_ROWVEC, B_S = TB ? BM_LM_COLSCL : BM_LM_ROWSCL;
#define BM_GO(LA_, LB_)                                                                                                  \
  do {                                                                                                                   \
    auto kern = &bm_gemm_kernel<BM, LA_, LB_, EPI>;                                                                      \
    static bool attr = false;                                                                                            \
    if (!attr) {                                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);  \
      attr = true;                                                                                                       \
    }                                                                                                                    \
    hipLaunchKernelGGL(kern, grid, dim3(BM_THREADS), lds, st, M, N, K, klen, A, B, E);                                   \
  } while (0)
  if (va && vb) BM_GO(A_V, B_V);
  else if (va) BM_GO(A_V, B_S);
  else if (vb) BM_GO(A_S, B_V);
  else BM_GO(A_S, B_S);
#undef BM_GO
  return pqn_check_launch("pqn_bigmlp gemm");
}

// Tile height and K split of a GEMM: 128-row tiles when they still give every second CU a workgroup, and as many K splits
// (<= max_split, each >= 128 long) as it takes to put ~3 workgroups on every CU -- the kernel hides its global-load
// latency behind OTHER workgroups' MFMAs.  Run-time switches "bm_tile" (64 / 128) and "bm_split" override (A/B runs).
struct BmPlan { int bm, nsplit; };
BmPlan bm_plan(int M, int N, int K, int max_split) {
  const long long t128 = (long long)((M + 127) / 128) * ((N + BM_BN - 1) / BM_BN);
  BmPlan p;
  p.bm = t128 >= 128 ? 128 : 64;
  if (pqn_opt(PQN_OPT_BM_TILE) == 64 || pqn_opt(PQN_OPT_BM_TILE) == 128) p.bm = pqn_opt(PQN_OPT_BM_TILE);
  const long long tiles = (long long)((M + p.bm - 1) / p.bm) * ((N + BM_BN - 1) / BM_BN);
  int s = (int)((768 + tiles - 1) / tiles);
  if (pqn_opt(PQN_OPT_BM_SPLIT) > 0) s = pqn_opt(PQN_OPT_BM_SPLIT);
  s = min(s, min(max_split, max(1, K / 128)));
  p.nsplit = max(1, s);
  return p;
}

template <bool TA, bool TB>
int bm_gemm(int M, int N, int K, const BmOperand &A, const BmOperand &B, BmEpilogue E, int max_split, int *nsplit_out, hipStream_t st) {
  const BmPlan p = bm_plan(M, N, K, max_split);
  const int klen = ((K + p.nsplit - 1) / p.nsplit + BM_KS - 1) / BM_KS * BM_KS;
  if (nsplit_out) *nsplit_out = (K + klen - 1) / klen;
  if (p.bm == 128) return bm_launch<128, TA, TB, BM_EPI_STORE>(M, N, K, p.nsplit, A, B, E, st);
  return bm_launch<64, TA, TB, BM_EPI_STORE>(M, N, K, p.nsplit, A, B, E, st);
}

BmOperand bm_op(const float *p, long long ld, int rows, int cols) {
  BmOperand o = {};
  o.p = p; o.ld = ld; o.rows = rows; o.cols = cols;
  return o;
}
BmEpilogue bm_store(float *out, long long ldc, const float *bias = nullptr, long long split_stride = 0) {
  BmEpilogue e = {};
  e.out = out; e.ldc = ldc; e.bias = bias; e.split_stride = split_stride;
  return e;
}

int align4(int x) { return (x + 3) & ~3; }

// workspace carve-up (floats); rows = forward rows (2 nb with next_obs, else nb), nb = rows that carry gradient
struct BmWs {
  long long coef, cspart, xn, xhat, z[PQN_BIGMLP_MAX_LAYERS], h[PQN_BIGMLP_MAX_LAYERS], stat[PQN_BIGMLP_MAX_LAYERS], q, dq, dz,
      zpart, dpart, wpart, lnpart, inpart, total;
  long long zstride, dstride, wstride;   // elements between K-split partials
  int ldq, ldx, n_cs, n_ln, n_in;
};
BmWs bm_ws(const pqn_bigmlp_layout_t &L, int rows, int nb) {
  BmWs w = {};
  long long off = 0;
  auto take = [&](long long n) { const long long o = off; off += (n + 3) & ~3ll; return o; };
  w.ldq = align4(L.a);
  w.ldx = align4(L.d);
  w.n_cs = (rows + BM_CS_ROWS - 1) / BM_CS_ROWS;
  w.n_ln = (nb + BM_LB_ROWS - 1) / BM_LB_ROWS;
  w.n_in = (nb + 63) / 64;
  w.coef = take(4ll * L.d);
  w.cspart = take(4ll * w.n_cs * L.d);   // f64 partials
  w.xn = take((long long)rows * w.ldx);
  w.xhat = take((long long)nb * w.ldx);
  for (int l = 0; l < L.layers; ++l) {
    w.z[l] = take((long long)rows * L.h);
    w.h[l] = take((long long)rows * L.h);
    w.stat[l] = take(2ll * rows);
  }
  w.q = take((long long)rows * w.ldq);
  w.dq = take((long long)nb * w.ldq);
  w.dz = take((long long)nb * L.h);
  w.zstride = (long long)rows * L.h;
  w.zpart = take(BM_MAX_SPLIT * w.zstride);
  w.dstride = (long long)nb * L.h;
  w.dpart = take(BM_MAX_SPLIT * w.dstride);
  w.wstride = (long long)align4(max(L.d, L.h)) * L.h;
  w.wpart = take(BM_MAX_SPLIT * w.wstride);
  w.lnpart = take(3ll * w.n_ln * L.h);
  w.inpart = take(2ll * BM_MAX_SPLIT * w.n_in * L.d);
  w.total = off;
  return w;
}

// forward from the (normalised, gathered) input xn through the hidden layers (z_l, h_l, stat_l kept) to Q[rows][ldq]
int bm_forward(const pqn_bigmlp_layout_t &L, int rows, const float *theta, float *ws, const BmWs &w, hipStream_t st) {
  for (int l = 0; l < L.layers; ++l) {
    const int kin = l ? L.h : L.d;
    const BmOperand A = l ? bm_op(ws + w.h[l - 1], L.h, rows, L.h) : bm_op(ws + w.xn, w.ldx, rows, L.d);
    const BmOperand B = bm_op(theta + L.off_w[l], L.h, kin, L.h);
    int ns = 1;
    const int rc = bm_gemm<false, true>(rows, L.h, kin, A, B, bm_store(ws + w.zpart, L.h, nullptr, w.zstride), BM_MAX_SPLIT, &ns, st);
    if (rc != PQN_OK) return rc;
    hipLaunchKernelGGL(bm_ln_relu_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, ws + w.zpart, ns, w.zstride, rows, L.h,
                       theta + L.off_b[l], theta + L.off_lns[l], theta + L.off_lnb[l], ws + w.z[l], ws + w.h[l], ws + w.stat[l]);
  }
  const int lo = L.layers;   // output layer: Q = h_last W_out + b_out (narrow: no K split)
  return bm_gemm<false, true>(rows, L.a, L.h, bm_op(ws + w.h[lo - 1], L.h, rows, L.h), bm_op(theta + L.off_w[lo], L.a, L.h, L.a),
                              bm_store(ws + w.q, w.ldq, theta + L.off_b[lo]), 1, nullptr, st);
}

}  // namespace

extern "C" int pqn_bigmlp_layout(int32_t d, int32_t h, int32_t layers, int32_t a, int32_t norm_input,
                                 pqn_bigmlp_layout_t *L) {
  PQN_REQUIRE(L, "pqn_bigmlp_layout: NULL layout");
  PQN_REQUIRE(d >= 8 && h >= 256 && h <= 4096 && h % 256 == 0 && layers >= 1 && layers <= PQN_BIGMLP_MAX_LAYERS && a >= 1 &&
                  a <= 64 && norm_input >= 0 && norm_input <= 2,
              "pqn_bigmlp_layout: unsupported shape d=%d h=%d layers=%d a=%d norm_input=%d (d >= 8; h: multiple of 256 in "
              "[256, 4096]; layers <= %d; a <= 64)", d, h, layers, a, norm_input, PQN_BIGMLP_MAX_LAYERS);
  *L = pqn_bigmlp_layout_t{};
  L->d = d; L->h = h; L->layers = layers; L->a = a; L->norm_input = norm_input;
  int off = 0;
  auto take = [&](int n) { const int o = off; off += align4(n); return o; };
  L->off_in_scale = take(d);
  L->off_in_bias = take(d);
  int kin = d;
  for (int l = 0; l < layers; ++l) {
    L->off_w[l] = take(kin * h);
    L->off_b[l] = take(h);
    L->off_lns[l] = take(h);
    L->off_lnb[l] = take(h);
    kin = h;
  }
  L->off_w[layers] = take(h * a);
  L->off_b[layers] = take(a);
  L->total = off;
  return PQN_OK;
}

extern "C" int64_t pqn_bigmlp_workspace_floats(const pqn_bigmlp_layout_t *L, int32_t rows, int32_t nb) {
  if (!L || rows <= 0 || nb <= 0 || nb > rows) return -1;
  return bm_ws(*L, rows, nb).total;
}

extern "C" int pqn_bigmlp_forward(const pqn_bigmlp_layout_t *L, int32_t n, const float *obs, const float *theta,
                                  float *in_mean, float *in_var, float *workspace, float *q, int32_t *action, float *qmax,
                                  float eps, uint64_t key, const float *eps_dev, const uint64_t *key_dev, void *stream) {
  PQN_REQUIRE(L && obs && theta && workspace, "pqn_bigmlp_forward: NULL argument");
  PQN_REQUIRE(n > 0 && (q || action || qmax), "pqn_bigmlp_forward: nothing to do (n=%d)", n);
  PQN_REQUIRE(L->norm_input == 0 || (in_mean && in_var), "pqn_bigmlp_forward: the input normalisation needs its running moments");
  hipStream_t st = (hipStream_t)stream;
  const BmWs w = bm_ws(*L, n, n);
  float *ws = workspace;
  const float *coef = nullptr;
  if (L->norm_input) {   // use_running_average: coefficients from the running moments
    hipLaunchKernelGGL(bm_instat_finish_kernel, dim3((L->d + 255) / 256), dim3(256), 0, st, (const double *)nullptr, 0, n, L->d,
                       theta + L->off_in_scale, theta + L->off_in_bias, in_mean, in_var, (const int32_t *)nullptr, 0,
                       L->norm_input == 2 ? 1 : 0, L->norm_input == 2 ? 1e-3f : 1e-5f, 0.0f, ws + w.coef);
    coef = ws + w.coef;
  }
  hipLaunchKernelGGL(bm_innorm_apply_kernel, dim3((unsigned)(((long long)n * (w.ldx / 4) + 255) / 256)), dim3(256), 0, st, obs,
                     (long long)L->d, (const int64_t *)nullptr, n, 0ll, n, L->d, w.ldx, coef, ws + w.xn, (float *)nullptr);
  const int rc = bm_forward(*L, n, theta, ws, w, st);
  if (rc != PQN_OK) return rc;
  hipLaunchKernelGGL(bm_epsgreedy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws + w.q, w.ldq, n, L->a, eps, key, eps_dev,
                     key_dev, action, qmax, q);
  return pqn_check_launch("pqn_bigmlp_forward");
}

extern "C" int pqn_bigmlp_grad(const pqn_bigmlp_layout_t *L, int32_t nb, const int64_t *idx, const float *obs,
                               int64_t next_offset, const int32_t *action, const float *target, const float *reward,
                               const uint8_t *done, float gamma, const float *theta, float *in_mean, float *in_var,
                               int32_t *in_steps, float *grad, float *workspace, float *loss_out, float *qv_out,
                               void *stream) {
  PQN_REQUIRE(L && idx && obs && action && theta && grad && workspace, "pqn_bigmlp_grad: NULL argument");
  PQN_REQUIRE(nb > 0 && next_offset >= 0, "pqn_bigmlp_grad: bad shape nb=%d next_offset=%lld", nb, (long long)next_offset);
  PQN_REQUIRE(next_offset > 0 ? (reward && done) : (target != nullptr),
              "pqn_bigmlp_grad: the 1-step loss needs reward + done, the Q(lambda) loss needs target");
  PQN_REQUIRE(L->norm_input == 0 || (in_mean && in_var && (L->norm_input == 1 || in_steps)),
              "pqn_bigmlp_grad: the input normalisation needs its running statistics");
  hipStream_t st = (hipStream_t)stream;
  const int rows = next_offset > 0 ? 2 * nb : nb;
  const BmWs w = bm_ws(*L, rows, nb);
  float *ws = workspace;
  const float *coef = nullptr;
  if (L->norm_input) {
    const bool renorm = L->norm_input == 2;
    hipLaunchKernelGGL(bm_colstats_kernel, dim3((L->d + 255) / 256, w.n_cs), dim3(256), 0, st, obs, (long long)L->d, idx, nb,
                       (long long)next_offset, rows, L->d, reinterpret_cast<double *>(ws + w.cspart));
    hipLaunchKernelGGL(bm_instat_finish_kernel, dim3((L->d + 255) / 256), dim3(256), 0, st,
                       reinterpret_cast<const double *>(ws + w.cspart), w.n_cs, rows, L->d, theta + L->off_in_scale,
                       theta + L->off_in_bias, in_mean, in_var, (const int32_t *)in_steps, 1, renorm ? 1 : 0,
                       renorm ? 1e-3f : 1e-5f, renorm ? 0.999f : 0.99f, ws + w.coef);
    if (renorm) hipLaunchKernelGGL(bm_steps_inc_kernel, dim3(1), dim3(1), 0, st, in_steps);
    coef = ws + w.coef;
  } else {   // the dummy input normalisation never receives gradient (pqn_craftax.py:47-49)
    if (hipMemsetAsync(grad + L->off_in_scale, 0, sizeof(float) * (size_t)(L->off_w[0] - L->off_in_scale), st) != hipSuccess) {
      pqn_set_error("pqn_bigmlp_grad: hipMemsetAsync failed");
      return PQN_E_HIP;
    }
  }
  hipLaunchKernelGGL(bm_innorm_apply_kernel, dim3((unsigned)(((long long)rows * (w.ldx / 4) + 255) / 256)), dim3(256), 0, st, obs,
                     (long long)L->d, idx, nb, (long long)next_offset, rows, L->d, w.ldx, coef, ws + w.xn,
                     coef ? ws + w.xhat : (float *)nullptr);
  int rc = bm_forward(*L, rows, theta, ws, w, st);
  if (rc != PQN_OK) return rc;
  const int lo = L->layers;
  hipLaunchKernelGGL(bm_loss_kernel, dim3(1), dim3(1024), 0, st, ws + w.q, w.ldq, nb, L->a, idx, action, target, reward, done,
                     gamma, next_offset > 0 ? 1 : 0, ws + w.dq, grad + L->off_b[lo], loss_out, qv_out);
  // backward over the first nb rows (the next_obs half carries no gradient: stop_gradient, pqn_craftax.py:301)
  float *dz = ws + w.dz;
  const BmOperand dQ = bm_op(ws + w.dq, w.ldq, nb, L->a);
  auto wgrad = [&](int kin, const BmOperand &Hin, const BmOperand &dZ, int n_out, float *gout) -> int {   // d W = Hin^T dZ
    int ns = 1;
    const long long cnt = (long long)kin * n_out;
    const bool direct = n_out < BM_BN || (cnt & 3);     // narrow output layer: one split, straight into the gradient
    const int r = bm_gemm<true, true>(kin, n_out, nb, Hin, dZ, direct ? bm_store(gout, n_out) : bm_store(ws + w.wpart, n_out, nullptr, w.wstride),
                                      direct ? 1 : BM_MAX_SPLIT, &ns, st);
    if (r != PQN_OK || direct) return r;
    hipLaunchKernelGGL(bm_sum_partials_kernel, dim3((unsigned)((cnt / 4 + 255) / 256)), dim3(256), 0, st, ws + w.wpart, ns, w.wstride,
                       cnt, gout);
    return PQN_OK;
  };
  // d W_out = h_last^T dQ;   d h_last = dQ W_out^T (K = a: one split)
  rc = wgrad(L->h, bm_op(ws + w.h[lo - 1], L->h, nb, L->h), dQ, L->a, grad + L->off_w[lo]);
  if (rc != PQN_OK) return rc;
  int nsd = 1;
  rc = bm_gemm<false, false>(nb, L->h, L->a, dQ, bm_op(theta + L->off_w[lo], L->a, L->h, L->a), bm_store(ws + w.dpart, L->h, nullptr, w.dstride),
                             1, &nsd, st);
  if (rc != PQN_OK) return rc;
  for (int l = lo - 1; l >= 0; --l) {
    // dpart (nsd K-split partials) = d loss / d h_l  ->  relu mask + LayerNorm backward: dz = d loss / d z_l
    hipLaunchKernelGGL(bm_ln_bwd_kernel, dim3(w.n_ln), dim3(256), 0, st, ws + w.dpart, nsd, w.dstride, dz, ws + w.z[l],
                       ws + w.stat[l], theta + L->off_lns[l], theta + L->off_lnb[l], nb, L->h, ws + w.lnpart);
    hipLaunchKernelGGL(bm_colreduce_kernel, dim3((3 * L->h + 63) / 64), dim3(1024), 0, st, ws + w.lnpart, w.n_ln, 3, L->h,
                       grad + L->off_lns[l], grad + L->off_lnb[l], grad + L->off_b[l]);
    const int kin = l ? L->h : L->d;
    const BmOperand dZ = bm_op(dz, L->h, nb, L->h);
    const BmOperand Hin = l ? bm_op(ws + w.h[l - 1], L->h, nb, L->h) : bm_op(ws + w.xn, w.ldx, nb, L->d);
    rc = wgrad(kin, Hin, dZ, L->h, grad + L->off_w[l]);   // d W_l = h_{l-1}^T dZ_l
    if (rc != PQN_OK) return rc;
    const BmOperand W = bm_op(theta + L->off_w[l], L->h, kin, L->h);
    if (l > 0) {   // d h_{l-1} = dZ_l W_l^T
      rc = bm_gemm<false, false>(nb, kin, L->h, dZ, W, bm_store(ws + w.dpart, L->h, nullptr, w.dstride), BM_MAX_SPLIT, &nsd, st);
      if (rc != PQN_OK) return rc;
    } else if (coef) {
      // d (input-normalisation scale, bias): column sums of (dZ_0 W_0^T) xhat and of dZ_0 W_0^T over the nb rows
      BmEpilogue E = {};
      E.out = ws + w.inpart; E.xhat = ws + w.xhat; E.ldx = w.ldx;
      const BmPlan p = bm_plan(nb, kin, L->h, BM_MAX_SPLIT);
      const int klen = ((L->h + p.nsplit - 1) / p.nsplit + BM_KS - 1) / BM_KS * BM_KS;
      const int nz = (L->h + klen - 1) / klen;
      rc = bm_launch<64, false, false, BM_EPI_INNORM>(nb, kin, L->h, p.nsplit, dZ, W, E, st);
      if (rc != PQN_OK) return rc;
      hipLaunchKernelGGL(bm_colreduce_kernel, dim3((2 * L->d + 63) / 64), dim3(1024), 0, st, ws + w.inpart, nz * w.n_in, 2, L->d,
                         grad + L->off_in_scale, grad + L->off_in_bias, (float *)nullptr);
    }
  }
  return pqn_check_launch("pqn_bigmlp_grad");
}

// where a forward intermediate lives in the workspace (tests / debugging): what 0 = normalised input xn [rows][ld],
// 1 = pre-activation z_l [rows][h], 2 = activation h_l = relu(LN(z_l)) [rows][h], 3 = (mean, rstd) of z_l [rows][2], 4 = q [rows][ld]
extern "C" int pqn_bigmlp_workspace_view(const pqn_bigmlp_layout_t *L, int32_t rows, int32_t nb, int32_t what, int32_t layer,
                                         int64_t *offset, int64_t *ld) {
  PQN_REQUIRE(L && offset && ld && rows > 0 && nb > 0 && nb <= rows && layer >= 0 && layer < L->layers && what >= 0 && what <= 4,
              "pqn_bigmlp_workspace_view: bad arguments");
  const BmWs w = bm_ws(*L, rows, nb);
  switch (what) {
    case 0: *offset = w.xn; *ld = w.ldx; break;
    case 1: *offset = w.z[layer]; *ld = L->h; break;
    case 2: *offset = w.h[layer]; *ld = L->h; break;
    case 3: *offset = w.stat[layer]; *ld = 2; break;
    default: *offset = w.q; *ld = w.ldq; break;
  }
  return PQN_OK;
}

// C[M][N] = op(A) op(B) (+ bias[N]) with f32-grade bf16x3 products -- the GEMM every Dense layer above runs (one K split),
// exposed for tests against a plain f32 / f64 matmul.  trans_a: A is stored [K][M] (else [M][K]); trans_b: B is stored
// [K][N] (else [N][K]).  nsplit > 1: K-split partial outputs, `c` then holds nsplit x [M][ldc] partials split_stride apart.
extern "C" int pqn_bigmlp_gemm(int32_t m, int32_t n, int32_t k, const float *a, int64_t lda, int32_t trans_a, const float *b,
                               int64_t ldb, int32_t trans_b, const float *bias, float *c, int64_t ldc, int32_t nsplit,
                               int64_t split_stride, int32_t tile_rows, void *stream) {
  PQN_REQUIRE(a && b && c && m > 0 && n > 0 && k > 0 && nsplit >= 1 && nsplit <= BM_MAX_SPLIT && (tile_rows == 64 || tile_rows == 128),
              "pqn_bigmlp_gemm: bad arguments");
  PQN_REQUIRE(nsplit == 1 || !bias, "pqn_bigmlp_gemm: the bias belongs to the consumer of K-split partials");
  hipStream_t st = (hipStream_t)stream;
  const BmOperand A = trans_a ? bm_op(a, lda, k, m) : bm_op(a, lda, m, k);
  const BmOperand B = trans_b ? bm_op(b, ldb, k, n) : bm_op(b, ldb, n, k);
  const BmEpilogue E = bm_store(c, ldc, bias, split_stride);
#define BM_T(TA_, TB_) (tile_rows == 128 ? bm_launch<128, TA_, TB_, BM_EPI_STORE>(m, n, k, nsplit, A, B, E, st) \
                                         : bm_launch<64, TA_, TB_, BM_EPI_STORE>(m, n, k, nsplit, A, B, E, st))
  if (trans_a) return trans_b ? BM_T(true, true) : BM_T(true, false);
  return trans_b ? BM_T(false, true) : BM_T(false, false);
#undef BM_T
}
