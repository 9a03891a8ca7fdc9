// pqn_bigmlp.hip -- the wide MLP Q-network of the Craftax script as gfx950 kernels.
//
// Network = QNetwork of the reference's purejaxql/pqn_craftax.py:33-62 with NORM_TYPE = layer_norm (the value of
// config/alg/pqn_craftax.yaml:12): [BatchRenorm | BatchNorm | nothing](x) -> NUM_LAYERS x (Dense(H) -> LayerNorm -> relu)
// -> Dense(A); C5 = 1345 -> 4 x 1024 -> 17 on 2 x 1024 rows per update (obs and next_obs as ONE batch, :287-304).
// These are GEMM-sized layers (2048 x 1345 x 1024, 2048 x 1024 x 1024), unlike the 16-sample tiles of pqn_mlp.hip whose
// activations live in LDS: here every Dense layer -- forward, input gradient, weight gradient -- is ONE tiled MFMA GEMM
// kernel, and everything elementwise sits in a handful of row / column kernels around it.
//
// Operand format ("planes").  The GEMMs evaluate f32 products on the bf16 matrix core as bf16x3 (pqn_qnet.hip: x = hi +
// mid + lo exactly, 6 x v_mfma_f32_16x16x32_bf16 per 32-wide K step, f32 accumulate).  Splitting inside the GEMM would
// repeat the ~5 VALU ops per element once per output tile that touches the element (16x for these shapes) and, measured,
// cost as much issue time as the MFMAs themselves; so every tensor that feeds a GEMM is split ONCE, by the kernel that
// produces it, into three bf16 planes, k padded with zeros to a multiple of 32.  A GEMM operand is "row r, K-values k ..
// k + 7 = 16 bytes per plane"; since round 5 those 16-B slots are stored FRAGMENT-MAJOR (bm_slot below: one contiguous 1 KB
// block per 16 rows x 32 K-values, laid out as the LDS image the MFMA fragments are read from), so that the GEMM moves a K
// step global -> LDS with one LDS-DMA instruction per block: no registers, no arithmetic, no bounds masks (row blocks are
// clamped: a clamped block only feeds outputs that are never stored).  Tensors needed with the other dimension as K (weights
// for the forward pass, activations and output gradients for the weight gradients) get a transposed plane copy from an
// LDS-tiled 2-byte transpose kernel.
//
//   bm_gemm_kernel          C[M,N] (+)= A[M,K] B[N,K]^T from planes; tile BM x 64 (BM = 64 | 128), 4 waves as 2 x 2, two LDS
//                           stages filled by LDS-DMA (global_load_lds_dwordx4, 6 | 9 per wave and K step), one barrier per
//                           K step; split-K over blockIdx.z (partial outputs, folded by the consumer) so that ~3
//                           workgroups sit on every CU and hide each other's latency; epilogue = (bias +) store, or
//                           the column sums that are the input-normalisation parameter gradients
//   bm_split_kernel         f32 matrix -> planes (weights; test entry);   bm_transpose_kernel  planes -> transposed planes;
//                           bm_split_transpose_multi: theta -> transposed weight planes in one pass (after an optimizer step)
//   bm_colstats* / bm_innorm_apply   batch moments of the gathered input rows, BatchRenorm / BatchNorm bookkeeping
//                           (utils/batch_renorm.py:95-116), normalised input as planes (+ xhat for the parameter gradients)
//   bm_ln_relu              z = sum of K-split partials + bias; LayerNorm (flax: var = E[x^2] - E[x]^2 clamped, eps 1e-6) +
//                           relu -> activation planes; z and (mean, rstd) kept for the backward pass
//   bm_qfold                Q = sum of the output layer's K-split partials + bias (the 17-column layer runs 8 K splits)
//   bm_loss                 TD loss of both branches of _loss_fn (pqn_craftax.py:277-312), dQ planes; per-workgroup records of
//                           d b_out / loss / mean chosen Q for bm_colreduce
//   bm_ln_bwd               relu mask + LayerNorm backward -> dz planes, column partial sums for d scale / d bias / d dense-bias
//   bm_colreduce / bm_sum_partials   fixed-order folds of per-workgroup column partials / K-split weight-gradient partials
//                           (_multi: every layer's fold in one launch)
// The row / column kernels issue all loads of a thread before the first use (straight-line code: K-split counts as template
// parameters, clamped indices instead of bounds branches) -- they move a few MB each and are latency-, not bandwidth-bound.
// Everything is deterministic (fixed summation orders, no atomics).
#include <stdlib.h>

#include "pqn_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned short bf16_t;   // raw bf16 bits

#define BM_THREADS 256
#define BM_KS 32
#define BM_LN_EPS 1e-6f
#define BM_MAX_SPLIT 4

PQN_HD int bm_pad32(int x) { return (x + 31) & ~31; }

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
// eight consecutive K-values -> one 16-B slot per plane
PQN_D void bm_split8(const float (&v)[8], u32x4 &h, u32x4 &m, u32x4 &l) {
  unsigned hh[4], mm[4], ll[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bm_split2(v[2 * q], v[2 * q + 1], hh[q], mm[q], ll[q]);
  h = u32x4{hh[0], hh[1], hh[2], hh[3]};
  m = u32x4{mm[0], mm[1], mm[2], mm[3]};
  l = u32x4{ll[0], ll[1], ll[2], ll[3]};
}
// tied-accumulator MFMA as volatile inline asm (see x3_mfma_tied in pqn_qnet.hip for why not the builtin)
PQN_D f32x4 bm_mfma(const u32x4 &a, const u32x4 &b, f32x4 c) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
PQN_D void bm_drain(f32x4 &a) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a)); }
// LDS-DMA (see pqn_qnet_pos.hip): 64 lanes x 16 B from uniform base + per-lane byte offset to LDS at a uniform address + 16 lane
typedef __attribute__((address_space(3))) char bm_lds_char_t;
PQN_D uint32_t bm_lds_addr(const void *p) { return (uint32_t)(uintptr_t)(bm_lds_char_t *)p; }
PQN_D void bm_dma16(uint32_t voff, const void *sbase, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
PQN_D void bm_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Planes of a logical matrix X[rows][k]: plane p at p + pl * pstride, leading dimension ld (bf16 elements, ld % 32 == 0,
// columns [k, ld) zero), 16-B aligned.  Round 5: the planes are stored FRAGMENT-MAJOR -- one contiguous 1 KB block per (16
// rows, 32 K-values), the blocks of a row block consecutive along K; inside a block the 16-B slot of (row r, K-values 8 kb ..
// 8 kb + 7) is number 4 r + (kb ^ (-(r >> 2) & 3)): a row's four slots are one 64-B run (a producer that owns a row writes
// whole 64-B pieces), and the block IS the LDS image a 16 x 32 MFMA operand fragment is read from conflict-free -- the XOR
// puts the 16 lanes of every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS) on 16 different slots mod 16 -- so that the
// GEMM moves it global -> LDS by LDS-DMA with fully contiguous reads and no register staging.  Every producer and the
// transposes address 16-B slots through bm_slot(); rows are allocated in whole blocks (bm_pad16).
PQN_HD int bm_pad16(int x) { return (x + 15) & ~15; }
PQN_HD long long bm_slot(long long ld, long long row, int k) {   // element offset of the slot holding (row, k .. k + 7), k % 8 == 0
  const int kb = (k >> 3) & 3, r = (int)row & 15;
  return ((((row >> 4) * (ld >> 5) + (k >> 5)) << 6) + 4 * r + (kb ^ ((0 - (r >> 2)) & 3))) << 3;
}
struct BmPlanes {
  const bf16_t *p;
  long long ld, pstride;
  int rows;
};
struct BmPlanesOut {
  bf16_t *p;
  long long ld, pstride;
};
PQN_D void bm_put(const BmPlanesOut &o, long long row, int k, const u32x4 &h, const u32x4 &m, const u32x4 &l) {
  bf16_t *q = o.p + bm_slot(o.ld, row, k);
  *reinterpret_cast<u32x4 *>(q) = h;
  *reinterpret_cast<u32x4 *>(q + o.pstride) = m;
  *reinterpret_cast<u32x4 *>(q + 2 * o.pstride) = l;
}

enum { BM_EPI_STORE = 0, BM_EPI_INNORM = 1 };
struct BmEpilogue {
  float *out;                // BM_EPI_STORE: C, row-major, leading dimension ldc;  BM_EPI_INNORM: partials [K split][m tile][2][N]
  long long ldc;
  const float *bias;         // BM_EPI_STORE: optional per-column bias (only without split-K)
  long long split_stride;    // BM_EPI_STORE: elements between the partial outputs of consecutive K splits
  // BM_EPI_INNORM (input-normalisation parameter gradients): d scale_c = sum_r C[r][c] xhat[r][c], d bias_c = sum_r C[r][c]
  const float *xhat;         // [M][ldx]: (x - m_c) k_c of the gradient rows
  long long ldx;
  unsigned long long *stamps;   // (rounds 3-4: cycle stamps of the register-staged K loop; the LDS-DMA loop of round 5 carries no vector-memory stores and records none)
};

template <int BM, int BN>
constexpr int bm_lds_bytes() { return 2 * 3 * (BM / 16 + BN / 16) * 64 * 16; }

// C[m][n] = sum_k A[m][k] B[n][k]; K padded (the planes hold zeros there), klen % 32 == 0.
// Workgroup tile BM x BN (128 x 64 | 64 x 64; 128 x 128 builds too and measured slower at one workgroup per CU), four
// waves as 2 x 2, wave tile (BM/2) x (BN/2) = MI x NI MFMA tiles; two (three) workgroups per CU.
// (A single-stage LDS plan with four 64 x 64 workgroups per CU was measured in round 5 and lost by 1 %:
// profiles/r05_v4_c5_gemm_single_stage_ab.txt.)
template <int BM, int BN, int EPI>
__global__ __launch_bounds__(BM_THREADS) void bm_gemm_kernel(int M, int N, int Kp, int klen, BmPlanes A, BmPlanes B, BmEpilogue E) {
  constexpr int NBA = BM / 16, NBB = BN / 16;
  constexpr int MI = BM / 32, NI = BN / 32;                             // 16 x 16 MFMA tiles per wave
  extern __shared__ __attribute__((aligned(16))) char bm_smem[];
  u32x4 *sA = reinterpret_cast<u32x4 *>(bm_smem);                         // [2][3][NBA][64]
  u32x4 *sB = sA + 2 * 3 * NBA * 64;                                      // [2][3][NBB][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order.  Workgroups go to the 8 XCDs round-robin by linear id, and each XCD has its own 4 MB L2: with
  // the plain (x, y, z) order the workgroups an XCD runs at a time touch many different A panels a few times each and
  // every panel is fetched by every XCD.  Here XCD k owns a contiguous eighth of the (split, m tile, n tile) sequence, n
  // fastest: its concurrent workgroups cover a few m tiles x all n tiles of one K split -- a panel set that fits its L2
  // (measured L2 hit rate of the kernel: 89 %).
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const unsigned tx = gridDim.x, ty = gridDim.y, total = tx * ty * gridDim.z;
    if ((total & 7u) == 0u) {
      const unsigned lin = blockIdx.x + tx * (blockIdx.y + ty * blockIdx.z);
      const unsigned q = (lin & 7u) * (total >> 3) + (lin >> 3);
      bz = q / (tx * ty);
      const unsigned r = q - bz * (tx * ty);
      by = r / tx;
      bx = r - by * tx;
    }
  }
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * klen, kend = min(Kp, kbeg + klen);
  // Global -> LDS by LDS-DMA.  The planes are fragment-major (bm_slot): the 3 (NBA + NBB) blocks a K step needs are 1 KB
  // runs of global memory, and one global_load_lds_dwordx4 per block (64 lanes x 16 B, lane-linear on both sides) drops each
  // into its place of the stage -- no VGPR staging, no ds_write, no address arithmetic in the loop.  (Round 4 staged through
  // registers from row-major planes: the LDS-DMA form needs lane = slot, which on row-major planes is a 16-B gather per lane
  // and measured 4x the issue time.)  Issued as inline asm so that the compiler's wait-count bookkeeping does not see it (it
  // would order every LDS read behind vmcnt(0), see pqn_qnet_pos.hip); the kernel drains it itself in front of the barrier
  // that publishes a stage.  Row blocks beyond the matrix are clamped to the last one: they only feed outputs that are never
  // stored.  tools/ubench/gemm_dma.hip is the measurement this rests on (1024^3, three K splits: 17.2 against 21.9 us).
  constexpr int NBLK = 3 * (NBA + NBB), PER_WAVE = NBLK / 4;
  static_assert(NBLK % 4 == 0, "blocks of a K step must divide over the four waves");
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int a_blocks = (A.rows + 15) >> 4, b_blocks = (B.rows + 15) >> 4;
  const long long a_kb = A.ld >> 5, b_kb = B.ld >> 5;
  const bf16_t *src[PER_WAVE];
  uint32_t dst[PER_WAVE];      // LDS byte offset inside a stage
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int q = PER_WAVE * wave_u + i;
    if (q < 3 * NBA) {
      const int pl = q / NBA, rb = q % NBA;
      src[i] = A.p + pl * A.pstride + (((long long)min(m0 / 16 + rb, a_blocks - 1) * a_kb + (kbeg >> 5)) << 9);
      dst[i] = (uint32_t)((pl * NBA + rb) * 1024);
    } else {
      const int qq = q - 3 * NBA, pl = qq / NBB, rb = qq % NBB;
      src[i] = B.p + pl * B.pstride + (((long long)min(n0 / 16 + rb, b_blocks - 1) * b_kb + (kbeg >> 5)) << 9);
      dst[i] = (uint32_t)((2 * 3 * NBA + pl * NBB + rb) * 1024);   // sB starts behind both stages of sA
    }
  }
  const uint32_t lds0 = bm_lds_addr(bm_smem);
  auto dma_step = [&](int ks) {
    const int buf = ks & 1;
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const bool is_a = PER_WAVE * wave_u + i < 3 * NBA;
      bm_dma16(lane * 16u, src[i] + ((long long)ks << 9), lds0 + dst[i] + (uint32_t)(buf * 3 * (is_a ? NBA : NBB) * 1024));
    }
  };
  // ONE accumulator per tile: the six products of a step meet in f32 in the order (l,h) (m,h) (h,l) (h,m) (m,m) (h,h)
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  // K loop: at the top of iteration ks stage ks & 1 holds step ks (landed, published by the last barrier); the iteration starts
  // the transfer of step ks + 1 into the other stage (every wave passed the barrier behind its last read of it), reads its
  // fragments, issues the MFMAs, waits for its own share of the transfer and meets the others at the one barrier of the step.
  const int nk = (kend - kbeg) / BM_KS;
  u32x4 a[MI][3], b[NI][3];
  auto fragload = [&](int buf) {
    const int fl = 4 * (lane & 15) + ((lane >> 4) ^ ((0 - ((lane & 15) >> 2)) & 3));   // fragment lane (row i, kq) -> its slot (bm_slot)
    const u32x4 *pa = sA + buf * 3 * NBA * 64 + (wm * MI) * 64 + fl;
    const u32x4 *pb = sB + buf * 3 * NBB * 64 + (wn * NI) * 64 + fl;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi][pl] = pa[(pl * NBA + mi) * 64];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni][pl] = pb[(pl * NBB + ni) * 64];
    }
  };
  if (nk > 0) dma_step(0);
  bm_dma_wait();
  __syncthreads();
#pragma unroll 1
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + 1 < nk) dma_step(ks + 1);
    fragload(ks & 1);
#define BM_PROD(PA_, PB_)                                                          \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = bm_mfma(a[mi][PA_], b[ni][PB_], acc[mi][ni]);
    BM_PROD(2, 0)
    BM_PROD(1, 0)
    BM_PROD(0, 2)
    BM_PROD(0, 1)
    BM_PROD(1, 1)
    BM_PROD(0, 0)
#undef BM_PROD
    bm_dma_wait();
    __syncthreads();
  }
  // ---- epilogue ----
  const int col_l = lane & 15, rq = lane >> 4;
  if (EPI == BM_EPI_STORE) {
    float *outp = E.out + (long long)bz * E.split_stride;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        bm_drain(acc[mi][ni]);
        const float tv[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
        const int col = n0 + (wn * NI + ni) * 16 + col_l;
        const float bias = E.bias ? E.bias[min(col, N - 1)] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + (wm * MI + mi) * 16 + 4 * rq + r;
          if (row < M && col < N) outp[(long long)row * E.ldc + col] = tv[r] + bias;
        }
      }
  } else {
    // input-normalisation parameter gradients: column sums over this tile's rows, folded lane -> row groups -> the two
    // wave rows through LDS in a fixed order; one partial record per (K split, m tile)
    __syncthreads();   // the operand buffers are dead: reuse them
    float *red = reinterpret_cast<float *>(bm_smem);   // [2 wm][BN cols][2]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int cl = (wn * NI + ni) * 16 + col_l, col = n0 + cl;
      const int cc = min(col, N - 1);
      float ss = 0.f, sb_ = 0.f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        bm_drain(acc[mi][ni]);
        const float tv[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + (wm * MI + mi) * 16 + 4 * rq + r;
          const float xh = E.xhat[(long long)min(row, M - 1) * E.ldx + cc];
          const float g = (row < M && col < N) ? tv[r] : 0.0f;
          ss += g * xh;
          sb_ += g;
        }
      }
      ss += __shfl_xor(ss, 16, 64); ss += __shfl_xor(ss, 32, 64);
      sb_ += __shfl_xor(sb_, 16, 64); sb_ += __shfl_xor(sb_, 32, 64);
      if (rq == 0) {
        red[(wm * BN + cl) * 2] = ss;
        red[(wm * BN + cl) * 2 + 1] = sb_;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < N) {
      float *po = E.out + ((long long)bz * gridDim.y + by) * 2 * N;
      po[n0 + tid] = red[tid * 2] + red[(BN + tid) * 2];
      po[N + n0 + tid] = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
    }
  }
}

// all-reduce over the 64 lanes of a wave, fixed butterfly order
// row of Q (ldq % 4 == 0, 16-B aligned, ldq <= 32) into registers: 8 unconditional vector loads, packs beyond the row
// re-read its last pack (those entries are never used: k >= ldq >= a)
PQN_D void bm_load_qrow(const float *__restrict__ row, int ldq, float (&out)[32]) {
#pragma unroll
  for (int k4 = 0; k4 < 8; ++k4) {
    const f32x4 t = *reinterpret_cast<const f32x4 *>(row + min(4 * k4, ldq - 4));
    out[4 * k4] = t.x; out[4 * k4 + 1] = t.y; out[4 * k4 + 2] = t.z; out[4 * k4 + 3] = t.w;
  }
}

PQN_D float bm_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// f32 matrix src[rows][cols] (leading dimension lds, any alignment) -> planes [rows][ld]; one thread per 8 columns,
// columns >= cols are written as zeros up to ld
PQN_D void bm_split_body(const float *__restrict__ src, long long lds, int rows, int cols, const BmPlanesOut &out, long long block) {
  const int ppr = (int)(out.ld / 8);
  const long long e = block * 256 + threadIdx.x;
  if (e >= (long long)rows * ppr) return;
  const int r = (int)(e / ppr), k = (int)(e % ppr) * 8;
  const float *sr = src + (long long)r * lds;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = sr[min(k + j, cols - 1)];
    v[j] = (k + j < cols) ? x : 0.0f;
  }
  u32x4 h, m, l;
  bm_split8(v, h, m, l);
  bm_put(out, r, k, h, m, l);
}
__global__ __launch_bounds__(256) void bm_split_kernel(const float *__restrict__ src, long long lds, int rows, int cols,
                                                       BmPlanesOut out) {
  bm_split_body(src, lds, rows, cols, out, blockIdx.x);
}
// several matrices in one launch (the Dense kernels of every layer after an optimizer step): job j owns blocks
// [first[j], first[j + 1])
#define BM_MAX_JOBS (2 * PQN_BIGMLP_MAX_LAYERS + 2)   // backward: the input, every h_l, dQ and every dz_l in one transpose launch
struct BmSplitJobs {
  int n;
  const float *src[BM_MAX_JOBS];
  long long lds[BM_MAX_JOBS];
  int rows[BM_MAX_JOBS], cols[BM_MAX_JOBS];
  BmPlanesOut out[BM_MAX_JOBS];
  long long first[BM_MAX_JOBS + 1];
};
__global__ __launch_bounds__(256) void bm_split_multi_kernel(BmSplitJobs J) {
  int j = 0;
  while (j + 1 < J.n && (long long)blockIdx.x >= J.first[j + 1]) ++j;
  bm_split_body(J.src[j], J.lds[j], J.rows[j], J.cols[j], J.out[j], (long long)blockIdx.x - J.first[j]);
}

// planes src[rows][ld_s] (logical columns `cols`) -> planes dst[cols][ld_d] (logical columns `rows`, zero beyond);
// 64 x 64 element tiles through LDS, 16-B global accesses both ways; grid (ceil(ld_d / 64), ceil(cols / 64), 3 planes)
PQN_D void bm_transpose_body(const BmPlanes &src, int cols, const BmPlanesOut &dst, int bx, int by, int bz, bf16_t (*tile)[72]) {
  const int tid = threadIdx.x;
  const int r0 = bx * 64, c0 = by * 64;   // source rows r0.., source columns c0..
  const bf16_t *sp = src.p + (long long)bz * src.pstride;
  bf16_t *dp = dst.p + (long long)bz * dst.pstride;
  u32x4 lv[2];   // load: 64 rows x 8 chunks of 8 columns (columns >= cols hold the source's zero padding); both loads in flight
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + 256 * q, r = e >> 3, ch = e & 7;
    const int rr = min(r0 + r, src.rows - 1), cc = min((long long)(c0 + 8 * ch), src.ld - 8);   // clamped: masked below
    lv[q] = *reinterpret_cast<const u32x4 *>(sp + bm_slot(src.ld, rr, (int)cc));
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + 256 * q, r = e >> 3, ch = e & 7;
    const bool ok = r0 + r < src.rows && c0 + 8 * ch < src.ld;
    *reinterpret_cast<u32x4 *>(&tile[r][8 * ch]) = ok ? lv[q] : u32x4{0u, 0u, 0u, 0u};
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {   // store: 64 destination rows (source columns) x 8 chunks of 8 source rows
    const int e = tid + 256 * q, c = e >> 3, ch = e & 7;
    const int cc = c0 + c, rr = r0 + 8 * ch;
    if (cc < cols && rr < dst.ld) {
      unsigned v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[8 * ch + j][c];   // rows >= src.rows were loaded as zeros
      u32x4 o;
      o.x = v[0] | (v[1] << 16); o.y = v[2] | (v[3] << 16); o.z = v[4] | (v[5] << 16); o.w = v[6] | (v[7] << 16);
      *reinterpret_cast<u32x4 *>(dp + bm_slot(dst.ld, cc, rr)) = o;
    }
  }
}
__global__ __launch_bounds__(256) void bm_transpose_kernel(BmPlanes src, int cols, BmPlanesOut dst) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];   // row stride 144 B: the 2-B column gathers below spread over the banks
  bm_transpose_body(src, cols, dst, blockIdx.x, blockIdx.y, blockIdx.z, tile);
}
// several plane sets in one launch; job j owns the linear blocks [first[j], first[j + 1]) = its (bx, by) grid, row-major in by
struct BmTransposeJobs {
  int n;
  BmPlanes src[BM_MAX_JOBS];
  int cols[BM_MAX_JOBS], nbx[BM_MAX_JOBS];
  BmPlanesOut dst[BM_MAX_JOBS];
  int first[BM_MAX_JOBS + 1];
};
__global__ __launch_bounds__(256) void bm_transpose_multi_kernel(BmTransposeJobs J) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
  int j = 0;
  while (j + 1 < J.n && (int)blockIdx.x >= J.first[j + 1]) ++j;
  const int b = (int)blockIdx.x - J.first[j];
  bm_transpose_body(J.src[j], J.cols[j], J.dst[j], b % J.nbx[j], b / J.nbx[j], blockIdx.z, tile);
}

// f32 matrix src[rows][cols] (leading dimension lds, any alignment) -> planes dst[cols][ld_d] of its TRANSPOSE (logical
// columns `rows`, zero beyond): split + transpose in one pass, 64 x 64 element tiles through LDS.  Job j owns the linear
// blocks [first[j], first[j + 1]) = its (bx over source rows, by over source columns) grid, row-major in by.
struct BmSplitTJobs {
  int n;
  const float *src[BM_MAX_JOBS];
  long long lds[BM_MAX_JOBS];
  int rows[BM_MAX_JOBS], cols[BM_MAX_JOBS], nbx[BM_MAX_JOBS];
  BmPlanesOut dst[BM_MAX_JOBS];
  int first[BM_MAX_JOBS + 1];
};
__global__ __launch_bounds__(256) void bm_split_transpose_multi_kernel(BmSplitTJobs J) {
  __shared__ float tile[64][65];
  int j = 0;
  while (j + 1 < J.n && (int)blockIdx.x >= J.first[j + 1]) ++j;
  const int b = (int)blockIdx.x - J.first[j];
  const int r0 = (b % J.nbx[j]) * 64, c0 = (b / J.nbx[j]) * 64;   // source rows r0.., source columns c0..
  const float *__restrict__ src = J.src[j];
  const long long lds = J.lds[j];
  const int rows = J.rows[j], cols = J.cols[j];
  const BmPlanesOut dst = J.dst[j];
  const int tid = threadIdx.x;
  float v[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {   // 16 coalesced 4-B loads per thread in flight
    const int e = tid + 256 * q, r = e >> 6, c = e & 63;
    v[q] = (r0 + r < rows && c0 + c < cols) ? src[(long long)(r0 + r) * lds + c0 + c] : 0.0f;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int e = tid + 256 * q;
    tile[e >> 6][e & 63] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q) {    // destination row = source column c, 8 source rows = 8 K-values per 16-B pack
    const int e = tid + 256 * q, c = e >> 3, ch = e & 7;
    const int cc = c0 + c, rr = r0 + 8 * ch;
    if (cc < cols && rr < dst.ld) {
      float y[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) y[k] = tile[8 * ch + k][c];   // rows >= `rows` were loaded as zeros
      u32x4 ph, pm, pl;
      bm_split8(y, ph, pm, pl);
      bm_put(dst, cc, rr, ph, pm, pl);
    }
  }
}

// z = sum of the K-split partials of the layer's GEMM + bias;  h = relu(LayerNorm(z)) row by row (one wave per row) written
// as planes; z and stat[r] = (mean, rstd) are kept for the backward pass.  n % 256 == 0, n <= 2048: lane owns the
// 8-column packs lane, lane + 64, ... of the row.
#define BM_MAXP 4   // packs per lane at the widest row (n <= 2048); the LayerNorm kernels are instantiated per count (KP)
// Straight-line loads: NS (the K-split count) is a template parameter and the packs beyond the row are clamped to its
// last pack instead of branched around, so every load of a lane (partials, bias, LayerNorm scale / bias) is in flight at
// once.  With a run-time split loop inside per-pack branches each (pack, half, split) was its own round trip: 10 us per
// launch at 1024 x 1024 where the traffic needs 4.
template <int NS, int KP>   // KP = 8-column packs per lane = ceil(n / 512): no loads or arithmetic for packs the row does not have
__global__ __launch_bounds__(256) void bm_ln_relu_kernel(const float *__restrict__ zpart, long long pstride, int m,
                                                         int n, const float *__restrict__ bias, const float *__restrict__ g,
                                                         const float *__restrict__ beta, float *__restrict__ z,
                                                         BmPlanesOut h, float *__restrict__ stat) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;
  const int np = n / 8;   // packs per row
  f32x4 v[KP][2], gv[KP][2], bv[KP][2];
#pragma unroll
  for (int k = 0; k < KP; ++k) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int c = 8 * min(lane + 64 * k, np - 1) + 4 * hh;
      f32x4 t = *reinterpret_cast<const f32x4 *>(zpart + (long long)row * n + c);
#pragma unroll
      for (int sp = 1; sp < NS; ++sp) t += *reinterpret_cast<const f32x4 *>(zpart + sp * pstride + (long long)row * n + c);
      t += *reinterpret_cast<const f32x4 *>(bias + c);
      v[k][hh] = t;
      gv[k][hh] = *reinterpret_cast<const f32x4 *>(g + c);
      bv[k][hh] = *reinterpret_cast<const f32x4 *>(beta + c);
    }
  }
  float s = 0.f, q = 0.f;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    if (lane + 64 * k < np) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x4 t = v[k][hh];
        *reinterpret_cast<f32x4 *>(z + (long long)row * n + 8 * (lane + 64 * k) + 4 * hh) = t;
        s += (t.x + t.y) + (t.z + t.w);
        q += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
      }
    }
  }
  s = bm_wave_sum(s);
  q = bm_wave_sum(q);
  const float mean = s / (float)n;
  const float var = fmaxf(q / (float)n - mean * mean, 0.0f);
  const float rstd = 1.0f / sqrtf(var + BM_LN_EPS);
  if (lane == 0) { stat[2 * row] = mean; stat[2 * row + 1] = rstd; }
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const int pk = lane + 64 * k;
    if (pk < np) {
      float y[8];
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x4 xh = (v[k][hh] - mean) * rstd;
        const f32x4 yy = xh * gv[k][hh] + bv[k][hh];
        y[4 * hh] = fmaxf(yy.x, 0.f); y[4 * hh + 1] = fmaxf(yy.y, 0.f); y[4 * hh + 2] = fmaxf(yy.z, 0.f); y[4 * hh + 3] = fmaxf(yy.w, 0.f);
      }
      u32x4 ph, pm, pl;
      bm_split8(y, ph, pm, pl);
      bm_put(h, row, 8 * pk, ph, pm, pl);
    }
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
// workgroup = 64 columns x 4 waves; wave w sums rows r0 + w, r0 + w + 4, ... of the chunk (16 loads in flight per lane instead
// of one lane walking 64 rows), the four wave sums are folded in a fixed order through LDS.  grid (ceil(d / 64), chunks).
__global__ __launch_bounds__(256) void bm_colstats_kernel(const float *__restrict__ x, long long ldx, const int64_t *idx, int nb,
                                                          long long next_off, int m, int d, double *__restrict__ part) {
  __shared__ double s_s[4][64], s_q[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * BM_CS_ROWS, r1 = min(r0 + BM_CS_ROWS, m);
  const int cc = min(c, d - 1);
  // lane i < 16 fetches the source row of the wave's i-th row, then the 16 column loads go out together (a loop of
  // "row index, wait, element, wait" was 2 x 16 dependent round trips)
  const int myr = r0 + wave + 4 * (lane & 15);
  const long long mysrc = myr < r1 ? bm_src_row(idx, nb, next_off, myr) : 0ll;
  float v[BM_CS_ROWS / 4];
#pragma unroll
  for (int i = 0; i < BM_CS_ROWS / 4; ++i) v[i] = x[__shfl(mysrc, i) * ldx + cc];   // rows >= r1: row 0, not added
  double s = 0.0, q = 0.0;
#pragma unroll
  for (int i = 0; i < BM_CS_ROWS / 4; ++i) {
    if (r0 + wave + 4 * i < r1) {
      const double d = (double)v[i];
      s += d;
      q += d * d;
    }
  }
  s_s[wave][lane] = s;
  s_q[wave][lane] = q;
  __syncthreads();
  if (wave == 0 && c < d) {
    part[((long long)blockIdx.y * 2) * d + c] = (s_s[0][lane] + s_s[1][lane]) + (s_s[2][lane] + s_s[3][lane]);
    part[((long long)blockIdx.y * 2 + 1) * d + c] = (s_q[0][lane] + s_q[1][lane]) + (s_q[2][lane] + s_q[3][lane]);
  }
}

// stage 2 + the bookkeeping of utils/batch_renorm.py:95-116 (renorm = 1) / flax nn.BatchNorm (renorm = 0), thread per
// column.  coef = [4][ldc] (ldc >= d, a multiple of 8; zero beyond d): m_c (mean used), a_c = k_c * scale_c, b_c = bias_c,
// k_c = 1 / sqrt(var used + eps).
// train = 0: coefficients from the running moments only (use_running_average).  steps[0] = BatchRenorm's train-call
// counter, steps[1] = a ticket word (zero between launches): the last workgroup to finish advances the counter, once
// every thread of the launch has read it.
__global__ __launch_bounds__(256) void bm_instat_finish_kernel(const double *__restrict__ part, int nparts, int m, int d,
                                                               const float *__restrict__ scale, const float *__restrict__ bias,
                                                               float *__restrict__ ra_mean, float *__restrict__ ra_var,
                                                               int32_t *__restrict__ steps, int train, int renorm, float eps,
                                                               float momentum, float *__restrict__ coef, int ldc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;   // 64-thread workgroups: d / 64 CUs share the partial loads
  if (c < d) {
    float mean = ra_mean[c], var = ra_var[c];
    if (train) {
      double s = 0.0, q = 0.0;
      for (int p0 = 0; p0 < nparts; p0 += 8) {   // 16 loads in flight; same order of additions
        double ts[8], tq[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int p = min(p0 + u, nparts - 1);
          ts[u] = part[((long long)p * 2) * d + c];
          tq[u] = part[((long long)p * 2 + 1) * d + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (p0 + u < nparts) { s += ts[u]; q += tq[u]; }
      }
      const float bmean = (float)(s / (double)m);
      const float bvar = fmaxf((float)(q / (double)m) - bmean * bmean, 0.0f);
      mean = bmean;
      var = bvar;
      if (renorm && steps[0] >= 1000) {
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
    coef[ldc + c] = k * scale[c];
    coef[2 * ldc + c] = bias[c];
    coef[3 * ldc + c] = k;
  } else if (c < ldc) {
    coef[c] = 0.0f; coef[ldc + c] = 0.0f; coef[2 * ldc + c] = 0.0f; coef[3 * ldc + c] = 0.0f;
  }
  if (train && renorm) {
    __shared__ int s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      s_last = atomicAdd(steps + 1, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (s_last && threadIdx.x == 0) { steps[0] += 1; steps[1] = 0; }
  }
}

// xn[r][c] = (x[src(r)][c] - m_c) a_c + b_c (NORM; coef = [4][xn.ld] of bm_instat_finish_kernel) or the gathered x itself, as
// planes [m][ld]; xhat[r][c] = (x - m_c) k_c in f32 for the gradient rows r < nb (xhat != NULL).  One thread per (row, 8
// columns); columns >= d zero.  All loads of a thread are independent (8 scalar x -- rows of odd length are not 16-B
// aligned -- and 8 vector coefficient loads): one memory round trip.  The first version branched on `coef` per element
// and the compiler serialised the eight (x, 4 coefficients) groups: 16 / 35 us at 1024 / 2048 rows of 1345.
template <bool NORM>
__global__ __launch_bounds__(256) void bm_innorm_apply_kernel(const float *__restrict__ x, long long ldx, const int64_t *idx,
                                                              int nb, long long next_off, int m, int d,
                                                              const float *__restrict__ coef, BmPlanesOut xn,
                                                              float *__restrict__ xhat, long long ldh) {
  const unsigned ppr = (unsigned)(xn.ld / 8);
  const unsigned e = blockIdx.x * 256u + threadIdx.x;
  if (e >= (unsigned)m * ppr) return;
  const unsigned ru = e / ppr;
  const int r = (int)ru, c0 = (int)(e - ru * ppr) * 8;
  const float *xr = x + bm_src_row(idx, nb, next_off, r) * ldx;
  float v[8], o[8], oh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = xr[min(c0 + j, d - 1)];
  if (NORM) {
    const long long ldc = xn.ld;
    f32x4 cm[2], ca[2], cb[2], ck[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      cm[hh] = *reinterpret_cast<const f32x4 *>(coef + c0 + 4 * hh);
      ca[hh] = *reinterpret_cast<const f32x4 *>(coef + ldc + c0 + 4 * hh);
      cb[hh] = *reinterpret_cast<const f32x4 *>(coef + 2 * ldc + c0 + 4 * hh);
      ck[hh] = *reinterpret_cast<const f32x4 *>(coef + 3 * ldc + c0 + 4 * hh);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xc = v[j] - cm[j >> 2][j & 3];
      const float y = xc * ca[j >> 2][j & 3] + cb[j >> 2][j & 3];
      const float yh = xc * ck[j >> 2][j & 3];
      o[j] = c0 + j < d ? y : 0.0f;
      oh[j] = c0 + j < d ? yh : 0.0f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = c0 + j < d ? v[j] : 0.0f; oh[j] = 0.0f; }
  }
  u32x4 h, mm, l;
  bm_split8(o, h, mm, l);
  bm_put(xn, r, c0, h, mm, l);
  if (NORM && xhat && r < nb) {   // ldh % 4 == 0: whole 16-B stores
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
      if (c0 + 4 * hh < ldh)
        *reinterpret_cast<f32x4 *>(xhat + (long long)r * ldh + c0 + 4 * hh) = f32x4{oh[4 * hh], oh[4 * hh + 1], oh[4 * hh + 2], oh[4 * hh + 3]};
  }
}

// TD loss.  Rows r < b of Q are the q-values of transition idx[r]; with next_rows the rows b + r hold Q(next_obs) and
// target = reward + (1 - done) gamma max_a Q_next (pqn_craftax.py:300-306), otherwise `target` is the given Q(lambda)
// target.  dQ[r][a] = [a == action] (q_a - target) / b as planes [b][32].  a <= 32.
// BM_LOSS_WG one-wave workgroups, workgroup w = rows 64 w + lane (+ 1024 per pass); each leaves a record in `part`:
//   part[w * a + k]            sum over its rows of dQ[.][k]           -> d b_out        (folded by bm_colreduce, fixed order)
//   part[BM_LOSS_WG * 32 + w]  sum of 0.5 (q_a - target)^2 / b         -> the loss
//   part[BM_LOSS_WG * 33 + w]  sum of q_a / b                          -> mean chosen Q
// (A single 1024-thread workgroup did all of it in round 3: 16 us on the critical chain of the backward pass -- 196 KB of
// plane stores from one CU and 17 six-step lane butterflies per wave for the column sums.)
#define BM_LOSS_WG 16
__global__ __launch_bounds__(64) void bm_loss_kernel(const float *__restrict__ q, int ldq, int b, int a,
                                                     const int64_t *__restrict__ idx, const int32_t *__restrict__ action,
                                                     const float *__restrict__ target, const float *__restrict__ reward,
                                                     const uint8_t *__restrict__ done, float gamma, int next_rows,
                                                     BmPlanesOut dq, float *__restrict__ part) {
  __shared__ __attribute__((aligned(16))) float s_g[64];
  __shared__ __attribute__((aligned(16))) int s_a[64];
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x, w = blockIdx.x;
  float lsum = 0.f, qsum = 0.f, bacc = 0.f;   // bacc: lane k accumulates this workgroup's column sum for action k
  const float inv_b = 1.0f / (float)b;
  for (int r0 = 0; r0 < b; r0 += 64 * BM_LOSS_WG) {
    const int r = r0 + 64 * w + lane;
    float g = 0.f;
    int act = -1;
    if (r < b) {
      // the Q rows are loaded whole, as vectors, before anything that depends on idx / action arrives
      float qr[32], qx[32];
      bm_load_qrow(q + (long long)r * ldq, ldq, qr);
      if (next_rows) bm_load_qrow(q + (long long)(b + r) * ldq, ldq, qx);
      const long long j = idx ? idx[r] : r;
      act = action[j];
      float qa = qr[0];
#pragma unroll
      for (int k = 1; k < 32; ++k) qa = k == act ? qr[k] : qa;
      float tgt;
      if (next_rows) {
        float mx = qx[0];
#pragma unroll
        for (int k = 1; k < 32; ++k) mx = k < a ? fmaxf(mx, qx[k]) : mx;
        tgt = reward[j] + (1.0f - (float)done[j]) * gamma * mx;
      } else {
        tgt = target[j];
      }
      const float diff = qa - tgt;
      g = diff * inv_b;
      lsum += 0.5f * diff * diff;
      qsum += qa;
#pragma unroll
      for (int k8 = 0; k8 < 4; ++k8) {   // the row of dQ: one non-zero among 32 padded columns
        float v[8];
#pragma unroll
        for (int j8 = 0; j8 < 8; ++j8) v[j8] = (8 * k8 + j8 == act) ? g : 0.0f;
        u32x4 h, m, l;
        bm_split8(v, h, m, l);
        bm_put(dq, r, 8 * k8, h, m, l);
      }
    }
    s_g[lane] = g;
    s_a[lane] = act;
    __syncthreads();
    if (lane < a) {   // column sum of action `lane` over the 64 rows, in row order
      float t = 0.f;
#pragma unroll
      for (int j4 = 0; j4 < 16; ++j4) {
        const f32x4 gv = *reinterpret_cast<const f32x4 *>(&s_g[4 * j4]);
        const i32x4 av = *reinterpret_cast<const i32x4 *>(&s_a[4 * j4]);
        t += av.x == lane ? gv.x : 0.f;
        t += av.y == lane ? gv.y : 0.f;
        t += av.z == lane ? gv.z : 0.f;
        t += av.w == lane ? gv.w : 0.f;
      }
      bacc += t;
    }
    __syncthreads();
  }
  lsum = bm_wave_sum(lsum);
  qsum = bm_wave_sum(qsum);
  if (lane < a) part[w * a + lane] = bacc;
  if (lane == 0) { part[BM_LOSS_WG * 32 + w] = lsum * inv_b; part[BM_LOSS_WG * 33 + w] = qsum * inv_b; }
}

// relu mask + LayerNorm backward: dz (planes, rows x n) <- f(sum of the dpart K-split partials), one wave per row,
// BM_LB_ROWS rows per workgroup.
//   y = xhat g + beta;  dy = d [y > 0];  dxh = dy g;  dz = rstd (dxh - mean(dxh) - xhat mean(dxh xhat))
// part[wg][0] = sum_rows dy xhat (d scale), [1] = sum_rows dy (d LN bias), [2] = sum_rows dz (d dense bias).  n <= 2048, n % 256 == 0.
#define BM_LB_ROWS 4   // one row per wave: 256 workgroups at 1024 gradient rows
template <int NS, int KP>   // the K-split count of dpart and the packs per lane: straight-line loads, see bm_ln_relu_kernel
__global__ __launch_bounds__(256) void bm_ln_bwd_kernel(const float *__restrict__ dpart, long long pstride,
                                                        BmPlanesOut dzp, const float *__restrict__ z,
                                                        const float *__restrict__ stat, const float *__restrict__ g,
                                                        const float *__restrict__ beta, int rows, int n,
                                                        float *__restrict__ part) {
  __shared__ float s_red[4][3 * 1024];   // 1024 columns per pass of the cross-wave fold
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int np = n / 8;                  // packs per row: lane owns packs lane, lane + 64, ...
  f32x4 a0[KP][2], a1[KP][2], a2[KP][2];
#pragma unroll
  for (int k = 0; k < KP; ++k)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) { a0[k][hh] = f32x4{0.f, 0.f, 0.f, 0.f}; a1[k][hh] = a0[k][hh]; a2[k][hh] = a0[k][hh]; }
  for (int rr = wave; rr < BM_LB_ROWS; rr += 4) {
    const int row = blockIdx.x * BM_LB_ROWS + rr;
    if (row >= rows) break;
    const float mean = stat[2 * row], rstd = stat[2 * row + 1];
    const float *zr = z + (long long)row * n;
    f32x4 xh[KP][2], dxh[KP][2], dy[KP][2];
    f32x4 zv[KP][2], dv[KP][2], gv[KP][2], bv[KP][2];
#pragma unroll
    for (int k = 0; k < KP; ++k) {   // every load of the row in flight at once (packs beyond the row: its last pack again)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int c = 8 * min(lane + 64 * k, np - 1) + 4 * hh;
        zv[k][hh] = *reinterpret_cast<const f32x4 *>(zr + c);
        f32x4 d = *reinterpret_cast<const f32x4 *>(dpart + (long long)row * n + c);   // d loss / d h: sum of the K-split partials
#pragma unroll
        for (int sp = 1; sp < NS; ++sp) d += *reinterpret_cast<const f32x4 *>(dpart + sp * pstride + (long long)row * n + c);
        dv[k][hh] = d;
        gv[k][hh] = *reinterpret_cast<const f32x4 *>(g + c);
        bv[k][hh] = *reinterpret_cast<const f32x4 *>(beta + c);
      }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int pk = lane + 64 * k;
      if (pk < np) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          xh[k][hh] = (zv[k][hh] - mean) * rstd;
          const f32x4 y = xh[k][hh] * gv[k][hh] + bv[k][hh];
          const f32x4 d = dv[k][hh];
          dy[k][hh] = f32x4{y.x > 0.f ? d.x : 0.f, y.y > 0.f ? d.y : 0.f, y.z > 0.f ? d.z : 0.f, y.w > 0.f ? d.w : 0.f};
          dxh[k][hh] = dy[k][hh] * gv[k][hh];
          s1 += (dxh[k][hh].x + dxh[k][hh].y) + (dxh[k][hh].z + dxh[k][hh].w);
          const f32x4 t = dxh[k][hh] * xh[k][hh];
          s2 += (t.x + t.y) + (t.z + t.w);
        }
      }
    }
    s1 = bm_wave_sum(s1) / (float)n;
    s2 = bm_wave_sum(s2) / (float)n;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int pk = lane + 64 * k;
      if (pk < np) {
        float dzv[8];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const f32x4 dz = (dxh[k][hh] - s1 - xh[k][hh] * s2) * rstd;
          dzv[4 * hh] = dz.x; dzv[4 * hh + 1] = dz.y; dzv[4 * hh + 2] = dz.z; dzv[4 * hh + 3] = dz.w;
          a0[k][hh] += dy[k][hh] * xh[k][hh];
          a1[k][hh] += dy[k][hh];
          a2[k][hh] += dz;
        }
        u32x4 ph, pm, pl;
        bm_split8(dzv, ph, pm, pl);
        bm_put(dzp, row, 8 * pk, ph, pm, pl);
      }
    }
  }
  // fold the four waves (fixed order) and write this workgroup's record
  for (int base = 0; base < n; base += 1024) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int pk = lane + 64 * k;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int c = 8 * pk + 4 * hh;
        if (pk < np && c >= base && c < base + 1024) {
          *reinterpret_cast<f32x4 *>(&s_red[wave][c - base]) = a0[k][hh];
          *reinterpret_cast<f32x4 *>(&s_red[wave][1024 + c - base]) = a1[k][hh];
          *reinterpret_cast<f32x4 *>(&s_red[wave][2048 + c - base]) = a2[k][hh];
        }
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
PQN_D void bm_colreduce_body(const float *__restrict__ part, int nparts, int nseg, int n, float *__restrict__ out0,
                             float *__restrict__ out1, float *__restrict__ out2, int block, float (*s_r)[64]) {
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = block * 64 + lane;
  const bool ok = e < nseg * n;
  const int ee = ok ? e : 0;
  const int k = ee / n, c = ee - k * n;
  float s = 0.f;
  for (int p0 = grp; p0 < nparts; p0 += 16 * 8) {   // 8 loads in flight; same order of additions
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = part[((long long)min(p0 + 16 * u, nparts - 1) * nseg + k) * n + c];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (p0 + 16 * u < nparts) s += t[u];
  }
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
__global__ __launch_bounds__(1024) void bm_colreduce_kernel(const float *__restrict__ part, int nparts, int nseg, int n,
                                                            float *__restrict__ out0, float *__restrict__ out1,
                                                            float *__restrict__ out2) {
  __shared__ float s_r[16][64];
  bm_colreduce_body(part, nparts, nseg, n, out0, out1, out2, blockIdx.x, s_r);
}
// the column sums of every layer in one launch (job j owns blocks [first[j], first[j + 1])): same body, same sums
struct BmColreduceJobs {
  int n;
  const float *part[BM_MAX_JOBS];
  int nparts[BM_MAX_JOBS], nseg[BM_MAX_JOBS], ncol[BM_MAX_JOBS];
  float *out[BM_MAX_JOBS][3];
  int first[BM_MAX_JOBS + 1];
};
__global__ __launch_bounds__(1024) void bm_colreduce_multi_kernel(BmColreduceJobs J) {
  __shared__ float s_r[16][64];
  int j = 0;
  while (j + 1 < J.n && (int)blockIdx.x >= J.first[j + 1]) ++j;
  bm_colreduce_body(J.part[j], J.nparts[j], J.nseg[j], J.ncol[j], J.out[j][0], J.out[j][1], J.out[j][2], (int)blockIdx.x - J.first[j], s_r);
}

// out[i] = sum of the nsplit K-split partials of a weight-gradient GEMM (i < n, n % 4 == 0, everything 16-B aligned)
PQN_D void bm_sum_partials_body(const float *__restrict__ part, int nsplit, long long pstride, long long n, float *__restrict__ out,
                                long long block) {
  const long long i = (block * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 v = *reinterpret_cast<const f32x4 *>(part + i);
  for (int sp = 1; sp < nsplit; ++sp) v += *reinterpret_cast<const f32x4 *>(part + sp * pstride + i);
  *reinterpret_cast<f32x4 *>(out + i) = v;
}
__global__ __launch_bounds__(256) void bm_sum_partials_kernel(const float *__restrict__ part, int nsplit, long long pstride,
                                                              long long n, float *__restrict__ out) {
  bm_sum_partials_body(part, nsplit, pstride, n, out, blockIdx.x);
}
struct BmSumJobs {
  int n;
  const float *part[BM_MAX_JOBS];
  int nsplit[BM_MAX_JOBS];
  long long pstride[BM_MAX_JOBS], cnt[BM_MAX_JOBS];
  float *out[BM_MAX_JOBS];
  long long first[BM_MAX_JOBS + 1];
};
__global__ __launch_bounds__(256) void bm_sum_partials_multi_kernel(BmSumJobs J) {
  int j = 0;
  while (j + 1 < J.n && (long long)blockIdx.x >= J.first[j + 1]) ++j;
  bm_sum_partials_body(J.part[j], J.nsplit[j], J.pstride[j], J.cnt[j], J.out[j], (long long)blockIdx.x - J.first[j]);
}

// Q = sum of the K-split partials of the output layer's GEMM + bias, [rows][ldq] (ldq % 4 == 0; columns >= a stay as the
// GEMM left them).  The output layer is 17 columns wide: unsplit it is 16 (32) workgroups walking all of K (24 us for
// 0.07 GFLOP), split 8 ways it is a 6-us launch plus this fold.  All loads of a thread in flight at once.
#define BM_HEAD_SPLIT 8
__global__ __launch_bounds__(256) void bm_qfold_kernel(const float *__restrict__ part, int nsplit, long long pstride, long long n4,
                                                       int ldq, const float *__restrict__ bias, int a, float *__restrict__ q) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 t[BM_HEAD_SPLIT];
#pragma unroll
  for (int sp = 0; sp < BM_HEAD_SPLIT; ++sp) t[sp] = *reinterpret_cast<const f32x4 *>(part + min(sp, nsplit - 1) * pstride + 4 * i);
  f32x4 v = t[0];
#pragma unroll
  for (int sp = 1; sp < BM_HEAD_SPLIT; ++sp)
    if (sp < nsplit) v += t[sp];
  const int c = (int)((4 * i) % ldq);
  float bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bb[j] = bias[min(c + j, a - 1)];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (c + j < a) v[j] += bb[j];
  *reinterpret_cast<f32x4 *>(q + 4 * i) = v;
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
  float qi[32];
  bm_load_qrow(q + (long long)i * ldq, ldq, qi);   // ldq % 4 == 0: the whole row in 8 vector loads
  int best = 0;
  float bv = qi[0];
#pragma unroll
  for (int j = 1; j < 32; ++j) {
    const bool up = j < a && qi[j] > bv;
    bv = up ? qi[j] : bv;
    best = up ? j : best;
  }
  if (q_out) {
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j < a) q_out[(long long)i * a + j] = qi[j];
  }
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
int align4(int x) { return (x + 3) & ~3; }

// a plane triple inside a bf16 arena: rows x ld elements per plane
BmPlanes bm_pl(const bf16_t *base, int rows, int ld) {
  BmPlanes p = {};
  p.p = base; p.ld = ld; p.pstride = (long long)bm_pad16(rows) * ld; p.rows = rows;
  return p;
}
BmPlanesOut bm_plo(bf16_t *base, int rows, int ld) {
  BmPlanesOut p = {};
  p.p = base; p.ld = ld; p.pstride = (long long)bm_pad16(rows) * ld;
  return p;
}
long long bm_pl_elems(int rows, int ld) { return 3ll * bm_pad16(rows) * ld; }

BmEpilogue bm_store(float *out, long long ldc, const float *bias = nullptr, long long split_stride = 0) {
  BmEpilogue e = {};
  e.out = out; e.ldc = ldc; e.bias = bias; e.split_stride = split_stride;
  return e;
}

// Tile and K split of a GEMM: 128 x 64 tiles when they give more than every second CU a workgroup (else 64 x 64), and the K
// split (<= max_split, each part >= 128 long) that fills whole "rounds" best -- a round = every CU holding as many
// workgroups as its LDS takes (two 128 x 64, three 64 x 64).  Run-time switches "bm_tile" (64 / 128) and "bm_split"
// override (A/B runs).
struct BmPlan { int bm, nsplit, klen; };
BmPlan bm_plan(int M, int N, int Kp, int max_split) {
  const long long t128 = (long long)((M + 127) / 128) * ((N + 63) / 64);
  BmPlan p;
  // 128 x 64 tiles only when they give MORE than 128 workgroups: at exactly 128 (1024 x 1024 outputs, C5's rollout forward,
  // input gradients and weight gradients) the 64 x 64 tile with 3 K splits = one round of 768 workgroups measured faster
  // inside the update than 128 x 64 with 4 splits (0.777 vs 0.809 ms per C5 update, profiles/r04_v7_c5_batched_backward.txt).
  const int topt = pqn_opt(PQN_OPT_BM_TILE);   // 64 / 128: forced tile height; > 128: the 128-row tile from that many tiles up
  p.bm = t128 >= (topt > 128 ? topt : 129) ? 128 : 64;
  if (topt == 64 || topt == 128) p.bm = topt;
  const long long tiles = (long long)((M + p.bm - 1) / p.bm) * ((N + 63) / 64);
  const long long round = 256ll * (p.bm == 128 ? 2 : 3);
  int s = 1;
  double best = 0.0;
  for (int c = 1; c <= max_split; ++c) {
    const double r = (double)(tiles * c) / (double)round, eff = r / (double)(long long)(r + 0.999999);
    if (eff > best + 0.02 || (r <= 1.0 && eff > best)) { best = eff; s = c; }   // below one round every extra split helps
  }
  if (pqn_opt(PQN_OPT_BM_SPLIT) > 0) s = pqn_opt(PQN_OPT_BM_SPLIT);
  s = max(1, min(s, min(max_split, max(1, Kp / 128))));
  p.klen = ((Kp + s - 1) / s + BM_KS - 1) / BM_KS * BM_KS;
  p.nsplit = (Kp + p.klen - 1) / p.klen;
  return p;
}

template <int EPI>
int bm_launch(int M, int N, int Kp, const BmPlan &p, const BmPlanes &A, const BmPlanes &B, const BmEpilogue &E, hipStream_t st) {
  const dim3 grid((N + 63) / 64, (M + p.bm - 1) / p.bm, p.nsplit);
#define BM_GO(BM_)                                                                                                      \
  do {                                                                                                                   \
    auto kern = &bm_gemm_kernel<BM_, 64, EPI>;                                                                           \
    constexpr int lds = bm_lds_bytes<BM_, 64>();                                                                        \
    static pqn_once_per_device attr;                                                                                            \
    if (attr.first()) {                                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);  \
    }                                                                                                                    \
    hipLaunchKernelGGL(kern, grid, dim3(BM_THREADS), lds, st, M, N, Kp, p.klen, A, B, E);                                \
  } while (0)
  const bool timed = pqn_prof_begin(2, st);
  if (p.bm == 128) BM_GO(128);
  else BM_GO(64);
  if (timed) pqn_prof_end(st);
#undef BM_GO
  return pqn_check_launch("pqn_bigmlp gemm");
}

// C = A B^T from planes with K padded to Kp; returns the number of K-split partials written through *nsplit_out
int bm_gemm(int M, int N, int Kp, const BmPlanes &A, const BmPlanes &B, const BmEpilogue &E, int max_split, int *nsplit_out,
            hipStream_t st) {
  const BmPlan p = bm_plan(M, N, Kp, max_split);
  if (nsplit_out) *nsplit_out = p.nsplit;
  return bm_launch<BM_EPI_STORE>(M, N, Kp, p, A, B, E, st);
}

void bm_split(const float *src, long long lds, int rows, int cols, const BmPlanesOut &out, hipStream_t st) {
  const long long n = (long long)rows * (out.ld / 8);
  hipLaunchKernelGGL(bm_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, lds, rows, cols, out);
}
// dst[cols][ld_d] <- transpose of src[rows][.] (logical columns `cols`); ld_d = pad32(rows)
void bm_transpose(const BmPlanes &src, int cols, const BmPlanesOut &dst, hipStream_t st) {
  hipLaunchKernelGGL(bm_transpose_kernel, dim3((unsigned)((dst.ld + 63) / 64), (unsigned)((cols + 63) / 64), 3), dim3(256), 0, st, src, cols,
                     dst);
}

struct BmTransposeBatch {
  BmTransposeJobs J = {};
  void add(const BmPlanes &src, int cols, const BmPlanesOut &dst) {
    const int j = J.n++;
    J.src[j] = src; J.cols[j] = cols; J.dst[j] = dst;
    J.nbx[j] = (int)((dst.ld + 63) / 64);
    J.first[j + 1] = J.first[j] + J.nbx[j] * ((cols + 63) / 64);
  }
  void launch(hipStream_t st) {
    if (J.n) hipLaunchKernelGGL(bm_transpose_multi_kernel, dim3((unsigned)J.first[J.n], 1, 3), dim3(256), 0, st, J);
  }
};

// ---- workspace carve-up: a float region followed by a bf16 region (offsets in floats / in bf16 elements) ----
struct BmWs {
  // f32
  long long coef, cspart, xhat, z[PQN_BIGMLP_MAX_LAYERS], stat[PQN_BIGMLP_MAX_LAYERS], q, qpart, losspart, zpart, dpart, wpart[PQN_BIGMLP_MAX_LAYERS],
      wpart_out, lnpart[PQN_BIGMLP_MAX_LAYERS], inpart, f_total;
  long long zstride, dstride, wstride, qstride, wostride;
  // bf16 planes (element offsets from the start of the bf16 region)
  long long xn, xnT, h[PQN_BIGMLP_MAX_LAYERS], hT[PQN_BIGMLP_MAX_LAYERS], dz[PQN_BIGMLP_MAX_LAYERS], dzT[PQN_BIGMLP_MAX_LAYERS], dq, dqT, b_total;
  int ldq, ldx, dp, nbp, n_cs, n_ln, n_in;
};
BmWs bm_ws(const pqn_bigmlp_layout_t &L, int rows, int nb) {
  BmWs w = {};
  long long off = 0;
  auto take = [&](long long n) { const long long o = off; off += (n + 3) & ~3ll; return o; };
  w.ldq = align4(L.a);
  w.ldx = align4(L.d);
  w.dp = bm_pad32(L.d);
  w.nbp = bm_pad32(nb);
  w.n_cs = (rows + BM_CS_ROWS - 1) / BM_CS_ROWS;
  w.n_ln = (nb + BM_LB_ROWS - 1) / BM_LB_ROWS;
  w.n_in = (nb + 63) / 64;
  w.coef = take(4ll * w.dp);   // [4][dp], zero beyond d
  w.cspart = take(4ll * w.n_cs * L.d);   // f64 partials
  w.xhat = take((long long)nb * w.ldx);
  for (int l = 0; l < L.layers; ++l) {
    w.z[l] = take((long long)rows * L.h);
    w.stat[l] = take(2ll * rows);
  }
  w.q = take((long long)rows * w.ldq);
  w.qstride = (long long)rows * w.ldq;
  w.qpart = take(BM_HEAD_SPLIT * w.qstride);   // K-split partials of the output layer
  w.losspart = take(BM_LOSS_WG * 34);          // records of bm_loss_kernel
  w.zstride = (long long)rows * L.h;
  w.zpart = take(BM_MAX_SPLIT * w.zstride);
  w.dstride = (long long)nb * L.h;
  w.dpart = take(BM_MAX_SPLIT * w.dstride);
  w.wstride = (long long)align4(max(L.d, L.h)) * L.h;
  // per layer: the weight-gradient side of layer l (column sums, transposes, d W GEMM, fold) runs on a second stream
  // beside the input-gradient chain of the layers below it, so none of its buffers may be reused by them
  for (int l = 0; l < L.layers; ++l) {
    w.wpart[l] = take(BM_MAX_SPLIT * w.wstride);
    w.lnpart[l] = take(3ll * w.n_ln * L.h);
  }
  w.wostride = (long long)align4(L.h * L.a);
  w.wpart_out = take(BM_HEAD_SPLIT * w.wostride);   // K-split partials of the output layer's weight gradient
  w.inpart = take(2ll * BM_MAX_SPLIT * w.n_in * L.d);
  w.f_total = off;
  long long bo = 0;
  auto takeb = [&](long long n) { const long long o = bo; bo += (n + 7) & ~7ll; return o; };
  w.xn = takeb(bm_pl_elems(rows, w.dp));
  w.xnT = takeb(bm_pl_elems(L.d, w.nbp));
  for (int l = 0; l < L.layers; ++l) {
    w.h[l] = takeb(bm_pl_elems(rows, L.h));
    w.hT[l] = takeb(bm_pl_elems(L.h, w.nbp));
  }
  for (int l = 0; l < L.layers; ++l) {
    w.dz[l] = takeb(bm_pl_elems(nb, L.h));
    w.dzT[l] = takeb(bm_pl_elems(L.h, w.nbp));
  }
  w.dq = takeb(bm_pl_elems(nb, 32));
  w.dqT = takeb(bm_pl_elems(L.a, w.nbp));
  w.b_total = bo;
  return w;
}
long long bm_ws_floats(const BmWs &w) { return w.f_total + (w.b_total + 1) / 2; }

// weight planes (bf16 element offsets inside the arena): per layer the natural copy Wn[kin][pad32(out)] (input gradient:
// K = out) and the transposed copy WT[out][pad32(kin)] (forward: K = kin)
struct BmWp { long long wn[PQN_BIGMLP_MAX_LAYERS + 1], wt[PQN_BIGMLP_MAX_LAYERS + 1], total; };
BmWp bm_wp(const pqn_bigmlp_layout_t &L) {
  BmWp w = {};
  long long bo = 0;
  auto takeb = [&](long long n) { const long long o = bo; bo += (n + 7) & ~7ll; return o; };
  for (int l = 0; l <= L.layers; ++l) {
    const int kin = l ? L.h : L.d, out = l < L.layers ? L.h : L.a;
    w.wn[l] = takeb(bm_pl_elems(kin, bm_pad32(out)));
    w.wt[l] = takeb(bm_pl_elems(out, bm_pad32(kin)));
  }
  w.total = bo;
  return w;
}

// (K-split count, packs per lane) -> template instance
#define BM_LN_CASES(KERN, NS_)                                                                              \
  switch (kp) {                                                                                             \
    case 1: hipLaunchKernelGGL((KERN<NS_, 1>), dim3(blocks), dim3(256), 0, st, args...); break;             \
    case 2: hipLaunchKernelGGL((KERN<NS_, 2>), dim3(blocks), dim3(256), 0, st, args...); break;             \
    case 3: hipLaunchKernelGGL((KERN<NS_, 3>), dim3(blocks), dim3(256), 0, st, args...); break;             \
    default: hipLaunchKernelGGL((KERN<NS_, 4>), dim3(blocks), dim3(256), 0, st, args...); break;            \
  }
template <typename... Args>
void bm_ln_relu(int ns, int n, int blocks, hipStream_t st, Args... args) {
  const int kp = (n / 8 + 63) / 64;
  switch (ns) {
    case 1: BM_LN_CASES(bm_ln_relu_kernel, 1) break;
    case 2: BM_LN_CASES(bm_ln_relu_kernel, 2) break;
    case 3: BM_LN_CASES(bm_ln_relu_kernel, 3) break;
    default: BM_LN_CASES(bm_ln_relu_kernel, 4) break;
  }
}
template <typename... Args>
void bm_ln_bwd(int ns, int n, int blocks, hipStream_t st, Args... args) {
  const int kp = (n / 8 + 63) / 64;
  switch (ns) {
    case 1: BM_LN_CASES(bm_ln_bwd_kernel, 1) break;
    case 2: BM_LN_CASES(bm_ln_bwd_kernel, 2) break;
    case 3: BM_LN_CASES(bm_ln_bwd_kernel, 3) break;
    default: BM_LN_CASES(bm_ln_bwd_kernel, 4) break;
  }
}
#undef BM_LN_CASES
template <typename... Args>
void bm_innorm_apply(bool norm, unsigned blocks, hipStream_t st, Args... args) {
  if (norm) hipLaunchKernelGGL(bm_innorm_apply_kernel<true>, dim3(blocks), dim3(256), 0, st, args...);
  else hipLaunchKernelGGL(bm_innorm_apply_kernel<false>, dim3(blocks), dim3(256), 0, st, args...);
}

// forward from the (normalised, gathered) input planes through the hidden layers (z_l, h_l planes, stat_l kept) to Q
int bm_forward(const pqn_bigmlp_layout_t &L, int rows, const float *theta, const bf16_t *wpl, float *ws, bf16_t *wb, const BmWs &w,
               hipStream_t st) {
  const BmWp wp = bm_wp(L);
  for (int l = 0; l < L.layers; ++l) {
    const int kin = l ? L.h : L.d, kp = bm_pad32(kin);
    const BmPlanes A = l ? bm_pl(wb + w.h[l - 1], rows, L.h) : bm_pl(wb + w.xn, rows, w.dp);
    const BmPlanes B = bm_pl(wpl + wp.wt[l], L.h, kp);
    int ns = 1;
    const int rc = bm_gemm(rows, L.h, kp, A, B, bm_store(ws + w.zpart, L.h, nullptr, w.zstride), BM_MAX_SPLIT, &ns, st);
    if (rc != PQN_OK) return rc;
    bm_ln_relu(ns, L.h, (rows + 3) / 4, st, (const float *)(ws + w.zpart), w.zstride, rows, L.h, theta + L.off_b[l], theta + L.off_lns[l],
               theta + L.off_lnb[l], ws + w.z[l], bm_plo(wb + w.h[l], rows, L.h), ws + w.stat[l]);
  }
  const int lo = L.layers;   // output layer: Q = h_last W_out + b_out, K split + fold (see bm_qfold_kernel)
  int nsq = 1;
  const int rc = bm_gemm(rows, L.a, L.h, bm_pl(wb + w.h[lo - 1], rows, L.h), bm_pl(wpl + wp.wt[lo], L.a, L.h),
                         bm_store(ws + w.qpart, w.ldq, nullptr, w.qstride), BM_HEAD_SPLIT, &nsq, st);
  if (rc != PQN_OK) return rc;
  const long long n4 = w.qstride / 4;
  hipLaunchKernelGGL(bm_qfold_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float *)(ws + w.qpart), nsq, w.qstride, n4,
                     w.ldq, theta + L.off_b[lo], L.a, ws + w.q);
  return pqn_check_launch("pqn_bigmlp forward");
}

// ---- second stream of the backward pass ----------------------------------------------------------------------------------
// The input-gradient chain (LayerNorm backward_l -> d h_{l-1} GEMM -> LayerNorm backward_{l-1} ...) is the critical path of
// the backward pass; everything that ends in a parameter gradient (column sums, plane transposes, the d W GEMMs, their
// K-split folds) hangs off it.  Those launches go to ONE process-wide side stream, ordered against the caller's stream by
// events (fork after the producer, one join at the end), so they fill the CUs the 1024-row GEMMs of the chain leave idle.
// Event record / wait pairs are capturable: inside a hipGraph capture the side stream joins the capture at the first wait
// and leaves it at the join.  Stream and events are created on first use (the first, eager update); if that fails, or with
// the option bm_overlap = 0 -- the DEFAULT -- everything stays on the caller's stream.  Measured on C5 (profiles/
// r03_v2_c5_overlap_ab.txt): 1.149 ms per update with the side stream, 1.137 ms without; the overlapped kernels only slow
// each other down (GEMM 29 -> 34 us, LayerNorm backward 16 -> 22 us on average): the update is throughput-bound on the
// L2 -> LDS path, not latency-bound with idle CUs.  Kept as an option (bit-identical results, tested) for other shapes.
struct BmFork {
  hipStream_t side = nullptr;
  hipEvent_t ev[PQN_BIGMLP_MAX_LAYERS + 4];
  bool ok = false, tried = false;
};
BmFork g_bm_fork;
BmFork *bm_fork() {
  if (pqn_opt(PQN_OPT_BM_OVERLAP) <= 0) return nullptr;
  BmFork &f = g_bm_fork;
  if (!f.tried) {
    f.tried = true;
    bool ok = hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < PQN_BIGMLP_MAX_LAYERS + 4; ++i) ok = hipEventCreateWithFlags(&f.ev[i], hipEventDisableTiming) == hipSuccess;
    f.ok = ok;
    if (!ok) (void)hipGetLastError();
  }
  return f.ok ? &f : nullptr;
}
// the side stream, ordered behind everything enqueued on `st` so far (event slot i); `st` itself when there is no side stream
hipStream_t bm_fork_at(BmFork *f, int i, hipStream_t st) {
  if (!f) return st;
  (void)hipEventRecord(f->ev[i], st);
  (void)hipStreamWaitEvent(f->side, f->ev[i], 0);
  return f->side;
}
void bm_join(BmFork *f, int i, hipStream_t st) {
  if (!f) return;
  (void)hipEventRecord(f->ev[i], f->side);
  (void)hipStreamWaitEvent(st, f->ev[i], 0);
}

}  // namespace

extern "C" int pqn_bigmlp_layout(int32_t d, int32_t h, int32_t layers, int32_t a, int32_t norm_input,
                                 pqn_bigmlp_layout_t *L) {
  PQN_REQUIRE(L, "pqn_bigmlp_layout: NULL layout");
  PQN_REQUIRE(d >= 8 && h >= 256 && h <= 2048 && h % 256 == 0 && layers >= 1 && layers <= PQN_BIGMLP_MAX_LAYERS && a >= 1 &&
                  a <= 32 && norm_input >= 0 && norm_input <= 2,
              "pqn_bigmlp_layout: unsupported shape d=%d h=%d layers=%d a=%d norm_input=%d (d >= 8; h: multiple of 256 in "
              "[256, 2048]; layers <= %d; a <= 32)", d, h, layers, a, norm_input, PQN_BIGMLP_MAX_LAYERS);
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
  return bm_ws_floats(bm_ws(*L, rows, nb));
}

extern "C" int64_t pqn_bigmlp_weight_plane_floats(const pqn_bigmlp_layout_t *L) {
  if (!L) return -1;
  return (bm_wp(*L).total + 1) / 2;
}

// the two plane sets of every Dense kernel straight from theta: the transposed copy (forward operand) on `st`, the natural
// copy (input-gradient operand) on `st_nat` -- the same stream, or a second one the caller orders itself
static int bm_refresh(const pqn_bigmlp_layout_t *L, const float *theta, float *wplanes, hipStream_t st, hipStream_t st_nat) {
  const BmWp wp = bm_wp(*L);
  bf16_t *base = reinterpret_cast<bf16_t *>(wplanes);
  BmSplitJobs S = {};   // one launch per orientation for all layers
  BmSplitTJobs T = {};
  for (int l = 0; l <= L->layers; ++l) {
    const int kin = l ? L->h : L->d, out = l < L->layers ? L->h : L->a;
    const BmPlanesOut wn = bm_plo(base + wp.wn[l], kin, bm_pad32(out));
    S.src[l] = theta + L->off_w[l]; S.lds[l] = out; S.rows[l] = kin; S.cols[l] = out; S.out[l] = wn;
    S.first[l + 1] = S.first[l] + ((long long)kin * (wn.ld / 8) + 255) / 256;
    const BmPlanesOut wt = bm_plo(base + wp.wt[l], out, bm_pad32(kin));
    T.src[l] = theta + L->off_w[l]; T.lds[l] = out; T.rows[l] = kin; T.cols[l] = out; T.dst[l] = wt;
    T.nbx[l] = (int)((wt.ld + 63) / 64);
    T.first[l + 1] = T.first[l] + T.nbx[l] * ((out + 63) / 64);
  }
  S.n = T.n = L->layers + 1;
  hipLaunchKernelGGL(bm_split_transpose_multi_kernel, dim3((unsigned)T.first[T.n]), dim3(256), 0, st, T);
  hipLaunchKernelGGL(bm_split_multi_kernel, dim3((unsigned)S.first[S.n]), dim3(256), 0, st_nat, S);
  return pqn_check_launch("pqn_bigmlp_refresh_planes");
}

extern "C" int pqn_bigmlp_refresh_planes(const pqn_bigmlp_layout_t *L, const float *theta, float *wplanes, void *stream) {
  PQN_REQUIRE(L && theta && wplanes, "pqn_bigmlp_refresh_planes: NULL argument");
  return bm_refresh(L, theta, wplanes, (hipStream_t)stream, (hipStream_t)stream);
}

extern "C" int pqn_bigmlp_refresh_planes_streams(const pqn_bigmlp_layout_t *L, const float *theta, float *wplanes, void *stream,
                                                 void *gradient_stream) {
  PQN_REQUIRE(L && theta && wplanes, "pqn_bigmlp_refresh_planes_streams: NULL argument");
  return bm_refresh(L, theta, wplanes, (hipStream_t)stream, (hipStream_t)gradient_stream);
}

extern "C" int pqn_bigmlp_forward(const pqn_bigmlp_layout_t *L, int32_t n, const float *obs, const float *theta,
                                  const float *wplanes, float *in_mean, float *in_var, float *workspace, float *q,
                                  int32_t *action, float *qmax, float eps, uint64_t key, const float *eps_dev,
                                  const uint64_t *key_dev, void *stream) {
  PQN_REQUIRE(L && obs && theta && wplanes && workspace, "pqn_bigmlp_forward: NULL argument");
  PQN_REQUIRE(n > 0 && (q || action || qmax), "pqn_bigmlp_forward: nothing to do (n=%d)", n);
  PQN_REQUIRE(L->norm_input == 0 || (in_mean && in_var), "pqn_bigmlp_forward: the input normalisation needs its running moments");
  hipStream_t st = (hipStream_t)stream;
  const BmWs w = bm_ws(*L, n, n);
  float *ws = workspace;
  bf16_t *wb = reinterpret_cast<bf16_t *>(workspace + w.f_total);
  const bf16_t *wpl = reinterpret_cast<const bf16_t *>(wplanes);
  const float *coef = nullptr;
  if (L->norm_input) {   // use_running_average: coefficients from the running moments
    hipLaunchKernelGGL(bm_instat_finish_kernel, dim3((w.dp + 63) / 64), dim3(64), 0, st, (const double *)nullptr, 0, n, L->d,
                       theta + L->off_in_scale, theta + L->off_in_bias, in_mean, in_var, (int32_t *)nullptr, 0,
                       L->norm_input == 2 ? 1 : 0, L->norm_input == 2 ? 1e-3f : 1e-5f, 0.0f, ws + w.coef, w.dp);
    coef = ws + w.coef;
  }
  bm_innorm_apply(coef != nullptr, (unsigned)(((long long)n * (w.dp / 8) + 255) / 256), st, obs, (long long)L->d, (const int64_t *)nullptr,
                  n, 0ll, n, L->d, coef, bm_plo(wb + w.xn, n, w.dp), (float *)nullptr, (long long)w.ldx);
  const int rc = bm_forward(*L, n, theta, wpl, ws, wb, w, st);
  if (rc != PQN_OK) return rc;
  hipLaunchKernelGGL(bm_epsgreedy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws + w.q, w.ldq, n, L->a, eps, key, eps_dev,
                     key_dev, action, qmax, q);
  return pqn_check_launch("pqn_bigmlp_forward");
}

extern "C" int pqn_bigmlp_grad(const pqn_bigmlp_layout_t *L, int32_t nb, const int64_t *idx, const float *obs,
                               int64_t next_offset, const int32_t *action, const float *target, const float *reward,
                               const uint8_t *done, float gamma, const float *theta, const float *wplanes, float *in_mean,
                               float *in_var, int32_t *in_steps, float *grad, float *workspace, float *loss_out,
                               float *qv_out, void *stream) {
  PQN_REQUIRE(L && idx && obs && action && theta && wplanes && grad && workspace, "pqn_bigmlp_grad: NULL argument");
  PQN_REQUIRE(nb > 0 && next_offset >= 0, "pqn_bigmlp_grad: bad shape nb=%d next_offset=%lld", nb, (long long)next_offset);
  PQN_REQUIRE(next_offset > 0 ? (reward && done) : (target != nullptr),
              "pqn_bigmlp_grad: the 1-step loss needs reward + done, the Q(lambda) loss needs target");
  PQN_REQUIRE(L->norm_input == 0 || (in_mean && in_var && (L->norm_input == 1 || in_steps)),
              "pqn_bigmlp_grad: the input normalisation needs its running statistics");
  hipStream_t st = (hipStream_t)stream;
  const int rows = next_offset > 0 ? 2 * nb : nb;
  const BmWs w = bm_ws(*L, rows, nb);
  const BmWp wp = bm_wp(*L);
  float *ws = workspace;
  bf16_t *wb = reinterpret_cast<bf16_t *>(workspace + w.f_total);
  const bf16_t *wpl = reinterpret_cast<const bf16_t *>(wplanes);
  const float *coef = nullptr;
  if (L->norm_input) {
    const bool renorm = L->norm_input == 2;
    hipLaunchKernelGGL(bm_colstats_kernel, dim3((L->d + 63) / 64, w.n_cs), dim3(256), 0, st, obs, (long long)L->d, idx, nb,
                       (long long)next_offset, rows, L->d, reinterpret_cast<double *>(ws + w.cspart));
    hipLaunchKernelGGL(bm_instat_finish_kernel, dim3((w.dp + 63) / 64), dim3(64), 0, st,
                       reinterpret_cast<const double *>(ws + w.cspart), w.n_cs, rows, L->d, theta + L->off_in_scale,
                       theta + L->off_in_bias, in_mean, in_var, in_steps, 1, renorm ? 1 : 0, renorm ? 1e-3f : 1e-5f,
                       renorm ? 0.999f : 0.99f, ws + w.coef, w.dp);
    coef = ws + w.coef;
  } else {   // the dummy input normalisation never receives gradient (pqn_craftax.py:47-49)
    if (hipMemsetAsync(grad + L->off_in_scale, 0, sizeof(float) * (size_t)(L->off_w[0] - L->off_in_scale), st) != hipSuccess) {
      pqn_set_error("pqn_bigmlp_grad: hipMemsetAsync failed");
      return PQN_E_HIP;
    }
  }
  bm_innorm_apply(coef != nullptr, (unsigned)(((long long)rows * (w.dp / 8) + 255) / 256), st, obs, (long long)L->d, idx, nb,
                  (long long)next_offset, rows, L->d, coef, bm_plo(wb + w.xn, rows, w.dp), coef ? ws + w.xhat : (float *)nullptr,
                  (long long)w.ldx);
  int rc = bm_forward(*L, rows, theta, wpl, ws, wb, w, st);
  if (rc != PQN_OK) return rc;
  // backward over the first nb rows (the next_obs half carries no gradient: stop_gradient, pqn_craftax.py:301).
  // Weight gradient d W = Hin^T dZ: both operands with the samples as K = the transposed plane copies [feature][sample].
  // Two launch orders, same kernels / buffers / sums (bit-identical results, tested):
  //  * default: the input-gradient chain first (loss -> d h_last -> LayerNorm backward_l -> d h_{l-1} ...), then everything
  //    that ends in a parameter gradient in FOUR batched steps: one transpose launch (input, h_l, dQ, dz_l), one column-sum
  //    launch (LayerNorm / bias gradients of every layer, the input-normalisation gradient), the d W GEMMs, one fold launch
  //    of their K-split partials -- 3 small launches instead of 15 (profiles/r04_v7_c5_batched_backward.txt);
  //  * option bm_overlap = 1: layer by layer on the side stream of bm_fork beside the chain (see there).
  BmFork *fk = bm_fork();
  const bool deferred = fk == nullptr;
  BmTransposeBatch TB;
  BmColreduceJobs CJ = {};
  BmSumJobs SJ = {};
  struct PendingWgrad { int l, kin, n_out; BmPlanes HinT, dZT; float *gout; } pend[PQN_BIGMLP_MAX_LAYERS + 1];
  int npend = 0;
  auto fold = [&](const float *part, int ns, long long pstride, long long cnt, float *gout, hipStream_t sd) {
    if (deferred) {
      const int j = SJ.n++;
      SJ.part[j] = part; SJ.nsplit[j] = ns; SJ.pstride[j] = pstride; SJ.cnt[j] = cnt; SJ.out[j] = gout;
      SJ.first[j + 1] = SJ.first[j] + (cnt / 4 + 255) / 256;
    } else {
      hipLaunchKernelGGL(bm_sum_partials_kernel, dim3((unsigned)((cnt / 4 + 255) / 256)), dim3(256), 0, sd, part, ns, pstride, cnt, gout);
    }
  };
  auto wgrad_now = [&](int l, int kin, const BmPlanes &HinT, const BmPlanes &dZT, int n_out, float *gout, hipStream_t sd) -> int {
    int ns = 1;
    const long long cnt = (long long)kin * n_out;
    const bool direct = (cnt & 3) != 0;                  // (odd element count: one split, straight into the gradient)
    const bool head = l == L->layers;                    // 17 columns: few tiles, so up to 8 K splits (see bm_qfold_kernel)
    float *part = ws + (head ? w.wpart_out : w.wpart[l]);
    const long long pstride = head ? w.wostride : w.wstride;
    const int r = bm_gemm(kin, n_out, w.nbp, HinT, dZT, direct ? bm_store(gout, n_out) : bm_store(part, n_out, nullptr, pstride),
                          direct ? 1 : (head ? BM_HEAD_SPLIT : BM_MAX_SPLIT), &ns, sd);
    if (r != PQN_OK || direct) return r;
    fold(part, ns, pstride, cnt, gout, sd);
    return PQN_OK;
  };
  auto wgrad = [&](int l, int kin, const BmPlanes &HinT, const BmPlanes &dZT, int n_out, float *gout, hipStream_t sd) -> int {
    if (!deferred) return wgrad_now(l, kin, HinT, dZT, n_out, gout, sd);
    pend[npend++] = PendingWgrad{l, kin, n_out, HinT, dZT, gout};
    return PQN_OK;
  };
  auto transpose = [&](const BmPlanes &src, int cols, const BmPlanesOut &dst, hipStream_t sd) {
    if (deferred) TB.add(src, cols, dst);
    else bm_transpose(src, cols, dst, sd);
  };
  auto colreduce = [&](const float *part, int nparts, int nseg, int ncol, float *o0, float *o1, float *o2, hipStream_t sd) {
    if (deferred) {
      const int j = CJ.n++;
      CJ.part[j] = part; CJ.nparts[j] = nparts; CJ.nseg[j] = nseg; CJ.ncol[j] = ncol;
      CJ.out[j][0] = o0; CJ.out[j][1] = o1; CJ.out[j][2] = o2;
      CJ.first[j + 1] = CJ.first[j] + (nseg * ncol + 63) / 64;
    } else {
      hipLaunchKernelGGL(bm_colreduce_kernel, dim3((nseg * ncol + 63) / 64), dim3(1024), 0, sd, part, nparts, nseg, ncol, o0, o1, o2);
    }
  };
  // transposed copies of the gradient rows of every layer input (the normalised input, h_0 .. h_{L-1}): they depend on the
  // forward pass only (side stream: all of them at once, beside the loss kernel)
  hipStream_t sd = bm_fork_at(fk, 0, st);
  {
    BmTransposeBatch T;
    BmTransposeBatch &TT = deferred ? TB : T;
    BmPlanes src = bm_pl(wb + w.xn, nb, w.dp);
    src.pstride = (long long)bm_pad16(rows) * w.dp;   // the planes hold all forward rows; only the first nb are transposed
    TT.add(src, L->d, bm_plo(wb + w.xnT, L->d, w.nbp));
    for (int l = 0; l < L->layers; ++l) {
      BmPlanes sh = bm_pl(wb + w.h[l], nb, L->h);
      sh.pstride = (long long)bm_pad16(rows) * L->h;
      TT.add(sh, L->h, bm_plo(wb + w.hT[l], L->h, w.nbp));
    }
    T.launch(sd);
  }
  auto hT = [&](int l) -> BmPlanes { return l < 0 ? bm_pl(wb + w.xnT, L->d, w.nbp) : bm_pl(wb + w.hT[l], L->h, w.nbp); };
  const int lo = L->layers;
  hipLaunchKernelGGL(bm_loss_kernel, dim3(BM_LOSS_WG), dim3(64), 0, st, ws + w.q, w.ldq, nb, L->a, idx, action, target, reward, done,
                     gamma, next_offset > 0 ? 1 : 0, bm_plo(wb + w.dq, nb, 32), ws + w.losspart);
  // output layer: d W_out = h_last^T dQ (side);   d h_last = dQ W_out^T (K = a padded to 32: one split)
  sd = bm_fork_at(fk, 1, st);
  // d b_out, the loss and the mean chosen Q: folds of the loss kernel's records
  colreduce(ws + w.losspart, BM_LOSS_WG, 1, L->a, grad + L->off_b[lo], (float *)nullptr, (float *)nullptr, sd);
  if (loss_out) colreduce(ws + w.losspart + BM_LOSS_WG * 32, BM_LOSS_WG, 1, 1, loss_out, (float *)nullptr, (float *)nullptr, sd);
  if (qv_out) colreduce(ws + w.losspart + BM_LOSS_WG * 33, BM_LOSS_WG, 1, 1, qv_out, (float *)nullptr, (float *)nullptr, sd);
  transpose(bm_pl(wb + w.dq, nb, 32), L->a, bm_plo(wb + w.dqT, L->a, w.nbp), sd);
  int rc2 = wgrad(lo, L->h, hT(lo - 1), bm_pl(wb + w.dqT, L->a, w.nbp), L->a, grad + L->off_w[lo], sd);
  if (rc2 != PQN_OK) return rc2;
  int nsd = 1;
  rc = bm_gemm(nb, L->h, 32, bm_pl(wb + w.dq, nb, 32), bm_pl(wpl + wp.wn[lo], L->h, 32), bm_store(ws + w.dpart, L->h, nullptr, w.dstride), 1,
               &nsd, st);
  if (rc != PQN_OK) return rc;
  for (int l = lo - 1; l >= 0; --l) {
    // dpart (nsd K-split partials) = d loss / d h_l  ->  relu mask + LayerNorm backward: dz planes = d loss / d z_l
    bm_ln_bwd(nsd, L->h, w.n_ln, st, (const float *)(ws + w.dpart), w.dstride, bm_plo(wb + w.dz[l], nb, L->h), (const float *)(ws + w.z[l]),
              (const float *)(ws + w.stat[l]), theta + L->off_lns[l], theta + L->off_lnb[l], nb, L->h, ws + w.lnpart[l]);
    const int kin = l ? L->h : L->d;
    const BmPlanes dZ = bm_pl(wb + w.dz[l], nb, L->h);
    // side: LayerNorm / bias gradients, d W_l = h_{l-1}^T dZ_l
    sd = bm_fork_at(fk, 2 + l, st);
    colreduce(ws + w.lnpart[l], w.n_ln, 3, L->h, grad + L->off_lns[l], grad + L->off_lnb[l], grad + L->off_b[l], sd);
    transpose(dZ, L->h, bm_plo(wb + w.dzT[l], L->h, w.nbp), sd);
    rc = wgrad(l, kin, hT(l - 1), bm_pl(wb + w.dzT[l], L->h, w.nbp), L->h, grad + L->off_w[l], sd);
    if (rc != PQN_OK) return rc;
    const BmPlanes Wn = bm_pl(wpl + wp.wn[l], kin, L->h);   // rows = input feature, K = output feature
    if (l > 0) {   // d h_{l-1} = dZ_l W_l^T
      rc = bm_gemm(nb, kin, L->h, dZ, Wn, bm_store(ws + w.dpart, L->h, nullptr, w.dstride), BM_MAX_SPLIT, &nsd, st);
      if (rc != PQN_OK) return rc;
    } else if (coef) {
      // d (input-normalisation scale, bias): column sums of (dZ_0 W_0^T) xhat and of dZ_0 W_0^T over the nb rows
      BmEpilogue E = {};
      E.out = ws + w.inpart; E.xhat = ws + w.xhat; E.ldx = w.ldx;
      const BmPlan p = bm_plan(nb, kin, L->h, BM_MAX_SPLIT);
      rc = bm_launch<BM_EPI_INNORM>(nb, kin, L->h, p, dZ, Wn, E, st);
      if (rc != PQN_OK) return rc;
      colreduce(ws + w.inpart, p.nsplit * ((nb + p.bm - 1) / p.bm), 2, L->d, grad + L->off_in_scale, grad + L->off_in_bias,
                (float *)nullptr, st);
    }
  }
  if (deferred) {
    TB.launch(st);
    if (CJ.n) hipLaunchKernelGGL(bm_colreduce_multi_kernel, dim3((unsigned)CJ.first[CJ.n]), dim3(1024), 0, st, CJ);
    for (int i = 0; i < npend; ++i) {
      const PendingWgrad &g = pend[i];
      rc = wgrad_now(g.l, g.kin, g.HinT, g.dZT, g.n_out, g.gout, st);
      if (rc != PQN_OK) return rc;
    }
    if (SJ.n) hipLaunchKernelGGL(bm_sum_partials_multi_kernel, dim3((unsigned)SJ.first[SJ.n]), dim3(256), 0, st, SJ);
  }
  bm_join(fk, PQN_BIGMLP_MAX_LAYERS + 3, st);
  return pqn_check_launch("pqn_bigmlp_grad");
}

// where a forward intermediate lives in the workspace (tests / debugging).  f32 tensors (what 1 = pre-activation z_l
// [rows][h], 3 = (mean, rstd) of z_l [rows][2], 4 = q [rows][ld]): *offset in floats.  bf16 plane triples (what 0 = the
// normalised input [rows][ld], 2 = h_l = relu(LN(z_l)) [rows][h]): *offset in bf16 elements from the start of the
// workspace, the three planes pad16(rows) * ld elements apart, fragment-major (bm_slot); value = hi + mid + lo.
extern "C" int pqn_bigmlp_workspace_view(const pqn_bigmlp_layout_t *L, int32_t rows, int32_t nb, int32_t what, int32_t layer,
                                         int64_t *offset, int64_t *ld) {
  PQN_REQUIRE(L && offset && ld && rows > 0 && nb > 0 && nb <= rows && layer >= 0 && layer < L->layers && what >= 0 && what <= 4,
              "pqn_bigmlp_workspace_view: bad arguments");
  const BmWs w = bm_ws(*L, rows, nb);
  switch (what) {
    case 0: *offset = 2 * w.f_total + w.xn; *ld = w.dp; break;
    case 1: *offset = w.z[layer]; *ld = L->h; break;
    case 2: *offset = 2 * w.f_total + w.h[layer]; *ld = L->h; break;
    case 3: *offset = w.stat[layer]; *ld = 2; break;
    default: *offset = w.q; *ld = w.ldq; break;
  }
  return PQN_OK;
}

static unsigned long long *g_bm_stamps = nullptr;   // profiling: PQN_BM_STAMPS=1
extern "C" int pqn_debug_bm_stamps(unsigned long long *out /* host, 128 entries */) {
  if (!g_bm_stamps) return PQN_E_INVALID;
  if (hipMemcpy(out, g_bm_stamps, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return PQN_E_HIP;
  return PQN_OK;
}

extern "C" int64_t pqn_bigmlp_gemm_scratch_floats(int32_t m, int32_t n, int32_t k) {
  if (m <= 0 || n <= 0 || k <= 0) return -1;
  const long long kp = bm_pad32(k);
  const long long tmp = 3ll * bm_pad16(k) * max(bm_pad32(m), bm_pad32(n));
  return (3 * ((long long)bm_pad16(m) * kp + (long long)bm_pad16(n) * kp) + tmp) / 2 + 64;
}

extern "C" int pqn_bigmlp_gemm(int32_t m, int32_t n, int32_t k, const float *a, int64_t lda, int32_t trans_a, const float *b,
                               int64_t ldb, int32_t trans_b, const float *bias, float *c, int64_t ldc, int32_t nsplit,
                               int64_t split_stride, int32_t tile_rows, float *scratch, void *stream) {
  PQN_REQUIRE(a && b && c && scratch && m > 0 && n > 0 && k > 0 && nsplit >= 1 && nsplit <= BM_MAX_SPLIT &&
                  (tile_rows == 64 || tile_rows == 128), "pqn_bigmlp_gemm: bad arguments");
  PQN_REQUIRE(nsplit == 1 || !bias, "pqn_bigmlp_gemm: the bias belongs to the consumer of K-split partials");
  hipStream_t st = (hipStream_t)stream;
  const int kp = bm_pad32(k);
  bf16_t *pa = reinterpret_cast<bf16_t *>(scratch);
  bf16_t *pb = pa + bm_pl_elems(m, kp), *tmp = pb + bm_pl_elems(n, kp);
  // -> planes [rows_out][kp] with K contiguous; a source stored [k][rows_out] is split as it lies, then transposed
  auto prep = [&](const float *src, long long ld, bool k_major, int rows_out, bf16_t *dst) {
    if (!k_major) {
      bm_split(src, ld, rows_out, k, bm_plo(dst, rows_out, kp), st);
    } else {
      const int rp = bm_pad32(rows_out);
      bm_split(src, ld, k, rows_out, bm_plo(tmp, k, rp), st);
      bm_transpose(bm_pl(tmp, k, rp), rows_out, bm_plo(dst, rows_out, kp), st);
    }
  };
  prep(a, lda, trans_a != 0, m, pa);
  prep(b, ldb, trans_b != 0, n, pb);
  BmPlan p;
  p.bm = tile_rows;
  p.klen = ((kp + nsplit - 1) / nsplit + BM_KS - 1) / BM_KS * BM_KS;
  p.nsplit = (kp + p.klen - 1) / p.klen;
  BmEpilogue E = bm_store(c, ldc, bias, split_stride);
  if (getenv("PQN_BM_STAMPS")) {
    if (!g_bm_stamps && (hipMalloc(&g_bm_stamps, 128 * sizeof(unsigned long long)) != hipSuccess ||
                         hipMemset(g_bm_stamps, 0, 128 * sizeof(unsigned long long)) != hipSuccess)) g_bm_stamps = nullptr;
    E.stamps = g_bm_stamps;
  }
  // the K range may need fewer splits than asked for (k = 45: two 32-wide ranges for nsplit = 3): the remaining partial outputs
  // are zero-filled, so that "nsplit partials, the caller sums them" holds for every shape
  for (int sp = p.nsplit; sp < nsplit; ++sp)
    if (hipMemset2DAsync(c + (long long)sp * split_stride, (size_t)ldc * sizeof(float), 0, (size_t)n * sizeof(float), (size_t)m, st) != hipSuccess) {
      pqn_set_error("pqn_bigmlp_gemm: hipMemset2DAsync failed");
      return PQN_E_HIP;
    }
  return bm_launch<BM_EPI_STORE>(m, n, kp, p, bm_pl(pa, m, kp), bm_pl(pb, n, kp), E, st);
}
