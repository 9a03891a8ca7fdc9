// pqn_qnet.hip -- fused MinAtar CNN Q-network kernels for gfx950 (f32, MFMA).
//
// Network = QNetwork(CNN) of the reference, purejaxql/pqn_minatar.py:24-69:
//   x/255 -> Conv3x3(C->16, VALID) -> LayerNorm(16) -> relu -> flatten(h,w,c)=1024
//         -> Dense(128) -> LayerNorm(128) -> relu -> Dense(A)
// The kernels never see the f32 [10,10,C] observation: they read the bit-packed
// grid (include/pqn_hotpath.h "obs_bits", 64 B/obs for Breakout instead of
// 1600 B) and evaluate the conv as a sum over the SET bits of each 3x3xC window.
//
// Work decomposition (one 256-thread workgroup = 16 samples = one MFMA M-tile):
//   phase 1  conv+LN+relu, one lane per (sample, output position); h1 tile -> LDS
//   phase 2  fc1 as v_mfma_f32_16x16x4_f32: A = h1 tile (LDS, ds_read_b128),
//            B = fc1 kernel streamed from L2 in MFMA-fragment order (1 KB per
//            wave-instruction, global_load_dwordx4); 4 waves x 2 column blocks
//   phase 3  LN(128)+relu+fc2 (+ eps-greedy epilogue), 16 lanes per sample
//
// Parameter layout ("kernel layout", pqn_cnn_layout): flax order, segment starts
// padded to 16 B, and the fc1 kernel W1[i][o] stored in MFMA C/D-fragment order
//   idx(i,o) = ((i/16 * 8 + o/16) * 64 + ((i%16)/4)*16 + o%16) * 4 + i%4
// which is at once (a) the B-operand fragment of the forward GEMM, with the K
// dimension permuted so one lane's float4 feeds 4 consecutive MFMAs, and (b) the
// accumulator layout in which the weight-gradient GEMM produces dW1, so
// gradient, moments and parameters share one layout and RAdam stays elementwise.
#include <stdlib.h>

#include "pqn_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define QN_TILE 16        // samples per workgroup
#define QN_H1 1024        // conv features (8*8*16)
#define QN_H1S 1032       // LDS row stride of the h1 tile (floats): conflict-free ds_read_b128
#define QN_HID 128
#define QN_ZS 132         // LDS row stride of the z tile
#define QN_LN_EPS 1e-6f   // flax nn.LayerNorm default
#define QN_MAXA 8

template <int C>
struct CnnCfg {
  static constexpr int OW = (((100 * C + 31) / 32) + 3) / 4 * 4;  // packed obs words (16-B multiple)
  static constexpr int ROWBITS = 3 * C;                           // bits of one window row
  static constexpr int KW = 9 * C;                                // conv reduction length
};

struct CnnSmem {
  float *h1;      // [QN_TILE][QN_H1S]
  float *z;       // [QN_TILE][QN_ZS]
  float *wc;      // [KW][16] conv kernel, then bias[16], ln0 scale[16], ln0 bias[16]
  uint32_t *bits; // [QN_TILE][OW]
};

PQN_D float rsqrt_exact(float x) { return 1.0f / sqrtf(x); }

// ---------------------------------------------------------------------------
// phase 1: conv (sparse over set bits) + LN(16) + relu -> h1 tile in LDS.
// lane <-> (sample m = p>>6, position pos = p&63); a wave covers one sample.
// If xhat_out != nullptr the caller wants (pre-relu) normalised values back.
// ---------------------------------------------------------------------------
template <int C>
PQN_D void conv_ln_point(const CnnSmem &s, int m, int pos, float (&y)[16], float (&xhat)[16], float &rstd,
                         uint32_t (&wmask)[3]) {
  using Cfg = CnnCfg<C>;
  const int py = pos >> 3, px = pos & 7;
  const float *wc = s.wc;
  const float *bc = s.wc + Cfg::KW * 16;
  float acc[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) acc[o] = bc[o];
  const uint32_t *b = s.bits + m * Cfg::OW;
  const float inv255 = 1.0f / 255.0f;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int sb = ((py + ky) * 10 + px) * C;
    const int w = sb >> 5, sh = sb & 31;
    const uint64_t v = (((uint64_t)b[w + 1] << 32) | b[w]) >> sh;
    uint32_t mk = (uint32_t)v & ((1u << Cfg::ROWBITS) - 1u);
    wmask[ky] = mk;
    while (mk) {
      const int bit = __builtin_ctz(mk);
      mk &= mk - 1;
      const f32x4 *wr = reinterpret_cast<const f32x4 *>(wc + (ky * Cfg::ROWBITS + bit) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 wv = wr[q];
        acc[4 * q + 0] = fmaf(inv255, wv.x, acc[4 * q + 0]);
        acc[4 * q + 1] = fmaf(inv255, wv.y, acc[4 * q + 1]);
        acc[4 * q + 2] = fmaf(inv255, wv.z, acc[4 * q + 2]);
        acc[4 * q + 3] = fmaf(inv255, wv.w, acc[4 * q + 3]);
      }
    }
  }
  // LayerNorm over the 16 channels of this position (flax: var = E[x^2]-E[x]^2, clamped)
  float sum = 0.f, sq = 0.f;
#pragma unroll
  for (int o = 0; o < 16; ++o) { sum += acc[o]; sq = fmaf(acc[o], acc[o], sq); }
  const float mean = sum * (1.0f / 16.0f);
  const float var = fmaxf(sq * (1.0f / 16.0f) - mean * mean, 0.0f);
  rstd = rsqrt_exact(var + QN_LN_EPS);
  const float *g0 = bc + 16, *b0 = bc + 32;
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    xhat[o] = (acc[o] - mean) * rstd;
    y[o] = fmaxf(fmaf(xhat[o], g0[o], b0[o]), 0.0f);
  }
}

template <int C>
PQN_D void phase1_conv(const CnnSmem &s, int tid) {
#pragma unroll 1
  for (int r = 0; r < (QN_TILE * 64) / 256; ++r) {
    const int p = tid + 256 * r;
    const int m = p >> 6, pos = p & 63;
    float y[16], xhat[16], rstd;
    uint32_t wm[3];
    conv_ln_point<C>(s, m, pos, y, xhat, rstd, wm);
    f32x4 *dst = reinterpret_cast<f32x4 *>(s.h1 + m * QN_H1S + pos * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = f32x4{y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]};
  }
}

// ---------------------------------------------------------------------------
// phase 2: z[16][128] = h1[16][1024] x W1 (packed) via v_mfma_f32_16x16x4_f32.
// wave w owns column blocks 2w, 2w+1.  Operand maps (cdna guide 3): A[i=l&15][k=l>>4],
// B[k=l>>4][j=l&15], D: col=l&15, row=4*(l>>4)+reg.
// ---------------------------------------------------------------------------
// ablate: 0 = normal; 2 = no global B loads; 3 = no MFMA (loads only).  Profiling hook.
template <int ABL = 0>
PQN_D void phase2_fc1(const CnnSmem &s, const float *__restrict__ w1p, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int cb0 = 2 * wave, cb1 = cb0 + 1;
  const f32x4 *wp = reinterpret_cast<const f32x4 *>(w1p);
  const float *arow = s.h1 + (lane & 15) * QN_H1S + 4 * (lane >> 4);
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  constexpr int PF = 8;
  f32x4 b0[PF], b1[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    b0[i] = wp[(i * 8 + cb0) * 64 + lane];
    b1[i] = wp[(i * 8 + cb1) * 64 + lane];
  }
#pragma unroll 1
  for (int g = 0; g < QN_H1 / 16; g += PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(arow + 16 * (g + i));
      const f32x4 x0 = b0[i], x1 = b1[i];
      if (ABL != 2 && g + i + PF < QN_H1 / 16) {
        b0[i] = wp[((g + i + PF) * 8 + cb0) * 64 + lane];
        b1[i] = wp[((g + i + PF) * 8 + cb1) * 64 + lane];
      }
      if (ABL == 3) {
        acc0 += x0 + a;
        acc1 += x1;
        continue;
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x0.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x1.x, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x0.y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x1.y, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x0.z, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x1.z, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x0.w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x1.w, acc1, 0, 0, 0);
    }
  }
  const int col = lane & 15, r0 = 4 * (lane >> 4);
  s.z[(r0 + 0) * QN_ZS + 16 * cb0 + col] = acc0.x;
  s.z[(r0 + 1) * QN_ZS + 16 * cb0 + col] = acc0.y;
  s.z[(r0 + 2) * QN_ZS + 16 * cb0 + col] = acc0.z;
  s.z[(r0 + 3) * QN_ZS + 16 * cb0 + col] = acc0.w;
  s.z[(r0 + 0) * QN_ZS + 16 * cb1 + col] = acc1.x;
  s.z[(r0 + 1) * QN_ZS + 16 * cb1 + col] = acc1.y;
  s.z[(r0 + 2) * QN_ZS + 16 * cb1 + col] = acc1.z;
  s.z[(r0 + 3) * QN_ZS + 16 * cb1 + col] = acc1.w;
}

PQN_D float group16_sum(float v) {
  v += __shfl_xor(v, 8, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 1, 16);
  return v;
}

// ---------------------------------------------------------------------------
// phase 3: per sample m (16 lanes each): z+b1 -> LN(128) -> relu -> fc2 -> q[A].
// Every lane of the 16-lane group ends up with all q values.  xh/rstd returned
// for the backward pass (lane owns features o = sub + 16 r).
// ---------------------------------------------------------------------------
PQN_D void phase3_head(const CnnSmem &s, const float *__restrict__ theta, const pqn_cnn_layout_t &L, int tid,
                       float (&q)[QN_MAXA], float (&h2)[8], float (&xh)[8], float &rstd) {
  const int m = tid >> 4, sub = tid & 15;
  float v[8];
  float sum = 0.f, sq = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int o = sub + 16 * r;
    v[r] = s.z[m * QN_ZS + o] + theta[L.off_b1 + o];
    sum += v[r];
    sq = fmaf(v[r], v[r], sq);
  }
  sum = group16_sum(sum);
  sq = group16_sum(sq);
  const float mean = sum * (1.0f / QN_HID);
  const float var = fmaxf(sq * (1.0f / QN_HID) - mean * mean, 0.0f);
  rstd = rsqrt_exact(var + QN_LN_EPS);
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int o = sub + 16 * r;
    xh[r] = (v[r] - mean) * rstd;
    h2[r] = fmaxf(fmaf(xh[r], theta[L.off_ln1s + o], theta[L.off_ln1b + o]), 0.0f);
  }
#pragma unroll
  for (int a = 0; a < QN_MAXA; ++a) {
    float part = 0.f;
    if (a < L.a) {
#pragma unroll
      for (int r = 0; r < 8; ++r) part = fmaf(h2[r], theta[L.off_w2 + (sub + 16 * r) * L.a + a], part);
    }
    q[a] = group16_sum(part) + (a < L.a ? theta[L.off_b2 + a] : 0.0f);
  }
}

template <int C>
PQN_D void load_tile_common(const CnnSmem &s, const float *__restrict__ theta, const pqn_cnn_layout_t &L, int tid) {
  using Cfg = CnnCfg<C>;
  for (int i = tid; i < Cfg::KW * 16 + 48; i += 256) s.wc[i] = theta[L.off_wc + i];  // kernel|bias|ln0s|ln0b contiguous
}

template <int C>
PQN_D CnnSmem carve_smem(char *base) {
  using Cfg = CnnCfg<C>;
  CnnSmem s;
  s.h1 = reinterpret_cast<float *>(base);
  s.z = s.h1 + QN_TILE * QN_H1S;
  s.wc = s.z + QN_TILE * QN_ZS;
  s.bits = reinterpret_cast<uint32_t *>(s.wc + ((Cfg::KW * 16 + 48 + 3) & ~3));
  return s;
}

template <int C>
constexpr size_t cnn_smem_bytes() {
  using Cfg = CnnCfg<C>;
  return sizeof(float) * (QN_TILE * QN_H1S + QN_TILE * QN_ZS + ((Cfg::KW * 16 + 48 + 3) & ~3)) +
         sizeof(uint32_t) * (QN_TILE * Cfg::OW + 4);
}

// ---------------------------------------------------------------------------
// forward (+ optional eps-greedy epilogue): network.apply(train=False) at
// pqn_minatar.py:184-196,227-234,380-390.
//   idx   (nullable): gather -- sample j of the launch reads obs_bits[idx[j]]
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void qnet_cnn_fwd_kernel(int n, const uint32_t *__restrict__ obs_bits,
                                                           const float *__restrict__ theta, pqn_cnn_layout_t L,
                                                           float *__restrict__ q_out, int32_t *__restrict__ action,
                                                           float *__restrict__ qmax, float eps, uint64_t key,
                                                           int ablate) {
  using Cfg = CnnCfg<C>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const CnnSmem s = carve_smem<C>(smem_raw);
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * QN_TILE;
  load_tile_common<C>(s, theta, L, tid);
  for (int i = tid; i < QN_TILE * Cfg::OW; i += 256) {
    const int le = i / Cfg::OW;
    s.bits[i] = (e0 + le < n) ? obs_bits[(size_t)e0 * Cfg::OW + i] : 0u;
  }
  if (tid < 4) s.bits[QN_TILE * Cfg::OW + tid] = 0u;  // b[w+1] guard word
  __syncthreads();
  if (ablate != 1) phase1_conv<C>(s, tid);
  __syncthreads();
  if (ablate == 0 || ablate == 1) phase2_fc1<0>(s, theta + L.off_w1, tid);
  else if (ablate == 2) phase2_fc1<2>(s, theta + L.off_w1, tid);
  else if (ablate == 3) phase2_fc1<3>(s, theta + L.off_w1, tid);
  __syncthreads();
  float q[QN_MAXA], h2[8], xh[8], rstd;
  phase3_head(s, theta, L, tid, q, h2, xh, rstd);
  const int m = tid >> 4, sub = tid & 15, e = e0 + m;
  if (sub == 0 && e < n) {
    int best = 0;
    float bv = q[0];
#pragma unroll
    for (int a = 1; a < QN_MAXA; ++a)
      if (a < L.a && q[a] > bv) { bv = q[a]; best = a; }
    if (q_out) {
#pragma unroll
      for (int a = 0; a < QN_MAXA; ++a)
        if (a < L.a) q_out[(size_t)e * L.a + a] = q[a];
    }
    if (qmax) qmax[e] = bv;
    if (action) {
      uint32_t o0, o1;
      pqn_bits(key, (uint32_t)e, PQN_STREAM_ACT, o0, o1);
      const float u = pqn_uniform(o0);
      const int rnd = (int)pqn_randint(o1, (uint32_t)L.a);
      action[e] = (u < eps) ? rnd : best;
    }
  }
}


// ===========================================================================
// TRAINING.  One optimizer step of _learn_phase (pqn_minatar.py:266-297) =
//   T1 qnet_cnn_train_kernel   per 16-sample tile: forward, loss gradient, backward through
//                              head / LN1 / fc1 (dgrad MFMA) / relu / LN0 / conv; emits dz^T
//                              and per-tile partial sums of every "small" gradient
//   T2 qnet_cnn_wgrad_kernel   dW1 = h1^T x dz as MFMA, h1 recomputed from the packed obs,
//                              split-K over 512-sample slabs, output in fragment layout
//   T3 pqn_grad_reduce_kernel  deterministic fold of the partials into the flat gradient
//                              (+ block sums of squares), then radam_apply (pqn_algo.hip).
// "Small" partial record per tile (floats): [conv kernel KW*16 | conv bias 16 | ln0 scale 16 |
// ln0 bias 16 | b1 128 | ln1 scale 128 | ln1 bias 128 | w2 128*A | b2 A | loss | sum q_a].
// ===========================================================================
template <int C>
struct TrainCfg {
  using Cfg = CnnCfg<C>;
  static constexpr int CONVBLK = Cfg::KW * 16 + 48;
  static constexpr int SCR = (3 * QN_TILE * QN_ZS > Cfg::KW * 64) ? 3 * QN_TILE * QN_ZS : Cfg::KW * 64;
};

__host__ __device__ inline int small_record_floats(int c, int a) { return 9 * c * 16 + 48 + 384 + 128 * a + a + 2; }

struct TrainSmem {
  CnnSmem n;
  float *scr;        // 3 x [16][QN_ZS] tiles, later [KW][4][16] conv-wgrad partials
  uint32_t *planes;  // [16][C][4] per-channel cell masks
  float *gs;         // [16] per-sample loss gradient g_m = (q_a - target)/B
  int *act;          // [16]
  float *red;        // [4][48] cross-wave reduction of conv bias / ln0 grads
  int *ctr;          // work-queue head for the conv weight gradient
};

template <int C>
PQN_D TrainSmem carve_train_smem(char *base) {
  using Cfg = CnnCfg<C>;
  TrainSmem t;
  t.n = carve_smem<C>(base);
  uint32_t *after_bits = t.n.bits + QN_TILE * Cfg::OW + 4;
  t.scr = reinterpret_cast<float *>(after_bits);
  t.planes = reinterpret_cast<uint32_t *>(t.scr + TrainCfg<C>::SCR);
  t.gs = reinterpret_cast<float *>(t.planes + QN_TILE * C * 4);
  t.act = reinterpret_cast<int *>(t.gs + QN_TILE);
  t.red = reinterpret_cast<float *>(t.act + QN_TILE);
  t.ctr = reinterpret_cast<int *>(t.red + 4 * 48);
  return t;
}

template <int C>
constexpr size_t train_smem_bytes() {
  return cnn_smem_bytes<C>() + sizeof(float) * (TrainCfg<C>::SCR + QN_TILE * C * 4 + 2 * QN_TILE + 4 * 48 + 4);
}

template <int C>
__global__ __launch_bounds__(256) void qnet_cnn_train_kernel(
    int nb, const int64_t *__restrict__ idx, const uint32_t *__restrict__ obs_bits, const int32_t *__restrict__ action,
    const float *__restrict__ target, const float *__restrict__ theta, const float *__restrict__ w1b,
    pqn_cnn_layout_t L, float inv_b, float *__restrict__ dzT, float *__restrict__ gpart, int ablate) {
  using Cfg = CnnCfg<C>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const TrainSmem ts = carve_train_smem<C>(smem_raw);
  const CnnSmem &s = ts.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * QN_TILE;
  const int rec = small_record_floats(C, L.a);
  float *gp = gpart + (size_t)blockIdx.x * rec;

  // ---- P0: gather inputs -------------------------------------------------------------------
  load_tile_common<C>(s, theta, L, tid);
  for (int i = tid; i < QN_TILE * Cfg::OW; i += 256) {
    const int le = i / Cfg::OW, w = i - le * Cfg::OW;
    s.bits[i] = (b0 + le < nb) ? obs_bits[(size_t)idx[b0 + le] * Cfg::OW + w] : 0u;
  }
  if (tid < 4) s.bits[QN_TILE * Cfg::OW + tid] = 0u;
  if (tid == 0) *ts.ctr = 0;
  __syncthreads();
  // per-channel cell masks (bit `cell` of plane c <=> obs bit cell*C+c), for the conv weight gradient
  for (int i = tid; i < QN_TILE * C * 4; i += 256) {
    const int m = i / (C * 4), c = (i / 4) % C, w = i & 3;
    uint32_t pm = 0u;
    for (int j = 0; j < 32; ++j) {
      const int cell = w * 32 + j;
      if (cell < 100) {
        const int bit = cell * C + c;
        pm |= ((s.bits[m * Cfg::OW + (bit >> 5)] >> (bit & 31)) & 1u) << j;
      }
    }
    ts.planes[i] = pm;
  }
  // ---- P1..P3: forward ---------------------------------------------------------------------
  phase1_conv<C>(s, tid);
  __syncthreads();
  phase2_fc1<0>(s, theta + L.off_w1, tid);
  __syncthreads();
  float q[QN_MAXA], h2[8], xh[8], rstd1;
  phase3_head(s, theta, L, tid, q, h2, xh, rstd1);
  const int m = tid >> 4, sub = tid & 15;
  const bool valid = (b0 + m) < nb;
  const int64_t src = valid ? idx[b0 + m] : 0;
  const int act = valid ? action[src] : 0;
  float chosen = q[0];
#pragma unroll
  for (int a = 1; a < QN_MAXA; ++a)
    if (a == act) chosen = q[a];
  const float diff = valid ? (chosen - target[src]) : 0.0f;
  const float gm = diff * inv_b;  // d loss / d q_a, loss = 0.5*mean(diff^2)  (pqn_minatar.py:285)
  if (sub == 0) {
    ts.gs[m] = gm;
    ts.act[m] = act;
  }
  // ---- head backward: fc2, relu, LN1 ----------------------------------------------------------
  float *tA = ts.scr, *tB = ts.scr + QN_TILE * QN_ZS, *tH = ts.scr + 2 * QN_TILE * QN_ZS;
  __syncthreads();  // everyone is done reading s.z (phase 3) before it is overwritten with dz
  {
    float dxh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int o = sub + 16 * r;
      const float dh2 = gm * theta[L.off_w2 + o * L.a + act];
      const float dy = h2[r] > 0.0f ? dh2 : 0.0f;
      tA[m * QN_ZS + o] = dy * xh[r];
      tB[m * QN_ZS + o] = dy;
      tH[m * QN_ZS + o] = h2[r];
      dxh[r] = dy * theta[L.off_ln1s + o];
      s1 += dxh[r];
      s2 = fmaf(dxh[r], xh[r], s2);
    }
    s1 = group16_sum(s1) * (1.0f / QN_HID);
    s2 = group16_sum(s2) * (1.0f / QN_HID);
#pragma unroll
    for (int r = 0; r < 8; ++r) s.z[m * QN_ZS + sub + 16 * r] = rstd1 * (dxh[r] - s1 - xh[r] * s2);
  }
  __syncthreads();
  // column sums over the 16 samples (fixed order -> deterministic)
  {
    const int o_b1 = 9 * C * 16 + 48;
    if (tid < QN_HID) {
      float a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int mm = 0; mm < QN_TILE; ++mm) {
        a1 += s.z[mm * QN_ZS + tid];
        a2 += tA[mm * QN_ZS + tid];
        a3 += tB[mm * QN_ZS + tid];
      }
      gp[o_b1 + tid] = a1;              // d b1
      gp[o_b1 + 128 + tid] = a2;        // d ln1 scale
      gp[o_b1 + 256 + tid] = a3;        // d ln1 bias
    } else {
      const int o = tid - QN_HID;       // d w2[o][a] = sum_m [act_m == a] g_m h2[m][o]
      for (int a = 0; a < L.a; ++a) {
        float acc = 0.f;
#pragma unroll
        for (int mm = 0; mm < QN_TILE; ++mm) acc += (ts.act[mm] == a) ? ts.gs[mm] * tH[mm * QN_ZS + o] : 0.0f;
        gp[o_b1 + 384 + o * L.a + a] = acc;
      }
    }
    if (tid < L.a) {
      float acc = 0.f;
      for (int mm = 0; mm < QN_TILE; ++mm) acc += (ts.act[mm] == tid) ? ts.gs[mm] : 0.0f;
      gp[o_b1 + 384 + 128 * L.a + tid] = acc;  // d b2
    }
    // loss / chosen-q partials (metrics td_loss, qvals: pqn_minatar.py:334-335)
    float l = (sub == 0) ? 0.5f * diff * diff : 0.0f, cq = (sub == 0 && valid) ? chosen : 0.0f;
    for (int off = 32; off > 0; off >>= 1) { l += __shfl_down(l, off, 64); cq += __shfl_down(cq, off, 64); }
    if (lane == 0) { ts.red[wave * 48] = l; ts.red[wave * 48 + 1] = cq; }
  }
  // dz^T for the weight-gradient GEMM: dzT[o][b0 + m], 16-B stores
  for (int i = tid; i < QN_HID * 4; i += 256) {
    const int o = i >> 2, mq = i & 3;
    const f32x4 v = {s.z[(4 * mq + 0) * QN_ZS + o], s.z[(4 * mq + 1) * QN_ZS + o], s.z[(4 * mq + 2) * QN_ZS + o],
                     s.z[(4 * mq + 3) * QN_ZS + o]};
    *reinterpret_cast<f32x4 *>(dzT + (size_t)o * nb + b0 + 4 * mq) = v;
  }
  __syncthreads();
  if (tid == 0) {
    const int o_l = 9 * C * 16 + 48 + 384 + 128 * L.a + L.a;
    gp[o_l] = (ts.red[0] + ts.red[48]) + (ts.red[96] + ts.red[144]);
    gp[o_l + 1] = (ts.red[1] + ts.red[49]) + (ts.red[97] + ts.red[145]);
  }
  // ---- P4: dgrad  dh1[m][i] = sum_o dz[m][o] W1[i][o]  (A = dz tile, B = W1 in dgrad fragment order) ----
  if (!(ablate & 1)) {
    const f32x4 *wb = reinterpret_cast<const f32x4 *>(w1b);
    f32x4 afr[8];
#pragma unroll
    for (int g = 0; g < 8; ++g)
      afr[g] = *reinterpret_cast<const f32x4 *>(s.z + (lane & 15) * QN_ZS + 16 * g + 4 * (lane >> 4));
    const int col = lane & 15, r0 = 4 * (lane >> 4);
    f32x4 bA[8], bB[8];
    const int ib_first = 16 * wave;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      bA[g] = wb[(g * 64 + ib_first) * 64 + lane];
      bB[g] = wb[(g * 64 + ib_first + 1) * 64 + lane];
    }
#pragma unroll 1
    for (int ip = 0; ip < 8; ++ip) {
      const int ib = ib_first + 2 * ip;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      f32x4 cA[8], cB[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) { cA[g] = bA[g]; cB[g] = bB[g]; }
      if (ip + 1 < 8) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          bA[g] = wb[(g * 64 + ib + 2) * 64 + lane];
          bB[g] = wb[(g * 64 + ib + 3) * 64 + lane];
        }
      }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].x, cA[g].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].x, cB[g].x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].y, cA[g].y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].y, cB[g].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].z, cA[g].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].z, cB[g].z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].w, cA[g].w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].w, cB[g].w, acc1, 0, 0, 0);
      }
      // relu mask (h1 > 0) and in-place overwrite of the h1 tile with d(pre-relu)
      float *p0 = s.h1 + r0 * QN_H1S + 16 * ib + col;
      float *p1 = p0 + 16;
      const float a0[4] = {acc0.x, acc0.y, acc0.z, acc0.w}, a1[4] = {acc1.x, acc1.y, acc1.z, acc1.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p0[r * QN_H1S] = p0[r * QN_H1S] > 0.0f ? a0[r] : 0.0f;
        p1[r * QN_H1S] = p1[r * QN_H1S] > 0.0f ? a1[r] : 0.0f;
      }
    }
  }
  __syncthreads();
  // ---- P5: LN0 backward per (sample, position); conv bias / ln0 grads; dx tile in place ------------
  if (!(ablate & 2)) {
    float gsc[16], gbi[16], gbc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) { gsc[o] = 0.f; gbi[o] = 0.f; gbc[o] = 0.f; }
    const float *g0 = s.wc + Cfg::KW * 16 + 16;
#pragma unroll 1
    for (int r = 0; r < (QN_TILE * 64) / 256; ++r) {
      const int p = tid + 256 * r;
      const int mm = p >> 6, pos = p & 63;
      float y[16], xhat[16], rstd;
      uint32_t wm[3];
      conv_ln_point<C>(s, mm, pos, y, xhat, rstd, wm);
      f32x4 *gptr = reinterpret_cast<f32x4 *>(s.h1 + mm * QN_H1S + pos * 16);
      float g[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = gptr[qd];
        g[4 * qd] = v.x; g[4 * qd + 1] = v.y; g[4 * qd + 2] = v.z; g[4 * qd + 3] = v.w;
      }
      float s1 = 0.f, s2 = 0.f, dxh[16];
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        gsc[o] = fmaf(g[o], xhat[o], gsc[o]);
        gbi[o] += g[o];
        dxh[o] = g[o] * g0[o];
        s1 += dxh[o];
        s2 = fmaf(dxh[o], xhat[o], s2);
      }
      s1 *= (1.0f / 16.0f);
      s2 *= (1.0f / 16.0f);
      float dx[16];
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        dx[o] = rstd * (dxh[o] - s1 - xhat[o] * s2);
        gbc[o] += dx[o];
      }
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) gptr[qd] = f32x4{dx[4 * qd], dx[4 * qd + 1], dx[4 * qd + 2], dx[4 * qd + 3]};
    }
    // wave tree reduction, then fixed-order fold of the 4 waves
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      for (int off = 32; off > 0; off >>= 1) {
        gsc[o] += __shfl_down(gsc[o], off, 64);
        gbi[o] += __shfl_down(gbi[o], off, 64);
        gbc[o] += __shfl_down(gbc[o], off, 64);
      }
    }
    __syncthreads();  // ts.red was read by tid 0 above; dx tile complete
    if (lane == 0) {
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        ts.red[wave * 48 + o] = gbc[o];
        ts.red[wave * 48 + 16 + o] = gsc[o];
        ts.red[wave * 48 + 32 + o] = gbi[o];
      }
    }
  }
  __syncthreads();
  if (tid < 48) gp[Cfg::KW * 16 + tid] = (ts.red[tid] + ts.red[48 + tid]) + (ts.red[96 + tid] + ts.red[144 + tid]);
  // ---- P6: conv weight gradient, sparse over the set cells of each channel plane --------------------
  //   dWc[ky][kx][c][o] = 1/255 * sum_m sum_{cell in plane c(m)} dx[m][cell - (ky,kx)][o]
  // work item = (k, sample quarter); 16 lanes = 16 output channels; partials [KW][4][16] in LDS.
  if (!(ablate & 4)) {
    float *part = ts.scr;
    const int o = tid & 15;
    for (;;) {
      int item = 0;
      if (o == 0) item = atomicAdd(ts.ctr, 1);  // which 16-lane group takes which item does not change the result
      item = __shfl(item, 0, 16);
      if (item >= Cfg::KW * 4) break;
      const int k = item >> 2, mq = item & 3;
      const int c = k % C, kx = (k / C) % 3, ky = k / (3 * C);
      float acc = 0.f;
      for (int mm = 4 * mq; mm < 4 * mq + 4; ++mm) {
        const uint32_t *pl = ts.planes + (mm * C + c) * 4;
        const float *dxm = s.h1 + mm * QN_H1S + o;
#pragma unroll 1
        for (int w = 0; w < 4; ++w) {
          uint32_t pm = pl[w];
          while (pm) {
            const int cell = w * 32 + __builtin_ctz(pm);
            pm &= pm - 1;
            const int y = cell / 10, x = cell - 10 * y;
            const int py = y - ky, px = x - kx;
            if ((unsigned)py < 8u && (unsigned)px < 8u) acc += dxm[(py * 8 + px) * 16];
          }
        }
      }
      part[item * 16 + o] = acc;
    }
  }
  __syncthreads();
  for (int i = tid; i < Cfg::KW * 16; i += 256) {
    const int k = i >> 4, o = i & 15;
    const float *pp = ts.scr + (k * 4) * 16 + o;
    gp[i] = ((pp[0] + pp[16]) + (pp[32] + pp[48])) * (1.0f / 255.0f);
  }
}

// ---------------------------------------------------------------------------
// T2: dW1[i][o] = sum_b h1[b][i] dz[b][o].  Workgroup (it, ks): rows i in [32 it, 32 it + 32)
// (= conv positions 2it, 2it+1, recomputed from the packed obs), samples [512 ks, 512 ks + 512).
// A = h1^T tile from LDS, B = dz^T from L2; output written in fragment layout into wpart[ks].
// ---------------------------------------------------------------------------
#define QW_CH 128   // samples per LDS chunk
#define QW_HS 132   // row stride of the h1^T chunk

template <int C>
__global__ __launch_bounds__(256) void qnet_cnn_wgrad_kernel(int nb, const int64_t *__restrict__ idx,
                                                             const uint32_t *__restrict__ obs_bits,
                                                             const float *__restrict__ theta, pqn_cnn_layout_t L,
                                                             const float *__restrict__ dzT, float *__restrict__ wpart) {
  using Cfg = CnnCfg<C>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *h1t = reinterpret_cast<float *>(smem_raw);                    // [32][QW_HS]
  CnnSmem s;
  s.h1 = nullptr;
  s.z = nullptr;
  s.wc = h1t + 32 * QW_HS;
  s.bits = reinterpret_cast<uint32_t *>(s.wc + ((Cfg::KW * 16 + 48 + 3) & ~3));  // [QW_CH][OW] + guard
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int it = blockIdx.x, ks = blockIdx.y;
  load_tile_common<C>(s, theta, L, tid);
  f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int cb0 = 2 * wave;
  const int bend = min(nb, ks * 512 + 512);
  for (int c0 = ks * 512; c0 < bend; c0 += QW_CH) {
    __syncthreads();  // previous chunk's MFMA reads are done
    for (int i = tid; i < QW_CH * Cfg::OW; i += 256) {
      const int le = i / Cfg::OW, w = i - le * Cfg::OW;
      s.bits[i] = (c0 + le < nb) ? obs_bits[(size_t)idx[c0 + le] * Cfg::OW + w] : 0u;
    }
    if (tid < 4) s.bits[QW_CH * Cfg::OW + tid] = 0u;
    __syncthreads();
    {
      const int bl = tid & (QW_CH - 1), pp = tid >> 7;
      float y[16], xhat[16], rstd;
      uint32_t wm[3];
      conv_ln_point<C>(s, bl, 2 * it + pp, y, xhat, rstd, wm);
#pragma unroll
      for (int o = 0; o < 16; ++o) h1t[(pp * 16 + o) * QW_HS + bl] = y[o];
    }
    __syncthreads();
    const int ngroups = min(QW_CH / 16, (bend - c0) / 16);
    for (int g = 0; g < ngroups; ++g) {
      const int boff = 16 * g + 4 * (lane >> 4);
      const f32x4 a0 = *reinterpret_cast<const f32x4 *>(h1t + (lane & 15) * QW_HS + boff);
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(h1t + (16 + (lane & 15)) * QW_HS + boff);
      const f32x4 x0 = *reinterpret_cast<const f32x4 *>(dzT + (size_t)(16 * cb0 + (lane & 15)) * nb + c0 + boff);
      const f32x4 x1 = *reinterpret_cast<const f32x4 *>(dzT + (size_t)(16 * cb0 + 16 + (lane & 15)) * nb + c0 + boff);
#define QW_STEP(comp)                                                                             \
  acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.comp, x0.comp, acc[0][0], 0, 0, 0);        \
  acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.comp, x1.comp, acc[0][1], 0, 0, 0);        \
  acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.comp, x0.comp, acc[1][0], 0, 0, 0);        \
  acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.comp, x1.comp, acc[1][1], 0, 0, 0);
      QW_STEP(x) QW_STEP(y) QW_STEP(z) QW_STEP(w)
#undef QW_STEP
    }
  }
  f32x4 *out = reinterpret_cast<f32x4 *>(wpart + (size_t)ks * QN_H1 * QN_HID);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) out[((2 * it + a) * 8 + cb0 + b) * 64 + lane] = acc[a][b];
}

// ---------------------------------------------------------------------------
// T3a: fold partials into the flat gradient (kernel layout) and emit per-block sums of
// squares + the *count snapshot for radam_apply (same scratch protocol as radam_norm_kernel).
// Blocks [0, QR_W1_BLOCKS): fc1 region, float4 per lane, sum over the K-splits.
// Remaining blocks: one WAVE per "small" element, lanes stride over the tiles, fixed-order tree.
// ---------------------------------------------------------------------------
#define QR_W1_BLOCKS (QN_H1 * QN_HID / 1024)

__host__ __device__ inline int grad_reduce_blocks(int total) { return QR_W1_BLOCKS + (total - QN_H1 * QN_HID + 3) / 4; }

__global__ __launch_bounds__(256) void qnet_grad_reduce_kernel(pqn_cnn_layout_t L, int ntiles, int nks, int rec,
                                                               const float *__restrict__ gpart,
                                                               const float *__restrict__ wpart, float *__restrict__ grad,
                                                               const int32_t *__restrict__ count,
                                                               float *__restrict__ scratch, float *__restrict__ loss_out,
                                                               float *__restrict__ qv_out, float inv_b) {
  __shared__ float s_part[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ss = 0.0f;
  if (blockIdx.x < QR_W1_BLOCKS) {
    const int j4 = blockIdx.x * 256 + threadIdx.x;  // float4 index inside the fc1 region
    f32x4 g = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nks; ++k) g += reinterpret_cast<const f32x4 *>(wpart + (size_t)k * QN_H1 * QN_HID)[j4];
    reinterpret_cast<f32x4 *>(grad + L.off_w1)[j4] = g;
    ss = fmaf(g.x, g.x, fmaf(g.y, g.y, fmaf(g.z, g.z, g.w * g.w)));
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
  } else {
    const int sidx = (blockIdx.x - QR_W1_BLOCKS) * 4 + wave;  // index among the non-fc1 elements
    const int i = sidx < L.off_w1 ? sidx : sidx + QN_H1 * QN_HID;
    if (i < L.total) {
      const int convblk = 9 * L.c * 16 + 48;
      int r = -1;  // index into the small record (-1: dummy BatchNorm / padding -> zero gradient)
      if (i >= L.off_wc && i < L.off_wc + convblk) r = i - L.off_wc;
      else if (i >= L.off_b1 && i < L.off_b1 + 384) r = convblk + (i - L.off_b1);
      else if (i >= L.off_w2 && i < L.off_w2 + 128 * L.a) r = convblk + 384 + (i - L.off_w2);
      else if (i >= L.off_b2 && i < L.off_b2 + L.a) r = convblk + 384 + 128 * L.a + (i - L.off_b2);
      float g = 0.0f;
      if (r >= 0)
        for (int t = lane; t < ntiles; t += 64) g += gpart[(size_t)t * rec + r];
      for (int off = 32; off > 0; off >>= 1) g += __shfl_down(g, off, 64);
      if (lane == 0) {
        grad[i] = g;
        ss = g * g;
      }
    }
    if (blockIdx.x == QR_W1_BLOCKS && wave == 0) {  // metrics td_loss / qvals (pqn_minatar.py:334-335)
      float l = 0.f, qv = 0.f;
      for (int t = lane; t < ntiles; t += 64) {
        l += gpart[(size_t)t * rec + rec - 2];
        qv += gpart[(size_t)t * rec + rec - 1];
      }
      for (int off = 32; off > 0; off >>= 1) { l += __shfl_down(l, off, 64); qv += __shfl_down(qv, off, 64); }
      if (lane == 0) {
        if (loss_out) *loss_out = l * inv_b;
        if (qv_out) *qv_out = qv * inv_b;
      }
    }
  }
  if (lane == 0) s_part[wave] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    scratch[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    if (blockIdx.x == 0) reinterpret_cast<int32_t *>(scratch)[1023] = *count;
  }
}

// ===========================================================================
// host side
// ===========================================================================
static int align4(int x) { return (x + 3) & ~3; }

extern "C" int pqn_cnn_layout(int32_t c, int32_t a, pqn_cnn_layout_t *L) {
  PQN_REQUIRE(L, "pqn_cnn_layout: layout is NULL");
  PQN_REQUIRE(c == 4 || c == 6 || c == 7 || c == 10, "pqn_cnn_layout: unsupported channel count %d", c);
  PQN_REQUIRE(a >= 1 && a <= QN_MAXA, "pqn_cnn_layout: num_actions %d out of range [1,%d]", a, QN_MAXA);
  int off = 0;
  L->c = c;
  L->a = a;
  L->off_bn = off; off = align4(off + 2 * c);
  L->off_wc = off; off += 9 * c * 16;   // conv kernel, bias, ln0 scale, ln0 bias stay contiguous
  L->off_bc = off; off += 16;
  L->off_ln0s = off; off += 16;
  L->off_ln0b = off; off = align4(off + 16);
  L->off_w1 = off; off += QN_H1 * QN_HID;
  L->off_b1 = off; off += QN_HID;
  L->off_ln1s = off; off += QN_HID;
  L->off_ln1b = off; off += QN_HID;
  L->off_w2 = off; off = align4(off + QN_HID * a);
  L->off_b2 = off; off = align4(off + a);
  L->total = off;
  return PQN_OK;
}

template <int C>
static int launch_fwd(int n, const uint32_t *bits, const float *theta, const pqn_cnn_layout_t &L, float *q,
                      int32_t *action, float *qmax, float eps, uint64_t key, hipStream_t st) {
  const size_t smem = cnn_smem_bytes<C>();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_fwd_kernel<C>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  static const int ablate = getenv("PQN_ABLATE") ? atoi(getenv("PQN_ABLATE")) : 0;  // profiling only
  hipLaunchKernelGGL((qnet_cnn_fwd_kernel<C>), dim3((n + QN_TILE - 1) / QN_TILE), dim3(256), smem, st, n, bits, theta, L,
                     q, action, qmax, eps, key, ablate);
  return pqn_check_launch("pqn_qnet_cnn_forward");
}

extern "C" int pqn_qnet_cnn_forward(const pqn_cnn_layout_t *L, int32_t n, const uint32_t *obs_bits, const float *theta,
                                    float *q, int32_t *action, float *qmax, float eps, uint64_t key, void *stream) {
  PQN_REQUIRE(L && obs_bits && theta, "pqn_qnet_cnn_forward: NULL argument");
  PQN_REQUIRE(n > 0, "pqn_qnet_cnn_forward: n must be > 0");
  PQN_REQUIRE(q || action || qmax, "pqn_qnet_cnn_forward: no output requested");
  hipStream_t st = (hipStream_t)stream;
  switch (L->c) {
    case 4: return launch_fwd<4>(n, obs_bits, theta, *L, q, action, qmax, eps, key, st);
    case 6: return launch_fwd<6>(n, obs_bits, theta, *L, q, action, qmax, eps, key, st);
    case 7: return launch_fwd<7>(n, obs_bits, theta, *L, q, action, qmax, eps, key, st);
    case 10: return launch_fwd<10>(n, obs_bits, theta, *L, q, action, qmax, eps, key, st);
    default: pqn_set_error("pqn_qnet_cnn_forward: unsupported channel count %d", L->c); return PQN_E_UNSUPPORTED;
  }
}

template <int C>
static int launch_train(const pqn_cnn_layout_t &L, int nb, const int64_t *idx, const uint32_t *bits,
                        const int32_t *action, const float *target, const float *theta, const float *w1b, float *grad,
                        const int32_t *count, float *ws, float *loss_out, float *qv_out, hipStream_t st) {
  const int ntiles = nb / QN_TILE, nks = (nb + 511) / 512, rec = small_record_floats(C, L.a);
  float *scratch = ws;
  float *dzT = ws + 1024;
  float *gpart = dzT + (size_t)QN_HID * nb;
  float *wpart = gpart + (size_t)ntiles * rec;
  const size_t smem1 = train_smem_bytes<C>();
  const size_t smem2 = sizeof(float) * (32 * QW_HS + ((CnnCfg<C>::KW * 16 + 48 + 3) & ~3)) +
                       sizeof(uint32_t) * (QW_CH * CnnCfg<C>::OW + 4);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_train_kernel<C>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_wgrad_kernel<C>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
    attr_set = true;
  }
  const float inv_b = 1.0f / (float)nb;
  static const int ablate = getenv("PQN_ABLATE_TRAIN") ? atoi(getenv("PQN_ABLATE_TRAIN")) : 0;  // profiling only
  hipLaunchKernelGGL((qnet_cnn_train_kernel<C>), dim3(ntiles), dim3(256), smem1, st, nb, idx, bits, action, target,
                     theta, w1b, L, inv_b, dzT, gpart, ablate);
  hipLaunchKernelGGL((qnet_cnn_wgrad_kernel<C>), dim3(32, nks), dim3(256), smem2, st, nb, idx, bits, theta, L, dzT,
                     wpart);
  hipLaunchKernelGGL(qnet_grad_reduce_kernel, dim3(grad_reduce_blocks(L.total)), dim3(256), 0, st, L, ntiles, nks, rec,
                     gpart, wpart, grad, count, scratch, loss_out, qv_out, inv_b);
  return pqn_check_launch("pqn_qnet_cnn_grad");
}

extern "C" int64_t pqn_qnet_cnn_workspace_floats(const pqn_cnn_layout_t *L, int32_t nb) {
  if (!L || nb <= 0) return -1;
  const int64_t ntiles = nb / QN_TILE, nks = (nb + 511) / 512;
  return 1024 + (int64_t)QN_HID * nb + ntiles * small_record_floats(L->c, L->a) + nks * (int64_t)QN_H1 * QN_HID;
}

extern "C" int pqn_qnet_cnn_grad(const pqn_cnn_layout_t *L, int32_t nb, const int64_t *idx, const uint32_t *obs_bits,
                                 const int32_t *action, const float *target, const float *theta, const float *w1b,
                                 float *grad, const int32_t *count, float *workspace, float *loss_out, float *qv_out,
                                 void *stream) {
  PQN_REQUIRE(L && idx && obs_bits && action && target && theta && w1b && grad && count && workspace,
              "pqn_qnet_cnn_grad: NULL argument");
  PQN_REQUIRE(nb > 0 && nb % QN_TILE == 0, "pqn_qnet_cnn_grad: minibatch size %d must be a positive multiple of %d", nb,
              QN_TILE);
  hipStream_t st = (hipStream_t)stream;
  switch (L->c) {
    case 4: return launch_train<4>(*L, nb, idx, obs_bits, action, target, theta, w1b, grad, count, workspace, loss_out, qv_out, st);
    case 6: return launch_train<6>(*L, nb, idx, obs_bits, action, target, theta, w1b, grad, count, workspace, loss_out, qv_out, st);
    case 7: return launch_train<7>(*L, nb, idx, obs_bits, action, target, theta, w1b, grad, count, workspace, loss_out, qv_out, st);
    case 10: return launch_train<10>(*L, nb, idx, obs_bits, action, target, theta, w1b, grad, count, workspace, loss_out, qv_out, st);
    default: pqn_set_error("pqn_qnet_cnn_grad: unsupported channel count %d", L->c); return PQN_E_UNSUPPORTED;
  }
}

extern "C" int pqn_qnet_cnn_apply(const pqn_cnn_layout_t *L, float *theta, float *w1b, const float *grad, float *m,
                                  float *v, int32_t *count, float lr_init, float lr_end, float lr_steps,
                                  float max_norm, float *workspace, float *gnorm_out, int32_t recompute_norm,
                                  void *stream) {
  PQN_REQUIRE(L && theta && w1b && grad && m && v && count && workspace, "pqn_qnet_cnn_apply: NULL argument");
  return pqn_launch_radam(theta, grad, m, v, L->total, count, lr_init, lr_end, lr_steps, max_norm, workspace, gnorm_out,
                          L->off_w1, w1b, recompute_norm, grad_reduce_blocks(L->total), (hipStream_t)stream);
}

__global__ void pack_w1b_kernel(const float *__restrict__ w1p, float *__restrict__ w1b) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= QN_H1 * QN_HID) return;
  const int frag = j >> 8, ln = (j >> 2) & 63, sx = j & 3;
  const int gi = frag >> 3, cb = frag & 7, kk = ln >> 4, jj = ln & 15;
  w1b[(((cb * 64 + gi) * 64 + (jj >> 2) * 16 + 4 * kk + sx) << 2) + (jj & 3)] = w1p[j];
}

extern "C" int pqn_qnet_cnn_pack_w1b(const pqn_cnn_layout_t *L, const float *theta, float *w1b, void *stream) {
  PQN_REQUIRE(L && theta && w1b, "pqn_qnet_cnn_pack_w1b: NULL argument");
  hipLaunchKernelGGL(pack_w1b_kernel, dim3(QN_H1 * QN_HID / 256), dim3(256), 0, (hipStream_t)stream,
                     theta + L->off_w1, w1b);
  return pqn_check_launch("pqn_qnet_cnn_pack_w1b");
}
