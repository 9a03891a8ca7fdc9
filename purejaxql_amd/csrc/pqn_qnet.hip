// pqn_qnet.hip -- fused MinAtar CNN Q-network kernels for gfx950 (f32, MFMA).
//
// Network = QNetwork(CNN) of the reference, purejaxql/pqn_minatar.py:24-69:
//   x/255 -> Conv3x3(C->16, VALID) -> LayerNorm(16) -> relu -> flatten(h,w,c)=1024
//         -> Dense(128) -> LayerNorm(128) -> relu -> Dense(A)
// The kernels never see the f32 [10,10,C] observation: they read the bit-packed
// grid (include/pqn_hotpath.h "obs_bits", 64 B/obs for Breakout instead of
// 1600 B) and build the conv's MFMA A operand from the window bits.
//
// Work decomposition (one 512-thread workgroup = 16 samples = one MFMA M-tile):
//   phase 1  conv as v_mfma_f32_16x16x4_f32 (16 positions x 16 channels per tile,
//            K = the 9C window bits), LDS transposition to one lane per position,
//            LayerNorm(16) + relu; h1 tile -> LDS
//   phase 2  fc1 as v_mfma_f32_16x16x4_f32: A = h1 tile (LDS, ds_read_b128),
//            B = fc1 kernel streamed from L2 in MFMA-fragment order (1 KB per
//            wave-instruction, global_load_dwordx4) through a 16-deep register ring;
//            8 waves x 1 column block.  (matmul_f16 layouts: v_mfma_f32_16x16x16_f16)
//   phase 3  LN(128)+relu+fc2 (+ eps-greedy epilogue), 16 / 32 lanes per sample
// The same phases run inside the persistent rollout kernel (one workgroup owns 16
// envs for all T steps) and as the forward half of the training kernel.
//
// Parameter layout ("kernel layout", pqn_cnn_layout): flax order, segment starts
// padded to 16 B, and the fc1 kernel W1[i][o] stored in MFMA C/D-fragment order
//   idx(i,o) = ((i/16 * 8 + o/16) * 64 + ((i%16)/4)*16 + o%16) * 4 + i%4
// which is at once (a) the B-operand fragment of the forward GEMM, with the K
// dimension permuted so one lane's float4 feeds 4 consecutive MFMAs, and (b) the
// accumulator layout in which the weight-gradient GEMM produces dW1, so
// gradient, moments and parameters share one layout and RAdam stays elementwise.
#include <stdlib.h>

#include <type_traits>

#include "pqn_common.h"
#include "pqn_env_rules.h"
#include "pqn_qnet_x3.h"
#include "pqn_qnet_pos.h"
#include "pqn_fold.h"
static bool pos_train_plan(const pqn_cnn_layout_t &L, int nb, const pqn_seeds_t &sd, pos_plan_t &plan);   // (defined next to the epoch gather)

struct CnnSmem {
  float *h1;      // [QN_TILE][QN_H1S]
  float *z;       // [QN_TILE][QN_ZS]
  float *wc;      // [KW][16] conv kernel, then bias[16], ln0 scale[16], ln0 bias[16]
  float *stg;     // [QN_WAVES][64 points][QN_STG] MFMA-layout -> point-per-lane transposition buffer
  float *hp;      // head parameters: b1[128] | ln1 scale[128] | ln1 bias[128] | w2[128][A] | b2[A]
  uint32_t *bits; // [QN_TILE][OW]
};

// ---------------------------------------------------------------------------
// conv 3x3xC -> 16 as MFMA.  A 16x16 output tile = 16 "points" (rows) x 16 channels; the
// reduction runs over the 9C window bits in steps of 4 (v_mfma_f32_16x16x4_f32):
//   A[i = l&15][kk = l>>4] = bit(point i, k = 4s+kk) ? 1/255 : 0     (built from the packed obs)
//   B[kk][j = l&15]        = Wc[k = 4s+kk][j]                         (held in registers)
//   D: lane (channel j = l&15) holds rows 4*(l>>4)+r, r = 0..3.
// Bit of point (py,px), window element k=(ky,kx,c): ((py+ky)*10 + px+kx)*C + c = base(point) + off(k).
// ---------------------------------------------------------------------------
// Per-point window masks: for position pos=(py,px) of one sample, word ky holds the 3C bits of window
// row ky (cells (py+ky, px..px+2), all channels) -- bit j of word ky is window element k = ky*3C + j.
// Computed once per sample by lane = pos; the MFMA A-operand builders below only shift and test.
template <int C>
PQN_D void window_masks(const uint32_t *row_bits, uint32_t *wm, int pos) {
  const int py = pos >> 3, px = pos & 7;
  uint32_t lo[3], hi[3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {   // all six LDS reads in flight before the first (possibly aliasing) store
    const int w = (((py + ky) * 10 + px) * C) >> 5;
    lo[ky] = row_bits[w];
    hi[ky] = row_bits[w + 1];
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int sh = (((py + ky) * 10 + px) * C) & 31;
    const uint64_t v = (((uint64_t)hi[ky] << 32) | lo[ky]) >> sh;
    wm[pos * 3 + ky] = (uint32_t)v & ((1u << (3 * C)) - 1u);
  }
}

// bit `sh` of w ? 1/255 : 0 in two VALU ops: v_bfe_i32 (1-bit signed field = 0 / ~0) and v_and_b32
PQN_D float bit_times_inv255(uint32_t w, int sh) {
  return __int_as_float(__builtin_amdgcn_sbfe((int)w, (uint32_t)sh, 1u) & __float_as_int(1.0f / 255.0f));
}

template <int C>
struct ConvMfma {
  static constexpr int NS = (9 * C + 3) / 4;
  static constexpr int RB = 3 * C;   // bits per window row
  int kyS[NS], shS[NS];              // lane constants: window row and bit of k = 4s + (lane>>4)
  float wk[NS];
  PQN_D void init(const float *wc, int lane) {
    const int kk = lane >> 4, o = lane & 15;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int k = 4 * s + kk;
      const bool ok = k < 9 * C;
      kyS[s] = ok ? k / RB : 0;
      shS[s] = ok ? k % RB : 0;
      wk[s] = ok ? wc[k * 16 + o] : 0.0f;
    }
  }
  PQN_D float a_of(const uint32_t (&m)[3], int s) const {
    // for C = 4 (RB = 12) the row index is the same for all four kk of a step: compile-time select
    const uint32_t w = (RB % 4 == 0) ? m[(4 * s) / RB] : (kyS[s] == 0 ? m[0] : (kyS[s] == 1 ? m[1] : m[2]));
    return bit_times_inv255(w, shS[s]);
  }
  // two tiles at once (independent accumulators hide the 40-cycle MFMA dependency); pA / pB = the
  // point (row of the window-mask table) this lane's A row stands for in each tile
  PQN_D void tile2(const uint32_t *wm, int pA, int pB, f32x4 &dA, f32x4 &dB) const {
    const uint32_t mA[3] = {wm[pA * 3], wm[pA * 3 + 1], wm[pA * 3 + 2]};
    const uint32_t mB[3] = {wm[pB * 3], wm[pB * 3 + 1], wm[pB * 3 + 2]};
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_of(mA, s), wk[s], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a_of(mB, s), wk[s], a1, 0, 0, 0);
    }
    dA = a0;
    dB = a1;
  }
};

// MFMA leaves a tile as (channel = lane&15, 4 points per lane); LayerNorm wants all 16 channels of a
// point in one lane (no cross-lane reductions, no 16x redundant statistics).  Transpose through LDS.
// in-place variant of the two helpers below: the staging slot of point p IS its 64-B output slot in the h1 row (stride
// 16, no pad); the 16-B quads of a slot are XOR-swizzled with (p >> 2) & 3, which keeps the per-lane ds_read_b128 of
// ln16 conflict-free without padding (the staged writes are 2-way conflicted b32 stores, 16 per sample)
PQN_D void stage_tile_inplace(float *row, int p0, const f32x4 &d, float bias, int lane) {
  const int o = lane & 15;
  const float v[4] = {d.x + bias, d.y + bias, d.z + bias, d.w + bias};
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int p = p0 + 4 * (lane >> 4) + rr;
    row[p * 16 + ((((o >> 2) ^ ((p >> 2) & 3))) << 2) + (o & 3)] = v[rr];
  }
}
PQN_D void ln16_point_inplace(const float *row, int p, float (&xhat)[16], float &rstd) {
  float v[16];
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const f32x4 t = *reinterpret_cast<const f32x4 *>(row + p * 16 + ((qd ^ ((p >> 2) & 3)) << 2));
    v[4 * qd] = t.x; v[4 * qd + 1] = t.y; v[4 * qd + 2] = t.z; v[4 * qd + 3] = t.w;
  }
  float sum = 0.f, sq = 0.f;
#pragma unroll
  for (int o = 0; o < 16; ++o) { sum += v[o]; sq = fmaf(v[o], v[o], sq); }
  const float mean = sum * (1.0f / 16.0f);
  const float var = fmaxf(sq * (1.0f / 16.0f) - mean * mean, 0.0f);
  rstd = rsqrt_exact(var + QN_LN_EPS);
#pragma unroll
  for (int o = 0; o < 16; ++o) xhat[o] = (v[o] - mean) * rstd;
}

PQN_D void stage_tile(float *stg, int p0, const f32x4 &d, float bias, int lane) {
  float *dst = stg + (p0 + 4 * (lane >> 4)) * QN_STG + (lane & 15);
  dst[0] = d.x + bias;
  dst[QN_STG] = d.y + bias;
  dst[2 * QN_STG] = d.z + bias;
  dst[3 * QN_STG] = d.w + bias;
}

// LayerNorm(16) statistics of staged point p (flax: var = E[x^2]-E[x]^2, clamped; eps inside rsqrt)
PQN_D void ln16_point(const float *stg, int p, float (&xhat)[16], float &rstd) {
  const f32x4 *src = reinterpret_cast<const f32x4 *>(stg + p * QN_STG);
  float v[16];
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const f32x4 t = src[qd];
    v[4 * qd] = t.x; v[4 * qd + 1] = t.y; v[4 * qd + 2] = t.z; v[4 * qd + 3] = t.w;
  }
  float sum = 0.f, sq = 0.f;
#pragma unroll
  for (int o = 0; o < 16; ++o) { sum += v[o]; sq = fmaf(v[o], v[o], sq); }
  const float mean = sum * (1.0f / 16.0f);
  const float var = fmaxf(sq * (1.0f / 16.0f) - mean * mean, 0.0f);
  rstd = rsqrt_exact(var + QN_LN_EPS);
#pragma unroll
  for (int o = 0; o < 16; ++o) xhat[o] = (v[o] - mean) * rstd;
}

// phase 1: h1 tile [16 samples][64 pos * 16 ch] = relu(LN(conv)).  Wave w owns samples QN_SPW*w ...
// KEEP: also return the normalised activations xhat[sample][channel] and 1/std of this lane's point, which the
// training kernel holds in registers until the LN0 backward (no conv recompute there).
// INPLACE: no staging buffer (s.stg unused) -- see stage_tile_inplace
template <int C, bool KEEP = false, bool X3 = false, bool INPLACE = false>
PQN_D void phase1_conv(const CnnSmem &s, int tid, float (*xkeep)[16] = nullptr, float *rkeep = nullptr) {
  using Cfg = CnnCfg<C>;
  const int lane = tid & 63, wave = tid >> 6;
  typename std::conditional<X3, ConvX3<C>, ConvMfma<C>>::type cv;
  cv.init(s.wc, lane);
  const float *bc = s.wc + Cfg::KW * 16;
  const float bias = bc[lane & 15];
  float *stg = s.stg + wave * 64 * QN_STG;
  uint32_t *wm = reinterpret_cast<uint32_t *>(s.z) + wave * 192;   // z tile is not live yet
  // Software pipeline over the wave's samples: the conv MFMAs of sample mm+1 are issued before the
  // LayerNorm / store of sample mm (VALU + LDS only), so the matrix core works in the shadow of the
  // VALU phase instead of idling while both waves of the SIMD normalise.
  const int i = lane & 15;
  f32x4 d[2][4];
  auto conv_mfma = [&](int mm, f32x4(&out)[4]) {
    window_masks<C>(s.bits + (QN_SPW * wave + mm) * Cfg::OW, wm, lane);
    if constexpr (X3) {
      cv.tile4(wm, i, lane >> 4, out);
    } else {
      cv.tile2(wm, i, 16 + i, out[0], out[1]);
      cv.tile2(wm, 32 + i, 48 + i, out[2], out[3]);
    }
  };
  conv_mfma(0, d[0]);
#pragma unroll
  for (int mm = 0; mm < QN_SPW; ++mm) {
    const int m = QN_SPW * wave + mm;
    if (mm + 1 < QN_SPW) conv_mfma(mm + 1, d[(mm + 1) & 1]);
    float xhat[16], rstd;
    if constexpr (INPLACE) {
      float *row = s.h1 + m * QN_H1S;
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) stage_tile_inplace(row, 16 * pb, d[mm & 1][pb], bias, lane);
      ln16_point_inplace(row, lane, xhat, rstd);
    } else {
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) stage_tile(stg, 16 * pb, d[mm & 1][pb], bias, lane);
      ln16_point(stg, lane, xhat, rstd);   // lane = position
    }
    if (KEEP) {
#pragma unroll
      for (int c = 0; c < 16; ++c) xkeep[mm][c] = xhat[c];
      rkeep[mm] = rstd;
    }
    f32x4 *dst = reinterpret_cast<f32x4 *>(s.h1 + m * QN_H1S + lane * 16);
    const int sw = (INPLACE && QN_H1_SWIZZLE) ? ((lane >> 2) & 3) : 0;   // h1_slot: the quads of position `lane`
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      f32x4 y;
      y.x = fmaxf(fmaf(xhat[4 * qd + 0], bc[16 + 4 * qd + 0], bc[32 + 4 * qd + 0]), 0.0f);
      y.y = fmaxf(fmaf(xhat[4 * qd + 1], bc[16 + 4 * qd + 1], bc[32 + 4 * qd + 1]), 0.0f);
      y.z = fmaxf(fmaf(xhat[4 * qd + 2], bc[16 + 4 * qd + 2], bc[32 + 4 * qd + 2]), 0.0f);
      y.w = fmaxf(fmaf(xhat[4 * qd + 3], bc[16 + 4 * qd + 3], bc[32 + 4 * qd + 3]), 0.0f);
      dst[qd ^ sw] = y;
    }
  }
}

// ---------------------------------------------------------------------------
// phase 2: z[16][128] = h1[16][1024] x W1 (packed) via v_mfma_f32_16x16x4_f32.
// wave w owns column blocks 2w, 2w+1.  Operand maps (cdna guide 3): A[i=l&15][k=l>>4],
// B[k=l>>4][j=l&15], D: col=l&15, row=4*(l>>4)+reg.
// ---------------------------------------------------------------------------
// ablate: 0 = normal; 2 = no global B loads; 3 = no MFMA (loads only).  Profiling hook.
template <int ABL = 0, int PF = 16>   // PF: K groups in flight per wave (16 KB at 16: covers an L2-miss round trip)
PQN_D void phase2_fc1(const CnnSmem &s, const float *__restrict__ w1p, int tid, int tile = -1) {
  constexpr int CBW = 8 / QN_WAVES > 0 ? 8 / QN_WAVES : 1;  // column blocks per wave
  static_assert(QN_WAVES * CBW == 8, "8 column blocks of 16 outputs");
  const int lane = tid & 63, wave = tid >> 6;
  const int cb0 = CBW * wave;
  // every workgroup streams the same 512 KB of W1: start each one at a different K group (and wrap) so
  // the CUs of an XCD do not sweep the same L2 channel at the same moment.  The K order only permutes
  // the f32 summation order of the tile.
  // (`tile` = index of the tile inside its seed when seeds are batched along grid.x, so that a seed's
  // summation order -- hence its results, bit for bit -- does not depend on how many seeds share the launch)
  if (tile < 0) tile = blockIdx.x;
  const int rot = (tile * 8 + (tile >> 3)) & 63;
  const f32x4 *wp = reinterpret_cast<const f32x4 *>(w1p);
  const float *arow = s.h1 + (lane & 15) * QN_H1S + 4 * (lane >> 4);
  // two accumulator sets (even / odd K groups): consecutive MFMAs of a wave are independent, so the
  // 40-cycle dependent-accumulator latency never gates the 32-cycle issue rate
  f32x4 acc[CBW], acc2[CBW];
#pragma unroll
  for (int c = 0; c < CBW; ++c) { acc[c] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  f32x4 b[PF][CBW];
#pragma unroll
  for (int i = 0; i < PF; ++i)
#pragma unroll
    for (int c = 0; c < CBW; ++c) b[i][c] = wp[(((i + rot) & 63) * 8 + cb0 + c) * 64 + lane];
  // the last PF groups are peeled (no prefetch): a conditional prefetch inside the loop makes the
  // compiler's vmcnt bookkeeping assume it was not issued and drain the ring once per iteration
  f32x4 a_next = *reinterpret_cast<const f32x4 *>(arow + 16 * (rot & 63));
  auto group_block = [&](int g, auto more_t) {
    constexpr bool more = decltype(more_t)::value;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const f32x4 a = a_next;   // A fragment one group ahead: its LDS latency hides behind the previous group's MFMAs
      a_next = *reinterpret_cast<const f32x4 *>(arow + 16 * ((g + i + 1 + rot) & 63));
      __builtin_amdgcn_sched_barrier(0);   // the read goes out before this group's MFMAs, not after them
      if (ABL == 3) {
#pragma unroll
        for (int c = 0; c < CBW; ++c) acc[c] += b[i][c] + a;
      } else {
#pragma unroll
        for (int c = 0; c < CBW; ++c) {
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[i][c].x, acc[c], 0, 0, 0);
          acc2[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[i][c].y, acc2[c], 0, 0, 0);
          acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[i][c].z, acc[c], 0, 0, 0);
          acc2[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[i][c].w, acc2[c], 0, 0, 0);
        }
      }
      // reload the ring slot in place, AFTER its consumers were issued: the loop-carried registers are then
      // the load destinations themselves (no copies that would have to wait for the data), and every wait
      // is exactly vmcnt(PF-1).  The scheduling barrier pins the reload to its group.
      if (ABL != 2 && more) {
#pragma unroll
        for (int c = 0; c < CBW; ++c) b[i][c] = wp[(((g + i + PF + rot) & 63) * 8 + cb0 + c) * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#pragma unroll 1
  for (int g = 0; g < QN_H1 / 16 - PF; g += PF) group_block(g, std::true_type{});
  group_block(QN_H1 / 16 - PF, std::false_type{});
#pragma unroll
  for (int c = 0; c < CBW; ++c) acc[c] += acc2[c];
  const int col = lane & 15, r0 = 4 * (lane >> 4);
#pragma unroll
  for (int c = 0; c < CBW; ++c) {
    float *zp = s.z + r0 * QN_ZS + 16 * (cb0 + c) + col;
    zp[0] = acc[c].x;
    zp[QN_ZS] = acc[c].y;
    zp[2 * QN_ZS] = acc[c].z;
    zp[3 * QN_ZS] = acc[c].w;
  }
}

// fp16-operand variant (pqn_cnn_layout_t.matmul_f16): same tile, same fragment order, one
// v_mfma_f32_16x16x16_f16 per 16-wide K group instead of four f32 MFMAs.  Lane (i = l&15, kk = l>>4) holds
// k = 16g + 4kk + 0..3 of its row as 4 halves -- exactly the 4 floats of the f32 fragment, so the A operand is
// the same ds_read_b128 converted on the fly and the B operand is the 8-byte half of the f32 fragment
// (w1h, kept in step by the optimizer).  Accumulation is f32.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
PQN_D f16x4 to_f16x4(const f32x4 v) { return f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; }
// v_mfma_f32_16x16x16_f16 with the accumulator TIED, as volatile inline asm -- the same discipline as x3_mfma_tied (the
// builtin let the compiler give this instruction a destination tuple partially overlapping its accumulator input in the
// 7- and 10-channel kernels, tools/check_mfma_overlap.py; the bf16 form of that pattern produced run-to-run different
// results).  `s_nop 1` covers a VALU-written operand; the result is read only by the next tied MFMA or after f16_drain.
PQN_D f32x4 f16_mfma_tied(const f16x4 &a, const f16x4 &b, f32x4 c) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
PQN_D void f16_drain(f32x4 &a, f32x4 &b) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b)); }
PQN_D void f16_drain(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }

template <int PF = 16>
PQN_D void phase2_fc1_f16(const CnnSmem &s, const _Float16 *__restrict__ w1h, int tid, int tile = -1) {
  static_assert(QN_WAVES == 8, "one column block of 16 outputs per wave");
  const int lane = tid & 63, wave = tid >> 6;
  if (tile < 0) tile = blockIdx.x;
  const int rot = (tile * 8 + (tile >> 3)) & 63;
  const f16x4 *wp = reinterpret_cast<const f16x4 *>(w1h);
  const float *arow = s.h1 + (lane & 15) * QN_H1S + 4 * (lane >> 4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  f16x4 b[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) b[i] = wp[(((i + rot) & 63) * 8 + wave) * 64 + lane];
  f32x4 a_next = *reinterpret_cast<const f32x4 *>(arow + 16 * (rot & 63));
  auto group_block = [&](int g, auto more_t) {
    constexpr bool more = decltype(more_t)::value;
#pragma unroll
    for (int i = 0; i < PF; i += 2) {
      const f32x4 a0 = a_next;
      const f32x4 a1 = *reinterpret_cast<const f32x4 *>(arow + 16 * ((g + i + 1 + rot) & 63));
      a_next = *reinterpret_cast<const f32x4 *>(arow + 16 * ((g + i + 2 + rot) & 63));
      __builtin_amdgcn_sched_barrier(0);
      acc = f16_mfma_tied(to_f16x4(a0), b[i], acc);
      acc2 = f16_mfma_tied(to_f16x4(a1), b[i + 1], acc2);
      if (more) {
        b[i] = wp[(((g + i + PF + rot) & 63) * 8 + wave) * 64 + lane];
        b[i + 1] = wp[(((g + i + 1 + PF + rot) & 63) * 8 + wave) * 64 + lane];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#pragma unroll 1
  for (int g = 0; g < QN_H1 / 16 - PF; g += PF) group_block(g, std::true_type{});
  group_block(QN_H1 / 16 - PF, std::false_type{});
  f16_drain(acc, acc2);
  acc += acc2;
  const int col = lane & 15, r0 = 4 * (lane >> 4);
  float *zp = s.z + r0 * QN_ZS + 16 * wave + col;
  zp[0] = acc.x;
  zp[QN_ZS] = acc.y;
  zp[2 * QN_ZS] = acc.z;
  zp[3 * QN_ZS] = acc.w;
}

// phase 2, bf16x3: z[16][128] = h1 x W1 on the bf16 matrix core.  Wave w = (K half kh = w&1, column-block pair
// cp = w>>1): 16 K steps of 32, per step ONE split of the h1 fragment (shared by the two column blocks) and 12 MFMAs;
// weight planes streamed through a PF-step register ring (6 dwordx4 per step).  The two K halves are added through
// LDS (the staging buffer is idle here).  Contains one __syncthreads(): every thread of the workgroup must call it.
// NT = 2 (the pair kernel): TWO 16-sample tiles share every weight fragment -- what bounds this phase is the 64 B/clk
// vector-memory path of the CU streaming the 768 KB of planes, so a second tile costs MFMAs and one more A split but
// no more weight traffic.  Partial sums of the two K halves are then folded IN PLACE in the z tiles (no park buffer).
struct Fc1NoSide { PQN_D void operator()(int, int) const {} };
// side(j, i), j = 0..15 (i = j % PF, a compile-time value after unrolling): independent work issued once per K step behind
// the step's weight reload (the pair kernel stores h1^T there: the stores ride in the shadow of the weight stream
// instead of a phase of their own)
template <int PF = 3, int NT = 1, class Side = Fc1NoSide>
PQN_D void phase2_fc1_x3(const CnnSmem &s, const float *__restrict__ planes, int tid, int tile = -1,
                         const CnnSmem *s2 = nullptr, Side side = Side()) {
  static_assert(QN_WAVES == 8, "2 K halves x 4 column-block pairs");
  constexpr int NS = 16;   // K steps per wave
  // accumulator copies per (tile, column block, kind).  ONE for both forms: the single-tile and the pair kernel must sum
  // in the same order, because a seed trained alone (single-tile kernel) and the same seed inside a 16-seed launch (pair
  // kernel) are required to agree bit for bit.  Reuse distance is then 4 MFMAs (single) / 8 (pair).
  constexpr int NK = 1;
  const int lane = tid & 63, wave = tid >> 6;
  const int kh = wave & 1, cp = wave >> 1;
  if (tile < 0) tile = blockIdx.x;
  // de-phase the weight stream across workgroups.  The rotation sets the order the K steps are summed in, so it is a
  // function of the PAIR index in both forms (NT = 1: tile = tile index, NT = 2: tile = pair index): a tile gets the same
  // bits from the single-tile and from the pair kernels.
  const int pr = NT == 2 ? tile : tile >> 1;
  const int rot = (pr * 5 + (pr >> 4)) & (NS - 1);
  const u32x4 *wf = reinterpret_cast<const u32x4 *>(planes);
  const float *arow[NT];
  float *zt[NT];
  arow[0] = s.h1 + (lane & 15) * QN_H1S;     // + h1_slot(32 st + kq4) (+ 16): the lane's two quads of K step st
  zt[0] = s.z;
  if (NT == 2) { arow[NT - 1] = s2->h1 + (lane & 15) * QN_H1S; zt[NT - 1] = s2->z; }
  const int kq4 = 4 * (lane >> 4);
  auto wfrag = [&](int p, int st, int c) {   // plane p, K step st (global), column block 2cp + c
    return wf[(size_t)p * (X3_PLANE / 8) + ((st * 8 + 2 * cp + c) * 64 + lane)];
  };
  // Independent accumulators {small, leading terms} x column block x tile: a dependent v_mfma_f32_16x16x32_bf16 issues
  // ~90 counter ticks after its producer and the pipe takes one per ~10 (tools/ubench/mfma_issue.hip); the MFMAs of a
  // step are ordered round-robin over them.  (The phase is bound by the weight stream, not by issue: 4 vs 8
  // accumulators made no measurable difference.)
  f32x4 acc_b[NT][2][NK], acc_s[NT][2][NK];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < NK; ++k) { acc_b[t][c][k] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_s[t][c][k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  auto step_mfma = [&](const X3Frag (&af)[NT], const X3Frag (&bq)[2]) {
    // products in the order (l,h) (m,h) (h,l) (h,m) (m,m) (h,h): small and leading terms alternate
#define FC1_PROD(AP, BP, ACC, K)                                                                                           \
    if constexpr (NT == 2)                                                                                                  \
      x3_grp4(ACC[0][0][K], af[0].AP, bq[0].BP, ACC[0][1][K], af[0].AP, bq[1].BP, ACC[NT - 1][0][K], af[NT - 1].AP, bq[0].BP, \
              ACC[NT - 1][1][K], af[NT - 1].AP, bq[1].BP);                                                                  \
    else                                                                                                                    \
      x3_grp2(ACC[0][0][K], af[0].AP, bq[0].BP, ACC[0][1][K], af[0].AP, bq[1].BP);
    FC1_PROD(l, h, acc_s, 0)
    FC1_PROD(m, h, acc_b, 0)
    FC1_PROD(h, l, acc_s, NK - 1)
    FC1_PROD(h, m, acc_b, NK - 1)
    FC1_PROD(m, m, acc_s, 0)
    FC1_PROD(h, h, acc_b, 0)
#undef FC1_PROD
  };
  u32x4 ring[PF][2][3];
#pragma unroll
  for (int i = 0; i < PF; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) ring[i][c][pl] = wfrag(pl, NS * kh + ((i + rot) & (NS - 1)), c);
  const int st0 = NS * kh + (rot & (NS - 1));
  f32x4 a0n[NT], a1n[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    a0n[t] = *reinterpret_cast<const f32x4 *>(arow[t] + h1_slot(32 * st0 + kq4));
    a1n[t] = *reinterpret_cast<const f32x4 *>(arow[t] + h1_slot(32 * st0 + kq4) + 16);
  }
  auto one_step = [&](int g, int i, bool reload) {
    f32x4 a0[NT], a1[NT];
    const int stn = NS * kh + ((g + i + 1 + rot) & (NS - 1));   // next step's A fragments (wraps harmlessly at the end)
    const int sln = h1_slot(32 * stn + kq4);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      a0[t] = a0n[t]; a1[t] = a1n[t];
      a0n[t] = *reinterpret_cast<const f32x4 *>(arow[t] + sln);
      a1n[t] = *reinterpret_cast<const f32x4 *>(arow[t] + sln + 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    X3Frag af[NT], bq[2];
#pragma unroll
    for (int t = 0; t < NT; ++t) af[t] = x3_split8(a0[t], a1[t]);
#pragma unroll
    for (int c = 0; c < 2; ++c) { bq[c].h = ring[i][c][0]; bq[c].m = ring[i][c][1]; bq[c].l = ring[i][c][2]; }
    step_mfma(af, bq);
    if (reload) {
      const int st = NS * kh + ((g + i + PF + rot) & (NS - 1));
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ring[i][c][pl] = wfrag(pl, st, c);
    }
    side(g + i, i);
    __builtin_amdgcn_sched_barrier(0);
  };
  int g = 0;
#pragma unroll 1
  for (; g + 2 * PF <= NS; g += PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) one_step(g, i, true);
  }
  // tail: the remaining NS - g steps; reload only while a later step still needs the slot (compile-time conditions:
  // a reload under a run-time condition makes the compiler drain the load queue)
  const int g0 = g;
#pragma unroll
  for (int q = 0; q < 2 * PF; ++q) {
    const int i = q % PF;
    if (g0 + q < NS) one_step(g0 + q - i, i, g0 + q + PF < NS);
  }
  // fold the two K halves
  f32x4 tot[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    x3_drain(acc_b[t][0][0], acc_s[t][0][0], acc_b[t][1][0], acc_s[t][1][0]);
    if (NK == 2) x3_drain(acc_b[t][0][NK - 1], acc_s[t][0][NK - 1], acc_b[t][1][NK - 1], acc_s[t][1][NK - 1]);
#pragma unroll
    for (int c = 0; c < 2; ++c)
      tot[t][c] = NK == 2 ? (acc_b[t][c][0] + acc_b[t][c][NK - 1]) + (acc_s[t][c][0] + acc_s[t][c][NK - 1])
                          : acc_b[t][c][0] + acc_s[t][c][0];
  }
  const int col = lane & 15, r0 = 4 * (lane >> 4);
  if (NT == 1) {   // kh = 1 parks its partial tiles in the (idle) staging buffer, kh = 0 adds and writes z
    f32x4 *park = reinterpret_cast<f32x4 *>(s.stg);
    if (kh == 1) {
#pragma unroll
      for (int c = 0; c < 2; ++c) park[(cp * 2 + c) * 64 + lane] = tot[0][c];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 acc = tot[0][c] + park[(cp * 2 + c) * 64 + lane];
        float *zp = s.z + r0 * QN_ZS + 16 * (2 * cp + c) + col;
        zp[0] = acc.x; zp[QN_ZS] = acc.y; zp[2 * QN_ZS] = acc.z; zp[3 * QN_ZS] = acc.w;
      }
    }
  } else {         // in place: kh = 1 writes its partial into the z tiles, kh = 0 adds its own on top
    if (kh == 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float *zp = zt[t] + r0 * QN_ZS + 16 * (2 * cp + c) + col;
          zp[0] = tot[t][c].x; zp[QN_ZS] = tot[t][c].y; zp[2 * QN_ZS] = tot[t][c].z; zp[3 * QN_ZS] = tot[t][c].w;
        }
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float *zp = zt[t] + r0 * QN_ZS + 16 * (2 * cp + c) + col;
          zp[0] += tot[t][c].x; zp[QN_ZS] += tot[t][c].y; zp[2 * QN_ZS] += tot[t][c].z; zp[3 * QN_ZS] += tot[t][c].w;
        }
    }
  }
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
    v[r] = s.z[m * QN_ZS + o] + s.hp[o];
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
    h2[r] = fmaxf(fmaf(xh[r], s.hp[128 + o], s.hp[256 + o]), 0.0f);
  }
#pragma unroll
  for (int a = 0; a < QN_MAXA; ++a) {
    float part = 0.f;
    if (a < L.a) {
#pragma unroll
      for (int r = 0; r < 8; ++r) part = fmaf(h2[r], s.hp[384 + (sub + 16 * r) * L.a + a], part);
    }
    q[a] = group16_sum(part) + (a < L.a ? s.hp[384 + 128 * L.a + a] : 0.0f);
  }
}

// The same block in two halves: all global loads first (into registers), LDS stores later -- the training kernel
// issues its dependent observation gather in between, so the two latencies overlap.
template <int C>
struct TileParams {
  using Cfg = CnnCfg<C>;
  static constexpr int NWC = (Cfg::KW * 16 + 48 + QN_THREADS - 1) / QN_THREADS;
  static constexpr int NW2 = (128 * QN_MAXA + QN_THREADS - 1) / QN_THREADS;
  float wc[NWC], hp, w2[NW2], b2;
  PQN_D void load(const float *__restrict__ theta, const pqn_cnn_layout_t &L, int tid) {
#pragma unroll
    for (int k = 0; k < NWC; ++k) {
      const int i = tid + k * QN_THREADS;
      wc[k] = i < Cfg::KW * 16 + 48 ? theta[L.off_wc + i] : 0.0f;
    }
    hp = tid < 384 ? theta[L.off_b1 + tid] : 0.0f;
#pragma unroll
    for (int k = 0; k < NW2; ++k) {
      const int i = tid + k * QN_THREADS;
      w2[k] = i < 128 * L.a ? theta[L.off_w2 + i] : 0.0f;
    }
    b2 = tid < L.a ? theta[L.off_b2 + tid] : 0.0f;
  }
  PQN_D void store(const CnnSmem &s, const pqn_cnn_layout_t &L, int tid) const {
#pragma unroll
    for (int k = 0; k < NWC; ++k) {
      const int i = tid + k * QN_THREADS;
      if (i < Cfg::KW * 16 + 48) s.wc[i] = wc[k];
    }
    if (tid < 384) s.hp[tid] = hp;
#pragma unroll
    for (int k = 0; k < NW2; ++k) {
      const int i = tid + k * QN_THREADS;
      if (i < 128 * L.a) s.hp[384 + i] = w2[k];
    }
    if (tid < L.a) s.hp[384 + 128 * L.a + tid] = b2;
  }
};

template <int C>
PQN_D void load_tile_common(const CnnSmem &s, const float *__restrict__ theta, const pqn_cnn_layout_t &L, int tid) {
  using Cfg = CnnCfg<C>;
  for (int i = tid; i < Cfg::KW * 16 + 48; i += QN_THREADS) s.wc[i] = theta[L.off_wc + i];  // kernel|bias|ln0s|ln0b contiguous
  // head parameters (read long after the prologue: their L2 latency is hidden behind phases 1-2)
  for (int i = tid; i < 384; i += QN_THREADS) s.hp[i] = theta[L.off_b1 + i];
  for (int i = tid; i < 128 * L.a; i += QN_THREADS) s.hp[384 + i] = theta[L.off_w2 + i];
  if (tid < L.a) s.hp[384 + 128 * L.a + tid] = theta[L.off_b2 + tid];
}

template <int C>
PQN_D CnnSmem carve_smem(char *base) {
  using Cfg = CnnCfg<C>;
  CnnSmem s;
  s.h1 = reinterpret_cast<float *>(base);
  s.z = s.h1 + QN_TILE * QN_H1S;
  s.wc = s.z + QN_TILE * QN_ZS;
  s.stg = s.wc + ((Cfg::KW * 16 + 48 + 3) & ~3);
  s.hp = s.stg + QN_WAVES * 64 * QN_STG;
  s.bits = reinterpret_cast<uint32_t *>(s.hp + QN_HP_FLOATS);
  return s;
}

template <int C>
constexpr size_t cnn_smem_bytes() {
  using Cfg = CnnCfg<C>;
  return sizeof(float) * (QN_TILE * QN_H1S + QN_TILE * QN_ZS + ((Cfg::KW * 16 + 48 + 3) & ~3) + QN_WAVES * 64 * QN_STG +
                          QN_HP_FLOATS) +
         sizeof(uint32_t) * (QN_TILE * Cfg::OW + 4);
}

// ---------------------------------------------------------------------------
// forward (+ optional eps-greedy epilogue): network.apply(train=False) at
// pqn_minatar.py:184-196,227-234,380-390.
//   idx   (nullable): gather -- sample j of the launch reads obs_bits[idx[j]]
// ---------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(QN_THREADS) void qnet_cnn_fwd_kernel(int n, const uint32_t *__restrict__ obs_bits,
                                                           const float *__restrict__ theta, pqn_cnn_layout_t L,
                                                           float *__restrict__ q_out, int32_t *__restrict__ action,
                                                           float *__restrict__ qmax, float eps, uint64_t key,
                                                           const float *__restrict__ eps_dev,
                                                           const uint64_t *__restrict__ key_dev, int ablate) {
  using Cfg = CnnCfg<C>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const CnnSmem s = carve_smem<C>(smem_raw);
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * QN_TILE;
  load_tile_common<C>(s, theta, L, tid);
  for (int i = tid; i < QN_TILE * Cfg::OW; i += QN_THREADS) {
    const int le = i / Cfg::OW;
    s.bits[i] = (e0 + le < n) ? obs_bits[(size_t)e0 * Cfg::OW + i] : 0u;
  }
  if (tid < 4) s.bits[QN_TILE * Cfg::OW + tid] = 0u;  // b[w+1] guard word
  __syncthreads();
  if (ablate != 1 && ablate != 6 && ablate != 7) {
    if (L.matmul_f16 == 2) phase1_conv<C, false, true, true>(s, tid);
    else phase1_conv<C>(s, tid);
  }
  __syncthreads();
  if (L.matmul_f16 == 1) phase2_fc1_f16<16>(s, reinterpret_cast<const _Float16 *>(theta + L.off_w1h), tid);
  else if (L.matmul_f16 == 2) phase2_fc1_x3<3>(s, theta + L.off_w1h, tid);
  else if (ablate == 0 || ablate == 1 || ablate == 5) phase2_fc1<0>(s, theta + L.off_w1, tid);
  else if (ablate == 2) phase2_fc1<2>(s, theta + L.off_w1, tid);
  else if (ablate == 3) phase2_fc1<3>(s, theta + L.off_w1, tid);
  __syncthreads();
  if (ablate == 5 || ablate == 7) return;   // profiling: no head
  if (tid >= 256) return;  // the head needs 16 lanes per sample
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
      if (key_dev) key = *key_dev;   // graph-replayable launches read key / eps from device memory
      if (eps_dev) eps = *eps_dev;
      uint32_t o0, o1;
      pqn_bits(key, (uint32_t)e, PQN_STREAM_ACT, o0, o1);
      const float u = pqn_uniform(o0);
      const int rnd = (int)pqn_randint(o1, (uint32_t)L.a);
      action[e] = (u < eps) ? rnd : best;
    }
  }
}


// ===========================================================================
// Persistent rollout: the whole SAMPLE PHASE of one update (`_step_env` scan, pqn_minatar.py:181-220) plus
// the bootstrap forward (:227-235) in ONE launch.  Envs are independent and the parameters are frozen
// during the rollout, so a workgroup keeps its 16 envs for all T steps: env + LogWrapper state stay in
// registers of the 16 owner lanes, the next observation is written as a packed row straight into the LDS
// tile the next forward reads, and only the transition record goes to HBM.  No kernel boundary, no
// prologue and no observation round trip between steps.
// ===========================================================================
template <int C, class Env, int MODE>
__global__ __launch_bounds__(QN_THREADS) void qnet_cnn_rollout_kernel(
    int n, int t_len, uint32_t *__restrict__ state, uint32_t *__restrict__ bits_all, const float *__restrict__ theta,
    pqn_cnn_layout_t L, int32_t *__restrict__ action, float *__restrict__ qmax, float *__restrict__ reward,
    uint8_t *__restrict__ done, float *__restrict__ discount, float *__restrict__ rer, int32_t *__restrict__ rel,
    int32_t *__restrict__ ts, float *__restrict__ last_q, const float *__restrict__ eps_dev,
    const uint64_t *__restrict__ keys, float rscale, int store_obs, int n_per_seed, long long theta_stride,
    int keys_stride) {
  using Cfg = CnnCfg<C>;
  static_assert(Cfg::OW == Env::OBS_WORDS, "packed observation width");
  int e_off = 0;          // first env of this tile's seed: the RNG counters use the env index inside the seed
  if (n_per_seed > 0) {   // seed batching: env tile -> seed (n_per_seed % 16 == 0); per-seed parameters and step keys
    const int seed = (blockIdx.x * QN_TILE) / n_per_seed;
    theta += seed * theta_stride;
    keys += (size_t)seed * keys_stride;
    e_off = seed * n_per_seed;
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const CnnSmem s = carve_smem<C>(smem_raw);
  const int tid = threadIdx.x;
  const int e0 = blockIdx.x * QN_TILE;
  const int m = (tid >> 4) & 15, sub = tid & 15, e = e0 + m;
  const int e_rng = e - e_off;
  const bool owner = tid < 256 && sub == 0 && e < n;   // the lane that owns env e
  const size_t bstride = (size_t)n * Cfg::OW;
  load_tile_common<C>(s, theta, L, tid);
  for (int i = tid; i < QN_TILE * Cfg::OW; i += QN_THREADS) {
    const int le = i / Cfg::OW;
    s.bits[i] = (e0 + le < n) ? bits_all[(size_t)e0 * Cfg::OW + i] : 0u;
  }
  if (tid < 4) s.bits[QN_TILE * Cfg::OW + tid] = 0u;
  Env env;
  LogRec log;
  if (owner) {
    uint32_t w[Env::ENV_WORDS];
#pragma unroll
    for (int i = 0; i < Env::ENV_WORDS; ++i) w[i] = state[(size_t)i * n + e];
    env.unpack(w);
    log.load(state, n, e, Env::ENV_WORDS);
  }
  const float eps = *eps_dev;
#pragma unroll 1
  for (int t = 0; t <= t_len; ++t) {
    __syncthreads();   // s.bits holds obs_t of the tile (and every previous reader of the tiles is done)
    if (MODE == 2) phase1_conv<C, false, true, true>(s, tid);
    else phase1_conv<C>(s, tid);
    __syncthreads();
    if (MODE == 1) phase2_fc1_f16<16>(s, reinterpret_cast<const _Float16 *>(theta + L.off_w1h), tid, (e0 - e_off) / QN_TILE);
    else if (MODE == 2) phase2_fc1_x3<2>(s, theta + L.off_w1h, tid, (e0 - e_off) / QN_TILE);
    else phase2_fc1<0, 8>(s, theta + L.off_w1, tid, (e0 - e_off) / QN_TILE);   // 8 in flight: the env state lives in registers across this loop
    __syncthreads();
    if (tid < 256) {
      float q[QN_MAXA], h2[8], xh[8], rstd;
      phase3_head(s, theta, L, tid, q, h2, xh, rstd);
      if (owner) {
        int best = 0;
        float bv = q[0];
#pragma unroll
        for (int a = 1; a < QN_MAXA; ++a)
          if (a < L.a && q[a] > bv) { bv = q[a]; best = a; }
        if (t == t_len) {
          if (last_q) last_q[e] = bv;                      // bootstrap value of obs_T
        } else {
          const uint64_t key = keys[t];
          uint32_t o0, o1;
          pqn_bits(key, (uint32_t)e_rng, PQN_STREAM_ACT, o0, o1);
          const int act = (pqn_uniform(o0) < eps) ? (int)pqn_randint(o1, (uint32_t)L.a) : best;
          int dn = 0;
          const float r = env.step(act, key, (uint32_t)e_rng, dn);
          log.step(r, dn);
          if (dn) env.reset(key, (uint32_t)e_rng);             // gymnax auto-reset
          const size_t o = (size_t)t * n + e;
          if (action) action[o] = act;
          if (qmax) qmax[o] = bv;
          if (reward) reward[o] = r * rscale;
          if (done) done[o] = (uint8_t)dn;
          if (discount) discount[o] = dn ? 0.0f : 1.0f;
          if (rer) rer[o] = log.ret_ret;
          if (rel) rel[o] = log.ret_len;
          if (ts) ts[o] = log.timestep;
          env.obs_bits(&s.bits[m * Cfg::OW]);              // obs_{t+1} straight into the LDS tile
        }
      }
    }
    if (t == t_len) break;
    __syncthreads();
    // transition record: packed obs_{t+1} of the tile (the training kernels gather from it); an evaluation
    // rollout (store_obs = 0) keeps only the running observation, in slot 0
    if (store_obs || t + 1 == t_len) {
      uint32_t *dst = bits_all + (store_obs ? (size_t)(t + 1) * bstride : (size_t)0);
      for (int i = tid; i < QN_TILE * Cfg::OW; i += QN_THREADS) {
        const int le = i / Cfg::OW;
        if (e0 + le < n) dst[(size_t)e0 * Cfg::OW + i] = s.bits[i];
      }
    }
  }
  if (owner) {
    uint32_t w[Env::ENV_WORDS];
    env.pack(w);
#pragma unroll
    for (int i = 0; i < Env::ENV_WORDS; ++i) state[(size_t)i * n + e] = w[i];
    log.store(state, n, e, Env::ENV_WORDS);
  }
}

// ===========================================================================
// TRAINING.  One optimizer step of _learn_phase (pqn_minatar.py:266-297) =
//   T1 qnet_cnn_train_kernel   per 16-sample tile: forward, loss gradient, backward through
//                              head / LN1 / fc1 (dgrad MFMA) / relu / LN0 / conv; emits dz^T
//                              and per-tile partial sums of every "small" gradient
//   T2 qnet_fc1_wgrad_kernel   dW1 = h1^T x dz as an LDS-staged MFMA GEMM on the operands T1 left in the
//                              workspace, split-K over 256-sample slabs, output in fragment layout
//                              (matmul_f16: qnet_fc1_wgrad_f16_kernel on tile-major fp16 operands)
//   T3 qnet_grad_reduce_kernel deterministic fold of the partials into the flat gradient
//                              (+ block sums of squares), then radam_apply (pqn_algo.hip).
// With seed batching (pqn_seeds_t) grid.y (T2: grid.z) is the seed and every pointer is offset by the
// seed's slice of the stacked buffers.
// "Small" partial record per tile (floats): [conv kernel KW*16 | conv bias 16 | ln0 scale 16 |
// ln0 bias 16 | b1 128 | ln1 scale 128 | ln1 bias 128 | w2 128*A | b2 A | loss | sum q_a].
// ===========================================================================
template <int C>
struct TrainCfg {
  using Cfg = CnnCfg<C>;
  static constexpr int CONVBLK = Cfg::KW * 16 + 48;
  static constexpr int SCR = (3 * QN_TILE * QN_ZS > Cfg::KW * 64) ? 3 * QN_TILE * QN_ZS : Cfg::KW * 64;
};

struct TrainSmem {
  CnnSmem n;
  float *scr;        // 3 x [16][QN_ZS] tiles, later [KW][4][16] conv-wgrad partials
  float *gs;         // [16] per-sample loss gradient g_m = (q_a - target)/B
  float *tgt;        // [16] per-sample target (gathered at kernel start, consumed by the head)
  int *act;          // [16]
  float *red;        // [QN_WAVES][48] cross-wave reduction of conv bias / ln0 grads
};

template <int C>
PQN_D TrainSmem carve_train_smem(char *base) {
  using Cfg = CnnCfg<C>;
  TrainSmem t;
  t.n = carve_smem<C>(base);
  uint32_t *after_bits = t.n.bits + QN_TILE * Cfg::OW + 4;
  t.scr = reinterpret_cast<float *>(after_bits);
  t.gs = t.scr + TrainCfg<C>::SCR;
  t.tgt = t.gs + QN_TILE;
  t.act = reinterpret_cast<int *>(t.tgt + QN_TILE);
  t.red = reinterpret_cast<float *>(t.act + QN_TILE);
  return t;
}

template <int C>
constexpr size_t train_smem_bytes() {
  return cnn_smem_bytes<C>() + sizeof(float) * (TrainCfg<C>::SCR + 3 * QN_TILE + QN_WAVES * 48 + 4);
}

// all-reduce sum over each aligned group of 32 lanes: DPP butterfly inside the 16-lane rows, then one
// ds_swizzle (xor 16) across the two rows
PQN_D float group32_sum(float v) {
  v = group16_sum(v);
  return v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
}

// stores of the T1 -> T2 hand-over operands (h1^T, dz^T: 18 MB per seed and launch, never read again by this kernel)
PQN_D void ws_store(f32x4 *p, const f32x4 &v) {
  __builtin_nontemporal_store(v, p);   // streaming: keeps the weight planes in L2 (-1 % on the 16-seed launch)
}

// Head of the training kernel.  Forward: z + b1 -> LN(128) -> relu -> fc2 -> q_a -> loss
// (pqn_minatar.py:271-285); backward through fc2 / relu / LN1 to dz (left in s.z for the dgrad and
// written transposed for the fc1 weight-gradient GEMM).  32 lanes per sample (all 8 waves): lane
// `sub` owns features o = sub + 32 r.  The sums over the 16 samples of the tile (d b1, d ln1 scale,
// d ln1 bias = column sums; d w2 = h2^T x G, G[m][a] = [act_m == a] g_m; d b2 = column sums of G)
// are MFMAs on the staged tiles -- an all-ones A operand turns a column sum into one 16x16x4 chain.
// NA: compile-time action count (0 = run-time L.a, up to QN_MAXA).
// NT = 2 (the pair kernel): the two tiles of the workgroup go through the head TOGETHER -- every thread runs its role
// for tile A and tile B back to back (two independent dependency chains, half the barriers); the per-sample arithmetic
// and its order are those of NT = 1, so a tile gets the same bits from both forms.
template <int C, int NA, int NT>
PQN_D void train_head_nt(const CnnSmem (&sv)[NT], const TrainSmem (&tv)[NT], const pqn_cnn_layout_t &L, int tid, int nb,
                         const int (&b0v)[NT], float inv_b, float *const (&gpv)[NT], float *__restrict__ dzT, float dz_scale) {
  constexpr int NAQ = NA ? NA : QN_MAXA;
  const int na = NA ? NA : L.a;
  const int lane = tid & 63, wave = tid >> 6;
  const int m = tid >> 5, sub = tid & 31;
  const float *hp = sv[0].hp;   // head parameters: one copy per workgroup
  float *tA[NT], *tB[NT], *tH[NT], *zrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    tA[t] = tv[t].scr + t * (3 * QN_TILE * QN_ZS);
    tB[t] = tA[t] + QN_TILE * QN_ZS;
    tH[t] = tA[t] + 2 * QN_TILE * QN_ZS;
    zrow[t] = sv[t].z + m * QN_ZS;
  }
  float v[NT][4], xh[NT][4], h2[NT][4], rstd[NT], sum[NT], sq[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    sum[t] = 0.f; sq[t] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = sub + 32 * r;
      v[t][r] = zrow[t][o] + hp[o];
      sum[t] += v[t][r];
      sq[t] = fmaf(v[t][r], v[t][r], sq[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) { sum[t] = group32_sum(sum[t]); sq[t] = group32_sum(sq[t]); }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float mean = sum[t] * (1.0f / QN_HID);
    const float var = fmaxf(sq[t] * (1.0f / QN_HID) - mean * mean, 0.0f);
    rstd[t] = rsqrt_exact(var + QN_LN_EPS);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = sub + 32 * r;
      xh[t][r] = (v[t][r] - mean) * rstd[t];
      h2[t][r] = fmaxf(fmaf(xh[t][r], hp[128 + o], hp[256 + o]), 0.0f);
    }
  }
  float qv[NT][NAQ];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < NAQ; ++a) {
      float part = 0.f;
      if (a < na) {
#pragma unroll
        for (int r = 0; r < 4; ++r) part = fmaf(h2[t][r], hp[384 + (sub + 32 * r) * na + a], part);
      }
      qv[t][a] = part;
    }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int a = 0; a < NAQ; ++a) qv[t][a] = group32_sum(qv[t][a]) + (a < na ? hp[384 + 128 * na + a] : 0.0f);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const bool valid = (b0v[t] + m) < nb;
    const int act = tv[t].act[m];
    float chosen = qv[t][0];
#pragma unroll
    for (int a = 1; a < NAQ; ++a)
      if (a == act) chosen = qv[t][a];
    const float diff = valid ? (chosen - tv[t].tgt[m]) : 0.0f;
    const float gm = diff * inv_b;  // d loss / d q_a, loss = 0.5*mean(diff^2)  (pqn_minatar.py:285)
    // backward: fc2, relu, LN1.  Every lane reads and rewrites only its own elements of the z tile.
    float dxh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = sub + 32 * r;
      const float dh2 = gm * hp[384 + o * na + act];
      const float dy = h2[t][r] > 0.0f ? dh2 : 0.0f;
      tA[t][m * QN_ZS + o] = dy * xh[t][r];
      tB[t][m * QN_ZS + o] = dy;
      tH[t][m * QN_ZS + o] = h2[t][r];
      dxh[r] = dy * hp[128 + o];
      s1 += dxh[r];
      s2 = fmaf(dxh[r], xh[t][r], s2);
    }
    s1 = group32_sum(s1) * (1.0f / QN_HID);
    s2 = group32_sum(s2) * (1.0f / QN_HID);
#pragma unroll
    for (int r = 0; r < 4; ++r) zrow[t][sub + 32 * r] = rstd[t] * (dxh[r] - s1 - xh[t][r] * s2);
    if (sub == 0) tv[t].gs[m] = gm;
    {  // loss / chosen-q partials of the wave's two samples (metrics td_loss, qvals: pqn_minatar.py:334-335)
      float l = (sub == 0) ? 0.5f * diff * diff : 0.0f, cq = (sub == 0 && valid) ? chosen : 0.0f;
      l += __shfl_xor(l, 32, 64);
      cq += __shfl_xor(cq, 32, 64);
      if (lane == 0) { tv[0].red[wave * 48 + 2 * t] = l; tv[0].red[wave * 48 + 2 * t + 1] = cq; }
    }
  }
  __syncthreads();
  // sums over the 16 samples as MFMA chains (K = 16 samples in 4 steps); wave w owns feature block 16w..16w+15
  {
    const int j = lane & 15, kk = lane >> 4;
    const int o_b1 = 9 * C * 16 + 48;
    f32x4 c1[NT], c2[NT], c3[NT], cw[NT], cb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) { c1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; c2[t] = c1[t]; c3[t] = c1[t]; cw[t] = c1[t]; cb[t] = c1[t]; }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int mr = 4 * st + kk;
        const int e = mr * QN_ZS + 16 * wave + j;
        const float gB = (tv[t].act[mr] == j) ? tv[t].gs[mr] : 0.0f;   // G[m][a = j]
        c1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, sv[t].z[e], c1[t], 0, 0, 0);
        c2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, tA[t][e], c2[t], 0, 0, 0);
        c3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, tB[t][e], c3[t], 0, 0, 0);
        cw[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tH[t][e], gB, cw[t], 0, 0, 0);   // A[i = o][k = m] = h2, B[k = m][j = a] = G
        if (wave == 0) cb[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, gB, cb[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float *gp = gpv[t];
      if (lane < 16) {   // every row of an all-ones chain holds the column sums: row 0 = register x of lanes 0..15
        gp[o_b1 + 16 * wave + lane] = c1[t].x;         // d b1
        gp[o_b1 + 128 + 16 * wave + lane] = c2[t].x;   // d ln1 scale
        gp[o_b1 + 256 + 16 * wave + lane] = c3[t].x;   // d ln1 bias
      }
      if (j < na) {      // d w2[o][a]: lane holds column a = j, rows o = 16w + 4kk + reg
        float *gw = gp + o_b1 + 384 + (16 * wave + 4 * kk) * na + j;
        gw[0] = cw[t].x; gw[na] = cw[t].y; gw[2 * na] = cw[t].z; gw[3 * na] = cw[t].w;
      }
      if (wave == 0 && lane < na) gp[o_b1 + 384 + 128 * na + lane] = cb[t].x;   // d b2
    }
  }
  // dz^T for the weight-gradient GEMM: dzT[o][b0 + m], 16-B stores (matmul_f16: tile-major fp16, scaled into
  // fp16's normal range by dz_scale)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const CnnSmem &s = sv[t];
    const int b0 = b0v[t];
    if (L.matmul_f16 == 1) {
      _Float16 *dzP = reinterpret_cast<_Float16 *>(dzT) + (size_t)(b0 / QN_TILE) * QN_HID * QN_TILE;
      if (tid < QN_HID * 2) {
        const int o = tid >> 1, hh = tid & 1;
        f16x8 vv;
#pragma unroll
        for (int r = 0; r < 8; ++r) vv[r] = (_Float16)(s.z[(8 * hh + r) * QN_ZS + o] * dz_scale);
        *reinterpret_cast<f16x8 *>(dzP + (size_t)o * QN_TILE + 8 * hh) = vv;
      }
    } else if (L.matmul_f16 == 2) {
      // bf16x3: dz leaves already split, in T2's operand order (dzw_index): T2 loads its B fragments with no arithmetic
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      unsigned short *dzw = reinterpret_cast<unsigned short *>(dzT + qw_dzw_offset(nb, C, L.a));
      const size_t ps = (size_t)qw_slabs(nb) * 256 * QN_HID;
      for (int i = tid; i < QN_HID * 4; i += QN_THREADS) {
        const int o = i >> 2, mq = i & 3;
        unsigned hh[2], mm[2], ll[2];
        x3_split2(s.z[(4 * mq + 0) * QN_ZS + o], s.z[(4 * mq + 1) * QN_ZS + o], hh[0], mm[0], ll[0]);
        x3_split2(s.z[(4 * mq + 2) * QN_ZS + o], s.z[(4 * mq + 3) * QN_ZS + o], hh[1], mm[1], ll[1]);
        unsigned short *pw = dzw + dzw_index(b0 + 4 * mq, o);
        __builtin_nontemporal_store(u2{hh[0], hh[1]}, reinterpret_cast<u2 *>(pw));
        __builtin_nontemporal_store(u2{mm[0], mm[1]}, reinterpret_cast<u2 *>(pw + ps));
        __builtin_nontemporal_store(u2{ll[0], ll[1]}, reinterpret_cast<u2 *>(pw + 2 * ps));
      }
      if ((nb & (QW_SLAB - 1)) != 0) {
        // ragged last slab: T2 multiplies whole 256-sample slabs, so the planes of the samples nb .. slabs * 256 - 1 must read
        // as zero.  The tail's 16-sample "virtual tiles" are shared out over the real tiles (tile t zeroes the virtual tiles
        // t + ntiles, t + 2 ntiles, ...): no extra launch, no host memset
        const int ntl = nb / QN_TILE, nvt = qw_slabs(nb) * (QW_SLAB / QN_TILE);
        for (int v = ntl + b0 / QN_TILE; v < nvt; v += ntl)
          for (int i = tid; i < QN_HID * 4; i += QN_THREADS) {
            unsigned short *pw = dzw + dzw_index(QN_TILE * v + 4 * (i & 3), i >> 2);
            __builtin_nontemporal_store(u2{0u, 0u}, reinterpret_cast<u2 *>(pw));
            __builtin_nontemporal_store(u2{0u, 0u}, reinterpret_cast<u2 *>(pw + ps));
            __builtin_nontemporal_store(u2{0u, 0u}, reinterpret_cast<u2 *>(pw + 2 * ps));
          }
      }
    } else
    for (int i = tid; i < QN_HID * 4; i += QN_THREADS) {
      const int o = i >> 2, mq = i & 3;
      const f32x4 vv = {s.z[(4 * mq + 0) * QN_ZS + o], s.z[(4 * mq + 1) * QN_ZS + o], s.z[(4 * mq + 2) * QN_ZS + o],
                        s.z[(4 * mq + 3) * QN_ZS + o]};
      ws_store(reinterpret_cast<f32x4 *>(dzT + (size_t)o * qw_ld(nb) + b0 + 4 * mq), vv);
    }
  }
  if (tid < NT) {
    const int t = tid;
    const float *red = tv[0].red + 2 * t;
    const int o_l = 9 * C * 16 + 48 + 384 + 128 * na + na;
    float *gp = t ? gpv[NT - 1] : gpv[0];
    gp[o_l] = ((red[0] + red[48]) + (red[96] + red[144])) + ((red[192] + red[240]) + (red[288] + red[336]));
    gp[o_l + 1] = ((red[1] + red[49]) + (red[97] + red[145])) + ((red[193] + red[241]) + (red[289] + red[337]));
  }
  __syncthreads();   // dz tiles complete (dgrad reads them); ts.red / scratch free again
}

template <int C, int NA>
PQN_D void train_head(const CnnSmem &s, const TrainSmem &ts, const pqn_cnn_layout_t &L, int tid, int nb, int b0,
                      float inv_b, float *__restrict__ gp, float *__restrict__ dzT, float dz_scale) {
  const CnnSmem sv[1] = {s};
  const TrainSmem tv[1] = {ts};
  const int b0v[1] = {b0};
  float *const gpv[1] = {gp};
  train_head_nt<C, NA, 1>(sv, tv, L, tid, nb, b0v, inv_b, gpv, dzT, dz_scale);
}

// profiling: per-phase s_memtime stamps of workgroup 0 / wave 0 (PQN_T1_STAMPS=1), read by tools/t1_stamps.py
#define T1_STAMP(k) do { if (stamps && threadIdx.x == 0 && blockIdx.x < 4) stamps[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)

// ---- P4 (bf16x3): dgrad  dh1[m][i] = relu'(h1[m][i]) * sum_o dz[m][o] W1[i][o]  written to out[m][i] (row stride
// QN_H1S).  A = the dz tile zt, split once per wave; B = the fc1 kernel's dgrad-order bf16 planes, streamed one
// i-block (12 dwordx4) ahead.  The relu mask is either the h1 tile itself, overwritten in place (out == the h1 tile,
// MASKBITS false), or a packed bit array (the pair kernel: the second tile's h1 region is reused before its backward).
template <bool MASKBITS>
PQN_D void t1_dgrad_x3(const float *zt, float *out, const uint32_t *mask, const float *__restrict__ theta,
                       const pqn_cnn_layout_t &L, int lane, int wave, int prot) {
    // bf16x3 split operands (see phase2_fc1_x3): the dz tile is split once per wave (4 K steps of 32 outputs, kept in
    // registers), the fc1 kernel's dgrad-order bf16 planes are streamed one i-block (12 dwordx4) ahead.
    const u32x4 *wd = reinterpret_cast<const u32x4 *>(theta + L.off_w1h) + 3 * (X3_PLANE / 8);
    X3Frag afr[4];
#pragma unroll
    for (int sK = 0; sK < 4; ++sK) {
      const f32x4 lo = *reinterpret_cast<const f32x4 *>(zt + (lane & 15) * QN_ZS + 32 * sK + 4 * (lane >> 4));
      const f32x4 hi = *reinterpret_cast<const f32x4 *>(zt + (lane & 15) * QN_ZS + 32 * sK + 16 + 4 * (lane >> 4));
      afr[sK] = x3_split8(lo, hi);
    }
    const int col = lane & 15, r0 = 4 * (lane >> 4);
    constexpr int IBW = 64 / QN_WAVES;
    const int ib_first = IBW * wave;
    auto frag = [&](int pl, int ibk, int sK) {   // plane pl, the wave's ibk-th i-block (rotated), K step sK
      const int ib = ib_first + ((ibk + prot) & (IBW - 1));
      return wd[(size_t)pl * (X3_PLANE / 8) + ((ib * 4 + sK) * 64 + lane)];
    };
    u32x4 ring[4][3];
#pragma unroll
    for (int sK = 0; sK < 4; ++sK)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) ring[sK][pl] = frag(pl, 0, sK);
    auto ib_step = [&](int ibk, auto more_t) {
      constexpr bool more = decltype(more_t)::value;
      const int ib = ib_first + ((ibk + prot) & (IBW - 1));
      // the dh1 / dx tile keeps the quad swizzle of the forward h1 tile (h1_slot): LayerNorm_0's backward reads and writes it
      // one position per lane with ds_read/write_b128 at a lane stride of 64 B -- conflict-free only in the swizzled layout
      float *p0 = out + r0 * QN_H1S + h1_slot(16 * ib + col);
      float m0, m1, m2, m3;   // relu mask (h1 > 0), read ahead of the MFMAs
      if (MASKBITS) {         // packed by the pair kernel's h1^T loop: 64-bit word [16-feature block ib][sample & 3],
                              // bit (feature & 15) * 4 + (sample >> 2)
        const unsigned long long *mw = reinterpret_cast<const unsigned long long *>(mask) + ib * 4;
        const int bsel = col * 4 + (lane >> 4);
        m0 = (float)((mw[0] >> bsel) & 1ull);
        m1 = (float)((mw[1] >> bsel) & 1ull);
        m2 = (float)((mw[2] >> bsel) & 1ull);
        m3 = (float)((mw[3] >> bsel) & 1ull);
      } else {
        m0 = p0[0]; m1 = p0[QN_H1S]; m2 = p0[2 * QN_H1S]; m3 = p0[3 * QN_H1S];   // the forward tile, in place (same slot map)
      }
      // four accumulators per K parity ({small, leading} x 2): reuse distance 6 in issue order (see phase2_fc1_x3)
      f32x4 acc_b[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, acc_s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      f32x4 acc_c[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int sK = 0; sK < 4; ++sK) {
        X3Frag bf;
        bf.h = ring[sK][0]; bf.m = ring[sK][1]; bf.l = ring[sK][2];
        const X3Frag &a = afr[sK];
        x3_grp6(acc_s[0], a.l, bf.h, acc_b[0], a.m, bf.h, acc_s[1], a.h, bf.l, acc_b[1], a.h, bf.m, acc_c[0], a.m, bf.m, acc_c[1], a.h, bf.h);
        if (more) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) ring[sK][pl] = frag(pl, ibk + 1, sK);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      x3_drain(acc_b[0], acc_s[0], acc_b[1], acc_s[1]);
      x3_drain(acc_c[0], acc_c[1]);
      const f32x4 acc0 = ((acc_b[0] + acc_b[1]) + acc_c[1]) + ((acc_s[0] + acc_s[1]) + acc_c[0]);
      p0[0] = m0 > 0.0f ? acc0.x : 0.0f;
      p0[QN_H1S] = m1 > 0.0f ? acc0.y : 0.0f;
      p0[2 * QN_H1S] = m2 > 0.0f ? acc0.z : 0.0f;
      p0[3 * QN_H1S] = m3 > 0.0f ? acc0.w : 0.0f;
    };
#pragma unroll 1
    for (int ibk = 0; ibk < IBW - 1; ++ibk) ib_step(ibk, std::true_type{});
    ib_step(IBW - 1, std::false_type{});
}

// sum over the 64 lanes of each of 16 per-lane values; afterwards lane l holds, in out[r] (r = 0..3), the wave total of
// value 8 (l >> 5) + 4 ((l >> 4) & 1) + r -- every lane of a 16-lane row holds the same four totals.  Halving by lane
// swaps (v_permlane32_swap / v_permlane16_swap: 8 + 4 swaps), then a DPP row sum of the remaining four: fixed order.
PQN_D void wave_sum16(const float (&v)[16], float (&out)[4]) {
  float w[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[c]), __float_as_uint(v[c + 8]), false, false);
    w[c] = __uint_as_float(r[0]) + __uint_as_float(r[1]);   // lanes < 32: value c, lanes >= 32: value c + 8
  }
  float u[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[c]), __float_as_uint(w[c + 4]), false, false);
    u[c] = __uint_as_float(r[0]) + __uint_as_float(r[1]);   // even rows: w[c], odd rows: w[c + 4]
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) out[c] = group16_sum(u[c]);
}

template <int C>
PQN_D void t1_ln0_bwd_ns(float *dh1, float *red, const float *__restrict__ theta, const pqn_cnn_layout_t &L,
                         const float (*xkeep)[16], const float *rkeep, float *__restrict__ gp, int tid) {
  using Cfg = CnnCfg<C>;
  const int lane = tid & 63, wave = tid >> 6;
  float ln0s[16];                                      // LayerNorm_0 scale: wave-uniform scalar loads
#pragma unroll
  for (int c = 0; c < 16; ++c) ln0s[c] = theta[L.off_ln0s + c];
  float G[16], GX[16], DX[16];                         // per lane (= position): sums over the wave's samples
#pragma unroll
  for (int c = 0; c < 16; ++c) { G[c] = 0.f; GX[c] = 0.f; DX[c] = 0.f; }
#pragma unroll
  for (int mm = 0; mm < QN_SPW; ++mm) {
    const int msamp = QN_SPW * wave + mm;
    f32x4 *gptr = reinterpret_cast<f32x4 *>(dh1 + msamp * QN_H1S + lane * 16);   // d relu-input (masked) -> dx in place
    const int sw = QN_H1_SWIZZLE ? ((lane >> 2) & 3) : 0;                          // quad swizzle of position `lane` (h1_slot)
    const float rstd = rkeep[mm];
    float g[16];
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const f32x4 v = gptr[qd ^ sw];
      g[4 * qd] = v.x; g[4 * qd + 1] = v.y; g[4 * qd + 2] = v.z; g[4 * qd + 3] = v.w;
    }
    float s1 = 0.f, s2 = 0.f, dxh[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      dxh[c] = g[c] * ln0s[c];
      s1 += dxh[c];
      s2 = fmaf(dxh[c], xkeep[mm][c], s2);
    }
    s1 *= (1.0f / 16.0f);
    s2 *= (1.0f / 16.0f);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      float d4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * qd + e;
        d4[e] = rstd * (dxh[c] - s1 - xkeep[mm][c] * s2);
        G[c] += g[c];
        GX[c] += g[c] * xkeep[mm][c];
        DX[c] += d4[e];
      }
      gptr[qd ^ sw] = f32x4{d4[0], d4[1], d4[2], d4[3]};
    }
  }
  float tb[4], ts[4], ti[4];
  wave_sum16(DX, tb);   // d conv bias
  wave_sum16(GX, ts);   // d ln0 scale
  wave_sum16(G, ti);    // d ln0 bias
  if ((lane & 15) == 0) {
    const int ch = 8 * (lane >> 5) + 4 * ((lane >> 4) & 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      red[wave * 48 + ch + r] = tb[r];
      red[wave * 48 + 16 + ch + r] = ts[r];
      red[wave * 48 + 32 + ch + r] = ti[r];
    }
  }
  __syncthreads();
  if (tid < 48) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < QN_WAVES; ++w) acc += red[w * 48 + tid];
    gp[Cfg::KW * 16 + tid] = acc;
  }
}

// ---- P5: LN0 backward of one 16-sample tile, in place on dh1 (d relu-input -> dx); ends with the workgroup barrier
// after which dx is complete and the 48 channel sums are in gp.  stg_base: QN_WAVES x 64 x QN_STG floats of scratch.
template <int C>
PQN_D void t1_ln0_bwd(float *dh1, float *stg_base, float *red, const float *__restrict__ theta, const pqn_cnn_layout_t &L,
                      const float (*xkeep)[16], const float *rkeep, float *__restrict__ gp, int tid, int ablate) {
  using Cfg = CnnCfg<C>;
  const int lane = tid & 63, wave = tid >> 6;
  // ---- P5: LN0 backward per point in one lane (xhat / rstd kept in registers since the forward conv);
  // channel sums (d conv-bias, d ln0-scale, d ln0-bias) in (channel = lane&15) layout. -----------
  if (!(ablate & 2)) {
    float ln0s[16];                                      // LayerNorm_0 scale: wave-uniform scalar loads
#pragma unroll
    for (int c = 0; c < 16; ++c) ln0s[c] = theta[L.off_ln0s + c];
    const int o = lane & 15, kk = lane >> 4;
    float *stg = stg_base + wave * 64 * QN_STG;
    float gsc = 0.f, gbi = 0.f, gbc = 0.f;
#pragma unroll
    for (int mm = 0; mm < QN_SPW; ++mm) {
      const int msamp = QN_SPW * wave + mm;
      float *gt = dh1 + msamp * QN_H1S;                 // d relu-input tile (masked by h1 > 0) -> dx in place
#pragma unroll
      for (int j = 0; j < 16; ++j) gbi += gt[(kk + 4 * j) * 16 + o];
      const float rstd = rkeep[mm];
      f32x4 *gptr = reinterpret_cast<f32x4 *>(gt + lane * 16);
      float g[16];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 v = gptr[qd];
        g[4 * qd] = v.x; g[4 * qd + 1] = v.y; g[4 * qd + 2] = v.z; g[4 * qd + 3] = v.w;
      }
      float s1 = 0.f, s2 = 0.f, dxh[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        dxh[c] = g[c] * ln0s[c];
        s1 += dxh[c];
        s2 = fmaf(dxh[c], xkeep[mm][c], s2);
      }
      s1 *= (1.0f / 16.0f);
      s2 *= (1.0f / 16.0f);
      f32x4 *sp = reinterpret_cast<f32x4 *>(stg + lane * QN_STG);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f32x4 dx, gx;
        dx.x = rstd * (dxh[4 * qd + 0] - s1 - xkeep[mm][4 * qd + 0] * s2); gx.x = g[4 * qd + 0] * xkeep[mm][4 * qd + 0];
        dx.y = rstd * (dxh[4 * qd + 1] - s1 - xkeep[mm][4 * qd + 1] * s2); gx.y = g[4 * qd + 1] * xkeep[mm][4 * qd + 1];
        dx.z = rstd * (dxh[4 * qd + 2] - s1 - xkeep[mm][4 * qd + 2] * s2); gx.z = g[4 * qd + 2] * xkeep[mm][4 * qd + 2];
        dx.w = rstd * (dxh[4 * qd + 3] - s1 - xkeep[mm][4 * qd + 3] * s2); gx.w = g[4 * qd + 3] * xkeep[mm][4 * qd + 3];
        gptr[qd] = dx;
        sp[qd] = gx;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        gsc += stg[(kk + 4 * j) * QN_STG + o];
        gbc += gt[(kk + 4 * j) * 16 + o];
      }
    }
    // fold the 4 position-groups of the wave (lanes o, o+16, o+32, o+48), then the 4 waves in fixed order
    gsc += __shfl_xor(gsc, 16, 64); gsc += __shfl_xor(gsc, 32, 64);
    gbi += __shfl_xor(gbi, 16, 64); gbi += __shfl_xor(gbi, 32, 64);
    gbc += __shfl_xor(gbc, 16, 64); gbc += __shfl_xor(gbc, 32, 64);
    if (lane < 16) {
      red[wave * 48 + lane] = gbc;
      red[wave * 48 + 16 + lane] = gsc;
      red[wave * 48 + 32 + lane] = gbi;
    }
  }
  __syncthreads();
  if (tid < 48) {
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < QN_WAVES; ++w) acc += red[w * 48 + tid];
    gp[Cfg::KW * 16 + tid] = acc;
  }
}

// ---- P6: conv weight gradient of one tile from dx (in the h1 region) and the packed observation bits; writes the
// tile's [KW][16] partial to gp.  wm_base: QN_WAVES x 192 words for the window masks; scr: TrainCfg<C>::SCR floats.
template <int C, int MODE>
PQN_D void t1_conv_wgrad(const float *dx, const uint32_t *bits, uint32_t *wm_base, float *scr, float *__restrict__ gp,
                         int tid, int ablate) {
  using Cfg = CnnCfg<C>;
  const int lane = tid & 63, wave = tid >> 6;
  // ---- P6: conv weight gradient as MFMA:  dWc[k][o] = sum_{m,pos} (bit(m,pos,k)/255) * dx[m][pos][o] ----
  //   A[i = k row][kk] = bit(m, pos = 4s+kk, k = 16rb+i)/255,  B[kk][o] = dx[m][4s+kk][o]  (LDS, contiguous)
  //   wave w reduces its samples 4w..4w+3; the 4 wave partials are folded in fixed order.
  if (!(ablate & 4)) {
    constexpr int NRB = (9 * C + 15) / 16;   // 16-row blocks of k
    constexpr int RB = 3 * C;
    // NRB odd (C = 4): wave = sample pair, all row blocks (no padded block);  NRB even: wave = (sample quad,
    // half of the row blocks).  NPART wave partials per row block are folded in fixed order.
    constexpr bool PAIR = (NRB % 2) == 1;
    constexpr int RBW = PAIR ? NRB : NRB / 2;      // row blocks per wave
    constexpr int SPW6 = PAIR ? 2 : 4;             // samples per wave
    constexpr int NPART = QN_TILE / SPW6;
    static_assert(NPART * NRB * 256 <= TrainCfg<C>::SCR, "conv-wgrad partials must fit the scratch tile");
    const int sg = PAIR ? wave : (wave & 3), rb0 = PAIR ? 0 : (wave >> 2) * RBW;
    const int i = lane & 15, kk = lane >> 4;
    int kyL[RBW], shL[RBW];                  // lane constants: window row / bit of k = 16*rb + i
#pragma unroll
    for (int j = 0; j < RBW; ++j) {
      const int k = 16 * (rb0 + j) + i;
      kyL[j] = (k < 9 * C) ? k / RB : 0;
      shL[j] = (k < 9 * C) ? k % RB : 31;    // padding rows test bit 31, which no window mask has (3C <= 30)
    }
    f32x4 acc[RBW];
#pragma unroll
    for (int j = 0; j < RBW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t *wm = wm_base + wave * 192;
    if constexpr (MODE == 2) {
      f32x4 accs[RBW];
#pragma unroll
      for (int j = 0; j < RBW; ++j) accs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // bf16 matrix core: A = window bits (exact in bf16, written as 2.0 = 0x4000), B = dx split exactly into three
      // bf16 planes; 2 K steps of 32 positions per sample, K slot (kq = kk, j) <-> position 32 st + 4 j + kk so that
      // the B reads stay the conflict-free dxm[64 * (8 st + j)] of the f32 path.  3 MFMAs per (step, row block)
      // instead of 8 at 1/4 the cycles each; 0.5/255 applied once to the accumulators.
#pragma unroll
      for (int mm = 0; mm < SPW6; ++mm) {
        const int msamp = SPW6 * sg + mm;
        window_masks<C>(bits + msamp * Cfg::OW, wm, lane);
        float bv[16];
        uint32_t wv[16][RBW];
#pragma unroll
        for (int q = 0; q < 16; ++q) {           // q = 8 st + j: all LDS reads of the sample in flight at once
          bv[q] = dx[msamp * QN_H1S + h1_slot(lane + 64 * q)];   // (dx keeps the quad swizzle of the h1 tile)
#pragma unroll
          for (int j = 0; j < RBW; ++j) wv[q][j] = wm[(4 * q + kk) * 3 + kyL[j]];
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const float *b = bv + 8 * st;
          const X3Frag bf = x3_split8(f32x4{b[0], b[1], b[2], b[3]}, f32x4{b[4], b[5], b[6], b[7]});
          u32x4 af[RBW];
#pragma unroll
          for (int j = 0; j < RBW; ++j) {
            uint32_t d[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const uint32_t b0 = __builtin_amdgcn_ubfe(wv[8 * st + 2 * jj][j], (uint32_t)shL[j], 1u);
              const uint32_t b1 = __builtin_amdgcn_ubfe(wv[8 * st + 2 * jj + 1][j], (uint32_t)shL[j], 1u);
              d[jj] = ((b1 << 16) | b0) << 14;
            }
            af[j] = u32x4{d[0], d[1], d[2], d[3]};
          }
          // {h plane} and {m + l planes} accumulate separately: 2 RBW independent chains (see phase2_fc1_x3)
          x3_grp_sameb<RBW>(accs, af, bf.l);
          x3_grp_sameb<RBW>(acc, af, bf.h);
          x3_grp_sameb<RBW>(accs, af, bf.m);
        }
      }
#pragma unroll
      for (int j = 0; j < RBW; ++j) {
        x3_drain(acc[j], accs[j]);
        acc[j] = (acc[j] + accs[j]) * (0.5f / 255.0f);
      }
    } else
#pragma unroll
    for (int mm = 0; mm < SPW6; ++mm) {
      const int msamp = SPW6 * sg + mm;
      window_masks<C>(bits + msamp * Cfg::OW, wm, lane);
      const float *dxm = dx + msamp * QN_H1S + lane;   // + 64*st : element (pos = 4st+kk, o = lane&15)
      float bv[16];
      uint32_t wv[16][RBW];
#pragma unroll
      for (int st = 0; st < 16; ++st) {          // all LDS reads of the sample in flight at once
        bv[st] = dxm[64 * st];
#pragma unroll
        for (int j = 0; j < RBW; ++j) wv[st][j] = wm[(4 * st + kk) * 3 + kyL[j]];
      }
#pragma unroll
      for (int st = 0; st < 16; ++st) {
#pragma unroll
        for (int j = 0; j < RBW; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bit_times_inv255(wv[st][j], shL[j]), bv[st], acc[j], 0, 0, 0);
      }
    }
    float *part = scr;  // [sample group][row block][16 rows][16 o]
#pragma unroll
    for (int j = 0; j < RBW; ++j) {
      float *pp = part + ((sg * NRB + rb0 + j) * 16 + 4 * kk) * 16 + i;
      pp[0] = acc[j].x; pp[16] = acc[j].y; pp[32] = acc[j].z; pp[48] = acc[j].w;
    }
    __syncthreads();
    for (int e = tid; e < Cfg::KW * 16; e += QN_THREADS) {
      const float *pp = scr + e;   // e = k*16 + o = (rb*16 + row)*16 + o
      if (NPART == 8)
        gp[e] = ((pp[0] + pp[NRB * 256]) + (pp[2 * NRB * 256] + pp[3 * NRB * 256])) +
                ((pp[4 * NRB * 256] + pp[5 * NRB * 256]) + (pp[6 * NRB * 256] + pp[7 * NRB * 256]));
      else
        gp[e] = (pp[0] + pp[NRB * 256]) + (pp[2 * NRB * 256] + pp[3 * NRB * 256]);
    }
  }
}

template <int C, int MODE>
__global__ __launch_bounds__(QN_THREADS) void qnet_cnn_train_kernel(
    int nb, const int64_t *__restrict__ idx, const uint32_t *__restrict__ obs_bits, const int32_t *__restrict__ action,
    const float *__restrict__ target, const float *__restrict__ theta, const float *__restrict__ w1b,
    pqn_cnn_layout_t L, float inv_b, float *__restrict__ dzT, float *__restrict__ h1T, float *__restrict__ gpart,
    int ablate, unsigned long long *__restrict__ stamps, pqn_seeds_t sd, float dz_scale) {
  using Cfg = CnnCfg<C>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int seed = blockIdx.y + sd.seed_base;   // seed slice of the stacked buffers (all strides 0 for a single seed)
  idx += seed * sd.idx_stride;
  theta += seed * sd.theta_stride;
  w1b += seed * sd.w1b_stride;
  dzT += seed * sd.ws_stride;
  h1T += seed * sd.ws_stride;
  gpart += seed * sd.ws_stride;
  // transition j = t*N + e of this seed -> row of the stacked [T][S*N] rollout record
  auto row_of = [&](int64_t key) -> int64_t {
    const uint32_t j = (uint32_t)(key & sd.idx_mask);
    if (sd.n_env_total == sd.n_env) return (int64_t)j;
    const uint32_t t = j / (uint32_t)sd.n_env;
    return (int64_t)t * sd.n_env_total + (int64_t)seed * sd.n_env + (int64_t)(j - t * (uint32_t)sd.n_env);
  };
  const TrainSmem ts = carve_train_smem<C>(smem_raw);
  const CnnSmem &s = ts.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * QN_TILE;
  const int rec = small_record_floats(C, L.a);
  float *gp = gpart + (size_t)blockIdx.x * rec;
  T1_STAMP(0);

  // ---- P0: gather inputs.  The permutation indices go out first, the parameter block is fetched while
  // they are in flight, then the dependent gathers: two memory round trips instead of three. ----------
  constexpr int NBI = (QN_TILE * Cfg::OW + QN_THREADS - 1) / QN_THREADS;
  int64_t ix[NBI];
#pragma unroll
  for (int k = 0; k < NBI; ++k) {
    const int i = tid + k * QN_THREADS, le = i / Cfg::OW;
    ix[k] = (i < QN_TILE * Cfg::OW && b0 + le < nb) ? row_of(idx[b0 + le]) : -1;
  }
  const int64_t src16 = (tid < QN_TILE && b0 + tid < nb) ? row_of(idx[b0 + tid]) : -1;
  TileParams<C> tp;
  tp.load(theta, L, tid);               // in flight together with the index loads above
  uint32_t gb[NBI];
#pragma unroll
  for (int k = 0; k < NBI; ++k) {       // needs only the (older) index loads: goes out while the parameters travel
    const int i = tid + k * QN_THREADS;
    gb[k] = (i < QN_TILE * Cfg::OW && ix[k] >= 0) ? obs_bits[(size_t)ix[k] * Cfg::OW + (i % Cfg::OW)] : 0u;
  }
  // action / target of the tile's samples (second level of the gather): consumed by the head, parked
  // in LDS after fc1 so the latency hides behind the forward pass
  int act_g = 0;
  float tgt_g = 0.0f;
  if (src16 >= 0) {
    act_g = action[src16];
    tgt_g = target[src16];
  }
  tp.store(s, L, tid);
#pragma unroll
  for (int k = 0; k < NBI; ++k) {
    const int i = tid + k * QN_THREADS;
    if (i < QN_TILE * Cfg::OW) s.bits[i] = gb[k];
  }
  if (tid < 4) s.bits[QN_TILE * Cfg::OW + tid] = 0u;
  __syncthreads();
  T1_STAMP(1);
  // ---- P1..P3: forward ---------------------------------------------------------------------
  float xkeep[QN_SPW][16], rkeep[QN_SPW];   // LN0 xhat / rstd of (sample QN_SPW*wave + mm, position lane): live until P5
  phase1_conv<C, true, MODE == 2, MODE == 2>(s, tid, xkeep, rkeep);
  __syncthreads();
  T1_STAMP(2);
  if (MODE == 1) phase2_fc1_f16<16>(s, reinterpret_cast<const _Float16 *>(theta + L.off_w1h), tid);
  else if (MODE == 2) {
    // h1^T for T2 (slab-major, h1s_index) leaves during fc1, one 512-thread slice at every second K step, in the
    // shadow of the weight stream (see the pair kernel)
    auto h1t_slice = [&](int j, int) {
      if (!(j & 1)) {
        const int e = tid + (j >> 1) * QN_THREADS, i = e >> 2, mq = e & 3;
        const float *src = s.h1 + (4 * mq) * QN_H1S + h1_slot(i);
        const f32x4 v = {src[0], src[QN_H1S], src[2 * QN_H1S], src[3 * QN_H1S]};
        ws_store(reinterpret_cast<f32x4 *>(h1T + h1s_index(i, b0 + 4 * mq)), v);
      }
    };
    phase2_fc1_x3<3>(s, theta + L.off_w1h, tid, -1, nullptr, h1t_slice);
  } else phase2_fc1<0>(s, theta + L.off_w1, tid);
  // h1^T for the fc1 weight-gradient GEMM (T2): h1T[i][b0 + m].  Issued here so the 64 KB of stores
  // drain while the (VALU-bound) head phase runs instead of queueing in front of the W1 stream.
  if (MODE == 1) {
    // fp16 operands for T2, tile-major: h1P[tile][i][16 samples] halves -- the wave writes 2 KB contiguous
    _Float16 *h1P = reinterpret_cast<_Float16 *>(h1T) + (size_t)blockIdx.x * QN_H1 * QN_TILE;
    for (int e = tid; e < QN_H1 * 2; e += QN_THREADS) {
      const int i = e >> 1, hh = e & 1;
      const float *src = s.h1 + (8 * hh) * QN_H1S + i;
      f16x8 v;
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = (_Float16)src[r * QN_H1S];
      *reinterpret_cast<f16x8 *>(h1P + (size_t)i * QN_TILE + 8 * hh) = v;
    }
  } else if (MODE == 0) {
    for (int e = tid; e < QN_H1 * 4; e += QN_THREADS) {
      const int i = e >> 2, mq = e & 3;
      const float *src = s.h1 + (4 * mq) * QN_H1S + i;
      const f32x4 v = {src[0], src[QN_H1S], src[2 * QN_H1S], src[3 * QN_H1S]};
      ws_store(reinterpret_cast<f32x4 *>(h1T + (size_t)i * qw_ld(nb) + b0 + 4 * mq), v);
    }
  }
  if (tid < QN_TILE) {
    ts.act[tid] = act_g;
    ts.tgt[tid] = tgt_g;
  }
  __syncthreads();
  T1_STAMP(3);
  // ---- head forward + backward (LN1 / fc2 / loss -> dz, parameter-gradient partials, dz^T) ----
  switch (L.a) {
    case 3: train_head<C, 3>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, dz_scale); break;
    case 4: train_head<C, 4>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, dz_scale); break;
    case 5: train_head<C, 5>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, dz_scale); break;
    case 6: train_head<C, 6>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, dz_scale); break;
    default: train_head<C, 0>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, dz_scale); break;
  }
  T1_STAMP(4);
  // ---- P4: dgrad  dh1[m][i] = sum_o dz[m][o] W1[i][o]  (A = dz tile, B = W1 in dgrad fragment order) ----
  if (MODE == 1) {
    // fp16 operands, f32 accumulation: one v_mfma_f32_16x16x16_f16 per 16-wide K group.  dz is O(1/B): it is
    // scaled by a power of two (>= B/2) into fp16's normal range and the product scaled back in f32.
    const float sc = dz_scale, isc = 1.0f / sc;
    const f16x4 *wb = reinterpret_cast<const f16x4 *>(reinterpret_cast<const _Float16 *>(theta + L.off_w1h) + QN_H1 * QN_HID);
    f16x4 afr[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4 *>(s.z + (lane & 15) * QN_ZS + 16 * g + 4 * (lane >> 4));
      afr[g] = to_f16x4(v * sc);
    }
    const int col = lane & 15, r0 = 4 * (lane >> 4);
    constexpr int IBW = 64 / QN_WAVES;
    const int ib_first = IBW * wave;
    const int prot = blockIdx.x & (IBW - 1);
    auto frag = [&](int n) { return wb[((n & 7) * 64 + ib_first + (((n >> 3) + prot) & (IBW - 1))) * 64 + lane]; };
    f16x4 ring[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) ring[n] = frag(n);
    auto pair_step = [&](int ip, auto more_t) {
      constexpr bool more = decltype(more_t)::value;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ib = ib_first + ((2 * ip + h + prot) & (IBW - 1));
        float *p0 = s.h1 + r0 * QN_H1S + 16 * ib + col;
        const float m0 = p0[0], m1 = p0[QN_H1S], m2 = p0[2 * QN_H1S], m3 = p0[3 * QN_H1S];
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 8; g += 2) {
          acc0 = f16_mfma_tied(afr[g], ring[8 * h + g], acc0);
          acc1 = f16_mfma_tied(afr[g + 1], ring[8 * h + g + 1], acc1);
          if (more) {
            ring[8 * h + g] = frag(16 * (ip + 1) + 8 * h + g);
            ring[8 * h + g + 1] = frag(16 * (ip + 1) + 8 * h + g + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        f16_drain(acc0, acc1);
        acc0 = (acc0 + acc1) * isc;
        p0[0] = m0 > 0.0f ? acc0.x : 0.0f;
        p0[QN_H1S] = m1 > 0.0f ? acc0.y : 0.0f;
        p0[2 * QN_H1S] = m2 > 0.0f ? acc0.z : 0.0f;
        p0[3 * QN_H1S] = m3 > 0.0f ? acc0.w : 0.0f;
      }
    };
#pragma unroll 1
    for (int ip = 0; ip < IBW / 2 - 1; ++ip) pair_step(ip, std::true_type{});
    pair_step(IBW / 2 - 1, std::false_type{});
  } else if (MODE == 2) {
    t1_dgrad_x3<false>(s.z, s.h1, nullptr, theta, L, lane, wave, blockIdx.x & (64 / QN_WAVES - 1));
  } else if (!(ablate & 1)) {
    const f32x4 *wb = reinterpret_cast<const f32x4 *>(w1b);
    f32x4 afr[8];
#pragma unroll
    for (int g = 0; g < 8; ++g)
      afr[g] = *reinterpret_cast<const f32x4 *>(s.z + (lane & 15) * QN_ZS + 16 * g + 4 * (lane >> 4));
    const int col = lane & 15, r0 = 4 * (lane >> 4);
    constexpr int IBW = 64 / QN_WAVES;        // i-blocks (16 conv features each) per wave
    constexpr int PF = 16;                    // fragments in flight: a rolling ring, one reload per 4 MFMAs
    const int ib_first = IBW * wave;
    const int prot = blockIdx.x & (IBW - 1);  // de-phase the W1 stream across workgroups (see phase2_fc1)
    auto frag = [&](int n) { return wb[((n & 7) * 64 + ib_first + (((n >> 3) + prot) & (IBW - 1))) * 64 + lane]; };
    f32x4 ring[PF];
#pragma unroll
    for (int n = 0; n < PF; ++n) ring[n] = frag(n);
    // a real loop over pairs of i-blocks (ring index = position inside the pair): fully unrolled, the
    // scheduler sinks the prefetches next to their uses and the ring degenerates to distance 1
    // (the last pair is peeled: a conditional prefetch inside the loop makes the compiler's vmcnt
    // bookkeeping assume it was not issued, draining the ring once per iteration)
    auto pair_step = [&](int ip, auto more_t) {
      constexpr bool more = decltype(more_t)::value;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ib = ib_first + ((2 * ip + h + prot) & (IBW - 1));
        // relu mask (h1 > 0), read ahead of the MFMAs; the h1 tile is overwritten in place with d(pre-relu)
        float *p0 = s.h1 + r0 * QN_H1S + 16 * ib + col;
        const float m0 = p0[0], m1 = p0[QN_H1S], m2 = p0[2 * QN_H1S], m3 = p0[3 * QN_H1S];
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const f32x4 c = ring[8 * h + g];
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].x, c.x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].y, c.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].z, c.z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[g].w, c.w, acc1, 0, 0, 0);
          if (more) ring[8 * h + g] = frag(16 * (ip + 1) + 8 * h + g);   // in-place reload after its consumers (see phase2_fc1)
          __builtin_amdgcn_sched_barrier(0);
        }
        acc0 += acc1;
        p0[0] = m0 > 0.0f ? acc0.x : 0.0f;
        p0[QN_H1S] = m1 > 0.0f ? acc0.y : 0.0f;
        p0[2 * QN_H1S] = m2 > 0.0f ? acc0.z : 0.0f;
        p0[3 * QN_H1S] = m3 > 0.0f ? acc0.w : 0.0f;
      }
    };
#pragma unroll 1
    for (int ip = 0; ip < IBW / 2 - 1; ++ip) pair_step(ip, std::true_type{});
    pair_step(IBW / 2 - 1, std::false_type{});
  }
  __syncthreads();
  T1_STAMP(5);
  // ---- P5: LN0 backward (t1_ln0_bwd; contains the barrier that ends the phase) ----
  if constexpr (MODE == 2) t1_ln0_bwd_ns<C>(s.h1, ts.red, theta, L, xkeep, rkeep, gp, tid);   // as the pair kernel: same bits
  else t1_ln0_bwd<C>(s.h1, s.stg, ts.red, theta, L, xkeep, rkeep, gp, tid, ablate);
  T1_STAMP(6);
  // ---- P6: conv weight gradient (t1_conv_wgrad) ----
  t1_conv_wgrad<C, MODE>(s.h1, s.bits, reinterpret_cast<uint32_t *>(s.stg), ts.scr, gp, tid, ablate);
  T1_STAMP(7);
}

// ---------------------------------------------------------------------------
// T1 for SMALL minibatches (f32 operand mode, nb <= KS_MAX_NB): the K-split form.  A 128-sample minibatch is 8 tiles:
// the single-tile kernel above then runs on 8 of 256 CUs and one optimizer step costs the latency of a whole tile pass
// (36 us: the yaml-default MinAtar run, 128 envs, spends 56 % of its time there).  Here a tile is cut along the conv
// POSITIONS -- the K dimension of fc1, the N dimension of its input gradient, the M dimension of its weight gradient --
// into NG = 64 / PG groups of PG positions, one 256-thread workgroup each, so that 8 tiles are 64 (PG = 8) workgroups:
//   ks_fwd   (group, tile): conv + LayerNorm_0 + relu of its PG positions x 16 samples, fc1 partial over its 16 PG features
//            against its 1/NG of W1 -> zpart[tile][group][16][128]
//   ks_head  (tile): z = sum of the NG partials, then the unchanged head of the training kernel (train_head): loss, head
//            parameter gradients, dz[16][128]
//   (default: ks_head and ks_bwd are ONE launch, qnet_cnn_ks_hb_kernel -- every group of a tile recomputes the head)
//   ks_bwd   (group, tile): conv + LayerNorm_0 again (cheaper than a round trip through memory), input gradient of ITS
//            features from dz and its 1/NG of W1 (dgrad fragment order), relu mask, LayerNorm_0 backward, conv weight
//            gradient partial, and its rows of the fc1 weight gradient for the tile -- straight in the layout of the
//            split-K slabs of T2, one slab per tile -- so there is no h1 hand-over and no T2.
// Everything stays in MFMA accumulator layout: ks_fwd runs the conv TRANSPOSED (rows = channels, columns = samples), which
// makes a lane's four conv outputs exactly its A fragment of the fc1 product; ks_bwd runs it the usual way (rows =
// samples, columns = channels), which is the layout the input gradient comes out in, the B fragment of the conv weight
// gradient and the A fragment of the fc1 weight gradient.  No LDS transposes, no staging.
// gfx950's f32-input MFMA (v_mfma_f32_16x16x4_f32) throughout: exact f32 products, as the large-batch f32 mode.
// ---------------------------------------------------------------------------
#define KS_MAX_NB 256
#define KS_NG_MAX 16             // position groups (workgroups) per tile at the finest cut (4 positions each)
#define KS_THREADS 256
#define KS_DZS 132   // LDS row stride of the dz tile

// window masks of ONE point (sample's packed row, position): word ky = the 3C bits of window row ky
template <int C>
PQN_D void window_masks_point(const uint32_t *row_bits, int pos, uint32_t (&m)[3]) {
  const int py = pos >> 3, px = pos & 7;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int b = ((py + ky) * 10 + px) * C;
    const uint64_t v = ((((uint64_t)row_bits[(b >> 5) + 1]) << 32) | row_bits[b >> 5]) >> (b & 31);
    m[ky] = (uint32_t)v & ((1u << (3 * C)) - 1u);
  }
}

// gather of the tile's packed observations + the window masks of (position pl of the group, sample) into LDS
template <int C, int PG>
PQN_D void ks_gather(int nb, int b0, int grp, const int64_t *__restrict__ idx, const uint32_t *__restrict__ obs_bits,
                     const pqn_seeds_t &sd, int seed, uint32_t *s_bits, uint32_t *s_wm, int tid) {
  using Cfg = CnnCfg<C>;
  auto row_of = [&](int64_t key) -> int64_t {
    const uint32_t j = (uint32_t)(key & sd.idx_mask);
    if (sd.n_env_total == sd.n_env) return (int64_t)j;
    const uint32_t t = j / (uint32_t)sd.n_env;
    return (int64_t)t * sd.n_env_total + (int64_t)seed * sd.n_env + (int64_t)(j - t * (uint32_t)sd.n_env);
  };
  for (int i = tid; i < QN_TILE * Cfg::OW; i += KS_THREADS) {
    const int le = i / Cfg::OW;
    s_bits[i] = (b0 + le < nb) ? obs_bits[(size_t)row_of(idx[b0 + le]) * Cfg::OW + (i % Cfg::OW)] : 0u;
  }
  if (tid < 4) s_bits[QN_TILE * Cfg::OW + tid] = 0u;
  __syncthreads();
  for (int t = tid; t < 16 * PG; t += KS_THREADS) {
    uint32_t m[3];
    window_masks_point<C>(s_bits + (t & 15) * Cfg::OW, grp * PG + (t >> 4), m);
    s_wm[t * 3] = m[0]; s_wm[t * 3 + 1] = m[1]; s_wm[t * 3 + 2] = m[2];
  }
  __syncthreads();
}

template <int C, int PG>
__global__ __launch_bounds__(KS_THREADS) void qnet_cnn_ks_fwd_kernel(int nb, const int64_t *__restrict__ idx,
                                                                     const uint32_t *__restrict__ obs_bits,
                                                                     const float *__restrict__ theta, pqn_cnn_layout_t L,
                                                                     float *__restrict__ zpart, pqn_seeds_t sd) {
  using Cfg = CnnCfg<C>;
  constexpr int NG = 64 / PG, PW = PG / 4;   // groups per tile, positions per wave
  static_assert(PG == 4 || PG == 8 || PG == 16, "positions per workgroup");
  __shared__ __attribute__((aligned(16))) uint32_t s_bits[QN_TILE * Cfg::OW + 4];
  __shared__ uint32_t s_wm[16 * PG * 3];
  __shared__ __attribute__((aligned(16))) float s_wc[Cfg::KW * 16 + 48];
  __shared__ __attribute__((aligned(16))) float s_z[4][QN_TILE][QN_HID + 4];
  const int seed = blockIdx.z + sd.seed_base;
  idx += seed * sd.idx_stride;
  theta += seed * sd.theta_stride;
  zpart += seed * sd.ws_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x, tile = blockIdx.y, b0 = tile * QN_TILE;
  for (int i = tid; i < Cfg::KW * 16 + 48; i += KS_THREADS) s_wc[i] = theta[L.off_wc + i];
  // the wave's rows of W1 (8 KB per position) go out FIRST: they do not depend on the gather, whose two dependent loads
  // and two barriers then run in their shadow
  const f32x4 *wp = reinterpret_cast<const f32x4 *>(theta + L.off_w1);
  f32x4 wfr[PW][8];
#pragma unroll
  for (int q = 0; q < PW; ++q)
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) wfr[q][cb] = wp[((grp * PG + wave * PW + q) * 8 + cb) * 64 + lane];
  ks_gather<C, PG>(nb, b0, grp, idx, obs_bits, sd, seed, s_bits, s_wm, tid);
  ConvMfma<C> cv;
  cv.init(s_wc, lane);                       // wk[s] = Wc[4 s + (lane >> 4)][lane & 15]: here the A operand (row = channel)
  const float *bc = s_wc + Cfg::KW * 16;
  const int smp = lane & 15, cq = 4 * (lane >> 4);   // this lane: sample column, channels cq .. cq + 3
  f32x4 acc[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < PW; ++q) {
    const int pl = wave * PW + q;
    const uint32_t m[3] = {s_wm[(pl * 16 + smp) * 3], s_wm[(pl * 16 + smp) * 3 + 1], s_wm[(pl * 16 + smp) * 3 + 2]};
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < ConvMfma<C>::NS; ++st) d = __builtin_amdgcn_mfma_f32_16x16x4f32(cv.wk[st], cv.a_of(m, st), d, 0, 0, 0);
    float v[4] = {d.x + bc[cq], d.y + bc[cq + 1], d.z + bc[cq + 2], d.w + bc[cq + 3]};
    float sum = (v[0] + v[1]) + (v[2] + v[3]);
    float sq = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
    sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);   // the 16 channels of (sample, position): 4 lanes x 4
    sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
    const float mean = sum * (1.0f / 16.0f);
    const float var = fmaxf(sq * (1.0f / 16.0f) - mean * mean, 0.0f);
    const float rstd = rsqrt_exact(var + QN_LN_EPS);
    float y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = fmaxf(fmaf((v[r] - mean) * rstd, bc[16 + cq + r], bc[32 + cq + r]), 0.0f);
    // fc1 partial: A[i = sample][K slot = lane >> 4] of sub-step x is feature 16 pos + 4 (lane >> 4) + x = y[x]
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[0], wfr[q][cb].x, acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[1], wfr[q][cb].y, acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[2], wfr[q][cb].z, acc[cb], 0, 0, 0);
      acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[3], wfr[q][cb].w, acc[cb], 0, 0, 0);
    }
  }
  // D: column = output 16 cb + (lane & 15), rows = samples 4 (lane >> 4) + r.  Fold the four waves in fixed order.
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) {
    float *zp = &s_z[wave][cq][16 * cb + smp];
    zp[0] = acc[cb].x; zp[QN_HID + 4] = acc[cb].y; zp[2 * (QN_HID + 4)] = acc[cb].z; zp[3 * (QN_HID + 4)] = acc[cb].w;
  }
  __syncthreads();
  float *dst = zpart + ((size_t)tile * NG + grp) * (QN_TILE * QN_HID);
  for (int e = tid; e < QN_TILE * QN_HID / 4; e += KS_THREADS) {
    const int mrow = e >> 5, c4 = (e & 31) * 4;
    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(&s_z[0][mrow][c4]), a1 = *reinterpret_cast<const f32x4 *>(&s_z[1][mrow][c4]);
    const f32x4 a2 = *reinterpret_cast<const f32x4 *>(&s_z[2][mrow][c4]), a3 = *reinterpret_cast<const f32x4 *>(&s_z[3][mrow][c4]);
    reinterpret_cast<f32x4 *>(dst)[e] = (a0 + a1) + (a2 + a3);
  }
}

// head of the K-split form: one 512-thread workgroup per tile, the head code of the training kernel unchanged
template <int C, int NG>
__global__ __launch_bounds__(QN_THREADS) void qnet_cnn_ks_head_kernel(
    int nb, const int64_t *__restrict__ idx, const int32_t *__restrict__ action, const float *__restrict__ target,
    const float *__restrict__ theta, pqn_cnn_layout_t L, float inv_b, const float *__restrict__ zpart, float *__restrict__ dzbuf,
    float *__restrict__ dzT, float *__restrict__ gpart, pqn_seeds_t sd, float dz_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int seed = blockIdx.y + sd.seed_base;
  idx += seed * sd.idx_stride;
  theta += seed * sd.theta_stride;
  zpart += seed * sd.ws_stride;
  dzbuf += seed * sd.ws_stride;
  dzT += seed * sd.ws_stride;
  gpart += seed * sd.ws_stride;
  auto row_of = [&](int64_t key) -> int64_t {
    const uint32_t j = (uint32_t)(key & sd.idx_mask);
    if (sd.n_env_total == sd.n_env) return (int64_t)j;
    const uint32_t t = j / (uint32_t)sd.n_env;
    return (int64_t)t * sd.n_env_total + (int64_t)seed * sd.n_env + (int64_t)(j - t * (uint32_t)sd.n_env);
  };
  const TrainSmem ts = carve_train_smem<C>(smem_raw);
  const CnnSmem &s = ts.n;
  const int tid = threadIdx.x, tile = blockIdx.x, b0 = tile * QN_TILE;
  const int rec = small_record_floats(C, L.a);
  float *gp = gpart + (size_t)tile * NG * rec;          // the record of group 0 carries the tile's head block
  const int64_t src16 = (tid < QN_TILE && b0 + tid < nb) ? row_of(idx[b0 + tid]) : -1;
  TileParams<C> tp;
  tp.load(theta, L, tid);
  if (src16 >= 0) {
    ts.act[tid] = action[src16];
    ts.tgt[tid] = target[src16];
  } else if (tid < QN_TILE) {
    ts.act[tid] = 0;
    ts.tgt[tid] = 0.0f;
  }
  tp.store(s, L, tid);
  const float *zp = zpart + (size_t)tile * NG * (QN_TILE * QN_HID);
  for (int e = tid; e < QN_TILE * QN_HID / 4; e += QN_THREADS) {
    f32x4 a = reinterpret_cast<const f32x4 *>(zp)[e];
#pragma unroll
    for (int g = 1; g < NG; ++g) a += reinterpret_cast<const f32x4 *>(zp + (size_t)g * (QN_TILE * QN_HID))[e];   // fixed order
    *reinterpret_cast<f32x4 *>(s.z + (e >> 5) * QN_ZS + (e & 31) * 4) = a;
  }
  __syncthreads();
  switch (L.a) {
    case 3: train_head<C, 3>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, dz_scale); break;
    case 4: train_head<C, 4>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, dz_scale); break;
    case 5: train_head<C, 5>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, dz_scale); break;
    case 6: train_head<C, 6>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, dz_scale); break;
    default: train_head<C, 0>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, dz_scale); break;
  }
  float *dz = dzbuf + (size_t)tile * (QN_TILE * QN_HID);   // train_head left dz in the z tile (and ended with a barrier)
  for (int e = tid; e < QN_TILE * QN_HID / 4; e += QN_THREADS)
    reinterpret_cast<f32x4 *>(dz)[e] = *reinterpret_cast<const f32x4 *>(s.z + (e >> 5) * QN_ZS + (e & 31) * 4);
}

// backward of ONE conv position x 16 samples by one wave (see the header above): recomputed conv + LayerNorm_0, input
// gradient of the position's 16 features, relu mask, LayerNorm_0 backward, conv weight-gradient and channel-sum
// accumulation, and the position's 16 rows of the tile's fc1 weight gradient (stored to `slab`).
//   wm_pl  window masks of (this position, sample) : wm_pl[sample * 3 + ky];  dz  the tile's dz in LDS, row stride dzs
template <int C>
struct KsBwdLane {
  static constexpr int NRB = (9 * C + 15) / 16, RB = 3 * C;
  int kyL[NRB], shL[NRB];     // conv weight gradient: this lane's row k = 16 rb + (lane & 15)
  f32x4 accw[NRB];
  float gsc, gbi, gbc, bias, ln_s, ln_b;
  PQN_D void init(const float *bc, int lane) {
    const int ch = lane & 15;
#pragma unroll
    for (int j = 0; j < NRB; ++j) {
      const int k = 16 * j + ch;
      kyL[j] = (k < 9 * C) ? k / RB : 0;
      shL[j] = (k < 9 * C) ? k % RB : 31;      // padding rows test bit 31, which no window mask has
      accw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    gsc = gbi = gbc = 0.f;
    bias = bc[ch]; ln_s = bc[16 + ch]; ln_b = bc[32 + ch];
  }
  PQN_D void position(const ConvMfma<C> &cv, const uint32_t *wm_pl, const f32x4 (&dzf)[8], const f32x4 (&wfr)[8], const float *dz,
                      int dzs, f32x4 *slab_pos, int lane) {
    const int ch = lane & 15, kk = lane >> 4;  // this lane: channel column, samples 4 kk .. 4 kk + 3 (D rows)
    // conv + LayerNorm_0 of (samples, this position): rows = samples (A = window bits of sample lane & 15), columns = channels
    const uint32_t mA[3] = {wm_pl[(lane & 15) * 3], wm_pl[(lane & 15) * 3 + 1], wm_pl[(lane & 15) * 3 + 2]};
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < ConvMfma<C>::NS; ++st) d = __builtin_amdgcn_mfma_f32_16x16x4f32(cv.a_of(mA, st), cv.wk[st], d, 0, 0, 0);
    const float v[4] = {d.x + bias, d.y + bias, d.z + bias, d.w + bias};
    float xh[4], rstd[4], y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {            // sample 4 kk + r: its 16 channels sit in the 16 lanes of this DPP row
      const float sum = group16_sum(v[r]), sq = group16_sum(v[r] * v[r]);
      const float mean = sum * (1.0f / 16.0f);
      const float var = fmaxf(sq * (1.0f / 16.0f) - mean * mean, 0.0f);
      rstd[r] = rsqrt_exact(var + QN_LN_EPS);
      xh[r] = (v[r] - mean) * rstd[r];
      y[r] = fmaxf(fmaf(xh[r], ln_s, ln_b), 0.0f);
    }
    // input gradient of this position's 16 features: D rows = samples, columns = channels -- the layout of v / xh / y
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dzf[cb].x, wfr[cb].x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dzf[cb].y, wfr[cb].y, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dzf[cb].z, wfr[cb].z, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(dzf[cb].w, wfr[cb].w, a1, 0, 0, 0);
    }
    a0 += a1;
    const float dh[4] = {a0.x, a0.y, a0.z, a0.w};
    float dx[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float g = y[r] > 0.0f ? dh[r] : 0.0f;       // relu mask
      gbi += g;
      gsc = fmaf(g, xh[r], gsc);
      const float dxh = g * ln_s;
      const float s1 = group16_sum(dxh) * (1.0f / 16.0f), s2 = group16_sum(dxh * xh[r]) * (1.0f / 16.0f);
      dx[r] = rstd[r] * (dxh - s1 - xh[r] * s2);
      gbc += dx[r];
    }
    // conv weight gradient: dWc[k][ch] += sum over samples of bit(sample, pos, k) / 255 * dx[sample][ch]; K slot kk of
    // sub-step r stands for sample 4 kk + r, which is register r of the lanes of row kk (B) -- A is built to match
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t *wmp = wm_pl + (4 * kk + r) * 3;
#pragma unroll
      for (int j = 0; j < NRB; ++j)
        accw[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bit_times_inv255(wmp[kyL[j]], shL[j]), dx[r], accw[j], 0, 0, 0);
    }
    // fc1 weight gradient rows 16 pos .. 16 pos + 15 of this tile: D rows = features (A: y[r] of lane (ch, kk) = h1[sample
    // 4 kk + r][ch]), columns = outputs (B: dz[sample 4 kk + r][16 cb + (lane & 15)]); stored as the fragment the
    // reduction kernel folds (the layout of the fc1 kernel itself)
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
      f32x4 w = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) w = __builtin_amdgcn_mfma_f32_16x16x4f32(y[r], dz[(4 * kk + r) * dzs + 16 * cb + ch], w, 0, 0, 0);
      slab_pos[cb * 64 + lane] = w;
    }
  }
  // channel sums of the wave -> red[48] (conv bias | ln0 scale | ln0 bias); conv weight-gradient tile -> cw[NRB * 256]
  PQN_D void finish(float *red, float *cw, int lane) {
    const int ch = lane & 15, kk = lane >> 4;
    gsc += __shfl_xor(gsc, 16, 64); gsc += __shfl_xor(gsc, 32, 64);   // the 4 sample rows: lanes ch, ch + 16, ch + 32, ch + 48
    gbi += __shfl_xor(gbi, 16, 64); gbi += __shfl_xor(gbi, 32, 64);
    gbc += __shfl_xor(gbc, 16, 64); gbc += __shfl_xor(gbc, 32, 64);
    if (lane < 16) {
      red[lane] = gbc;
      red[16 + lane] = gsc;
      red[32 + lane] = gbi;
    }
#pragma unroll
    for (int j = 0; j < NRB; ++j) {            // D: row k = 16 j + 4 kk + reg, column ch
      float *pp = cw + (j * 16 + 4 * kk) * 16 + ch;
      pp[0] = accw[j].x; pp[16] = accw[j].y; pp[32] = accw[j].z; pp[48] = accw[j].w;
    }
  }
};

template <int C, int PG>
__global__ __launch_bounds__(KS_THREADS) void qnet_cnn_ks_bwd_kernel(int nb, const int64_t *__restrict__ idx,
                                                                     const uint32_t *__restrict__ obs_bits,
                                                                     const float *__restrict__ theta, const float *__restrict__ w1b,
                                                                     pqn_cnn_layout_t L, const float *__restrict__ dzbuf,
                                                                     float *__restrict__ gpart, float *__restrict__ wpart,
                                                                     pqn_seeds_t sd) {
  using Cfg = CnnCfg<C>;
  constexpr int NG = 64 / PG, PW = PG / 4;
  constexpr int NRB = (9 * C + 15) / 16;   // 16-row blocks of the conv kernel's k index
  __shared__ __attribute__((aligned(16))) uint32_t s_bits[QN_TILE * Cfg::OW + 4];
  __shared__ uint32_t s_wm[16 * PG * 3];
  __shared__ __attribute__((aligned(16))) float s_wc[Cfg::KW * 16 + 48];
  __shared__ __attribute__((aligned(16))) float s_dz[QN_TILE * KS_DZS];
  __shared__ float s_red[4][48];
  __shared__ float s_cw[4][NRB * 256];
  const int seed = blockIdx.z + sd.seed_base;
  idx += seed * sd.idx_stride;
  theta += seed * sd.theta_stride;
  w1b += seed * sd.w1b_stride;
  dzbuf += seed * sd.ws_stride;
  gpart += seed * sd.ws_stride;
  wpart += seed * sd.ws_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x, tile = blockIdx.y, b0 = tile * QN_TILE;
  const int rec = small_record_floats(C, L.a);
  float *gp = gpart + ((size_t)tile * NG + grp) * rec;
  for (int i = tid; i < Cfg::KW * 16 + 48; i += KS_THREADS) s_wc[i] = theta[L.off_wc + i];
  const f32x4 *wb = reinterpret_cast<const f32x4 *>(w1b);
  f32x4 wfr[PW][8];                          // W1[16 pos + ch][16 cb + 4 kk + x]: out before the gather (see ks_fwd)
#pragma unroll
  for (int q = 0; q < PW; ++q)
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) wfr[q][cb] = wb[(cb * 64 + grp * PG + wave * PW + q) * 64 + lane];
  {
    const float *dz = dzbuf + (size_t)tile * (QN_TILE * QN_HID);
    for (int e = tid; e < QN_TILE * QN_HID / 4; e += KS_THREADS)
      *reinterpret_cast<f32x4 *>(s_dz + (e >> 5) * KS_DZS + (e & 31) * 4) = reinterpret_cast<const f32x4 *>(dz)[e];
  }
  ks_gather<C, PG>(nb, b0, grp, idx, obs_bits, sd, seed, s_bits, s_wm, tid);
  ConvMfma<C> cv;
  cv.init(s_wc, lane);                       // B operand: wk[s] = Wc[4 s + (lane >> 4)][channel = lane & 15]
  KsBwdLane<C> bl;
  bl.init(s_wc + Cfg::KW * 16, lane);
  // A fragments of the input-gradient product: dz[sample = lane & 15][16 cb + 4 kk + x]
  f32x4 dzf[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) dzf[cb] = *reinterpret_cast<const f32x4 *>(s_dz + (lane & 15) * KS_DZS + 16 * cb + 4 * (lane >> 4));
  f32x4 *slab = reinterpret_cast<f32x4 *>(wpart + (size_t)tile * QN_H1 * QN_HID);
#pragma unroll
  for (int q = 0; q < PW; ++q) {
    const int pl = wave * PW + q, pos = grp * PG + pl;
    bl.position(cv, s_wm + pl * 16 * 3, dzf, wfr[q], s_dz, KS_DZS, slab + (size_t)pos * 8 * 64, lane);
  }
  bl.finish(s_red[wave], s_cw[wave], lane);   // then the 4 waves in fixed order
  __syncthreads();
  for (int e = tid; e < Cfg::KW * 16; e += KS_THREADS) gp[e] = (s_cw[0][e] + s_cw[1][e]) + (s_cw[2][e] + s_cw[3][e]);
  if (tid < 48) gp[Cfg::KW * 16 + tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
  if (grp != 0)                              // the head block and the loss live in group 0's record (ks_head)
    for (int e = Cfg::KW * 16 + 48 + tid; e < rec; e += KS_THREADS) gp[e] = 0.0f;
}

// head + backward in ONE launch: a 512-thread workgroup per (group of 8 positions, tile) first runs the tile's head --
// every one of the 8 groups of a tile recomputes it (16 samples x 128 features: cheap) instead of waiting for a launch
// of its own, which costs as much as the whole kernel at these sizes -- then each of its 8 waves takes one position.
// Only group 0's record keeps the head block.
template <int C, int NGF>   // NGF: position groups of the forward launch (partials to sum)
__global__ __launch_bounds__(QN_THREADS) void qnet_cnn_ks_hb_kernel(
    int nb, const int64_t *__restrict__ idx, const uint32_t *__restrict__ obs_bits, const int32_t *__restrict__ action,
    const float *__restrict__ target, const float *__restrict__ theta, const float *__restrict__ w1b, pqn_cnn_layout_t L,
    float inv_b, const float *__restrict__ zpart, float *__restrict__ dzT, float *__restrict__ gpart, float *__restrict__ wpart,
    pqn_seeds_t sd) {
  using Cfg = CnnCfg<C>;
  constexpr int PG = 8, NG = 8;
  constexpr int NRB = (9 * C + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int seed = blockIdx.z + sd.seed_base;
  idx += seed * sd.idx_stride;
  theta += seed * sd.theta_stride;
  w1b += seed * sd.w1b_stride;
  zpart += seed * sd.ws_stride;
  dzT += seed * sd.ws_stride;
  gpart += seed * sd.ws_stride;
  wpart += seed * sd.ws_stride;
  auto row_of = [&](int64_t key) -> int64_t {
    const uint32_t j = (uint32_t)(key & sd.idx_mask);
    if (sd.n_env_total == sd.n_env) return (int64_t)j;
    const uint32_t t = j / (uint32_t)sd.n_env;
    return (int64_t)t * sd.n_env_total + (int64_t)seed * sd.n_env + (int64_t)(j - t * (uint32_t)sd.n_env);
  };
  const TrainSmem ts = carve_train_smem<C>(smem_raw);
  const CnnSmem &s = ts.n;
  // the h1 tile of the training kernel's LDS plan is not used here: window masks and the cross-wave folds live in it
  uint32_t *s_wm = reinterpret_cast<uint32_t *>(s.h1);            // [16 * PG * 3]
  float (*s_red)[48] = reinterpret_cast<float (*)[48]>(s.h1 + 512);       // [8][48]
  float (*s_cw)[NRB * 256] = reinterpret_cast<float (*)[NRB * 256]>(s.h1 + 1024);   // [8][NRB * 256]
  static_assert(1024 + 8 * NRB * 256 <= QN_TILE * QN_H1S, "scratch must fit the h1 tile");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x, tile = blockIdx.y, b0 = tile * QN_TILE;
  const int rec = small_record_floats(C, L.a);
  float *gp = gpart + ((size_t)tile * NG + grp) * rec;
  const int pos = grp * PG + wave;
  const f32x4 *wb = reinterpret_cast<const f32x4 *>(w1b);
  f32x4 wfr[8];                              // out first: nothing below depends on them until the backward
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) wfr[cb] = wb[(cb * 64 + pos) * 64 + lane];
  // gather: indices -> observation rows, action, target; parameters; the forward partials
  constexpr int NBI = (QN_TILE * Cfg::OW + QN_THREADS - 1) / QN_THREADS;
  uint32_t gb[NBI];
#pragma unroll
  for (int k = 0; k < NBI; ++k) {
    const int i = tid + k * QN_THREADS, le = i / Cfg::OW;
    gb[k] = (i < QN_TILE * Cfg::OW && b0 + le < nb) ? obs_bits[(size_t)row_of(idx[b0 + le]) * Cfg::OW + (i % Cfg::OW)] : 0u;
  }
  const int64_t src16 = (tid < QN_TILE && b0 + tid < nb) ? row_of(idx[b0 + tid]) : -1;
  TileParams<C> tp;
  tp.load(theta, L, tid);
  if (tid < QN_TILE) {
    ts.act[tid] = src16 >= 0 ? action[src16] : 0;
    ts.tgt[tid] = src16 >= 0 ? target[src16] : 0.0f;
  }
  tp.store(s, L, tid);
#pragma unroll
  for (int k = 0; k < NBI; ++k) {
    const int i = tid + k * QN_THREADS;
    if (i < QN_TILE * Cfg::OW) s.bits[i] = gb[k];
  }
  if (tid < 4) s.bits[QN_TILE * Cfg::OW + tid] = 0u;
  const float *zp = zpart + (size_t)tile * NGF * (QN_TILE * QN_HID);
  for (int e = tid; e < QN_TILE * QN_HID / 4; e += QN_THREADS) {
    f32x4 a = reinterpret_cast<const f32x4 *>(zp)[e];
#pragma unroll
    for (int g = 1; g < NGF; ++g) a += reinterpret_cast<const f32x4 *>(zp + (size_t)g * (QN_TILE * QN_HID))[e];   // fixed order
    *reinterpret_cast<f32x4 *>(s.z + (e >> 5) * QN_ZS + (e & 31) * 4) = a;
  }
  __syncthreads();
  if (tid < 16 * PG) {
    uint32_t m[3];
    window_masks_point<C>(s.bits + (tid & 15) * Cfg::OW, grp * PG + (tid >> 4), m);
    s_wm[tid * 3] = m[0]; s_wm[tid * 3 + 1] = m[1]; s_wm[tid * 3 + 2] = m[2];
  }
  switch (L.a) {   // ends with a barrier; leaves dz in the z tile
    case 3: train_head<C, 3>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, 1.0f); break;
    case 4: train_head<C, 4>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, 1.0f); break;
    case 5: train_head<C, 5>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, 1.0f); break;
    case 6: train_head<C, 6>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, 1.0f); break;
    default: train_head<C, 0>(s, ts, L, tid, nb, b0, inv_b, gp, dzT, 1.0f); break;
  }
  ConvMfma<C> cv;
  cv.init(s.wc, lane);
  KsBwdLane<C> bl;
  bl.init(s.wc + Cfg::KW * 16, lane);
  f32x4 dzf[8];
#pragma unroll
  for (int cb = 0; cb < 8; ++cb) dzf[cb] = *reinterpret_cast<const f32x4 *>(s.z + (lane & 15) * QN_ZS + 16 * cb + 4 * (lane >> 4));
  f32x4 *slab = reinterpret_cast<f32x4 *>(wpart + (size_t)tile * QN_H1 * QN_HID);
  bl.position(cv, s_wm + wave * 16 * 3, dzf, wfr, s.z, QN_ZS, slab + (size_t)pos * 8 * 64, lane);
  bl.finish(s_red[wave], s_cw[wave], lane);   // then the 8 waves in fixed order
  __syncthreads();
  for (int e = tid; e < Cfg::KW * 16; e += QN_THREADS)
    gp[e] = ((s_cw[0][e] + s_cw[1][e]) + (s_cw[2][e] + s_cw[3][e])) + ((s_cw[4][e] + s_cw[5][e]) + (s_cw[6][e] + s_cw[7][e]));
  if (tid < 48)
    gp[Cfg::KW * 16 + tid] = ((s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid])) +
                             ((s_red[4][tid] + s_red[5][tid]) + (s_red[6][tid] + s_red[7][tid]));
  if (grp != 0)                              // only group 0's record keeps the head block and the loss
    for (int e = Cfg::KW * 16 + 48 + tid; e < rec; e += QN_THREADS) gp[e] = 0.0f;
}

// ---------------------------------------------------------------------------
// T1, pair form (bf16x3 mode): one workgroup runs TWO 16-sample tiles and shares the fc1 weight stream between them
// (phase2_fc1_x3<.., 2>): the forward fc1 of a tile is bounded by the CU's 64 B/clk vector-memory path moving 768 KB of
// weight planes, whichever tile they are for.  Everything else is the single-tile code, run once per tile, with the
// LDS laid out so that both h1 tiles are resident through fc1:
//   [h1 A 66 KB][h1 B 66 KB][z A][z B][conv / head parameters][obs bits A, B][relu mask bits of B][act/tgt/gs x2][red]
//   - conv staging is in place in the destination rows (stage_tile_inplace), the window masks live in the z tiles;
//   - after fc1 and the h1^T stores, tile B's h1 is only needed as relu mask for its dgrad: it is packed to 2 KB of
//     bits and the 66 KB region becomes the scratch of the heads and of both backward passes;
//   - backward of A runs in place on h1 A; backward of B writes its dgrad into the (then free) h1 A region.
// ---------------------------------------------------------------------------
template <int C>
struct PairSmem {
  using Cfg = CnnCfg<C>;
  static constexpr int WCN = (Cfg::KW * 16 + 48 + 3) & ~3;
  static constexpr int BITN = QN_TILE * Cfg::OW + 4;
  // the head-parameter block is sized by the ACTUAL action count (round 4): with the 8-action block of the single-tile
  // kernels the plan of C = 6 missed the 160 KB by 1,024 B and that of C = 7 by 2,112 B; SpaceInvaders (C = 6, A = 4) and
  // Freeway (C = 7, A = 3) now take the pair form too
  static constexpr int hp_floats(int a) { return (384 + 128 * a + a + 3) & ~3; }
  static constexpr int HP_MIN = hp_floats(1);
  static constexpr size_t bytes(int a) {
    return sizeof(float) * (2 * QN_TILE * QN_H1S + 2 * QN_TILE * QN_ZS + WCN + hp_floats(a)) +
           sizeof(uint32_t) * (2 * BITN + QN_TILE * 32) + sizeof(float) * (6 * QN_TILE + QN_WAVES * 48);
  }
  // the freed h1 B region must hold the head / conv-wgrad scratch followed by the window masks, and the LN0-bwd staging
  static_assert(TrainCfg<C>::SCR + QN_WAVES * 192 <= QN_TILE * QN_H1S && QN_WAVES * 64 * QN_STG <= QN_TILE * QN_H1S, "scratch must fit h1 B");
  static_assert(2 * 3 * QN_TILE * QN_ZS <= QN_TILE * QN_H1S, "the staged tiles of both heads (train_head_nt<.., 2>) must fit h1 B");
};

template <int C>
__global__ __launch_bounds__(QN_THREADS) void qnet_cnn_train_pair_kernel(
    int nb, const int64_t *__restrict__ idx, const uint32_t *__restrict__ obs_bits, const int32_t *__restrict__ action,
    const float *__restrict__ target, const float *__restrict__ theta, pqn_cnn_layout_t L, float inv_b,
    float *__restrict__ dzT, float *__restrict__ h1T, float *__restrict__ gpart, int ablate, pqn_seeds_t sd, float dz_scale,
    unsigned long long *__restrict__ stamps) {
  using Cfg = CnnCfg<C>;
  using PS = PairSmem<C>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  // XCD-aware (seed, pair) of this workgroup: workgroups go to the 8 XCDs round-robin in dispatch order, so with a plain
  // (x = pair, y = seed) grid every XCD's L2 sees the weight planes of every seed in flight.  When the seeds of the launch
  // divide over the XCDs, XCD k runs seeds [k S/8, (k+1) S/8) one after the other and its L2 holds one seed's 1.5 MB of
  // planes at a time (ablate bit 8 = plain mapping, for measurement).
  int pair_id = blockIdx.x, seed_l = blockIdx.y;
  if ((gridDim.y & 7) == 0 && !(ablate & 256)) {
    const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7u, slot = lin >> 3;
    seed_l = xcd * (gridDim.y >> 3) + slot / gridDim.x;
    pair_id = slot % gridDim.x;
  }
  const int seed = seed_l + sd.seed_base;
  idx += seed * sd.idx_stride;
  theta += seed * sd.theta_stride;
  dzT += seed * sd.ws_stride;
  h1T += seed * sd.ws_stride;
  gpart += seed * sd.ws_stride;
  auto row_of = [&](int64_t key) -> int64_t {
    const uint32_t j = (uint32_t)(key & sd.idx_mask);
    if (sd.n_env_total == sd.n_env) return (int64_t)j;
    const uint32_t t = j / (uint32_t)sd.n_env;
    return (int64_t)t * sd.n_env_total + (int64_t)seed * sd.n_env + (int64_t)(j - t * (uint32_t)sd.n_env);
  };
  // ---- LDS ----
  float *h1A = reinterpret_cast<float *>(smem_raw), *h1B = h1A + QN_TILE * QN_H1S;
  float *zA = h1B + QN_TILE * QN_H1S, *zB = zA + QN_TILE * QN_ZS;
  float *wc = zB + QN_TILE * QN_ZS, *hp = wc + PS::WCN;
  uint32_t *bitsA = reinterpret_cast<uint32_t *>(hp + PS::hp_floats(L.a)), *bitsB = bitsA + PS::BITN, *maskB = bitsB + PS::BITN;
  float *small = reinterpret_cast<float *>(maskB + QN_TILE * 32);   // gs[2][16] | tgt[2][16] | act[2][16] | red[8][48]
  float *red = small + 6 * QN_TILE;
  float *scr = h1B;                                                 // once h1 B has been packed to maskB
  uint32_t *wmS = reinterpret_cast<uint32_t *>(h1B + TrainCfg<C>::SCR);
  CnnSmem sT[2];
  TrainSmem tsT[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    sT[t].h1 = t ? h1B : h1A; sT[t].z = t ? zB : zA; sT[t].wc = wc; sT[t].stg = h1B; sT[t].hp = hp; sT[t].bits = t ? bitsB : bitsA;
    tsT[t].n = sT[t]; tsT[t].scr = scr; tsT[t].gs = small + t * QN_TILE; tsT[t].tgt = small + (2 + t) * QN_TILE;
    tsT[t].act = reinterpret_cast<int *>(small + (4 + t) * QN_TILE); tsT[t].red = red;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rec = small_record_floats(C, L.a);
  const int tile0 = 2 * pair_id;
  const int b0T[2] = {tile0 * QN_TILE, (tile0 + 1) * QN_TILE};
  float *gpT[2] = {gpart + (size_t)tile0 * rec, gpart + (size_t)(tile0 + 1) * rec};

  T1_STAMP(0);
  // ---- P0: gather the inputs of both tiles (same scheme as the single-tile kernel) ----
  constexpr int NBI = (QN_TILE * Cfg::OW + QN_THREADS - 1) / QN_THREADS;
  int64_t ix[2][NBI];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int k = 0; k < NBI; ++k) {
      const int i = tid + k * QN_THREADS, le = i / Cfg::OW;
      ix[t][k] = (i < QN_TILE * Cfg::OW && b0T[t] + le < nb) ? row_of(idx[b0T[t] + le]) : -1;
    }
  const int b32 = b0T[0] + tid;                                     // tid < 32: sample tid of the pair
  const int64_t src32 = (tid < 2 * QN_TILE && b32 < nb) ? row_of(idx[b32]) : -1;
  TileParams<C> tp;
  tp.load(theta, L, tid);
  uint32_t gb[2][NBI];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int k = 0; k < NBI; ++k) {
      const int i = tid + k * QN_THREADS;
      gb[t][k] = (i < QN_TILE * Cfg::OW && ix[t][k] >= 0) ? obs_bits[(size_t)ix[t][k] * Cfg::OW + (i % Cfg::OW)] : 0u;
    }
  int act_g = 0;
  float tgt_g = 0.0f;
  if (src32 >= 0) {
    act_g = action[src32];
    tgt_g = target[src32];
  }
  tp.store(sT[0], L, tid);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int k = 0; k < NBI; ++k) {
      const int i = tid + k * QN_THREADS;
      if (i < QN_TILE * Cfg::OW) sT[t].bits[i] = gb[t][k];
    }
    if (tid < 4) sT[t].bits[QN_TILE * Cfg::OW + tid] = 0u;
  }
  __syncthreads();
  T1_STAMP(1);
  // ---- forward: conv of both tiles, then fc1 of both against one pass over the weight planes ----
  float xkA[QN_SPW][16], rkA[QN_SPW], xkB[QN_SPW][16], rkB[QN_SPW];
  phase1_conv<C, true, true, true>(sT[0], tid, xkA, rkA);
  T1_STAMP(2);
  phase1_conv<C, true, true, true>(sT[1], tid, xkB, rkB);
  __syncthreads();
  T1_STAMP(3);
  {
    // h1^T for T2 (slab-major, h1s_index) leaves DURING fc1: K step j of every wave also stores one 512-thread slice
    // (tile j & 1, slice j >> 1) of the two h1 tiles, and tile B leaves its relu mask as bits on the way.  h1 is only
    // read in this phase, so the order against the MFMA operand reads does not matter.
    auto h1t_slice = [&](int j, int jc) {   // jc = j % 2 at compile time (PF = 2)
      const int t = jc & 1, e = tid + (j >> 1) * QN_THREADS;
      const int i = e >> 2, mq = e & 3;
      const float *src = (t ? h1B : h1A) + (4 * mq) * QN_H1S + h1_slot(i);
      const f32x4 v = {src[0], src[QN_H1S], src[2 * QN_H1S], src[3 * QN_H1S]};
      ws_store(reinterpret_cast<f32x4 *>(h1T + h1s_index(i, b0T[0] + QN_TILE * t + 4 * mq)), v);
      if (t == 1) {
        // lane = (feature i & 15) * 4 + mq, sample 4 mq + rr: one 64-bit ballot per rr and 16-feature block
        const unsigned long long bx = __ballot(v.x > 0.0f), by = __ballot(v.y > 0.0f), bz = __ballot(v.z > 0.0f), bw = __ballot(v.w > 0.0f);
        if (lane == 0) {
          unsigned long long *mw = reinterpret_cast<unsigned long long *>(maskB) + (i >> 4) * 4;
          mw[0] = bx; mw[1] = by; mw[2] = bz; mw[3] = bw;
        }
      }
    };
    phase2_fc1_x3<2, 2>(sT[0], theta + L.off_w1h, tid, pair_id, &sT[1], h1t_slice);
  }
  T1_STAMP(4);
  if (tid < 2 * QN_TILE) {
    tsT[0].act[tid] = act_g;     // act[2][16] / tgt[2][16] are contiguous: sample tid of the pair
    tsT[0].tgt[tid] = tgt_g;
  }
  __syncthreads();   // z tiles complete; h1 B is free from here on (scratch)
  T1_STAMP(5);
  // ---- heads (both tiles together) ----
  {
    float *const gpv[2] = {gpT[0], gpT[1]};
    switch (L.a) {
      case 3: train_head_nt<C, 3, 2>(sT, tsT, L, tid, nb, b0T, inv_b, gpv, dzT, dz_scale); break;
      case 4: train_head_nt<C, 4, 2>(sT, tsT, L, tid, nb, b0T, inv_b, gpv, dzT, dz_scale); break;
      case 6: train_head_nt<C, 6, 2>(sT, tsT, L, tid, nb, b0T, inv_b, gpv, dzT, dz_scale); break;
      default: train_head_nt<C, 0, 2>(sT, tsT, L, tid, nb, b0T, inv_b, gpv, dzT, dz_scale); break;
    }
  }
  T1_STAMP(6);
  const int prot = pair_id & (64 / QN_WAVES - 1);
  // ---- backward of A in place on h1 A, then of B into the same region ----
  t1_dgrad_x3<false>(zA, h1A, nullptr, theta, L, lane, wave, prot);
  __syncthreads();
  T1_STAMP(7);
  // LayerNorm_0 backward without the staging buffer (channel sums in registers + permlane swaps): -2.7 % per launch
  // against the staged form in an in-call A/B (profiles/r03_v4_ln0ns_ab.txt); the single-tile bf16x3 kernel uses the same
  // function, so a tile gets the same bits from both forms
  t1_ln0_bwd_ns<C>(h1A, red, theta, L, xkA, rkA, gpT[0], tid);
  t1_conv_wgrad<C, 2>(h1A, bitsA, wmS, scr, gpT[0], tid, ablate);
  __syncthreads();   // dx of A and the scratch fully consumed
  T1_STAMP(8);
  t1_dgrad_x3<true>(zB, h1A, maskB, theta, L, lane, wave, prot);
  __syncthreads();
  T1_STAMP(9);
  t1_ln0_bwd_ns<C>(h1A, red, theta, L, xkB, rkB, gpT[1], tid);
  t1_conv_wgrad<C, 2>(h1A, bitsB, wmS, scr, gpT[1], tid, ablate);
  T1_STAMP(10);
}

// ---------------------------------------------------------------------------
// Persistent rollout, pair form (bf16x3 mode): one workgroup owns 32 envs = two 16-env tiles that share every fc1
// weight fragment (phase2_fc1_x3<.., 2>), as in the training pair kernel; the head, epsilon-greedy draw and transition
// rule of tile A run on threads 0..255, those of tile B on threads 256..511 (the single-tile kernel leaves that half
// idle).  Same arithmetic, same draws, same record as qnet_cnn_rollout_kernel -- bit-identical results.
// ---------------------------------------------------------------------------
template <int C, class Env>
__global__ __launch_bounds__(QN_THREADS) void qnet_cnn_rollout_pair_kernel(
    int n, int t_len, uint32_t *__restrict__ state, uint32_t *__restrict__ bits_all, const float *__restrict__ theta,
    pqn_cnn_layout_t L, int32_t *__restrict__ action, float *__restrict__ qmax, float *__restrict__ reward,
    uint8_t *__restrict__ done, float *__restrict__ discount, float *__restrict__ rer, int32_t *__restrict__ rel,
    int32_t *__restrict__ ts, float *__restrict__ last_q, const float *__restrict__ eps_dev,
    const uint64_t *__restrict__ keys, float rscale, int store_obs, int n_per_seed, long long theta_stride,
    int keys_stride) {
  using Cfg = CnnCfg<C>;
  using PS = PairSmem<C>;
  static_assert(Cfg::OW == Env::OBS_WORDS, "packed observation width");
  int e_off = 0;
  if (n_per_seed > 0) {   // seed batching: both tiles of a pair belong to one seed (n_per_seed % 32 == 0)
    const int seed = (blockIdx.x * 2 * QN_TILE) / n_per_seed;
    theta += seed * theta_stride;
    keys += (size_t)seed * keys_stride;
    e_off = seed * n_per_seed;
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *h1A = reinterpret_cast<float *>(smem_raw), *h1B = h1A + QN_TILE * QN_H1S;
  float *zA = h1B + QN_TILE * QN_H1S, *zB = zA + QN_TILE * QN_ZS;
  float *wc = zB + QN_TILE * QN_ZS, *hp = wc + PS::WCN;
  uint32_t *bitsA = reinterpret_cast<uint32_t *>(hp + QN_HP_FLOATS), *bitsB = bitsA + PS::BITN;
  CnnSmem sA, sB;
  sA.h1 = h1A; sA.z = zA; sA.wc = wc; sA.stg = nullptr; sA.hp = hp; sA.bits = bitsA;
  sB.h1 = h1B; sB.z = zB; sB.wc = wc; sB.stg = nullptr; sB.hp = hp; sB.bits = bitsB;
  const int tid = threadIdx.x;
  const int half = tid >> 8, htid = tid & 255;            // tile (0 = A, 1 = B) and thread index inside its half
  const int e0 = blockIdx.x * 2 * QN_TILE;                // first env of the pair; tile B starts at e0 + 16
  const int m = (htid >> 4) & 15, sub = htid & 15, e = e0 + half * QN_TILE + m;
  const int e_rng = e - e_off;
  const bool owner = sub == 0 && e < n;                   // the lane that owns env e
  const CnnSmem &sH = half ? sB : sA;
  const size_t bstride = (size_t)n * Cfg::OW;
  load_tile_common<C>(sA, theta, L, tid);
  for (int i = tid; i < 2 * QN_TILE * Cfg::OW; i += QN_THREADS) {   // the two tiles' rows are contiguous in bits_all
    const int t = i / (QN_TILE * Cfg::OW), r = i - t * (QN_TILE * Cfg::OW), le = t * QN_TILE + r / Cfg::OW;
    (t ? bitsB : bitsA)[r] = (e0 + le < n) ? bits_all[(size_t)e0 * Cfg::OW + i] : 0u;
  }
  if (tid < 4) { bitsA[QN_TILE * Cfg::OW + tid] = 0u; bitsB[QN_TILE * Cfg::OW + tid] = 0u; }
  Env env;
  LogRec log;
  if (owner) {
    uint32_t w[Env::ENV_WORDS];
#pragma unroll
    for (int i = 0; i < Env::ENV_WORDS; ++i) w[i] = state[(size_t)i * n + e];
    env.unpack(w);
    log.load(state, n, e, Env::ENV_WORDS);
  }
  const float eps = *eps_dev;
  const int tile_in_seed = (e0 - e_off) / QN_TILE;        // even: the weight-stream rotation of the pair
#pragma unroll 1
  for (int t = 0; t <= t_len; ++t) {
    __syncthreads();   // the bits tiles hold obs_t (and every previous reader of the tiles is done)
    phase1_conv<C, false, true, true>(sA, tid);
    phase1_conv<C, false, true, true>(sB, tid);
    __syncthreads();
    phase2_fc1_x3<2, 2>(sA, theta + L.off_w1h, tid, tile_in_seed >> 1, &sB);   // (a 3-deep ring spills 38 VGPRs here)
    __syncthreads();
    {
      float q[QN_MAXA], h2[8], xh[8], rstd;
      phase3_head(sH, theta, L, htid, q, h2, xh, rstd);
      if (owner) {
        int best = 0;
        float bv = q[0];
#pragma unroll
        for (int a = 1; a < QN_MAXA; ++a)
          if (a < L.a && q[a] > bv) { bv = q[a]; best = a; }
        if (t == t_len) {
          if (last_q) last_q[e] = bv;                      // bootstrap value of obs_T
        } else {
          const uint64_t key = keys[t];
          uint32_t o0, o1;
          pqn_bits(key, (uint32_t)e_rng, PQN_STREAM_ACT, o0, o1);
          const int act = (pqn_uniform(o0) < eps) ? (int)pqn_randint(o1, (uint32_t)L.a) : best;
          int dn = 0;
          const float r = env.step(act, key, (uint32_t)e_rng, dn);
          log.step(r, dn);
          if (dn) env.reset(key, (uint32_t)e_rng);             // gymnax auto-reset
          const size_t o = (size_t)t * n + e;
          if (action) action[o] = act;
          if (qmax) qmax[o] = bv;
          if (reward) reward[o] = r * rscale;
          if (done) done[o] = (uint8_t)dn;
          if (discount) discount[o] = dn ? 0.0f : 1.0f;
          if (rer) rer[o] = log.ret_ret;
          if (rel) rel[o] = log.ret_len;
          if (ts) ts[o] = log.timestep;
          env.obs_bits(&sH.bits[m * Cfg::OW]);             // obs_{t+1} straight into the LDS tile
        }
      }
    }
    if (t == t_len) break;
    __syncthreads();
    if (store_obs || t + 1 == t_len) {
      uint32_t *dst = bits_all + (store_obs ? (size_t)(t + 1) * bstride : (size_t)0);
      for (int i = tid; i < 2 * QN_TILE * Cfg::OW; i += QN_THREADS) {
        const int tl = i / (QN_TILE * Cfg::OW), r = i - tl * (QN_TILE * Cfg::OW), le = tl * QN_TILE + r / Cfg::OW;
        if (e0 + le < n) dst[(size_t)e0 * Cfg::OW + i] = (tl ? bitsB : bitsA)[r];
      }
    }
  }
  if (owner) {
    uint32_t w[Env::ENV_WORDS];
    env.pack(w);
#pragma unroll
    for (int i = 0; i < Env::ENV_WORDS; ++i) state[(size_t)i * n + e] = w[i];
    log.store(state, n, e, Env::ENV_WORDS);
  }
}

// ---------------------------------------------------------------------------
// T2: dW1[i][o] = sum_b h1[b][i] dz[b][o]  -- a plain MFMA GEMM on the two transposed operands T1
// left in the workspace (h1T [1024][nb], dzT [128][nb]).  Workgroup (it, ks): 64 rows of i x all
// 128 o, K-slab of 256 samples; wave w owns column block w and 4 row blocks.  Per 16-sample K group a
// lane loads 5 dwordx4 (4 A + 1 B fragment) for 16 MFMAs; the output tile is already in the
// fragment layout of the parameter buffer, one dwordx4 store per block into wpart[ks].
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(QN_THREADS) void qnet_fc1_wgrad_kernel(int nb, const float *__restrict__ h1T,
                                                                    const float *__restrict__ dzT,
                                                                    float *__restrict__ wpart, long long ws_stride) {
  static_assert(QN_WAVES == 8, "one output column block per wave");
  h1T += blockIdx.z * ws_stride;   // seed slice
  dzT += blockIdx.z * ws_stride;
  wpart += blockIdx.z * ws_stride;
  // operands are staged through LDS so each element is fetched once per workgroup (the 8 waves share
  // the 64-row A tile); double-buffered, one barrier per 32-sample step.
  constexpr int KS = 32;                 // samples per step
  constexpr int ROWS = 64 + 128;         // A rows (i) then B rows (o)
  constexpr int RS = KS + 4;             // padded LDS row stride (floats): conflict-free b128 fragment reads
  __shared__ __attribute__((aligned(16))) float tile[2][ROWS * RS];
  const int tid = threadIdx.x, lane = tid & 63, cb = tid >> 6;
  const int it = blockIdx.x, ks = blockIdx.y;
  const int o = lane & 15, kq = 4 * (lane >> 4);
  const int c0 = ks * QW_SLAB;
  const int nsteps = (min(QW_SLAB, nb - c0) + KS - 1) / KS;   // last step may be half full (nb % 16 == 0)
  const int ld = qw_ld(nb);
  // loader mapping: ROWS*KS/4 = 1536 float4 per step = 3 per thread
  const float *src[3];
  int dst[3], col[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int e = tid + QN_THREADS * j;
    const int row = e >> 3, c4 = (e & 7) * 4;           // 8 float4 per 32-sample row
    src[j] = (row < 64 ? h1T + (size_t)(64 * it + row) * ld : dzT + (size_t)(row - 64) * ld) + c0 + c4;
    dst[j] = row * RS + c4;
    col[j] = c0 + c4;                                   // sample index of the float4 at step 0
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 pre[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) pre[j] = (col[j] < nb) ? *reinterpret_cast<const f32x4 *>(src[j]) : zero4;
  for (int st = 0; st < nsteps; ++st) {
    float *t = tile[st & 1];
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<f32x4 *>(t + dst[j]) = pre[j];
    if (st + 1 < nsteps) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        pre[j] = (col[j] + KS * (st + 1) < nb) ? *reinterpret_cast<const f32x4 *>(src[j] + KS * (st + 1)) : zero4;
    }
    __syncthreads();   // tile[st&1] complete; tile[(st+1)&1] was last read two steps ago
#pragma unroll
    for (int g = 0; g < KS / 16; ++g) {
      const f32x4 b = *reinterpret_cast<const f32x4 *>(t + (64 + 16 * cb + o) * RS + 16 * g + kq);
      f32x4 a4[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) a4[a] = *reinterpret_cast<const f32x4 *>(t + (16 * a + o) * RS + 16 * g + kq);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[a].x, b.x, acc[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[a].y, b.y, acc[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[a].z, b.z, acc[a], 0, 0, 0);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[a].w, b.w, acc[a], 0, 0, 0);
    }
  }
  f32x4 *out = reinterpret_cast<f32x4 *>(wpart + (size_t)ks * QN_H1 * QN_HID);
#pragma unroll
  for (int a = 0; a < 4; ++a) out[((4 * it + a) * 8 + cb) * 64 + lane] = acc[a];
}

// bf16x3 split-operand variant (matmul_f16 == 2; see phase2_fc1_x3): the same f32 operands T1 leaves in the workspace
// and the same output layout as the f32 kernel.  Design, from measurements on MI355X (profiles/r02_t2_*):
//   - operand traffic, not the matrix core, bounded the first version (both operands streamed per (it, ks) workgroup:
//     805 MB of loads per 16-seed launch, 2/3 of them the dz slab re-read by all 16 row blocks);
//   - VALU work issued between MFMAs is NOT hidden behind them (tools/ubench/mfma_issue.hip: ~2.8 counter ticks per
//     VALU op on top of ~10 per MFMA at two waves per SIMD), so splitting dz fragments at every use cost more than
//     the MFMAs themselves.
// Hence "dz in registers": wave w owns output column block w (16 of the 128 fc1 outputs) for the workgroup's whole
// 256-sample slab.  Its dz fragments -- 8 steps x 8 values per lane -- are loaded from global memory straight in MFMA
// operand layout, split into the three bf16 planes ONCE, and stay in 96 VGPRs while the workgroup walks over G row
// blocks `it` of h1T (G = 16 when the launch has enough seeds: dz is then read from memory exactly once).  Per step
// only the 64 x 32 h1T tile moves: one float4 per thread, 8 steps (64 KB per CU) in flight in registers, split by the
// loading thread and staged as planes in LDS (3 x 64 rows x 64 B per step; a ring of two 4-step groups, 96 KB); a wave
// reads the 4 row-block fragments (12 ds_read_b128) for its 24 MFMAs.
// K slot (kg = lane>>4, j) of a 32-sample step stands for sample 16 (j>>2) + 4 kg + (j&3) (two float4 per lane of a
// dz row); the A planes are stored in that order with an XOR swizzle of the 16-B quads, quad = kg ^ {0,3,2,1}[(row>>2)&3],
// which makes the unpadded 64-B rows conflict-free for the four 16-lane groups of ds_read_b128 (MI355X guide).
#define QY_SLAB QW_SLAB
#define QY_KS 32
#define QY_APL (64 * 64)                     // one A plane: 64 rows x 32 bf16
#define QY_ABUF (3 * QY_APL)                 // 12,288 B
#define QY_LDS (8 * QY_ABUF)                  // 98,304 B (dynamic LDS)
#define QY_LDS_ACC (QY_LDS + 4 * 16 * QN_THREADS)   // + the running sums of the ACC form
static_assert(QY_SLAB == 256, "8 steps of 32 samples");
PQN_D int qy_swz(int row) { return (0x1230 >> (((row >> 2) & 3) * 4)) & 3; }
// Round 4: dz arrives as bf16 planes in B-fragment order (dzw_index; split once by T1's head), so the fragments are
// plain dwordx4 loads.  Two forms of one body:
//   ACC = false  workgroup = (group of G row blocks, slab ks, seed): dz of ONE slab resident, walks over G row blocks and
//                writes a partial tile per row block into split-K slab ks (folded in slab order by qnet_grad_reduce_kernel);
//   ACC = true   workgroup = (row block it, seed), walks over ALL slabs: per slab a partial tile P_ks from zero, then
//                tot += P_ks in slab order -- the very sum the reduction forms from the partial slabs, bit for bit -- and
//                writes tot into slab 0 ONCE: no split-K partials in HBM at all (16 seeds x 4096 samples: 134 MB written
//                + 134 MB re-read per optimizer step before).  The dz fragments of step u are re-loaded IN PLACE for the
//                next slab right behind the step's last MFMA (8 steps of latency cover).  Taken when row blocks x seeds
//                fill the chip (launch_train).
#ifndef T2_ABL
#define T2_ABL 0   // profiling builds only (-DT2_ABL=mask): 1 no MFMAs, 2 no operand split in stage_a, 4 no dz reload, 8 no LDS fragment reads, 16 no barrier
#endif
template <bool ACC>
__global__ __launch_bounds__(QN_THREADS) void qnet_fc1_wgrad_x3_kernel(int nb, const float *__restrict__ h1T,
                                                                       const unsigned short *__restrict__ dzw,
                                                                       float *__restrict__ wpart, long long ws_stride,
                                                                       int G, unsigned long long *__restrict__ stamps) {
#define T2_STAMP(k) do { if (stamps && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) stamps[k] = __builtin_readcyclecounter(); } while (0)
  static_assert(QN_WAVES == 8, "one column block per wave");
  extern __shared__ __attribute__((aligned(16))) char abuf[];   // QY_LDS: two groups x four step tiles
  T2_STAMP(0);
  // ACC: XCD-aware (seed, row block).  Workgroups go to the 8 XCDs round-robin in dispatch order; when the seeds divide
  // over the XCDs, XCD k runs all 16 row blocks of seeds [k S/8, (k+1) S/8): a seed's dz planes (3 MB, read by each of its
  // 16 workgroups) then cross ONE L2 instead of eight.
  int seed = blockIdx.z, it0 = blockIdx.x * G, ks0 = blockIdx.y;
  if (ACC) {
    it0 = blockIdx.x;
    ks0 = 0;
    if ((gridDim.z & 7) == 0) {
      const unsigned lin = blockIdx.z * gridDim.x + blockIdx.x, xcd = lin & 7u, slot = lin >> 3;
      seed = xcd * (gridDim.z >> 3) + slot / gridDim.x;
      it0 = slot % gridDim.x;
    }
  }
  h1T += seed * ws_stride;   // seed slice
  dzw += 2 * seed * ws_stride;
  wpart += seed * ws_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int nks = qw_slabs(nb);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  constexpr int NST = QY_SLAB / QY_KS;   // 8 steps per (row block, slab) = the prefetch distance
  const int NQ = ACC ? nks : G;          // outer iterations: slabs (ACC) or row blocks
  // h1T loader: thread -> (row, 4 consecutive samples) of the 64 x 32 step tile
  const int arow = tid >> 3, q8 = tid & 7;
  const int a_dst = arow * 64 + (((q8 & 3) ^ qy_swz(arow)) << 4) + ((q8 >> 2) << 3);
  // Every prefetch is UNCONDITIONAL (row block / slab clamped into range, the value masked when it is consumed):
  // a load under a condition, or a select right behind it, makes the compiler drain the whole queue
  // (s_waitcnt vmcnt(0)) at every step, which serialises the HBM round trips.
  auto fetch = [&](int qn, int u) -> f32x4 {   // h1 slab-major (h1s_index): step tile = 8 KB contiguous, 16 B per thread
    const int qc = min(qn, NQ - 1);
    const size_t ks = ACC ? qc : ks0, it = ACC ? it0 : it0 + qc;
    return *reinterpret_cast<const f32x4 *>(h1T + (((ks * 16 + it) * 8 + u) * 2048 + 4 * tid));
  };
  // samples past nb (ragged last slab): h1 masked to zero here; the dz planes of that range are zeroed by the host
  auto masked = [&](const f32x4 &v, int qn, int u) -> f32x4 {
    const int ks = ACC ? min(qn, NQ - 1) : ks0;
    return (ks * QY_SLAB + 4 * q8 + QY_KS * u < nb) ? v : zero4;
  };
  f32x4 pre[NST];
#pragma unroll
  for (int u = 0; u < NST; ++u) pre[u] = fetch(0, u);
  // the wave's dz fragments of one slab: column block `wave`, 8 steps x 3 planes x one dwordx4 (96 VGPRs)
  const u32x4 *dzq = reinterpret_cast<const u32x4 *>(dzw);
  const size_t psq = (size_t)nks * (QY_SLAB * QN_HID / 8);   // plane stride in dwordx4
  auto dz_frag = [&](int ks, int u) -> X3Frag {
    const size_t e = (((size_t)ks * 8 + wave) * 8 + u) * 64 + lane;
    X3Frag f;
    f.h = dzq[e]; f.m = dzq[psq + e]; f.l = dzq[2 * psq + e];
    return f;
  };
  X3Frag bp[NST];
#pragma unroll
  for (int u = 0; u < NST; ++u) bp[u] = dz_frag(ks0, u);
  int a_off[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int row = 16 * a + r;
    a_off[a] = row * 64 + ((kg ^ qy_swz(row)) << 4);
  }
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  auto stage_a = [&](char *t, const f32x4 &v) {
    unsigned h0, m0, l0, h1, m1, l1;
    if (T2_ABL & 2) {
      h0 = __float_as_uint(v.x); m0 = __float_as_uint(v.y); l0 = h0 ^ m0; h1 = __float_as_uint(v.z); m1 = __float_as_uint(v.w); l1 = h1 ^ m1;
    } else {
      x3_split2(v.x, v.y, h0, m0, l0);
      x3_split2(v.z, v.w, h1, m1, l1);
    }
    *reinterpret_cast<u32x2 *>(t + a_dst) = u32x2{h0, h1};
    *reinterpret_cast<u32x2 *>(t + QY_APL + a_dst) = u32x2{m0, m1};
    *reinterpret_cast<u32x2 *>(t + 2 * QY_APL + a_dst) = u32x2{l0, l1};
  };
  u32x4 ah[4], am[4], al[4];
  auto load_plane = [&](const char *t, int plane, u32x4 (&dst)[4]) {
    if ((T2_ABL & 8) && t != abuf) return;
#pragma unroll
    for (int a = 0; a < 4; ++a) dst[a] = *reinterpret_cast<const u32x4 *>(t + plane * QY_APL + a_off[a]);
  };
  // Steps are grouped in fours (128 samples): ONE barrier per group.  A per-step barrier keeps all eight waves in
  // lockstep -- MFMA bursts, splits and LDS latency then add up instead of overlapping, measured 420 of 1310 counter
  // ticks per step -- so a group's planes (4 x 12 KB) are staged while the previous group computes, into the other half
  // of an 8-tile LDS ring; within a group a wave runs free: it reloads each fragment plane from the NEXT step's tile
  // as soon as the current step's MFMAs on that plane have issued.
  auto tile = [&](int half, int slot) -> char * { return abuf + (half * 4 + slot) * QY_ABUF; };
  // prologue: group 0 of the first iteration
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    stage_a(tile(0, u), masked(pre[u], 0, u));
    pre[u] = fetch(1, u);
  }
  __syncthreads();
  load_plane(tile(0, 0), 2, al);
  load_plane(tile(0, 0), 1, am);
  load_plane(tile(0, 0), 0, ah);
  T2_STAMP(1);
  // ACC: the running sum over the slabs lives in LDS (4 x 16 B per thread behind the staging ring, touched once per slab):
  // 16 more live registers would spill (the kernel sits at 239 of 256 with the dz fragments resident)
  f32x4 *totl = reinterpret_cast<f32x4 *>(abuf + QY_LDS) + tid;
  f32x4 acc_b[4], acc_s[4];
  if (ACC) {
#pragma unroll
    for (int a = 0; a < 4; ++a) totl[a * QN_THREADS] = zero4;
  }
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int a = 0; a < 4; ++a) { acc_b[a] = zero4; acc_s[a] = zero4; }
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      // All NST steps always run (samples past nb are zeros in both operands) and the last group of the last iteration
      // stages / reads harmless extra tiles: no data-dependent control flow inside the pipeline.
      const int half = (u >> 2) & 1, slot = u & 3;
      const bool last = (slot == 3);                          // last step of its group: the next tile is behind the barrier
      const char *tnext = last ? tile(half ^ 1, 0) : tile(half, slot + 1);
      const int us = (u + 4) % NST, qs = q + (u + 4) / NST;   // step staged now: same slot of the next group
#define QX_ROW(AP, BP, ACC_)                           \
      if (!(T2_ABL & 1)) x3_grp4(ACC_[0], AP[0], bp[u].BP, ACC_[1], AP[1], bp[u].BP, ACC_[2], AP[2], bp[u].BP, ACC_[3], AP[3], bp[u].BP);
      QX_ROW(al, h, acc_s)
      __builtin_amdgcn_sched_barrier(0);
      if (!last) load_plane(tnext, 2, al);
      __builtin_amdgcn_sched_barrier(0);
      QX_ROW(am, m, acc_s)
      QX_ROW(am, h, acc_b)
      __builtin_amdgcn_sched_barrier(0);
      if (!last) load_plane(tnext, 1, am);
      __builtin_amdgcn_sched_barrier(0);
      QX_ROW(ah, l, acc_s)
      QX_ROW(ah, m, acc_b)
      QX_ROW(ah, h, acc_b)
#undef QX_ROW
      __builtin_amdgcn_sched_barrier(0);
      if (!last) load_plane(tnext, 0, ah);
      if (ACC && !(T2_ABL & 4)) bp[u] = dz_frag(min(q + 1, NQ - 1), u);   // next slab's fragments of this step, in place (unconditional)
      stage_a(tile(half ^ 1, slot), masked(pre[us], qs, us));   // split once, stored as planes
      pre[us] = fetch(qs + 1, us);                               // refill the slot: the same step one iteration further on
      if (last) {
        if (!(T2_ABL & 16)) __syncthreads();   // the next group's planes complete; this group's half may be overwritten from now on
        load_plane(tnext, 2, al);
        load_plane(tnext, 1, am);
        load_plane(tnext, 0, ah);
      }
      if (q == 0) T2_STAMP(2 + u);
    }
    x3_drain(acc_b[0], acc_b[1], acc_b[2], acc_b[3]);
    x3_drain(acc_s[0], acc_s[1], acc_s[2], acc_s[3]);
    if (ACC) {
#pragma unroll
      for (int a = 0; a < 4; ++a) totl[a * QN_THREADS] += acc_b[a] + acc_s[a];   // 0 + P_0 + P_1 + ...: the reduction's own order
    } else {
      f32x4 *out = reinterpret_cast<f32x4 *>(wpart + (size_t)ks0 * QN_H1 * QN_HID);
#pragma unroll
      for (int a = 0; a < 4; ++a) out[((4 * (it0 + q) + a) * 8 + wave) * 64 + lane] = acc_b[a] + acc_s[a];
    }
    T2_STAMP(10 + min(q, 15));
  }
  if (ACC) {
    f32x4 *out = reinterpret_cast<f32x4 *>(wpart);
#pragma unroll
    for (int a = 0; a < 4; ++a) out[((4 * it0 + a) * 8 + wave) * 64 + lane] = totl[a * QN_THREADS];
  }
}

// fp16-operand variant (matmul_f16): operands packed tile-major by T1 (h1P[tile][1024][16], dzP[tile][128][16]
// halves, dz pre-scaled by dz_scale), one v_mfma_f32_16x16x16_f16 per row block and 16-sample tile, f32
// accumulation, result scaled back.  Same grid and output layout as the f32 kernel.
__global__ __launch_bounds__(QN_THREADS) void qnet_fc1_wgrad_f16_kernel(int nb, const float *__restrict__ h1T,
                                                                        const float *__restrict__ dzT,
                                                                        float *__restrict__ wpart, long long ws_stride,
                                                                        float inv_scale) {
  constexpr int TS = 4;                  // 16-sample tiles per step (64 samples)
  constexpr int ROWS = 64 + 128;
  constexpr int RS = TS * 16 + 8;        // LDS row stride in halves (144 B): conflict-free 8-B fragment reads
  __shared__ __attribute__((aligned(16))) _Float16 tile[2][ROWS * RS];
  const _Float16 *h1P = reinterpret_cast<const _Float16 *>(h1T + blockIdx.z * ws_stride);
  const _Float16 *dzP = reinterpret_cast<const _Float16 *>(dzT + blockIdx.z * ws_stride);
  wpart += blockIdx.z * ws_stride;
  const int tid = threadIdx.x, lane = tid & 63, cb = tid >> 6;
  const int it = blockIdx.x, ks = blockIdx.y;
  const int o = lane & 15, kq = 4 * (lane >> 4);
  const int ntiles = nb / QN_TILE;
  const int t0 = ks * (QW_SLAB / QN_TILE);
  const int tiles = min(QW_SLAB / QN_TILE, ntiles - t0);
  const int nsteps = (tiles + TS - 1) / TS;
  // loader: 1536 16-B pieces per step (A: 64 rows x 4 tiles x 2, B: 128 rows x 4 tiles x 2) = 3 per thread
  const _Float16 *src[3];
  int dst[3], tl[3];
  size_t adv[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int u = tid + QN_THREADS * j;
    if (u < 512) {
      const int t = u >> 7, rem = u & 127, r = rem >> 1, hh = rem & 1;
      src[j] = h1P + ((size_t)(t0 + t) * QN_H1 + 64 * it + r) * QN_TILE + 8 * hh;
      dst[j] = r * RS + t * 16 + 8 * hh;
      tl[j] = t;
      adv[j] = (size_t)TS * QN_H1 * QN_TILE;
    } else {
      const int v = u - 512, t = v >> 8, rem = v & 255, r = rem >> 1, hh = rem & 1;
      src[j] = dzP + ((size_t)(t0 + t) * QN_HID + r) * QN_TILE + 8 * hh;
      dst[j] = (64 + r) * RS + t * 16 + 8 * hh;
      tl[j] = t;
      adv[j] = (size_t)TS * QN_HID * QN_TILE;
    }
  }
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  f32x4 acc[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  f16x8 pre[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) pre[j] = (tl[j] < tiles) ? *reinterpret_cast<const f16x8 *>(src[j]) : zero8;
  for (int st = 0; st < nsteps; ++st) {
    _Float16 *t = tile[st & 1];
#pragma unroll
    for (int j = 0; j < 3; ++j) *reinterpret_cast<f16x8 *>(t + dst[j]) = pre[j];
    if (st + 1 < nsteps) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
        pre[j] = (tl[j] + TS * (st + 1) < tiles) ? *reinterpret_cast<const f16x8 *>(src[j] + adv[j] * (st + 1)) : zero8;
    }
    __syncthreads();   // tile[st&1] complete; tile[(st+1)&1] was last read two steps ago
#pragma unroll
    for (int g = 0; g < TS; ++g) {
      const f16x4 b = *reinterpret_cast<const f16x4 *>(t + (64 + 16 * cb + o) * RS + 16 * g + kq);
      f16x4 a4[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) a4[a] = *reinterpret_cast<const f16x4 *>(t + (16 * a + o) * RS + 16 * g + kq);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = f16_mfma_tied(a4[a], b, acc[a]);
    }
  }
  f16_drain(acc[0], acc[1], acc[2], acc[3]);
  f32x4 *out = reinterpret_cast<f32x4 *>(wpart + (size_t)ks * QN_H1 * QN_HID);
#pragma unroll
  for (int a = 0; a < 4; ++a) out[((4 * it + a) * 8 + cb) * 64 + lane] = acc[a] * inv_scale;
}

// ---------------------------------------------------------------------------
// T3a: fold partials into the flat gradient (kernel layout) and emit per-block sums of
// squares + the *count snapshot for radam_apply (same scratch protocol as radam_norm_kernel).
// Blocks [0, QR_W1_BLOCKS): fc1 region, float4 per lane, sum over the K-splits.
// Remaining blocks: 64 "small" elements each (lane = element), the four waves split the tile records, fixed order.
// ---------------------------------------------------------------------------
static_assert(QN_H1 * QN_HID == PQN_FC1_ELEMS, "pqn_fold.h");   // the fold itself: pqn_fold.h (shared with radam_apply_kernel<true>, pqn_algo.hip)

__global__ __launch_bounds__(256) void qnet_grad_reduce_kernel(pqn_fold_args_t fa, float *__restrict__ grad,
                                                               const int32_t *__restrict__ count, float *__restrict__ scratch,
                                                               long long theta_stride) {
  __shared__ float s_part[4];
  __shared__ float s_red[4][64];
  const long long s = blockIdx.y;   // seed slice
  scratch += s * fa.ws_stride;
  pqn_f4 g4;
  float g_small, ss;
  int i_small;
  pqn_fold_block(fa, (int)blockIdx.x, s, grad + s * theta_stride, s_red, g4, g_small, i_small, ss);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    scratch[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    if (blockIdx.x == 0) reinterpret_cast<int32_t *>(scratch)[1023] = count[s];
  }
}

// ===========================================================================
// host side
// ===========================================================================
static int align4(int x) { return (x + 3) & ~3; }

extern "C" int pqn_cnn_layout(int32_t c, int32_t a, pqn_cnn_layout_t *L) { return pqn_cnn_layout_ex(c, a, 0, L); }

extern "C" int pqn_cnn_layout_ex(int32_t c, int32_t a, int32_t matmul_f16, pqn_cnn_layout_t *L) {
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
  PQN_REQUIRE(matmul_f16 >= 0 && matmul_f16 <= 3, "pqn_cnn_layout_ex: operand mode %d (0 f32, 1 f16, 2 bf16x3, 3 f16x2)", matmul_f16);
  L->pos_f16x2 = matmul_f16 == 3;                          // 3: bf16x3 everywhere but the position-parallel kernels, which run f16x2
  if (matmul_f16 == 3) matmul_f16 = 2;
  L->matmul_f16 = matmul_f16;                              // 0: f32 MFMA; 1: fp16 operands; 2: bf16x3 split operands
  L->off_w1h = off;                                        // mode 1: 2 x 131072 halves = 131072 floats behind the parameters
  L->alloc = matmul_f16 == 1 ? off + QN_H1 * QN_HID            // two fp16 copies
                             : (matmul_f16 == 2 ? off + 3 * QN_H1 * QN_HID : off);   // mode 2: 6 bf16 planes (3 forward + 3 dgrad order)
  if (L->pos_f16x2) L->alloc += 2 * QN_H1 * QN_HID;        // + 4 fp16 planes (H2_PLANES_OFF behind off_w1h)
  return PQN_OK;
}

template <int C>
static int launch_fwd(int n, const uint32_t *bits, const float *theta, const pqn_cnn_layout_t &L, float *q,
                      int32_t *action, float *qmax, float eps, uint64_t key, const float *eps_dev,
                      const uint64_t *key_dev, hipStream_t st) {
  const size_t smem = cnn_smem_bytes<C>();
  static pqn_once_per_device attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_fwd_kernel<C>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int ablate = pqn_opt(PQN_OPT_ABLATE);  // profiling only
  hipLaunchKernelGGL((qnet_cnn_fwd_kernel<C>), dim3((n + QN_TILE - 1) / QN_TILE), dim3(QN_THREADS), smem, st, n, bits, theta, L,
                     q, action, qmax, eps, key, eps_dev, key_dev, ablate);
  return pqn_check_launch("pqn_qnet_cnn_forward");
}

int pqn_qnet_cnn_forward_dyn(const pqn_cnn_layout_t &L, int n, const uint32_t *obs_bits, const float *theta, float *q,
                             int32_t *action, float *qmax, float eps, uint64_t key, const float *eps_dev,
                             const uint64_t *key_dev, hipStream_t st) {
  switch (L.c) {
    case 4: return launch_fwd<4>(n, obs_bits, theta, L, q, action, qmax, eps, key, eps_dev, key_dev, st);
    case 6: return launch_fwd<6>(n, obs_bits, theta, L, q, action, qmax, eps, key, eps_dev, key_dev, st);
    case 7: return launch_fwd<7>(n, obs_bits, theta, L, q, action, qmax, eps, key, eps_dev, key_dev, st);
    case 10: return launch_fwd<10>(n, obs_bits, theta, L, q, action, qmax, eps, key, eps_dev, key_dev, st);
    default: pqn_set_error("pqn_qnet_cnn_forward: unsupported channel count %d", L.c); return PQN_E_UNSUPPORTED;
  }
}

extern "C" int pqn_qnet_cnn_forward(const pqn_cnn_layout_t *L, int32_t n, const uint32_t *obs_bits, const float *theta,
                                    float *q, int32_t *action, float *qmax, float eps, uint64_t key, void *stream) {
  PQN_REQUIRE(L && obs_bits && theta, "pqn_qnet_cnn_forward: NULL argument");
  PQN_REQUIRE(n > 0, "pqn_qnet_cnn_forward: n must be > 0");
  PQN_REQUIRE(q || action || qmax, "pqn_qnet_cnn_forward: no output requested");
  return pqn_qnet_cnn_forward_dyn(*L, n, obs_bits, theta, q, action, qmax, eps, key, nullptr, nullptr,
                                  (hipStream_t)stream);
}


template <int C, class Env>
static int launch_rollout(int env_id, const pqn_cnn_layout_t &L, int n, int t_len, uint32_t *state, uint32_t *bits, const float *theta,
                          const pqn_step_out_t &rec, int32_t *action, float *qmax, float *last_q, const float *eps_dev,
                          const uint64_t *keys, float rscale, int store_obs, hipStream_t st, int n_per_seed,
                          long long theta_stride, int keys_stride, int pin_form) {
  // position-structure form (pqn_qnet_pos.hip, round 5): one workgroup per 256 envs -- taken when the launch gives most CUs a
  // workgroup (option rollout_pos: 0 never, 1 auto, 2 whenever the shape allows; pin_form: from the envs per seed alone)
  {
    const int rp = pqn_opt(PQN_OPT_ROLLOUT_POS);
    const int nps = n_per_seed > 0 ? n_per_seed : n;
    // waves per workgroup (32 envs each): 8; f16x2 layouts outside pin_form: 4 / 2 when that is what puts >= 160 workgroups on the chip
    int nw = 0;
    if (pin_form) nw = nps >= 2048 ? 8 : 0;
    else if (nps % 256 == 0 && n / 256 >= 160) nw = 8;
    else if (L.pos_f16x2 && nps % 128 == 0 && n / 128 >= 160) nw = 4;
    else if (L.pos_f16x2 && nps % 64 == 0 && n / 64 >= 160) nw = 2;
    if (rp == 2 && nw == 0) nw = nps % 256 == 0 ? 8 : (L.pos_f16x2 ? (nps % 128 == 0 ? 4 : 2) : 0);
    if (nw != 0 && L.pos_f16x2 && !pin_form) {
      const int ow = pqn_opt(PQN_OPT_POS_WAVES);
      if ((ow == 8 || ow == 4 || ow == 2) && nps % (32 * ow) == 0) nw = ow;
    }
    if (rp && L.matmul_f16 == 2 && nw != 0 && pqn_cnn_pos_rollout_supported(env_id, C, L.a, n, n_per_seed, nw)) {
      pqn_note_kernel_form(1, PQN_FORM_POS);
      return pqn_cnn_pos_rollout(env_id, L, n, t_len, state, bits, theta, rec, action, qmax, last_q, eps_dev, keys, rscale, store_obs, st,
                                 n_per_seed, theta_stride, keys_stride, nw);
    }
  }
  const size_t smem = cnn_smem_bytes<C>();
  static pqn_once_per_device attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_rollout_kernel<C, Env, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_rollout_kernel<C, Env, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_rollout_kernel<C, Env, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  // one instantiation per operand mode (as the training kernel): the env state lives in registers across the T-step
  // loop, and carrying all three fc1 / conv variants in one kernel cost 27 spilled VGPRs
  // pair form (32 envs per workgroup, shared fc1 weight stream): bf16x3 mode, when its LDS layout fits, the envs (of
  // each seed) come in pairs of tiles and the grid still gives every CU a workgroup; PQN_ROLLOUT_PAIR=0 / 2: never / always
  const int rp_env = pqn_opt(PQN_OPT_ROLLOUT_PAIR);
  constexpr size_t pair_smem = sizeof(float) * (2 * QN_TILE * QN_H1S + 2 * QN_TILE * QN_ZS + PairSmem<C>::WCN + QN_HP_FLOATS) +
                               sizeof(uint32_t) * 2 * PairSmem<C>::BITN;
  const bool use_pair = rp_env && L.matmul_f16 == 2 && pair_smem <= 160 * 1024 && n % (2 * QN_TILE) == 0 &&
                        (n_per_seed <= 0 || n_per_seed % (2 * QN_TILE) == 0) && (n / (2 * QN_TILE) >= 256 || rp_env == 2);
  if (use_pair) {
    static pqn_once_per_device pattr;
    if (pattr.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_rollout_pair_kernel<C, Env>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_smem);
    }
    hipLaunchKernelGGL((qnet_cnn_rollout_pair_kernel<C, Env>), dim3(n / (2 * QN_TILE)), dim3(QN_THREADS), pair_smem, st, n,
                       t_len, state, bits, theta, L, action, qmax, rec.reward, rec.done, rec.discount,
                       rec.returned_episode_returns, rec.returned_episode_lengths, rec.timestep, last_q, eps_dev, keys,
                       rscale, store_obs, n_per_seed, theta_stride, keys_stride);
    pqn_note_kernel_form(1, PQN_FORM_PAIR);
    return pqn_check_launch("pqn_qnet_cnn_rollout");
  }
  pqn_note_kernel_form(1, PQN_FORM_SINGLE);
  auto kern = L.matmul_f16 == 2 ? &qnet_cnn_rollout_kernel<C, Env, 2>
                                : (L.matmul_f16 == 1 ? &qnet_cnn_rollout_kernel<C, Env, 1> : &qnet_cnn_rollout_kernel<C, Env, 0>);
  hipLaunchKernelGGL(kern, dim3((n + QN_TILE - 1) / QN_TILE), dim3(QN_THREADS), smem, st, n,
                     t_len, state, bits, theta, L, action, qmax, rec.reward, rec.done, rec.discount,
                     rec.returned_episode_returns, rec.returned_episode_lengths, rec.timestep, last_q, eps_dev, keys,
                     rscale, store_obs, n_per_seed, theta_stride, keys_stride);
  return pqn_check_launch("pqn_qnet_cnn_rollout");
}

// internal (pqn_update.hip): rec.* point at the [T][n] transition arrays
int pqn_qnet_cnn_rollout(int env_id, const pqn_cnn_layout_t &L, int n, int t_len, uint32_t *state, uint32_t *bits,
                         const float *theta, const pqn_step_out_t &rec, int32_t *action, float *qmax, float *last_q,
                         const float *eps_dev, const uint64_t *keys, float rscale, int store_obs, hipStream_t st,
                         int n_per_seed, long long theta_stride, int keys_stride, int pin_form) {
  if (n_per_seed > 0 && n_per_seed % QN_TILE != 0) {
    pqn_set_error("pqn_qnet_cnn_rollout: seed batching needs NUM_ENVS %% %d == 0 (got %d)", QN_TILE, n_per_seed);
    return PQN_E_INVALID;
  }
  switch (env_id) {
    case PQN_ENV_BREAKOUT:
      if (L.c == 4) return launch_rollout<4, Breakout>(env_id, L, n, t_len, state, bits, theta, rec, action, qmax, last_q, eps_dev, keys, rscale, store_obs, st, n_per_seed, theta_stride, keys_stride, pin_form);
      break;
    case PQN_ENV_ASTERIX:
      if (L.c == 4) return launch_rollout<4, Asterix>(env_id, L, n, t_len, state, bits, theta, rec, action, qmax, last_q, eps_dev, keys, rscale, store_obs, st, n_per_seed, theta_stride, keys_stride, pin_form);
      break;
    case PQN_ENV_FREEWAY:
      if (L.c == 7) return launch_rollout<7, Freeway>(env_id, L, n, t_len, state, bits, theta, rec, action, qmax, last_q, eps_dev, keys, rscale, store_obs, st, n_per_seed, theta_stride, keys_stride, pin_form);
      break;
    case PQN_ENV_SPACEINVADERS:
      if (L.c == 6) return launch_rollout<6, SpaceInvaders>(env_id, L, n, t_len, state, bits, theta, rec, action, qmax, last_q, eps_dev, keys, rscale, store_obs, st, n_per_seed, theta_stride, keys_stride, pin_form);
      break;
    default: break;
  }
  pqn_set_error("pqn_qnet_cnn_rollout: env %d / %d channels has no fused rollout", env_id, L.c);
  return PQN_E_UNSUPPORTED;
}

extern "C" int pqn_cnn_rollout(int env_id, const pqn_cnn_layout_t *layout, int32_t num_envs, int32_t num_steps,
                               uint32_t *state, uint32_t *obs_bits, int32_t store_obs, const float *theta,
                               const pqn_step_out_t *rec, int32_t *action, float *qmax, float *last_q,
                               const float *eps_dev, const uint64_t *keys_dev, float rew_scale, void *stream) {
  PQN_REQUIRE(layout && state && obs_bits && theta && eps_dev && keys_dev, "pqn_cnn_rollout: NULL argument");
  PQN_REQUIRE(num_envs > 0 && num_steps > 0, "pqn_cnn_rollout: bad shape n=%d T=%d", num_envs, num_steps);
  pqn_step_out_t none = {};
  const pqn_step_out_t &r = rec ? *rec : none;
  PQN_REQUIRE(r.obs == nullptr && r.obs_bits == nullptr,
              "pqn_cnn_rollout: observations are recorded through obs_bits[T+1][n][OW], rec->obs / rec->obs_bits must be NULL");
  return pqn_qnet_cnn_rollout(env_id, *layout, num_envs, num_steps, state, obs_bits, theta, r, action, qmax, last_q, eps_dev,
                              keys_dev, rew_scale, store_obs != 0, (hipStream_t)stream, 0, 0, 0, pqn_opt(PQN_OPT_PIN_FORM));
}

extern "C" int pqn_cnn_rollout_seeds(int env_id, const pqn_cnn_layout_t *layout, int32_t num_seeds, int32_t envs_per_seed,
                                     int32_t num_steps, uint32_t *state, uint32_t *obs_bits, int32_t store_obs,
                                     const float *theta, int64_t theta_stride, const pqn_step_out_t *rec, int32_t *action,
                                     float *qmax, float *last_q, const float *eps_dev, const uint64_t *keys_dev,
                                     int32_t keys_stride, float rew_scale, void *stream) {
  PQN_REQUIRE(layout && state && obs_bits && theta && eps_dev && keys_dev, "pqn_cnn_rollout_seeds: NULL argument");
  PQN_REQUIRE(num_seeds >= 1 && envs_per_seed > 0 && num_steps > 0 && keys_stride >= num_steps && theta_stride >= 0,
              "pqn_cnn_rollout_seeds: bad shape seeds=%d n=%d T=%d", num_seeds, envs_per_seed, num_steps);
  pqn_step_out_t none = {};
  const pqn_step_out_t &r = rec ? *rec : none;
  PQN_REQUIRE(r.obs == nullptr && r.obs_bits == nullptr, "pqn_cnn_rollout_seeds: rec->obs / rec->obs_bits must be NULL");
  return pqn_qnet_cnn_rollout(env_id, *layout, num_seeds * envs_per_seed, num_steps, state, obs_bits, theta, r, action, qmax,
                              last_q, eps_dev, keys_dev, rew_scale, store_obs != 0, (hipStream_t)stream, envs_per_seed,
                              theta_stride, keys_stride, pqn_opt(PQN_OPT_PIN_FORM));
}

// ---------------------------------------------------------------------------
// kernel timer: HIP events recorded on the launch stream around the dominant kernel (T1), read back
// by bench.py for the roofline line.  Off by default; costs two event records per launch when on.
// ---------------------------------------------------------------------------
#define PQN_PROF_MAX 4096
static struct {
  bool on = false, created = false;
  int mode = 0;   // pqn_prof_enable(mode): 1 = the CNN training kernel (T1), 2 = the wide-MLP GEMM kernel
  int n = 0;
  hipEvent_t s[PQN_PROF_MAX], e[PQN_PROF_MAX];
} g_prof;
// other translation units (pqn_bigmlp.hip) bracket their kernel with these; begin returns false when this launch is not timed
bool pqn_prof_begin(int mode, hipStream_t st) {
  if (!g_prof.on || g_prof.mode != mode || g_prof.n >= PQN_PROF_MAX) return false;
  (void)hipEventRecord(g_prof.s[g_prof.n], st);
  return true;
}
void pqn_prof_end(hipStream_t st) { (void)hipEventRecord(g_prof.e[g_prof.n++], st); }

extern "C" int pqn_prof_enable(int32_t on) {
  if (on && !g_prof.created) {
    for (int i = 0; i < PQN_PROF_MAX; ++i) {
      if (hipEventCreate(&g_prof.s[i]) != hipSuccess || hipEventCreate(&g_prof.e[i]) != hipSuccess) {
        pqn_set_error("pqn_prof_enable: hipEventCreate failed");
        return PQN_E_HIP;
      }
    }
    g_prof.created = true;
  }
  g_prof.on = on != 0;
  g_prof.mode = on;
  g_prof.n = 0;
  return PQN_OK;
}

extern "C" int pqn_prof_read(int32_t *count, float *total_ms) {
  PQN_REQUIRE(count && total_ms, "pqn_prof_read: NULL argument");
  float tot = 0.0f;
  for (int i = 0; i < g_prof.n; ++i) {
    float ms = 0.0f;
    if (hipEventSynchronize(g_prof.e[i]) != hipSuccess || hipEventElapsedTime(&ms, g_prof.s[i], g_prof.e[i]) != hipSuccess) {
      pqn_set_error("pqn_prof_read: event query failed");
      return PQN_E_HIP;
    }
    tot += ms;
  }
  *count = g_prof.n;
  *total_ms = tot;
  g_prof.n = 0;
  return PQN_OK;
}

static unsigned long long *g_t1_stamps = nullptr;   // profiling only (PQN_T1_STAMPS=1)
static unsigned long long *g_t2_stamps = nullptr;   // likewise, bf16x3 T2: [0] start, [1] slab resident, [2..9] steps of the first row block, [10..] row blocks
extern "C" int pqn_debug_t2_stamps(unsigned long long *out /* host, 32 entries */) {
  if (!g_t2_stamps) return PQN_E_INVALID;
  if (hipMemcpy(out, g_t2_stamps, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return PQN_E_HIP;
  return PQN_OK;
}

extern "C" int pqn_debug_t1_stamps(unsigned long long *out /* host, 64 entries */) {
  if (!g_t1_stamps) return PQN_E_INVALID;
  if (hipMemcpy(out, g_t1_stamps, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return PQN_E_HIP;
  return PQN_OK;
}

// seeds per T1 -> T2 launch pair (see launch_train); also how many seeds one timed T1 launch covers (pqn_prof_read)
extern "C" int pqn_cnn_seed_group(int matmul_mode, int nseeds) {
  const int env_gs = pqn_opt(PQN_OPT_SEED_GROUP);   // profiling override
  if (env_gs > 0) return min(env_gs, nseeds);
  (void)matmul_mode;
  return nseeds;   // measured (16 seeds x 4096 samples, bf16x3): groups of 16 / 8 / 4 / 2 -> 50.4 / 50.7 / 51.3 / 55.5 ms per update
}

// the fold of the partials: its own launch (qnet_grad_reduce_kernel, then radam_apply_kernel), or -- `defer` given -- handed to the
// optimizer kernel, which folds, clips and applies in one launch (radam_apply_kernel<true>, pqn_algo.hip / pqn_fold.h)
static void launch_fold(const pqn_cnn_layout_t &L, int ntiles, int nks, int rec, const float *gpart, const float *wpart, float *grad,
                        const int32_t *count, float *scratch, float *loss_out, float *qv_out, float inv_b, const pqn_seeds_t &sd,
                        const float *gpos, int npos, hipStream_t st, pqn_fold_args_t *defer) {
  pqn_fold_args_t fa = {};
  fa.L = L; fa.ntiles = ntiles; fa.nks = nks; fa.rec = rec; fa.gpart = gpart; fa.wpart = wpart; fa.loss_out = loss_out; fa.qv_out = qv_out;
  fa.inv_b = inv_b; fa.gpos = gpos; fa.npos = npos; fa.ws_stride = sd.ws_stride; fa.lq_stride = sd.lq_stride; fa.valid = 1;
  if (defer) {
    *defer = fa;
    return;
  }
  hipLaunchKernelGGL(qnet_grad_reduce_kernel, dim3(grad_reduce_blocks(L.total), sd.nseeds), dim3(256), 0, st, fa, grad, count, scratch,
                     sd.theta_stride);
}

// a deferred fold launched by itself after all (the optimizer kernel's one-launch form did not take the shape)
int pqn_cnn_fold_launch(const pqn_fold_args_t &fa, float *grad, const int32_t *count, float *scratch, int nseeds, long long theta_stride,
                        hipStream_t st) {
  hipLaunchKernelGGL(qnet_grad_reduce_kernel, dim3(grad_reduce_blocks(fa.L.total), nseeds), dim3(256), 0, st, fa, grad, count, scratch,
                     theta_stride);
  return pqn_check_launch("pqn_qnet_cnn_grad (fold)");
}

template <int C>
static int launch_train(const pqn_cnn_layout_t &L, int nb, const int64_t *idx, const uint32_t *bits,
                        const int32_t *action, const float *target, const float *theta, const float *w1b, float *grad,
                        const int32_t *count, float *ws, float *loss_out, float *qv_out, const pqn_seeds_t &sd,
                        hipStream_t st, bool with_reduce, int part, int epoch_mb = -1, int epoch_nmb = 0,
                        pqn_fold_args_t *defer = nullptr) {
  // epoch_mb >= 0 (round 6): the rows / actions / targets of ALL epoch_nmb minibatches of this epoch were gathered once, by
  // pqn_qnet_cnn_epoch_gather, into the epoch region behind the workspace; this is minibatch epoch_mb of them
  // part: 0 = the whole gradient; 1 = the compute-bound training kernel(s) only; 2 = the HBM-bound rest (fc1 weight
  // gradient + fold of the partials) only -- pqn_cnn_update_seed_groups runs the two parts of a seed group on different
  // streams so that one group's tail overlaps the other group's training kernel
  const int ntiles = nb / QN_TILE, rec = small_record_floats(C, L.a);
  const int nks = (nb + QW_SLAB - 1) / QW_SLAB;
  float *scratch = ws;
  float *dzT = ws + 1024;
  float *h1T = dzT + (size_t)QN_HID * qw_ld(nb);
  float *gpart = h1T + (size_t)QN_H1 * qw_h1_cols(nb);
  // K-split form for small minibatches in the f32 operand mode; PQN_T1_KSPLIT=0 keeps the single-tile kernel
  // 0 off; 1 (default, measured best) = two launches: forward partial with 4 positions per workgroup, then head + backward
  // in one (8 positions per workgroup); three launches (forward partial, head, backward) with 4 (2), 8 (3) or 16 (4)
  // positions per workgroup
  const int ks_opt = pqn_opt(PQN_OPT_T1_KSPLIT);
  // ... and only while the launch is small: with many tiles in flight (seeds batched into the launch) the chip is filled
  // by whole tiles and the K-split form's recomputation costs more than the latency it hides (yaml-default run, S seeds x 8
  // tiles per launch, K-split vs single-tile: S = 1: 5.9 vs 9.9 s, 2: 6.9 vs 10.3, 4: 7.8 vs 11.1, 6: 10.2 vs 11.6, 10: 14.1
  // vs 12.4).  A seed therefore gets the same bits alone and inside a batch only as long
  // as both launches fall on the same side of this threshold (option t1_ksplit_tiles).
  // (sd.pin_form: the caller asked for a batch that is bit-identical to its solo runs -- the form then follows the solo rule)
  const bool use_ks = L.matmul_f16 == 0 && nb <= KS_MAX_NB && with_reduce && ks_opt != 0 &&
                      ntiles * (sd.pin_form ? 1 : sd.nseeds) <= pqn_opt(PQN_OPT_T1_KSPLIT_TILES);
  const bool ks_hb = ks_opt == 1 || ks_opt > 4;
  const int ks_ng = ks_hb ? 8 : (ks_opt == 3 ? 8 : (ks_opt == 4 ? 4 : 16));   // records (and backward groups) per tile
  float *wpart = gpart + (size_t)ntiles * rec;
  const size_t smem1 = train_smem_bytes<C>();
  if (use_ks) {
    // the K-split carve-up (pqn_qnet_cnn_workspace_floats covers it at every nb): behind the dz^T region (which the
    // shared head code still writes) zpart [tile][group][16][128], dz [tile][16][128], a record per (tile, group), a
    // weight-gradient slab per tile
    float *zpart = h1T, *dzbuf = zpart + (size_t)ntiles * ks_ng * QN_TILE * QN_HID;
    gpart = dzbuf + (size_t)ntiles * QN_TILE * QN_HID;
    wpart = gpart + (size_t)ntiles * ks_ng * rec;
    const float inv_b_ks = 1.0f / (float)nb;
    if (part == 2) return PQN_OK;   // the K-split form has no separate tail: part 1 ran everything
    pqn_note_kernel_form(0, PQN_FORM_KSPLIT);
#define KS_LAUNCH(PG_)                                                                                                               \
    do {                                                                                                                               \
      static pqn_once_per_device ks_attr;                                                                                                     \
      if (ks_attr.first()) {                                                                                                                  \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_ks_head_kernel<C, 64 / PG_>),                              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1);                                             \
      }                                                                                                                                \
      hipLaunchKernelGGL((qnet_cnn_ks_fwd_kernel<C, PG_>), dim3(64 / PG_, ntiles, sd.nseeds), dim3(KS_THREADS), 0, st, nb, idx, bits,  \
                         theta, L, zpart, sd);                                                                                         \
      hipLaunchKernelGGL((qnet_cnn_ks_head_kernel<C, 64 / PG_>), dim3(ntiles, sd.nseeds), dim3(QN_THREADS), smem1, st, nb, idx, action, \
                         target, theta, L, inv_b_ks, zpart, dzbuf, dzT, gpart, sd, 1.0f);                                              \
      hipLaunchKernelGGL((qnet_cnn_ks_bwd_kernel<C, PG_>), dim3(64 / PG_, ntiles, sd.nseeds), dim3(KS_THREADS), 0, st, nb, idx, bits,  \
                         theta, w1b, L, dzbuf, gpart, wpart, sd);                                                                      \
    } while (0)
    if (ks_hb) {
      static pqn_once_per_device hb_attr;
      if (hb_attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_ks_hb_kernel<C, 16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)smem1);
      }
      // the forward partials use 16 groups per tile: zpart needs ntiles * 16 tiles of [16][128], which is what the carve-up
      // reserves at the finest cut (dzbuf / gpart / wpart were placed for ks_ng = 8: move them behind the 16-group zpart)
      dzbuf = zpart + (size_t)ntiles * 16 * QN_TILE * QN_HID;
      gpart = dzbuf + (size_t)ntiles * QN_TILE * QN_HID;
      wpart = gpart + (size_t)ntiles * ks_ng * rec;
      hipLaunchKernelGGL((qnet_cnn_ks_fwd_kernel<C, 4>), dim3(16, ntiles, sd.nseeds), dim3(KS_THREADS), 0, st, nb, idx, bits, theta, L, zpart,
                         sd);
      hipLaunchKernelGGL((qnet_cnn_ks_hb_kernel<C, 16>), dim3(8, ntiles, sd.nseeds), dim3(QN_THREADS), smem1, st, nb, idx, bits, action, target,
                         theta, w1b, L, inv_b_ks, zpart, dzT, gpart, wpart, sd);
    } else if (ks_ng == 16) KS_LAUNCH(4);
    else if (ks_ng == 4) KS_LAUNCH(16);
    else KS_LAUNCH(8);
#undef KS_LAUNCH
    launch_fold(L, ntiles * ks_ng, ntiles, rec, gpart, wpart, grad, count, scratch, loss_out, qv_out, inv_b_ks, sd, nullptr, 0, st, defer);
    return pqn_check_launch("pqn_qnet_cnn_grad");
  }
  static pqn_once_per_device attr_set;
  if (attr_set.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_train_kernel<C, 0>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_train_kernel<C, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_train_kernel<C, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1);
  }
  const float inv_b = 1.0f / (float)nb;
  const int ablate = pqn_opt(PQN_OPT_ABLATE_TRAIN);  // profiling only
  // ---- the position-parallel form (pqn_qnet_pos.hip; round 5): gather -> forward (wave = 32 samples, z in registers, W1
  // planes shared through LDS) -> backward (wave = conv position, its W1 / dW1 rows resident) -> fold.  Option bwd_pos:
  // 0 never, 1 when the launch fills the chip, 2 whenever the shape allows (tests).  One workgroup per 256 samples (forward) and per (8 positions, chunk)
  // (backward): 16 seeds x 4096 samples give 256 + 256.  The summation orders depend on the minibatch size only, never on the
  // number of seeds in the launch; sd.pin_form takes the form from the minibatch size alone (a solo run then equals its batch).
  {
    pos_plan_t plan;
    const bool pos_full = with_reduce && pos_train_plan(L, nb, sd, plan);
    if (pos_full) {
      const int nch = plan.nch;
      const pos_ws_t PW = pos_ws_layout(nb, C, L.a);
      // the form's buffers live in the region that holds h1^T in the other forms (pqn_qnet_cnn_workspace_floats sizes it)
      PQN_REQUIRE(PW.end <= (long long)QN_H1 * qw_h1_cols(nb), "pqn_qnet_cnn_grad: position-parallel layout (%lld floats) exceeds the h1^T region",
                  PW.end);
      pqn_note_kernel_form(0, PQN_FORM_POS);
      if (part != 2) {
        // kernel timer (pqn_prof_enable): mode 1 = the dominant kernel (the backward), 3 = the forward kernel, 4 = gather +
        // forward + backward together (the whole value_and_grad of the step)
        const bool prof = g_prof.on && g_prof.n < PQN_PROF_MAX;
        const bool t_all = prof && g_prof.mode == 4, t_fwd = prof && g_prof.mode == 3, t_bwd = prof && g_prof.mode == 1;
        if (t_all) (void)hipEventRecord(g_prof.s[g_prof.n], st);
        int rc = PQN_OK;
        pos_ws_t PM = PW;        // the minibatch's view: its gathered rows either in the form's own region or inside the epoch region
        if (epoch_mb >= 0) {
          const pos_epoch_t E = pos_epoch_layout(nb, epoch_nmb, C);
          const long long e0 = pqn_qnet_cnn_workspace_floats(&L, nb) - (long long)(h1T - ws);   // epoch region, relative to h1T
          PM.mb_bits = e0 + E.bits + (long long)epoch_mb * nb * CnnCfg<C>::OW;
          PM.t32 = e0 + E.t32 + (long long)epoch_mb * (nb / 32) * pos_t32_words(C);
          PM.act = e0 + E.act + (long long)epoch_mb * nb;
          PM.tgt = e0 + E.tgt + (long long)epoch_mb * nb;
        } else {
          rc = pqn_cnn_pos_gather(L, nb, idx, bits, action, target, h1T, PW, sd, sd.nseeds, st);
        }
        if (t_fwd) (void)hipEventRecord(g_prof.s[g_prof.n], st);
        if (rc == PQN_OK) rc = pqn_cnn_pos_forward(L, nb, theta, inv_b, h1T, PM, sd, sd.nseeds, st, plan.nw);
        if (t_fwd) (void)hipEventRecord(g_prof.e[g_prof.n++], st);
        if (t_bwd) (void)hipEventRecord(g_prof.s[g_prof.n], st);
        if (rc == PQN_OK) rc = pqn_cnn_pos_backward(L, nb, nch, theta, h1T, wpart, PM, sd, sd.nseeds, st);
        if (t_bwd || t_all) (void)hipEventRecord(g_prof.e[g_prof.n++], st);
        if (rc != PQN_OK) return rc;
      }
      if (part != 1)
        launch_fold(L, nb / (32 * plan.nw), nch, rec, h1T + PW.recs, wpart, grad, count, scratch, loss_out, qv_out, inv_b, sd, h1T + PW.gpos,
                    8 * nch, st, defer);
      return pqn_check_launch("pqn_qnet_cnn_grad");
    }
  }
  if (!g_t1_stamps && getenv("PQN_T1_STAMPS") && pqn_not_capturing(st)) {   // (profiling only; an allocation is illegal under stream capture)
    if (hipMalloc(&g_t1_stamps, 64 * sizeof(unsigned long long)) != hipSuccess) g_t1_stamps = nullptr;
  }
  // matmul_f16: dz (O(1/nb)) is scaled by a power of two into fp16's normal range; products are scaled back in f32
  const float dz_scale = exp2f(floorf(log2f((float)nb)));
  // one instantiation per operand mode of the fc1 / conv products (pqn_cnn_layout_t.matmul_f16)
  auto t1 = L.matmul_f16 == 2 ? &qnet_cnn_train_kernel<C, 2> : (L.matmul_f16 == 1 ? &qnet_cnn_train_kernel<C, 1> : &qnet_cnn_train_kernel<C, 0>);
  // Seed groups: the launches can be cut into T1 -> T2 pairs over groups of seeds (PQN_SEED_GROUP, profiling) so that
  // the h1 a group hands from T1 to T2 (16 MB per seed at a 4096-sample minibatch) stays inside the 256 MB Infinity
  // Cache.  Measured, it does not pay (see pqn_cnn_seed_group): the default is one group.
  // pair form of T1 (two tiles per workgroup, shared fc1 weight stream): bf16x3 mode, when its LDS layout fits and the
  // minibatch has an even number of tiles; PQN_T1_PAIR=0 keeps the single-tile kernel (profiling / A-B runs)
  const int pair_env = pqn_opt(PQN_OPT_T1_PAIR);
  // (and the launch still has a workgroup for every CU: a single 4096-sample seed is 128 pairs, half a chip)
  const size_t pair_bytes = PairSmem<C>::bytes(L.a);
  const bool use_pair = pair_env && L.matmul_f16 == 2 && pair_bytes <= 160 * 1024 && ntiles >= 2 && (ntiles % 2) == 0 &&
                        ((ntiles / 2) * sd.nseeds >= 256 || pair_env == 2);   // PQN_T1_PAIR=2: pair form at any size (tests)
  if (use_pair) {
    static pqn_once_per_device pair_attr;
    if (pair_attr.first()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_cnn_train_pair_kernel<C>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
  }
  pqn_note_kernel_form(0, use_pair ? PQN_FORM_PAIR : PQN_FORM_SINGLE);
  const int gs_max = pqn_cnn_seed_group(L.matmul_f16, sd.nseeds);
  // bf16x3 fc1 weight gradient without split-K partials (qnet_fc1_wgrad_x3_kernel<true>): option t2_acc = 0 never, 1 (default)
  // when row blocks x seeds of a launch give every CU a workgroup, 2 always.  Both forms sum the slabs in the same order, so
  // a seed's bits do not depend on which one its launch took.
  const int acc_opt = pqn_opt(PQN_OPT_T2_ACC);
  const bool t2_acc = L.matmul_f16 == 2 && (acc_opt == 2 || (acc_opt == 1 && 16 * min(gs_max, sd.nseeds) >= 256));
  // (ragged minibatches: the tail of the last slab of the dz planes is zeroed by T1's head itself, see train_head_nt)
  for (int s0 = 0; s0 < sd.nseeds; s0 += gs_max) {
    const int gs = min(gs_max, sd.nseeds - s0);
    pqn_seeds_t sg = sd;
    sg.seed_base = s0;
    const long long wo = (long long)s0 * sd.ws_stride;
    const bool timed = part != 2 && g_prof.on && g_prof.mode == 1 && g_prof.n < PQN_PROF_MAX;
    if (timed) (void)hipEventRecord(g_prof.s[g_prof.n], st);
    if (part == 2) {
    } else if (use_pair)
      hipLaunchKernelGGL((qnet_cnn_train_pair_kernel<C>), dim3(ntiles / 2, gs), dim3(QN_THREADS), pair_bytes, st, nb, idx, bits,
                         action, target, theta, L, inv_b, dzT, h1T, gpart, ablate, sg, dz_scale, g_t1_stamps);
    else
    hipLaunchKernelGGL(t1, dim3(ntiles, gs), dim3(QN_THREADS), smem1, st, nb, idx, bits, action,
                       target, theta, w1b, L, inv_b, dzT, h1T, gpart, ablate, g_t1_stamps, sg, dz_scale);
    if (timed) (void)hipEventRecord(g_prof.e[g_prof.n++], st);
    if (part == 1) continue;
    if (L.matmul_f16 == 1)
      hipLaunchKernelGGL(qnet_fc1_wgrad_f16_kernel, dim3(16, nks, gs), dim3(QN_THREADS), 0, st, nb, h1T + wo, dzT + wo, wpart + wo,
                         sd.ws_stride, 1.0f / dz_scale);
    else if (L.matmul_f16 == 2) {
      static pqn_once_per_device x3_attr;
      if (x3_attr.first()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_fc1_wgrad_x3_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, QY_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&qnet_fc1_wgrad_x3_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, QY_LDS_ACC);
      }
      if (!g_t2_stamps && getenv("PQN_T1_STAMPS")) {
        if (hipMalloc(&g_t2_stamps, 32 * sizeof(unsigned long long)) != hipSuccess) g_t2_stamps = nullptr;
      }
      const unsigned short *dzw = reinterpret_cast<const unsigned short *>(dzT + qw_dzw_offset(nb, C, L.a) + wo);
      if (t2_acc) {   // one workgroup per (row block, seed) walks over every slab: no split-K partials (see the kernel)
        hipLaunchKernelGGL(qnet_fc1_wgrad_x3_kernel<true>, dim3(16, 1, gs), dim3(QN_THREADS), QY_LDS_ACC, st, nb, h1T + wo, dzw, wpart + wo,
                           sd.ws_stride, 1, g_t2_stamps);
      } else {
        // row blocks per workgroup: as many as still leave one workgroup per CU (16 = the dz slab is read once)
        int G = 16;
        while (G > 1 && (16 / G) * nks * gs < 256) G >>= 1;
        hipLaunchKernelGGL(qnet_fc1_wgrad_x3_kernel<false>, dim3(16 / G, nks, gs), dim3(QN_THREADS), QY_LDS, st, nb, h1T + wo, dzw,
                           wpart + wo, sd.ws_stride, G, g_t2_stamps);
      }
    } else
      hipLaunchKernelGGL(qnet_fc1_wgrad_kernel, dim3(16, nks, gs), dim3(QN_THREADS), 0, st, nb, h1T + wo, dzT + wo, wpart + wo,
                         sd.ws_stride);
  }
  if (with_reduce && part != 1)
    launch_fold(L, ntiles, t2_acc ? 1 : nks, rec, gpart, wpart, grad, count, scratch, loss_out, qv_out, inv_b, sd, nullptr, 0, st, defer);
  return pqn_check_launch("pqn_qnet_cnn_grad");
}

extern "C" int64_t pqn_qnet_cnn_workspace_floats(const pqn_cnn_layout_t *L, int32_t nb) {
  if (!L || nb <= 0) return -1;
  const int64_t rec = small_record_floats(L->c, L->a);
  const int64_t ntiles = nb / QN_TILE, nks = (nb + QW_SLAB - 1) / QW_SLAB;
  // (the dz planes of the bf16x3 mode sit behind the split-K slabs: qw_dzw_offset)
  const int64_t std_layout = 1024 + (int64_t)QN_HID * qw_ld(nb) + (int64_t)QN_H1 * qw_h1_cols(nb) + ntiles * rec +
                             nks * (int64_t)QN_H1 * QN_HID + (int64_t)qw_dzw_floats(nb);
  // small minibatches (K-split form of T1): a record per (tile, position group) and a weight-gradient slab per tile.
  // Callers size the workspace once for their LARGEST minibatch and may pass smaller ones, so the result is monotonic:
  // the K-split layout of min(nb, KS_MAX_NB) is covered at every nb.
  const int nbk = min(nb, KS_MAX_NB) / QN_TILE * QN_TILE;
  const int64_t kt = nbk / QN_TILE;
  const int64_t ks_layout = nbk ? 1024 + (int64_t)QN_HID * qw_ld(nbk) + kt * (KS_NG_MAX + 1) * (QN_TILE * QN_HID) + kt * KS_NG_MAX * rec + kt * (int64_t)QN_H1 * QN_HID : 0;
  return max(std_layout, ks_layout);
}

extern "C" int pqn_qnet_cnn_grad(const pqn_cnn_layout_t *L, int32_t nb, const int64_t *idx, const uint32_t *obs_bits,
                                 const int32_t *action, const float *target, const float *theta, const float *w1b,
                                 float *grad, const int32_t *count, float *workspace, float *loss_out, float *qv_out,
                                 void *stream) {
  PQN_REQUIRE(L && idx && obs_bits && action && target && theta && w1b && grad && count && workspace,
              "pqn_qnet_cnn_grad: NULL argument");
  PQN_REQUIRE(nb > 0 && nb % QN_TILE == 0, "pqn_qnet_cnn_grad: minibatch size %d must be a positive multiple of %d", nb,
              QN_TILE);
  return pqn_qnet_cnn_grad_seeds_dyn(*L, nb, idx, obs_bits, action, target, theta, w1b, grad, count, workspace, loss_out, qv_out,
                                 pqn_one_seed(), (hipStream_t)stream);
}

extern "C" int pqn_qnet_cnn_grad_seeds(const pqn_cnn_layout_t *L, int32_t num_seeds, int32_t nb, const int64_t *idx,
                                       int64_t idx_stride, int32_t n_env, int32_t n_env_total, const uint32_t *obs_bits,
                                       const int32_t *action, const float *target, const float *theta,
                                       int64_t theta_stride, const float *w1b, float *grad, const int32_t *count,
                                       float *workspace, int64_t ws_stride, float *loss_out, float *qv_out, void *stream) {
  PQN_REQUIRE(L && idx && obs_bits && action && target && theta && w1b && grad && count && workspace,
              "pqn_qnet_cnn_grad_seeds: NULL argument");
  PQN_REQUIRE(nb > 0 && nb % QN_TILE == 0, "pqn_qnet_cnn_grad_seeds: minibatch size %d must be a positive multiple of %d", nb,
              QN_TILE);
  PQN_REQUIRE(num_seeds >= 1 && num_seeds <= 128 && n_env > 0 && n_env_total >= n_env && idx_stride >= nb &&
                  theta_stride >= L->alloc && ws_stride >= pqn_qnet_cnn_workspace_floats(L, nb),
              "pqn_qnet_cnn_grad_seeds: bad seed batch (seeds=%d n_env=%d/%d strides idx=%lld theta=%lld ws=%lld)", num_seeds,
              n_env, n_env_total, (long long)idx_stride, (long long)theta_stride, (long long)ws_stride);
  pqn_seeds_t sd = pqn_one_seed();
  sd.nseeds = num_seeds;
  sd.n_env = n_env;
  sd.n_env_total = n_env_total;
  sd.idx_stride = idx_stride;
  sd.theta_stride = theta_stride;
  sd.w1b_stride = QN_H1 * QN_HID;
  sd.ws_stride = ws_stride;
  sd.lq_stride = 1;
  sd.idx_mask = 0x7FFFFFFFll;
  return pqn_qnet_cnn_grad_seeds_dyn(*L, nb, idx, obs_bits, action, target, theta, w1b, grad, count, workspace, loss_out, qv_out,
                                     sd, (hipStream_t)stream);
}

// internal (pqn_update.hip): S seeds per launch, buffers = slices of stacked allocations (pqn_seeds_t)
int pqn_qnet_cnn_grad_seeds_dyn(const pqn_cnn_layout_t &L, int nb, const int64_t *idx, const uint32_t *obs_bits,
                            const int32_t *action, const float *target, const float *theta, const float *w1b, float *grad,
                            const int32_t *count, float *workspace, float *loss_out, float *qv_out, const pqn_seeds_t &sd,
                            hipStream_t st, bool with_reduce, int part, int epoch_mb, int epoch_nmb, pqn_fold_args_t *defer) {
  switch (L.c) {
    case 4: return launch_train<4>(L, nb, idx, obs_bits, action, target, theta, w1b, grad, count, workspace, loss_out, qv_out, sd, st, with_reduce, part, epoch_mb, epoch_nmb, defer);
    case 6: return launch_train<6>(L, nb, idx, obs_bits, action, target, theta, w1b, grad, count, workspace, loss_out, qv_out, sd, st, with_reduce, part, epoch_mb, epoch_nmb, defer);
    case 7: return launch_train<7>(L, nb, idx, obs_bits, action, target, theta, w1b, grad, count, workspace, loss_out, qv_out, sd, st, with_reduce, part, epoch_mb, epoch_nmb, defer);
    case 10: return launch_train<10>(L, nb, idx, obs_bits, action, target, theta, w1b, grad, count, workspace, loss_out, qv_out, sd, st, with_reduce, part, -1, 0, defer);
    default: pqn_set_error("pqn_qnet_cnn_grad: unsupported channel count %d", L.c); return PQN_E_UNSUPPORTED;
  }
}

int pqn_cnn_grad_reduce_blocks(int total) { return grad_reduce_blocks(total); }

// ---- the minibatch gather of the position-parallel form, once per EPOCH (round 6) ----
// pos_gather_kernel ran in front of every optimizer step (64 launches of ~9 us per update at the headline shape).  The rows of a whole
// epoch are the shuffled batch cut into num_minibatches slices, so ONE launch over all T*N samples of the epoch produces every
// minibatch's rows / bit-transposes / actions / targets; launch_train then points the forward / backward kernels at slice `mb` of
// the epoch region (same bytes at other addresses: results are bit-identical).  The region sits behind the per-minibatch workspace
// (pqn_qnet_cnn_workspace_floats); pqn_cnn_update_workspace_floats() sizes both.  Callers whose workspace stride is smaller keep
// the per-minibatch gather.
// Does a training launch of this shape take the position-parallel form, and cut how (waves per forward workgroup, sample chunks of the
// backward)?  A pure function of its arguments and the options: launch_train, the epoch gather and the optimizer's plane policy all ask it.
static bool pos_train_plan(const pqn_cnn_layout_t &L, int nb, const pqn_seeds_t &sd, pos_plan_t &plan) {
  const int pos_opt = pqn_opt(PQN_OPT_BWD_POS);
  plan = pos_plan(L.pos_f16x2 != 0, nb, sd.nseeds, sd.pin_form != 0);
  if (!(L.matmul_f16 == 2 && pos_opt != 0 && nb % 64 == 0 && pos_shape_ok(nb) && pqn_cnn_pos_forward_supported(L.c, L.a) && L.c != 10)) return false;
  bool taken = sd.pin_form ? (nb >= 2048 && nb % 256 == 0) : plan.nw != 0;
  if (sd.pin_form) plan.nw = 8;
  if (pos_opt == 2 && !taken) {               // "whenever the shape allows": 8 waves if the minibatch has whole 256-sample blocks, else the finest cut
    plan.nw = nb % 256 == 0 ? 8 : (L.pos_f16x2 ? (nb % 128 == 0 ? 4 : 2) : 0);
    taken = plan.nw != 0;
  }
  if (taken && L.pos_f16x2 && !sd.pin_form) {  // test / measurement overrides
    const int ow = pqn_opt(PQN_OPT_POS_WAVES), oc = pqn_opt(PQN_OPT_POS_CHUNKS);
    if ((ow == 8 || ow == 4 || ow == 2) && nb % (32 * ow) == 0) plan.nw = ow;
    if (oc >= 1 && oc <= POS_MAX_CHUNKS && (oc & (oc - 1)) == 0 && nb % (64 * oc) == 0 && oc <= (nb + 255) / 256) plan.nch = oc;
  }
  return taken;
}
static bool pos_form_taken(const pqn_cnn_layout_t &L, int nb, const pqn_seeds_t &sd) {
  pos_plan_t plan;
  return pos_train_plan(L, nb, sd, plan);
}
bool pqn_qnet_cnn_pos_form_taken(const pqn_cnn_layout_t &L, int nb, const pqn_seeds_t &sd) { return pos_form_taken(L, nb, sd); }
long long pqn_qnet_cnn_epoch_floats(const pqn_cnn_layout_t &L, int nb, int nmb) {
  if (L.matmul_f16 != 2 || nb % 64 != 0 || !pos_shape_ok(nb) || !pqn_cnn_pos_forward_supported(L.c, L.a) || L.c == 10) return 0;
  return pos_epoch_layout(nb, nmb, L.c).end;
}
// true when the update of this shape gathers per epoch: the launch takes the position-parallel form and the caller's workspace stride
// has room for the epoch region (a pure function of its arguments and the options: pqn_cnn_update_phase callers decide the same
// way in their SHUFFLE and GRAD phases)
bool pqn_qnet_cnn_epoch_applies(const pqn_cnn_layout_t &L, int nb, int nmb, const pqn_seeds_t &sd) {
  if (!pos_form_taken(L, nb, sd) || (long long)nmb * nb > (1ll << 25)) return false;
  return sd.ws_stride >= pqn_qnet_cnn_workspace_floats(&L, nb) + pqn_qnet_cnn_epoch_floats(L, nb, nmb);
}
int pqn_qnet_cnn_epoch_gather(const pqn_cnn_layout_t &L, int nb, int nmb, const int64_t *idx_epoch, const uint32_t *obs_bits,
                              const int32_t *action, const float *target, float *workspace, const pqn_seeds_t &sd, hipStream_t st) {
  PQN_REQUIRE(pqn_qnet_cnn_epoch_applies(L, nb, nmb, sd), "pqn_qnet_cnn_epoch_gather: not applicable to this launch");
  float *h1T = workspace + 1024 + (size_t)QN_HID * qw_ld(nb);
  const pos_epoch_t E = pos_epoch_layout(nb, nmb, L.c);
  const long long e0 = pqn_qnet_cnn_workspace_floats(&L, nb) - (long long)(h1T - workspace);
  pos_ws_t W = pos_ws_layout(nb, L.c, L.a);
  W.mb_bits = e0 + E.bits; W.t32 = e0 + E.t32; W.act = e0 + E.act; W.tgt = e0 + E.tgt;
  const bool timed = pqn_prof_begin(5, st);   // kernel timer mode 5: the epoch gather
  const int rc = pqn_cnn_pos_gather(L, nmb * nb, idx_epoch, obs_bits, action, target, h1T, W, sd, sd.nseeds, st);
  if (timed) pqn_prof_end(st);
  return rc;
}
extern "C" int64_t pqn_cnn_update_workspace_floats(const pqn_cnn_layout_t *L, int32_t num_envs, int32_t num_steps, int32_t num_minibatches) {
  if (!L || num_envs <= 0 || num_steps <= 0 || num_minibatches <= 0 || ((int64_t)num_envs * num_steps) % num_minibatches) return -1;
  const int nb = (int)((int64_t)num_envs * num_steps / num_minibatches);
  const int64_t base = pqn_qnet_cnn_workspace_floats(L, nb);
  return base < 0 ? base : base + pqn_qnet_cnn_epoch_floats(*L, nb, num_minibatches);
}

extern "C" int pqn_qnet_cnn_apply(const pqn_cnn_layout_t *L, float *theta, float *w1b, const float *grad, float *m,
                                  float *v, int32_t *count, float lr_init, float lr_end, double lr_steps,
                                  float max_norm, float *workspace, float *gnorm_out, int32_t recompute_norm,
                                  void *stream) {
  PQN_REQUIRE(L && theta && w1b && grad && m && v && count && workspace, "pqn_qnet_cnn_apply: NULL argument");
  return pqn_launch_radam(theta, grad, m, v, L->total, count, lr_init, lr_end, lr_steps, max_norm, workspace, gnorm_out,
                          L->off_w1, w1b, recompute_norm, grad_reduce_blocks(L->total), (hipStream_t)stream, 1, 0, 0, 0,
                          L->matmul_f16 != 0 ? L->off_w1h : 0, L->matmul_f16 + L->pos_f16x2);
}

// w1b: f32 dgrad-fragment copy (nullable);  w1h: fp16 forward-fragment copy + fp16 dgrad-fragment copy (nullable);
// x3: the six bf16 planes of the bf16x3 mode (nullable)
__global__ void pack_w1b_kernel(const float *__restrict__ w1p, float *__restrict__ w1b, _Float16 *__restrict__ w1h,
                                unsigned short *__restrict__ x3, _Float16 *__restrict__ h2) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= QN_H1 * QN_HID) return;
  const int frag = j >> 8, ln = (j >> 2) & 63, sx = j & 3;
  const int gi = frag >> 3, cb = frag & 7, kk = ln >> 4, jj = ln & 15;
  const int jb = (((cb * 64 + gi) * 64 + (jj >> 2) * 16 + 4 * kk + sx) << 2) + (jj & 3);
  const float w = w1p[j];
  if (w1b) w1b[jb] = w;
  if (w1h) {
    w1h[j] = (_Float16)w;
    w1h[QN_H1 * QN_HID + jb] = (_Float16)w;
  }
  if (x3) pqn_x3_store_planes(x3, 16 * gi + 4 * kk + sx, 16 * cb + jj, w);
  if (h2) pqn_h2_store_planes(h2, 16 * gi + 4 * kk + sx, 16 * cb + jj, w);
}

// theta is written only in its tail (the fp16 copies of a matmul_f16 layout); pass w1b = NULL to refresh just those
extern "C" int pqn_qnet_cnn_pack_w1b(const pqn_cnn_layout_t *L, float *theta, float *w1b, void *stream) {
  PQN_REQUIRE(L && theta, "pqn_qnet_cnn_pack_w1b: NULL argument");
  PQN_REQUIRE(w1b || L->matmul_f16 != 0, "pqn_qnet_cnn_pack_w1b: nothing to do");
  hipLaunchKernelGGL(pack_w1b_kernel, dim3(QN_H1 * QN_HID / 256), dim3(256), 0, (hipStream_t)stream, theta + L->off_w1, w1b,
                     L->matmul_f16 == 1 ? reinterpret_cast<_Float16 *>(theta + L->off_w1h) : (_Float16 *)nullptr,
                     L->matmul_f16 == 2 ? reinterpret_cast<unsigned short *>(theta + L->off_w1h) : (unsigned short *)nullptr,
                     L->pos_f16x2 ? reinterpret_cast<_Float16 *>(theta + L->off_w1h + H2_PLANES_OFF) : (_Float16 *)nullptr);
  return pqn_check_launch("pqn_qnet_cnn_pack_w1b");
}
