// pqn_qnet_pos.hip -- the position-parallel form of the MinAtar CNN training step (bf16x3 operand mode, gfx950).
//
// One optimizer step of _learn_phase (purejaxql/pqn_minatar.py:266-297) on a seed's minibatch of nb samples.  Everything
// below fc1 is local to a conv position: h1[m][16 pos + ch] depends only on sample m's 3x3 window at pos, LayerNorm_0
// (pqn_minatar.py:31-36, flax LayerNorm over the 16 channels) is per point, and the dgrad columns / dW1 rows of a position
// need only that position's 16 rows of the fc1 kernel W1.  The backward therefore gives every WAVE one conv position and
// keeps that position's slice of W1 (48 VGPRs of bf16 planes) and of dW1 (32 accumulator VGPRs) resident while the
// workgroup walks over the samples: no weight-plane stream, no h1 hand-over, no separate fc1 weight-gradient GEMM.
//
//   pos_gather_kernel    (super-tile of 32 samples, seed): the minibatch's packed observation rows in minibatch order +
//                        their bit-transpose T32[bit] = one word over the 32 samples (the conv weight gradient's operand)
//   cnn_pos_fwd_kernel   (256 samples, seed), wave = 32 samples for all 64 positions: conv on the fly (transposed, so that its
//                        outputs are fc1's A fragments), z[32][128] in registers, the W1 planes of a K step through ONE
//                        LDS-DMA per workgroup; head, loss and the head's backward on the accumulator layout; writes dz
//                        as bf16 planes, LayerNorm_0's (mean, 1/std) and one head record per workgroup
//   cnn_pos_bwd_kernel   (8 positions, chunk of samples, seed), wave = position: per super-tile the conv is recomputed in
//                        MFMA accumulator layout (lane = channel, 4 samples per lane and tile) and normalised with the
//                        forward's statistics, then dgrad against the resident planes, relu mask, LayerNorm_0 backward,
//                        and -- with h1 / dx split ONCE, in registers -- the position's rows of dW1 and its conv
//                        weight-gradient tile.  dz (bf16 planes, ONE image for both operand orders: ds_read_b128 /
//                        ds_read_b64_tr_b16), the packed rows, T32 and the statistics arrive per super-tile through
//                        LDS-DMA (global_load_lds) into a four-slot ring: one barrier per two super-tiles, no staging
//                        registers, no ds_write.
//   cnn_pos_rollout_kernel  the persistent rollout in the forward kernel's structure (256 envs per workgroup)
//
#include <stdlib.h>

#include "pqn_env_rules.h"
#include "pqn_qnet_x3.h"
#include "pqn_qnet_pos.h"

#define POS_THREADS 512
// Barrier period of the backward: ONE barrier per TWO super-tiles over a four-slot ring, the favoured half of the waves
// (s_setprio 1) alternating between the two super-tiles of a period -- the older wave of a SIMD then leads in one and trails in
// the other, and both reach the barrier together instead of one of them idling a quarter of every iteration (stamps: 2.3k of
// 8.5k cycles).  In-call A/B (profiles/r05_v2_pair_sync_ab.txt): backward 264 -> 248 us; the same period WITHOUT the
// priority alternation gains nothing, and so does the alternation with a barrier per super-tile.  -DPOS_PAIR_SYNC=0 = one
// barrier per super-tile, two slots.
#ifndef POS_PAIR_SYNC
#define POS_PAIR_SYNC 1
#endif
#define POS_ST 32              // samples per super-tile (two 16-sample MFMA tiles)
// Round 6: the elementwise arithmetic of the backward's two long vector stretches (LayerNorm_0 normalise, relu mask / LayerNorm_0
// backward) written on EXPLICIT pairs (f32x2 -> v_pk_add / mul / fma_f32): inside a long VALU-only stretch a packed instruction costs
// about one issue slot for two values, next to MFMAs it costs ~10 cycles per cluster (profiles/r06_ubench_issue_kinds.txt) -- so the
// file is built without the compiler's SLP packing and packs by hand exactly where it pays.  Same operations per value, same order:
// bit-identical results.  For the pairs the forward keeps LayerNorm_0's statistics as [position][mean | 1/std][32 samples].
#ifndef POS_BWD_PK
#define POS_BWD_PK 1
#endif
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

// LDS image of a super-tile's dz planes: [plane][32 samples][16 slots of 16 B], slot = quad ^ pos_sigma(sample & 15), where
// quad = 4 sK + kq names the 8 outputs 32 sK + 16 (j >> 2) + 4 kq + (j & 3) (dz_planes_a).  ONE image serves both products
// of the backward: the dgrad's A fragment (row = sample, 8 K slots = one quad) is a ds_read_b128, the weight gradient's
// B fragment (column = output, K slots = samples) is two ds_read_b64_tr_b16 (the transposing LDS read of gfx950: lane i of a
// 16-lane group fetches 8 B of row i >> 2, the result in lane n, element j is row j / column n of the 4 x 16 block).
// pos_sigma is the GF(2)-linear map (found by search over all 20,160) under which the ds_read_b128 pattern is conflict-free in
// the instruction's real lane groups AND the transposing reads hit every 16-B slot at most twice per 32-lane pass (the
// minimum: they use one 8-B half of every slot).
PQN_HD int pos_sigma(int v) { return v ^ ((v & 1) * 12) ^ ((v & 2) << 2); }
typedef short pos_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) pos_s16x4 pos_lds_s16x4;
PQN_D u32x2_t pos_tr_read(uint32_t lds_byte_addr) {   // ds_read_b64_tr_b16 (builtin: the compiler counts it in lgkmcnt)
  return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((pos_lds_s16x4 *)(uintptr_t)lds_byte_addr));
}

// 8 observation bits -> the 8 bf16 slots of an MFMA operand (2.0 = 0x4000 per set bit, ConvX3::expand8) as ONE 16-B LDS
// read from a 256-entry table instead of nine VALU instructions: both kernels are bound by instruction issue, and the
// operand build was 12-18 % of their vector instructions.  The table is filled once per workgroup (4 KB).
#ifndef POS_LUT
#define POS_LUT 1
#endif
PQN_D void pos_lut_fill(u32x4 *lut, int tid, int nthreads) {
  for (int b = tid; b < 256; b += nthreads) {
    const uint32_t y = ((uint32_t)b << 15) | (uint32_t)b;
    lut[b] = u32x4{(y << 14) & 0x40004000u, (y << 12) & 0x40004000u, (y << 10) & 0x40004000u, (y << 8) & 0x40004000u};
  }
}
template <int C>
PQN_D u32x4 pos_expand8(const u32x4 *lut, uint32_t byte) {
  if constexpr (POS_LUT != 0) return lut[byte];
  else return ConvX3<C>::expand8(byte);
}

template <int C, int NPL = 3>      // NPL: planes per split operand -- 3 = bf16x3, 2 = f16x2 (pqn_qnet_x3.h)
struct PosCfg {
  using Cfg = CnnCfg<C>;
  static constexpr int OW = Cfg::OW;                 // packed observation words per sample (multiple of 4)
  static constexpr int ROWCH = OW / 4;               // 16-B chunks per packed row
  static constexpr int ROWSTRIDE = ROWCH + 1;        // LDS row stride in chunks (one pad chunk: spreads the rows over the banks)
  static constexpr int NBITS = 100 * C;              // observation bits per sample
  static constexpr int TW = pos_t32_words(C);        // words of T32 per super-tile (zero word at NBITS, padded to 1 KB)
  static constexpr int KW = 9 * C, RB = 3 * C, NRB = (KW + 15) / 16;
  static constexpr int CONVBLK = KW * 16 + 48;
  // ring slot, in 16-B chunks: dz planes | rows | T32 | LN0 statistics
  static constexpr int N_DZ = NPL * 512;             // planes x 32 samples x 16 quads
  static constexpr int N_ROWS = (POS_ST * ROWSTRIDE + 63) / 64 * 64;
  static constexpr int N_T32 = TW / 4;
  static constexpr int N_STAT = 128;                 // 8 positions x 32 samples x {mean, rstd}
  static constexpr int O_ROWS = N_DZ, O_T32 = O_ROWS + N_ROWS, O_STAT = O_T32 + N_T32;
  static constexpr int SLOT = O_STAT + N_STAT;       // chunks per slot (a multiple of 64: whole DMA instructions)
  static constexpr int NI = SLOT / 64;               // 1-KB DMA instructions per slot
  static constexpr int NTAIL = NI - 8 * NPL;         // instructions behind the 8 per plane of dz
  // f16x2: the per-sample factors 2^-k / 128 of the dz rows (forward kernel) ride in the last 32 words of the super-tile's T32 block
  static constexpr int F_OFF = TW - 32;
  static_assert(NPL == 3 || TW - (NBITS + 1) >= 32, "no room for the dz row factors behind the bit words");
  static_assert(NTAIL >= 1 && NTAIL <= 16, "at most two tail instructions per wave");
  static constexpr int NCS = (KW + 31) / 32;          // K steps of the conv product
  static constexpr int RS = POS_PAIR_SYNC ? 4 : 2;    // ring slots
  static constexpr size_t lds_bytes() { return (size_t)RS * SLOT * 16 + sizeof(float) * ((CONVBLK + 3) & ~3) + (size_t)NCS * NPL * 64 * 16 + 4096; }
};

// sorted shuffle key -> row of the stacked [T][S * N] rollout record (see pqn_seeds_t)
PQN_D int64_t pos_row_of(int64_t key, const pqn_seeds_t &sd, int seed) {
  const uint32_t j = (uint32_t)(key & sd.idx_mask);
  if (sd.n_env_total == sd.n_env) return (int64_t)j;
  const uint32_t t = j / (uint32_t)sd.n_env;
  return (int64_t)t * sd.n_env_total + (int64_t)seed * sd.n_env + (int64_t)(j - t * (uint32_t)sd.n_env);
}

// ---------------------------------------------------------------------------
// minibatch gather: rows in minibatch order + the bit-transpose per super-tile
// ---------------------------------------------------------------------------
// G super-tiles per workgroup (round 6: 4 where the launch is large): the workgroup's G x 32 key loads, then its row loads, are all
// requested before any is consumed -- a super-tile alone is a chain of two dependent memory round trips per 64 + 16 B it moves, and the
// epoch launch (65,536 super-tiles of 16 seeds) ran 32 such chains back to back per CU slot (180 -> 143 us; 8 per workgroup: 145; profiles/r06_v10_gather_group.txt).
// Same bytes to the same addresses.
template <int C, int G>
__global__ __launch_bounds__(256) void pos_gather_kernel(int nb, const int64_t *__restrict__ idx, const uint32_t *__restrict__ obs_bits,
                                                         const int32_t *__restrict__ action, const float *__restrict__ target,
                                                         float *__restrict__ wsx, pos_ws_t W, pqn_seeds_t sd) {
  using P = PosCfg<C>;
  constexpr int NW = G * POS_ST * P::OW, NIT = (NW + 255) / 256;
  __shared__ uint32_t rows[NW];
  const int seed = blockIdx.y + sd.seed_base;
  idx += seed * sd.idx_stride;
  wsx += seed * sd.ws_stride;
  uint32_t *mb_bits = reinterpret_cast<uint32_t *>(wsx + W.mb_bits);
  uint32_t *t32 = reinterpret_cast<uint32_t *>(wsx + W.t32);
  const int st0 = blockIdx.x * G, tid = threadIdx.x;
  int64_t key[NIT];
#pragma unroll
  for (int q = 0; q < NIT; ++q) {
    const int i = min(tid + 256 * q, NW - 1);
    key[q] = idx[st0 * POS_ST + i / P::OW];
  }
  int64_t akey = 0;
  if (tid < G * POS_ST) akey = idx[st0 * POS_ST + tid];
  uint32_t v[NIT];
#pragma unroll
  for (int q = 0; q < NIT; ++q) {
    const int i = min(tid + 256 * q, NW - 1);
    v[q] = obs_bits[(size_t)pos_row_of(key[q], sd, seed) * P::OW + (i % P::OW)];
  }
  int32_t a = 0;
  float tg = 0.0f;
  if (tid < G * POS_ST) {
    const int64_t src = pos_row_of(akey, sd, seed);
    a = action[src];
    tg = target[src];
  }
#pragma unroll
  for (int q = 0; q < NIT; ++q) {
    const int i = tid + 256 * q;
    if (i < NW) {
      rows[i] = v[q];
      mb_bits[(size_t)st0 * POS_ST * P::OW + i] = v[q];
    }
  }
  if (tid < G * POS_ST) {
    reinterpret_cast<int32_t *>(wsx + W.act)[st0 * POS_ST + tid] = a;
    (wsx + W.tgt)[st0 * POS_ST + tid] = tg;
  }
  __syncthreads();
  for (int b = tid; b < G * P::TW; b += 256) {
    const int g = b / P::TW, bb = b - g * P::TW;
    uint32_t word = 0u;
    if (bb < P::NBITS) {
#pragma unroll
      for (int s = 0; s < POS_ST; ++s) word |= ((rows[(g * POS_ST + s) * P::OW + (bb >> 5)] >> (bb & 31)) & 1u) << s;
    }
    t32[(size_t)(st0 + g) * P::TW + bb] = word;
  }
}

// ---------------------------------------------------------------------------
// LDS-DMA: one wave-instruction (global_load_lds_dwordx4) moves 64 x 16 B from per-lane global addresses (uniform 64-bit
// base + per-lane 32-bit byte offset) to LDS at a wave-uniform byte address + 16 lane.  Issued as inline asm: through the
// builtin, hipcc orders every later LDS read behind `s_waitcnt vmcnt(0)` (it cannot tell which LDS bytes the DMA writes),
// which serialises the ring -- the transfer of super-tile j + 1 would have to land before super-tile j is even read.
// The asm form is invisible to the compiler's wait-count bookkeeping, so the kernel drains it itself (pos_dma_wait) in
// front of the barrier that publishes the slot.  M0 (the LDS destination) is saved and restored inside the statement.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) char lds_char_t;
PQN_D uint32_t pos_lds_addr(const void *p) { return (uint32_t)(uintptr_t)(lds_char_t *)p; }
PQN_D void pos_dma16(uint32_t voff, const void *sbase, uint32_t lds_dst) {
  // round 6: M0 is left holding the LDS destination.  Nothing else in these kernels reads M0 (gfx950 DS instructions do not; there is no
  // s_movrel / sendmsg / GWS / LDS-DMA builtin in this file: tests/test_host_cpu.py scans the assembly), and the save / restore pair
  // was two of the four scalar instructions of every transfer
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
PQN_D void pos_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// wave max over all 64 lanes (prologue / epilogue use only)
PQN_D float pos_wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// f16x2: the conv kernel's scale (a power of two from max |w|; every wave derives the same value from the LDS copy) and
// LayerNorm_0's output scale from 4 max|scale| + max|bias| (|xhat| <= sqrt(15))
template <int C>
PQN_D float pos_h2_conv_scale(const float *s_wc, int lane) {
  float m = 0.0f;
  for (int i = lane; i < CnnCfg<C>::KW * 16; i += 64) m = fmaxf(m, fabsf(s_wc[i]));
  return h2_pow2_below(pos_wave_max(m));
}
template <int C>
PQN_D float pos_h2_h1_scale(const float *s_wc, int lane) {
  const float g = fabsf(s_wc[CnnCfg<C>::KW * 16 + 16 + (lane & 15)]), b = fabsf(s_wc[CnnCfg<C>::KW * 16 + 32 + (lane & 15)]);
  return h2_pow2_below(4.0f * pos_wave_max(g) + pos_wave_max(b));
}
// the conv kernel as B-fragment planes [K step][plane][lane] in LDS, written by ONE wave (bf16x3: h, m, l; f16x2: h, l of scale * w)
template <int C, int NPL>
PQN_D void pos_conv_planes(const float *s_wc, u32x4 *s_cvw, int lane, float scale) {
  constexpr int NK = 9 * C, NS = (NK + 31) / 32;
  const int kq = lane >> 4, o = lane & 15;
#pragma unroll
  for (int sx = 0; sx < NS; ++sx) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * sx + 8 * kq + j;
      v[j] = (k < NK) ? s_wc[k * 16 + o] : 0.0f;
    }
    if constexpr (NPL == 2) {
      const H2Frag f = h2_split8(f32x4{v[0], v[1], v[2], v[3]} * scale, f32x4{v[4], v[5], v[6], v[7]} * scale);
      s_cvw[(sx * 2 + 0) * 64 + lane] = f.h;
      s_cvw[(sx * 2 + 1) * 64 + lane] = f.l;
    } else {
      const X3Frag f = x3_split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
      s_cvw[(sx * 3 + 0) * 64 + lane] = f.h;
      s_cvw[(sx * 3 + 1) * 64 + lane] = f.m;
      s_cvw[(sx * 3 + 2) * 64 + lane] = f.l;
    }
  }
}

#ifdef POS_STAMPS   // variant builds only: the stamp stores are VMEM operations the compiler counts, and its vmcnt waits would drain the DMA
#define POSB_STAMP(k) do { if (stamps && j == 4 && lane == 0 && blockIdx.x == 0 && wave == 0) stamps[(k)] = __builtin_readcyclecounter(); } while (0)
#else
#define POSB_STAMP(k) do { } while (0)
#endif

// LayerNorm_0 statistics (mean, 1/std per (sample, position)) are the forward kernel's record: xhat and the relu mask are then
// the forward's own, bit for bit
template <int C, int NPL>
__global__ __launch_bounds__(POS_THREADS) void cnn_pos_bwd_kernel(int nb, int nch, const float *__restrict__ theta, pqn_cnn_layout_t L,
                                                                  float *__restrict__ wsx, float *__restrict__ w1out,
                                                                  pos_ws_t W, pqn_seeds_t sd, unsigned long long *__restrict__ stamps) {
  using P = PosCfg<C, NPL>;
  using Cfg = CnnCfg<C>;
  using M = PosMM<NPL>;
  constexpr bool H2 = NPL == 2;
  static_assert(!H2 || POS_BWD_PK, "the f16x2 backward is written on the paired form");
  constexpr int NRB = P::NRB, RB = P::RB, CONVBLK = P::CONVBLK;
  extern __shared__ __attribute__((aligned(16))) char pos_smem[];
  u32x4 *ring = reinterpret_cast<u32x4 *>(pos_smem);
  float *s_wc = reinterpret_cast<float *>(ring + P::RS * P::SLOT);
  u32x4 *s_cvw = reinterpret_cast<u32x4 *>(s_wc + ((CONVBLK + 3) & ~3));   // conv kernel as bf16-plane B fragments [K step][plane][lane]: one copy for all waves
  u32x4 *s_lut = s_cvw + P::NCS * NPL * 64;                                 // pos_expand8 table
  // (seed, position group, chunk) of this workgroup.  Workgroups go to the 8 XCDs round-robin by linear id; the 8 nch
  // workgroups of a seed all read that seed's dz planes, so they are placed on ONE XCD's L2 when the seeds divide over the XCDs.
  const int wps = 8 * nch, nsl = gridDim.x / wps;
  int seed_l, within;
  {
    const int lin = blockIdx.x;
    if ((nsl & 7) == 0) {
      const int xcd = lin & 7, k = lin >> 3;
      seed_l = xcd * (nsl >> 3) + k / wps;
      within = k % wps;
    } else {
      seed_l = lin / wps;
      within = lin % wps;
    }
  }
  const int pg = within & 7, chunk = within >> 3;
  const int seed = seed_l + sd.seed_base;
  theta += seed * sd.theta_stride;
  wsx += seed * sd.ws_stride;
  w1out += seed * sd.ws_stride + (size_t)chunk * QN_H1 * QN_HID;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ch = lane & 15, kq = lane >> 4;
  const int nst = nb / (POS_ST * nch), st0 = chunk * nst;   // super-tiles of this chunk, first super-tile
  const int p = 8 * pg + wave, py = p >> 3, px = p & 7;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- DMA plan of this wave: instruction i = wave + 8 k fills chunks [64 i, 64 i + 64) of the slot ----
  const u32x4 *g_dza = reinterpret_cast<const u32x4 *>(wsx + W.dz);                      // [NPL][nb][16 quads]
  const u32x4 *g_rows = reinterpret_cast<const u32x4 *>(wsx + W.mb_bits);                // [nb][ROWCH]
  const u32x4 *g_t32 = reinterpret_cast<const u32x4 *>(wsx + W.t32);                     // [super-tile][N_T32]
  const u32x4 *g_stat = reinterpret_cast<const u32x4 *>(wsx + W.stats);                  // [super-tile][64 pos][16]
  const size_t pa = (size_t)nb * 16;                                                     // dzA plane stride in chunks
  // per-lane byte offsets; plane / instruction index and the super-tile advance the UNIFORM base.  dzA: instruction
  // wave + 8 k covers plane k, samples 4 wave .. 4 wave + 3: chunk (sample, slot) holds quad = slot ^ pos_sigma(sample & 15)
  uint32_t offT[2];
  int strT[2], tailq[2];         // wave-uniform: chunks per super-tile of a tail instruction's source, its slot chunk (-1: none)
  const uint32_t offA = (uint32_t)(((4 * wave + (lane >> 4)) * 16 + ((lane & 15) ^ pos_sigma((4 * wave + (lane >> 4)) & 15))) * 16);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int ti = wave + 8 * k;                          // tail instruction index (wave-uniform)
    const int q0 = (8 * NPL + ti) * 64, q = q0 + lane;    // the instruction's 64 chunks are all of one kind
    tailq[k] = ti < P::NTAIL ? q0 : -1;
    uint32_t off = 0u;
    int str = 0;
    if (ti < P::NTAIL) {
      if (q0 < P::O_T32) {                                // packed rows, padded row stride
        const int r = q - P::O_ROWS, row = r / P::ROWSTRIDE, cc = r - row * P::ROWSTRIDE;
        const bool real = row < POS_ST && cc < P::ROWCH;
        off = (uint32_t)((const char *)(g_rows + (real ? row * P::ROWCH + cc : 0)) - (const char *)g_dza);
        str = POS_ST * P::ROWCH;
      } else if (q0 < P::O_STAT) {
        off = (uint32_t)((const char *)(g_t32 + (q - P::O_T32)) - (const char *)g_dza);
        str = P::N_T32;
      } else {
        off = (uint32_t)((const char *)(g_stat + (size_t)pg * 128 + (q - P::O_STAT)) - (const char *)g_dza);
        str = 1024;
      }
    }
    offT[k] = off;
    strT[k] = str;
  }
  const uint32_t ring_lds = pos_lds_addr(ring);
  auto dma_slot = [&](int j) {   // super-tile st0 + min(j, nst - 1) -> slot j % RS
    const int g = st0 + min(j, nst - 1);
    const uint32_t slot = ring_lds + (uint32_t)((j & (P::RS - 1)) * P::SLOT * 16);
    const u32x4 *bA = g_dza + (size_t)g * (POS_ST * 16);
#pragma unroll
    for (int k = 0; k < 2; ++k)
      if (tailq[k] >= 0) pos_dma16(offT[k], g_dza + (size_t)g * strT[k], slot + (uint32_t)(tailq[k] * 16));
#pragma unroll
    for (int k = 0; k < NPL; ++k) pos_dma16(offA, bA + (size_t)k * pa, slot + (uint32_t)((wave + 8 * k) * 1024));
  };
  dma_slot(0);
  if constexpr (POS_PAIR_SYNC != 0) dma_slot(1);

  // ---- per-wave constants ----
  for (int i = tid; i < CONVBLK; i += POS_THREADS) s_wc[i] = theta[L.off_wc + i];
  pos_lut_fill(s_lut, tid, POS_THREADS);
  u32x4 wfr[4][NPL];             // the position's 16 rows of W1 as dgrad-order planes: B fragments of dh1 = dz W1p^T
  {
    const u32x4 *wd = H2 ? reinterpret_cast<const u32x4 *>(theta + L.off_w1h + H2_PLANES_OFF) + 2 * (X3_PLANE / 8)
                         : reinterpret_cast<const u32x4 *>(theta + L.off_w1h) + 3 * (X3_PLANE / 8);
#pragma unroll
    for (int sK = 0; sK < 4; ++sK)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) wfr[sK][pl] = wd[(size_t)pl * (X3_PLANE / 8) + ((p * 4 + sK) * 64 + lane)];
  }
  // f16x2: the largest dz row factor of this chunk (the forward left one per sample behind the bit words of T32): the weight
  // gradient's h1 operand carries factor / largest <= 1 per sample, the epilogue puts the largest back
  float *scr = reinterpret_cast<float *>(ring + (P::RS - 1) * P::SLOT);   // the last ring slot is idle until the loop
  if constexpr (H2) {
    const float *tf = wsx + W.t32 + (size_t)st0 * P::TW + P::F_OFF;
    float fm = 0.0f;
    for (int i = tid; i < nst * POS_ST; i += POS_THREADS) fm = fmaxf(fm, tf[(size_t)(i >> 5) * P::TW + (i & 31)]);
    fm = pos_wave_max(fm);
    if (lane == 0) scr[wave] = fm;
  }
  int bw[3], bs[3];              // window row ky of this position: word and bit offset inside a packed row (wave-uniform)
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int bp = ((py + ky) * 10 + px) * C;
    bw[ky] = bp >> 5;
    bs[ky] = bp & 31;
  }
  const int sig_ch = pos_sigma(ch);
  // transposing reads: lane (i = ch, g = kq) fetches 8 B of sample row 4 g + (i >> 2), quad 4 sK + (i & 3); per sK the swizzled
  // slot differs by more than a constant, so four lane offsets; plane (+8192 B), tile (+4096 B), half (+8 B) are immediates
  uint32_t tr_sk[4];
  {
    const int row = 4 * kq + (ch >> 2), sg = pos_sigma(row);
#pragma unroll
    for (int sK = 0; sK < 4; ++sK) tr_sk[sK] = (uint32_t)((row * 16 + ((4 * sK + (ch & 3)) ^ sg)) * 16);
  }
  int tb[NRB];                   // conv weight gradient: T32 word of this lane's row k = 16 rb + ch (padding rows: the zero word)
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) {
    const int k = 16 * rb + ch;
    tb[rb] = k < P::KW ? ((py + k / RB) * 10 + px) * C + (k % RB) : P::NBITS;
  }
  pos_dma_wait();
  __syncthreads();               // s_wc complete, slot 0 landed
  float oscale = ConvX3<C>::OUT_SCALE, kh1 = 1.0f;      // conv output scale; f16x2: the h1 operand's scale (LayerNorm_0 bound / largest row factor)
  if constexpr (H2) {
    const float sc = pos_h2_conv_scale<C>(s_wc, lane);
    oscale = ConvX3<C>::OUT_SCALE / sc;
    float fm = scr[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) fm = fmaxf(fm, scr[w]);
    kh1 = pos_h2_h1_scale<C>(s_wc, lane) / fmaxf(fm, 1e-30f);
    if (wave == 0) pos_conv_planes<C, NPL>(s_wc, s_cvw, lane, sc);
  } else {
    if (wave == 0) pos_conv_planes<C, NPL>(s_wc, s_cvw, lane, 1.0f);   // the same for every position: split once, shared through LDS
  }
  __syncthreads();
  const float bias = s_wc[Cfg::KW * 16 + ch], g0 = s_wc[Cfg::KW * 16 + 16 + ch], be0 = s_wc[Cfg::KW * 16 + 32 + ch];
  f32x4 dw[8], cw[NRB];
#pragma unroll
  for (int c = 0; c < 8; ++c) dw[c] = zero4;
#pragma unroll
  for (int rb = 0; rb < NRB; ++rb) cw[rb] = zero4;
  float gbi = 0.f, gsc = 0.f, gbc = 0.f;
  // the plane fragments must have ARRIVED before the loop: a compiler-counted load still pending at the loop's first use
  // would put `s_waitcnt vmcnt(N)` inside the loop, and with the (uncounted) DMA in flight that wait drains the DMA
#pragma unroll
  for (int sK = 0; sK < 4; ++sK)
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) asm volatile("" : "+v"(wfr[sK][pl]));

#pragma unroll 1
  for (int j = 0; j < nst; ++j) {
    if constexpr (POS_PAIR_SYNC != 0) {
      if ((j & 1) == 0) { dma_slot(j + 2); dma_slot(j + 3); }   // the slots of the previous period: every wave passed its barrier
      if ((wave >= 4) == ((j & 1) != 0)) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    } else {
      dma_slot(j + 1);           // the other slot was last read an iteration ago, before that iteration's barrier
    }
    POSB_STAMP(0);
    const u32x4 *slot = ring + (j & (P::RS - 1)) * P::SLOT;
    const u32x4 *ldA = slot;
    const uint32_t trb = ring_lds + (uint32_t)((j & (P::RS - 1)) * P::SLOT * 16);   // transposing reads of the same planes
    const uint32_t *rowsL = reinterpret_cast<const uint32_t *>(slot + P::O_ROWS);
    const uint32_t *t32L = reinterpret_cast<const uint32_t *>(slot + P::O_T32);
    // ---- window masks of (sample lane & 15 of each tile, this position) ----
    uint32_t mk[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint32_t *row = rowsL + (16 * t + ch) * (P::ROWSTRIDE * 4);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const uint32_t lo = row[bw[ky]], hi = row[bw[ky] + 1];
        mk[t][ky] = (uint32_t)(((((uint64_t)hi) << 32) | lo) >> bs[ky]) & ((1u << RB) - 1u);
      }
    }
    // ---- conv: rows = samples, columns = channels; the two tiles interleaved ----
    f32x4 cb_[2], cs_[2];       // (round 6) the first product of every chain runs on C = 0: no zeroed register tuples
#pragma unroll
    for (int sx = 0; sx < ConvX3<C>::NS; ++sx) {
      u32x4 fa[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) fa[t] = pos_expand8<C>(s_lut, __builtin_amdgcn_ubfe(ConvX3<C>::word(mk[t], sx), 8u * kq, 8u));
      const u32x4 wh = s_cvw[(sx * NPL + 0) * 64 + lane], wl = s_cvw[(sx * NPL + NPL - 1) * 64 + lane];
      if (sx == 0) {
        M::grp2_zero(cs_[0], fa[0], wl, cs_[1], fa[1], wl);
        M::grp2_zero(cb_[0], fa[0], wh, cb_[1], fa[1], wh);
      } else {
        M::grp2(cs_[0], fa[0], wl, cs_[1], fa[1], wl);
        M::grp2(cb_[0], fa[0], wh, cb_[1], fa[1], wh);
      }
      if constexpr (!H2) {
        const u32x4 wm = s_cvw[(sx * 3 + 1) * 64 + lane];
        x3_grp2(cs_[0], fa[0], wm, cs_[1], fa[1], wm);
      }
    }
    // dgrad operands of both tiles from LDS while the conv drains; ONE register set, each plane re-read for the next K step
    // right behind the last MFMA group that uses it (l after the first group, m after the fifth, h after the sixth)
    u32x4 az[2][NPL];            // planes in storage order: bf16x3 h, m, l; f16x2 h, l
    auto load_az = [&](int sK, int pl) {
#pragma unroll
      for (int t = 0; t < 2; ++t) az[t][pl] = ldA[(pl * 32 + 16 * t + ch) * 16 + ((sK * 4 + kq) ^ sig_ch)];
    };
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) load_az(0, pl);

    POSB_STAMP(1);
    x3_drain(cb_[0], cs_[0], cb_[1], cs_[1]);
    float xh[2][4], rs[2][4], dxv[2][4];
    f32x4 fs4[2];               // f16x2: the dz row factors of this lane's eight samples
    (void)fs4;
#if POS_BWD_PK
    f32x2 xh2[2][2], rs2[2][2], y2[2][2];      // [tile][pair]: values r = 2 pair, 2 pair + 1 of the lane
    const f32x2 bias2 = {bias, bias}, g02 = {g0, g0};
    // f16x2: relu's argument is formed with scale / bias times kh1 (a power of two: same sign, and max(.., 0) is the h1 operand's scaling)
    const f32x2 gy2 = {g0 * kh1, g0 * kh1}, by2 = {be0 * kh1, be0 * kh1};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 *sp = reinterpret_cast<const f32x4 *>(slot + P::O_STAT) + ((wave * 2 * POS_ST + 16 * t + 4 * kq) >> 2);
      const f32x4 mean4 = sp[0], rs4 = sp[POS_ST / 4];
      const f32x2 cbl = {cb_[t].x, cb_[t].y}, cbh = {cb_[t].z, cb_[t].w}, csl = {cs_[t].x, cs_[t].y}, csh = {cs_[t].z, cs_[t].w};
      const f32x2 sc2 = {oscale, oscale};
      const f32x2 vl = (cbl + csl) * sc2 + bias2, vh = (cbh + csh) * sc2 + bias2;      // -ffp-contract=off: add, mul, add as in the forward
      rs2[t][0] = f32x2{rs4.x, rs4.y}; rs2[t][1] = f32x2{rs4.z, rs4.w};
      xh2[t][0] = (vl - f32x2{mean4.x, mean4.y}) * rs2[t][0];
      xh2[t][1] = (vh - f32x2{mean4.z, mean4.w}) * rs2[t][1];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        y2[t][h] = __builtin_elementwise_fma(xh2[t][h], gy2, by2);
        xh[t][2 * h] = xh2[t][h].x; xh[t][2 * h + 1] = xh2[t][h].y;
        rs[t][2 * h] = rs2[t][h].x; rs[t][2 * h + 1] = rs2[t][h].y;
      }
    }
#else
    for (int t = 0; t < 2; ++t) {
      const f32x4 cvo = (cb_[t] + cs_[t]) * ConvX3<C>::OUT_SCALE;
      const float v[4] = {cvo.x + bias, cvo.y + bias, cvo.z + bias, cvo.w + bias};
      float mean[4];
      {
        const f32x4 *sp = reinterpret_cast<const f32x4 *>(slot + P::O_STAT) + ((wave * POS_ST + 16 * t + 4 * kq) >> 1);
        const f32x4 s0 = sp[0], s1 = sp[1];
        mean[0] = s0.x; rs[t][0] = s0.y; mean[1] = s0.z; rs[t][1] = s0.w;
        mean[2] = s1.x; rs[t][2] = s1.y; mean[3] = s1.z; rs[t][3] = s1.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xh[t][r] = (v[r] - mean[r]) * rs[t][r];
      }
    }
#endif
    POSB_STAMP(2);
    // ---- dgrad: dh1[sample][feature of this position], eight independent accumulators ----
    f32x4 gb[2], gs[2];         // {leading, small} terms per tile: four independent chains, each started on C = 0
#pragma unroll
    for (int sK = 0; sK < 4; ++sK) {
      if constexpr (H2) {      // small terms l h + h l on gs, the leading h h on gb
        if (sK == 0) h2_grp2_zero(gs[0], az[0][1], wfr[sK][0], gs[1], az[1][1], wfr[sK][0]);
        else h2_grp2(gs[0], az[0][1], wfr[sK][0], gs[1], az[1][1], wfr[sK][0]);
        if (sK + 1 < 4) load_az(sK + 1, 1);
        if (sK == 0) h2_grp2_zero(gb[0], az[0][0], wfr[sK][0], gb[1], az[1][0], wfr[sK][0]);
        else h2_grp2(gb[0], az[0][0], wfr[sK][0], gb[1], az[1][0], wfr[sK][0]);
        h2_grp2(gs[0], az[0][0], wfr[sK][1], gs[1], az[1][0], wfr[sK][1]);
        if (sK + 1 < 4) load_az(sK + 1, 0);
      } else {
        if (sK == 0) x3_grp2_zero(gs[0], az[0][2], wfr[sK][0], gs[1], az[1][2], wfr[sK][0]);
        else x3_grp2(gs[0], az[0][2], wfr[sK][0], gs[1], az[1][2], wfr[sK][0]);
        if (sK + 1 < 4) load_az(sK + 1, 2);
        if (sK == 0) x3_grp2_zero(gb[0], az[0][1], wfr[sK][0], gb[1], az[1][1], wfr[sK][0]);
        else x3_grp2(gb[0], az[0][1], wfr[sK][0], gb[1], az[1][1], wfr[sK][0]);
        x3_grp4(gs[0], az[0][0], wfr[sK][NPL - 1], gs[1], az[1][0], wfr[sK][NPL - 1], gb[0], az[0][0], wfr[sK][1], gb[1], az[1][0], wfr[sK][1]);
        x3_grp2(gs[0], az[0][1], wfr[sK][1], gs[1], az[1][1], wfr[sK][1]);
        if (sK + 1 < 4) load_az(sK + 1, 1);
        x3_grp2(gb[0], az[0][0], wfr[sK][0], gb[1], az[1][0], wfr[sK][0]);
        if (sK + 1 < 4) load_az(sK + 1, 0);
      }
    }
    POSB_STAMP(3);
    x3_drain(gb[0], gs[0], gb[1], gs[1]);
    // ---- relu mask + LayerNorm_0 backward (per point = per (kq, r); sums over the 16 channel lanes, four at a time) ----
#if POS_BWD_PK
    {
      f32x2 dxh2[2][2], dxx2[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x2 dhl = f32x2{gb[t].x, gb[t].y} + f32x2{gs[t].x, gs[t].y}, dhh = f32x2{gb[t].z, gb[t].w} + f32x2{gs[t].z, gs[t].w};
        if constexpr (H2) {      // undo the rows' dz scaling (and the fc1 planes' 2^7): factor = 2^-k / 128 per sample
          fs4[t] = *reinterpret_cast<const f32x4 *>(t32L + P::F_OFF + 16 * t + 4 * kq);
          dhl *= f32x2{fs4[t].x, fs4[t].y};
          dhh *= f32x2{fs4[t].z, fs4[t].w};
        }
        const f32x2 dh2[2] = {dhl, dhh};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 g2 = {y2[t][h].x > 0.0f ? dh2[h].x : 0.0f, y2[t][h].y > 0.0f ? dh2[h].y : 0.0f};
          gbi += g2.x; gbi += g2.y;                                  // (the three running sums keep their element order)
          gsc = fmaf(g2.x, xh2[t][h].x, gsc); gsc = fmaf(g2.y, xh2[t][h].y, gsc);
          dxh2[t][h] = g2 * g02;
          dxx2[t][h] = dxh2[t][h] * xh2[t][h];
        }
      }
      float s1[2][4], s2[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        group16_sum4_from(s1[t][0], s1[t][1], s1[t][2], s1[t][3], dxh2[t][0].x, dxh2[t][0].y, dxh2[t][1].x, dxh2[t][1].y);
        group16_sum4_from(s2[t][0], s2[t][1], s2[t][2], s2[t][3], dxx2[t][0].x, dxx2[t][0].y, dxx2[t][1].x, dxx2[t][1].y);
      }
      const f32x2 k16 = {1.0f / 16.0f, 1.0f / 16.0f};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x2 s1p = {s1[t][2 * h], s1[t][2 * h + 1]}, s2p = {s2[t][2 * h], s2[t][2 * h + 1]};
          const f32x2 d2 = rs2[t][h] * ((dxh2[t][h] - s1p * k16) - xh2[t][h] * (s2p * k16));
          dxv[t][2 * h] = d2.x; dxv[t][2 * h + 1] = d2.y;
          gbc += d2.x; gbc += d2.y;
        }
    }
#else
    {
      float dxh[2][4], dxx[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 dh4 = gb[t] + gs[t];
        const float dh[4] = {dh4.x, dh4.y, dh4.z, dh4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = fmaf(xh[t][r], g0, be0) > 0.0f ? dh[r] : 0.0f;
          gbi += g;
          gsc = fmaf(g, xh[t][r], gsc);
          dxh[t][r] = g * g0;
          dxx[t][r] = dxh[t][r] * xh[t][r];
        }
      }
      float s1[2][4], s2[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        group16_sum4_from(s1[t][0], s1[t][1], s1[t][2], s1[t][3], dxh[t][0], dxh[t][1], dxh[t][2], dxh[t][3]);
        group16_sum4_from(s2[t][0], s2[t][1], s2[t][2], s2[t][3], dxx[t][0], dxx[t][1], dxx[t][2], dxx[t][3]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dxv[t][r] = rs[t][r] * (dxh[t][r] - s1[t][r] * (1.0f / 16.0f) - xh[t][r] * (s2[t][r] * (1.0f / 16.0f)));
          gbc += dxv[t][r];
        }
    }
#endif
    POSB_STAMP(4);
    // h1 and dx of the 32 samples as bf16 planes, split ONCE: K slot j = sample 16 (j >> 2) + 4 kq + (j & 3), i.e. exactly
    // this lane's eight values -- already the A fragment of dW1p = h1^T dz and the B fragment of dWc = bits^T dx
    float h1v[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#if POS_BWD_PK
        h1v[t][r] = fmaxf((r & 1) ? y2[t][r >> 1].y : y2[t][r >> 1].x, 0.0f);      // the same fma(xh, g0, be0) as the relu mask's
        if constexpr (H2) h1v[t][r] *= fs4[t][r];                                  // (scaled by kh1) x the row factor: <= 2^15
#else
        h1v[t][r] = fmaxf(fmaf(xh[t][r], g0, be0), 0.0f);
#endif
      }
    const X3Frag fd = x3_split8(f32x4{dxv[0][0], dxv[0][1], dxv[0][2], dxv[0][3]}, f32x4{dxv[1][0], dxv[1][1], dxv[1][2], dxv[1][3]});
    // ---- dW1p[feature][o] += sum over the 32 samples of h1[sample][feature] dz[sample][o] ----
    // B fragments of column blocks 4 hf .. 4 hf + 3 (sK = cb >> 1, half h = cb & 1 of each quad): K slots 0..3 = the lane group's
    // four samples of tile 0, 4..7 = of tile 1 -- two transposing reads per (column block, plane)
    // (the four lane addresses are formed ONCE and hidden from the optimiser: it otherwise folds slot base + plane / half offset into
    // scalar registers and spends a v_add per read pair -- 16 per super-tile -- where the instruction's offset field does it for free)
    uint32_t trv[4];
#pragma unroll
    for (int sK = 0; sK < 4; ++sK) {
      trv[sK] = trb + tr_sk[sK];
      asm volatile("" : "+v"(trv[sK]));
    }
    auto tr_frag = [&](int cb, int pl) {
      const uint32_t a0 = trv[cb >> 1] + 8u * (cb & 1) + 8192u * pl;
      const u32x2_t v0 = pos_tr_read(a0), v1 = pos_tr_read(a0 + 4096u);
      return u32x4{v0.x, v0.y, v1.x, v1.y};
    };
    if constexpr (H2) {
      const H2Frag fh = h2_split8(f32x4{h1v[0][0], h1v[0][1], h1v[0][2], h1v[0][3]}, f32x4{h1v[1][0], h1v[1][1], h1v[1][2], h1v[1][3]});
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        u32x4 bh[4], bl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { bh[c] = tr_frag(4 * hf + c, 0); bl[c] = tr_frag(4 * hf + c, 1); }
        f32x4 *d = dw + 4 * hf;
        h2_grp4(d[0], fh.l, bh[0], d[1], fh.l, bh[1], d[2], fh.l, bh[2], d[3], fh.l, bh[3]);
        h2_grp4(d[0], fh.h, bl[0], d[1], fh.h, bl[1], d[2], fh.h, bl[2], d[3], fh.h, bl[3]);
        h2_grp4(d[0], fh.h, bh[0], d[1], fh.h, bh[1], d[2], fh.h, bh[2], d[3], fh.h, bh[3]);
      }
    } else {
      const X3Frag fh = x3_split8(f32x4{h1v[0][0], h1v[0][1], h1v[0][2], h1v[0][3]}, f32x4{h1v[1][0], h1v[1][1], h1v[1][2], h1v[1][3]});
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        u32x4 bh[4], bm[4], bl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { bh[c] = tr_frag(4 * hf + c, 0); bm[c] = tr_frag(4 * hf + c, 1); bl[c] = tr_frag(4 * hf + c, NPL - 1); }
        f32x4 *d = dw + 4 * hf;
        x3_grp4(d[0], fh.l, bh[0], d[1], fh.l, bh[1], d[2], fh.l, bh[2], d[3], fh.l, bh[3]);
        x3_grp4(d[0], fh.h, bl[0], d[1], fh.h, bl[1], d[2], fh.h, bl[2], d[3], fh.h, bl[3]);
        x3_grp4(d[0], fh.m, bm[0], d[1], fh.m, bm[1], d[2], fh.m, bm[2], d[3], fh.m, bm[3]);
        x3_grp4(d[0], fh.m, bh[0], d[1], fh.m, bh[1], d[2], fh.m, bh[2], d[3], fh.m, bh[3]);
        x3_grp4(d[0], fh.h, bm[0], d[1], fh.h, bm[1], d[2], fh.h, bm[2], d[3], fh.h, bm[3]);
        x3_grp4(d[0], fh.h, bh[0], d[1], fh.h, bh[1], d[2], fh.h, bh[2], d[3], fh.h, bh[3]);
      }
    }
    POSB_STAMP(5);
    // ---- conv weight gradient: dWc[k][ch] += sum over samples of bit(sample, k) dx[sample][ch]; the bits of window
    // element k over the 32 samples are ONE word of T32: nibble kq of each half = this lane's eight K slots ----
    // (the bit operand is two dependent LDS round trips -- T32 word -> byte -> table row -- right in front of its use.  Round 6 moved the
    // fetch up into the conv's / the dgrad's shadow: this wave's stamps improved, 4588 -> 4432 cycles per super-tile, and the KERNEL got
    // slower, 160.5 -> 168.2 / 165.1 us: the early LDS reads queue in front of the operand reads of the MFMA phases.  It stays here.)
    u32x4 fw[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      const uint32_t wv = t32L[tb[rb]];
      const uint32_t byte = __builtin_amdgcn_ubfe(wv, 4u * kq, 4u) | (__builtin_amdgcn_ubfe(wv, 16u + 4u * kq, 4u) << 4);
      fw[rb] = pos_expand8<C>(s_lut, byte);
    }
    x3_grp_sameb<NRB>(cw, fw, fd.l);   // NRB independent chains per plane: one asm group (one pad) each
    x3_grp_sameb<NRB>(cw, fw, fd.m);
    x3_grp_sameb<NRB>(cw, fw, fd.h);
    POSB_STAMP(6);
    if (POS_PAIR_SYNC == 0 || (j & 1) != 0) {
      pos_dma_wait();            // this wave's share of the next slot(s) has landed ...
      __syncthreads();           // ... and after the barrier everyone's has; every wave is done reading this period's slot(s)
    }
    POSB_STAMP(7);
  }

  // ---- epilogue: the position's rows of dW1 (kernel / fragment layout) and the conv-block record of the workgroup ----
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) x3_drain(dw[4 * hf], dw[4 * hf + 1], dw[4 * hf + 2], dw[4 * hf + 3]);
  {
    f32x4 *out = reinterpret_cast<f32x4 *>(w1out);
    // f16x2: the accumulators hold sum (kh1 h1 f) (dz / (128 f)) = (kh1 / 128) dW1 -- kh1 a power of two: the product is exact
    const float wsc = H2 ? H2_W_SCALE / kh1 : 1.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) out[(p * 8 + c) * 64 + lane] = H2 ? dw[c] * wsc : dw[c];
  }
  float *part = reinterpret_cast<float *>(pos_smem);   // [8 waves][CONVBLK]: the ring is dead (last barrier passed)
  static_assert((size_t)8 * CONVBLK * sizeof(float) <= (size_t)P::RS * P::SLOT * 16, "conv partials must fit the ring");
  {
    float *pr = part + wave * CONVBLK;
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
      x3_drain(cw[rb]);
      const f32x4 a = cw[rb] * (0.5f / 255.0f);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = 16 * rb + 4 * kq + r;
        if (k < Cfg::KW) pr[k * 16 + ch] = av[r];
      }
    }
    gbi += __shfl_xor(gbi, 16, 64); gbi += __shfl_xor(gbi, 32, 64);
    gsc += __shfl_xor(gsc, 16, 64); gsc += __shfl_xor(gsc, 32, 64);
    gbc += __shfl_xor(gbc, 16, 64); gbc += __shfl_xor(gbc, 32, 64);
    if (lane < 16) {
      pr[Cfg::KW * 16 + lane] = gbc; pr[Cfg::KW * 16 + 16 + lane] = gsc; pr[Cfg::KW * 16 + 32 + lane] = gbi;
    }
  }
  __syncthreads();
  float *gpos = wsx + W.gpos + (size_t)(chunk * 8 + pg) * CONVBLK;
  for (int e = tid; e < CONVBLK; e += POS_THREADS)
    gpos[e] = ((part[e] + part[CONVBLK + e]) + (part[2 * CONVBLK + e] + part[3 * CONVBLK + e])) +
              ((part[4 * CONVBLK + e] + part[5 * CONVBLK + e]) + (part[6 * CONVBLK + e] + part[7 * CONVBLK + e]));
}

// ---------------------------------------------------------------------------
// Forward half: conv -> LayerNorm_0 -> relu -> fc1 -> LayerNorm_1 -> relu -> head -> loss, and the head's backward down to
// dz (pqn_minatar.py:24-69, 271-285).  Workgroup = 256 samples of one seed's minibatch, wave = 32 samples (two MFMA tiles)
// for ALL 64 conv positions.  Per K step of fc1 (= two conv positions) a wave
//   - runs the conv TRANSPOSED (rows = channels, columns = samples), so that a lane's four conv outputs of a position are
//     four K slots of its fc1 A fragment: LayerNorm_0 is three in-lane adds and two lane swaps per sum, the activations
//     are split into bf16 planes ONCE, in registers, and no h1 tile exists anywhere;
//   - multiplies them against the K step's 24 KB of W1 planes, which ONE LDS-DMA per workgroup brings in for all eight
//     waves (two-slot ring): the plane stream per sample is 1/8 of what a 32-sample workgroup pulls through its CU;
//   - accumulates z[32][128] in 64 VGPRs over all 32 K steps: no split-K partials, no z tile in LDS.
// The head then works on the accumulator layout directly (lane = 8 output columns x 8 sample rows; row sums are DPP
// butterflies over the 16 column lanes), leaves its parameter-gradient sums as ONE record per workgroup, and writes dz as
// pre-split bf16 planes in the dgrad's operand order (dz_planes_a; the weight gradient reads the same image through the
// transposing LDS read) plus LayerNorm_0's (mean, 1/std).
// Inputs in minibatch order (pos_gather_kernel): packed rows, action, target.
// ---------------------------------------------------------------------------
// all-reduce of TWO per-lane values over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} with three lane swaps:
// fixed order ((l&15) + (l&15)+32) + ((l&15)+16 + (l&15)+48), the same bits in all four lanes
PQN_D void pos_quad_sum2(float &a, float &b) {
  const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const float w = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);        // lanes < 32: a over the halves, lanes >= 32: b
  const auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(w), __float_as_uint(w), false, false);
  const float u = __uint_as_float(r2[0]) + __uint_as_float(r2[1]);        // ... and over the row pairs
  const auto r3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(u), __float_as_uint(u), false, false);
  a = __uint_as_float(r3[0]);                                             // the lower half's total, everywhere
  b = __uint_as_float(r3[1]);
}
PQN_D float pos_quad_sum1(float a) {
  const auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
  const float w = __uint_as_float(r1[0]) + __uint_as_float(r1[1]);
  const auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(w), __float_as_uint(w), false, false);
  return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}

#ifndef POS_PAIR_SYNC_FWD
#define POS_PAIR_SYNC_FWD 1     // forward kernel: one barrier per two K steps over a four-slot ring (see POS_PAIR_SYNC)
#endif
// NW = waves per workgroup of the forward / rollout kernels (32 samples each): 8 in rounds 5-6; 4 and 2 (round 6, f16x2 only) give launches
// of fewer seeds or smaller minibatches a workgroup on every CU -- a workgroup still streams ALL of the fc1 planes through its LDS ring
template <int C, int NPL = 3, int NW = 8>
struct PosFwdCfg {
  using P = PosCfg<C, NPL>;
  static_assert(NW == 8 || NW == 4 || NW == 2, "waves per forward workgroup");
  static constexpr int N_W = NPL * 512;                             // chunks of one K step's planes: [NPL][8 cb][64]
  static constexpr int ROWW = POS_ST * P::ROWSTRIDE * 4;            // words of a wave's packed rows
  static constexpr int DZS = 132;                                   // LDS row stride of the dz transposition tile
  static constexpr int RS = POS_PAIR_SYNC_FWD ? 4 : 2;              // ring slots
  static constexpr size_t ring_bytes = (size_t)RS * N_W * 16;
  static constexpr size_t rows_bytes = (size_t)NW * ROWW * 4;
  static constexpr size_t misc_floats(int a) { return ((P::CONVBLK + 3) & ~3) + ((384 + 128 * a + a + 3) & ~3) + NW * 64 + 1024; }   // + the pos_expand8 table
  static constexpr size_t loop_bytes(int a) { return ring_bytes + rows_bytes + (size_t)P::NCS * NPL * 64 * 16 + sizeof(float) * misc_floats(a); }
  static constexpr size_t tail_bytes = (size_t)NW * POS_ST * DZS * 4;  // dz tiles of the waves (over ring + rows)
  static constexpr size_t lds_bytes(int a) {
    const size_t fixed = (size_t)P::NCS * NPL * 64 * 16 + sizeof(float) * misc_floats(a);
    const size_t front = ring_bytes + rows_bytes > tail_bytes ? ring_bytes + rows_bytes : tail_bytes;
    return front + fixed;
  }
};

#ifdef POS_STAMPS
#define POSF_STAMP(k) do { if (stamps && (s == 5 || (k) >= 8) && lane == 0 && blockIdx.x == 0 && wave == 0) stamps[16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define POSF_STAMP(k) do { } while (0)
#endif
// The K loop of fc1 for one wave's 32 samples: 32 K steps (two conv positions each), z accumulated in `zacc`.  Shared by the
// training forward kernel and the persistent rollout kernel (which runs it once per env step; kk0 = the number of K steps the
// workgroup has run before, so that the weight ring and the barrier period continue across env steps).  On entry the ring
// holds K steps kk0 and kk0 + 1 (visible to every wave); on exit it holds kk0 + 32 and kk0 + 33, i.e. steps 0 and 1 again.
template <int C>
struct PosFwdCtx {
  const u32x4 *wf;            // forward-order W1 planes of this seed [NPL][32 steps][8 cb][64]
  const u32x4 *ring;          // LDS ring of K-step plane sets [RS][NPL][8][64]
  uint32_t ring_lds;
  const u32x4 *s_cvw, *s_lut; // conv kernel planes [K step][plane][lane]; pos_expand8 table
  const uint32_t *rowsW;      // this wave's packed rows [32][ROWSTRIDE * 4]
  float *g_stat;              // LayerNorm_0 statistics of this wave's super-tile [64 pos][32][2] (STATS only)
  int lane, wave, i16, g;
  float cbias[4], cg0[4], cbe0[4];   // conv bias / LayerNorm_0 scale, bias of channels 4 g .. 4 g + 3 (f16x2: scale, bias times the h1 operand scale)
  float oscale;                      // conv accumulator -> conv output (bf16x3: 0.5 / 255; f16x2: that over the conv kernel's scale)
  float zscale;                      // f16x2: fc1 accumulator -> z (1 / (h1 scale * 128))
  unsigned long long *stamps;
};
// per-wave constants of the K loop from the LDS copy of the conv block (after the barrier that completes it)
template <int C, int NPL>
PQN_D float pos_fwd_consts(PosFwdCtx<C> &cx, const float *s_wc) {   // returns the conv kernel's scale (the plane builder's argument)
  using Cfg = CnnCfg<C>;
  float sc = 1.0f, sh = 1.0f;
  cx.oscale = ConvX3<C>::OUT_SCALE;
  cx.zscale = 1.0f;
  if constexpr (NPL == 2) {
    sc = pos_h2_conv_scale<C>(s_wc, cx.lane);
    sh = pos_h2_h1_scale<C>(s_wc, cx.lane);
    cx.oscale = ConvX3<C>::OUT_SCALE / sc;
    cx.zscale = 1.0f / (sh * H2_W_SCALE);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    cx.cbias[r] = s_wc[Cfg::KW * 16 + 4 * cx.g + r];
    cx.cg0[r] = s_wc[Cfg::KW * 16 + 16 + 4 * cx.g + r] * sh;
    cx.cbe0[r] = s_wc[Cfg::KW * 16 + 32 + 4 * cx.g + r] * sh;
  }
  return sc;
}
template <int C, int NPL, int NW>
PQN_D void pos_fwd_dma_step(const PosFwdCtx<C> &cx, int kk) {   // K step kk & 31 -> slot kk % RS; wave w moves column blocks w, w + NW, .. of each plane
  using F = PosFwdCfg<C, NPL, NW>;
  const uint32_t offW = (uint32_t)(cx.lane * 16);
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
    for (int cbk = 0; cbk < 8 / NW; ++cbk) {
      const int cb = cx.wave + NW * cbk;
      pos_dma16(offW, cx.wf + (size_t)pl * (X3_PLANE / 8) + ((kk & 31) * 8 + cb) * 64,
                cx.ring_lds + (uint32_t)(((kk & (F::RS - 1)) * F::N_W + (pl * 8 + cb) * 64) * 16));
    }
}
template <int C, int NPL, int NW, bool STATS>
PQN_D void pos_fwd_kloop(const PosFwdCtx<C> &cx, int kk0, f32x4 (&zacc)[2][8]) {
  using P = PosCfg<C, NPL>;
  using F = PosFwdCfg<C, NPL, NW>;
  using M = PosMM<NPL>;
  constexpr bool H2 = NPL == 2;
  constexpr int RB = P::RB, NCS = P::NCS;
  unsigned long long *stamps = cx.stamps;
  const int lane = cx.lane, wave = cx.wave;
  (void)stamps; (void)lane; (void)wave;
  auto dma_step = [&](int kk) { pos_fwd_dma_step<C, NPL, NW>(cx, kk); };
  // window masks of sample (16 t + lane & 15) at the two positions of K step sn: p0 = 8 py + pxb, p1 = p0 + 1 (same window
  // rows, one column apart) -- in two halves: the LDS reads, and (once they have arrived) the shifts
  auto mask_words = [&](int sn, uint32_t (&lo)[3][2], uint32_t (&hi)[3][2]) {
    sn = min(sn, 31);
    const int py = sn >> 2, pxb = 2 * (sn & 3);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int bwd = (((py + ky) * 10 + pxb) * C) >> 5;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t *row = cx.rowsW + (16 * t + cx.i16) * (P::ROWSTRIDE * 4);
        lo[ky][t] = row[bwd];
        hi[ky][t] = row[bwd + 1];
      }
    }
  };
  auto mask_finish = [&](int sn, const uint32_t (&lo)[3][2], const uint32_t (&hi)[3][2], uint32_t (&m)[2][2][3]) {
    sn = min(sn, 31);
    const int py = sn >> 2, pxb = 2 * (sn & 3);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int bsh = (((py + ky) * 10 + pxb) * C) & 31;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t v = (uint32_t)(((((uint64_t)hi[ky][t]) << 32) | lo[ky][t]) >> bsh);
        m[0][t][ky] = v & ((1u << RB) - 1u);
        m[1][t][ky] = (v >> C) & ((1u << RB) - 1u);
      }
    }
  };
  uint32_t mk[2][2][3];                                 // [position][tile][window row] of the CURRENT K step
  {
    uint32_t lo0[3][2], hi0[3][2];
    mask_words(0, lo0, hi0);
    mask_finish(0, lo0, hi0, mk);
  }

#pragma unroll 1
  for (int s = 0; s < 32; ++s) {
    if constexpr (POS_PAIR_SYNC_FWD != 0) {
      if ((s & 1) == 0) { dma_step(kk0 + s + 2); dma_step(kk0 + s + 3); }
      if ((cx.wave >= NW / 2) == ((s & 1) != 0)) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    } else {
      dma_step(kk0 + s + 1);
    }
    POSF_STAMP(0);
    const u32x4 *slot = cx.ring + ((kk0 + s) & (F::RS - 1)) * F::N_W + cx.lane;
    POSF_STAMP(1);
    // ---- conv (transposed) of the two positions x two tiles at once: eight independent accumulator chains ----
    f32x4 cb_[2][2], cs_[2][2];   // (round 6) the first product of every chain runs on C = 0: no zeroed register tuples
#pragma unroll
    for (int sx = 0; sx < NCS; ++sx) {
      u32x4 fa[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[q][t] = pos_expand8<C>(cx.s_lut, __builtin_amdgcn_ubfe(ConvX3<C>::word(mk[q][t], sx), 8u * cx.g, 8u));
      const u32x4 wh = cx.s_cvw[(sx * NPL + 0) * 64 + cx.lane], wl = cx.s_cvw[(sx * NPL + NPL - 1) * 64 + cx.lane];
      if (sx == 0) {
        M::grp4_zero(cs_[0][0], wl, fa[0][0], cs_[0][1], wl, fa[0][1], cs_[1][0], wl, fa[1][0], cs_[1][1], wl, fa[1][1]);
        M::grp4_zero(cb_[0][0], wh, fa[0][0], cb_[0][1], wh, fa[0][1], cb_[1][0], wh, fa[1][0], cb_[1][1], wh, fa[1][1]);
      } else {
        M::grp4(cs_[0][0], wl, fa[0][0], cs_[0][1], wl, fa[0][1], cs_[1][0], wl, fa[1][0], cs_[1][1], wl, fa[1][1]);
        M::grp4(cb_[0][0], wh, fa[0][0], cb_[0][1], wh, fa[0][1], cb_[1][0], wh, fa[1][0], cb_[1][1], wh, fa[1][1]);
      }
      if constexpr (!H2) {
        const u32x4 wm = cx.s_cvw[(sx * 3 + 1) * 64 + cx.lane];
        x3_grp4(cs_[0][0], wm, fa[0][0], cs_[0][1], wm, fa[0][1], cs_[1][0], wm, fa[1][0], cs_[1][1], wm, fa[1][1]);
      }
    }
    // the next K step's window words (the rows are this wave's own: always resident) go out now, in the conv's shadow
    uint32_t nlo[3][2], nhi[3][2];
    mask_words(s + 1, nlo, nhi);
    x3_drain(cb_[0][0], cs_[0][0], cb_[0][1], cs_[0][1]);
    x3_drain(cb_[1][0], cs_[1][0], cb_[1][1], cs_[1][1]);
    // ---- LayerNorm_0 + relu of the four (position, tile) combinations: independent chains, interleaved by the scheduler ----
    float y[2][2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 cvo = (cb_[q][t] + cs_[q][t]) * cx.oscale;
        const float v[4] = {cvo.x + cx.cbias[0], cvo.y + cx.cbias[1], cvo.z + cx.cbias[2], cvo.w + cx.cbias[3]};
        float sum = (v[0] + v[1]) + (v[2] + v[3]);
        float sq = fmaf(v[3], v[3], fmaf(v[2], v[2], fmaf(v[1], v[1], v[0] * v[0])));
        pos_quad_sum2(sum, sq);                          // the 16 channels of (sample, position): 4 lanes x 4
        const float mean = sum * (1.0f / 16.0f);
        const float var = fmaxf(sq * (1.0f / 16.0f) - mean * mean, 0.0f);
        const float rstd = rsqrt_exact(var + QN_LN_EPS);
#pragma unroll
        for (int r = 0; r < 4; ++r) y[q][t][r] = fmaxf(fmaf((v[r] - mean) * rstd, cx.cg0[r], cx.cbe0[r]), 0.0f);
#if POS_BWD_PK
        if (STATS && cx.g < 2)     // LayerNorm_0 statistics for the backward: [position][mean | 1/std][sample]; lanes g = 0 store the mean, g = 1 1/std
          cx.g_stat[((size_t)(2 * s + q) * 2 + cx.g) * POS_ST + 16 * t + cx.i16] = cx.g == 0 ? mean : rstd;
#else
        if (STATS && cx.g == 0) {                                    // LayerNorm_0 statistics for the backward: [position][sample][2]
          f32x2 ms = {mean, rstd};                                   // (all four g lanes hold the same bits; storing from all of them was measured SLOWER: 4x the store requests)
          *reinterpret_cast<f32x2 *>(cx.g_stat + ((size_t)(2 * s + q) * POS_ST + 16 * t + cx.i16) * 2) = ms;
        }
#endif
      }
    mask_finish(s + 1, nlo, nhi, mk);
    POSF_STAMP(2);
    // ---- fc1: K slots 0..3 = position p0's channels 4 g .., 4..7 = p1's (x3_fwd_index); the fragments of column-block pair
    // c + 1 are read while the 24 MFMAs of pair c issue (two register sets) ----
    // (f16x2: y already carries the h1 operand scale through cg0 / cbe0)
    u32x4 afh[2], afm[2], afl[2];
    (void)afm;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 ya = {y[0][t][0], y[0][t][1], y[0][t][2], y[0][t][3]}, yb = {y[1][t][0], y[1][t][1], y[1][t][2], y[1][t][3]};
      if constexpr (H2) {
        const H2Frag f = h2_split8(ya, yb);
        afh[t] = f.h; afl[t] = f.l;
      } else {
        const X3Frag f = x3_split8(ya, yb);
        afh[t] = f.h; afm[t] = f.m; afl[t] = f.l;
      }
    }
    POSF_STAMP(3);
    u32x4 bf[2][2 * NPL];                                // [set][planes of block c | planes of block c + 1], storage order (h, m, l / h, l)
    auto load_b = [&](int c, u32x4 (&d)[2 * NPL]) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) { d[pl] = slot[(pl * 8 + c) * 64]; d[NPL + pl] = slot[(pl * 8 + c + 1) * 64]; }
    };
    load_b(0, bf[0]);
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
      if (c + 2 < 8) load_b(c + 2, bf[((c >> 1) + 1) & 1]);
      const u32x4 (&b)[2 * NPL] = bf[(c >> 1) & 1];
      f32x4 &z00 = zacc[0][c], &z10 = zacc[1][c], &z01 = zacc[0][c + 1], &z11 = zacc[1][c + 1];
      if constexpr (H2) {
        h2_grp4(z00, afl[0], b[0], z10, afl[1], b[0], z01, afl[0], b[2], z11, afl[1], b[2]);
        h2_grp4(z00, afh[0], b[1], z10, afh[1], b[1], z01, afh[0], b[3], z11, afh[1], b[3]);
        h2_grp4(z00, afh[0], b[0], z10, afh[1], b[0], z01, afh[0], b[2], z11, afh[1], b[2]);
      } else {
        x3_grp4(z00, afl[0], b[0], z10, afl[1], b[0], z01, afl[0], b[3], z11, afl[1], b[3]);
        x3_grp4(z00, afh[0], b[2], z10, afh[1], b[2], z01, afh[0], b[5], z11, afh[1], b[5]);
        x3_grp4(z00, afm[0], b[1], z10, afm[1], b[1], z01, afm[0], b[4], z11, afm[1], b[4]);
        x3_grp4(z00, afm[0], b[0], z10, afm[1], b[0], z01, afm[0], b[3], z11, afm[1], b[3]);
        x3_grp4(z00, afh[0], b[1], z10, afh[1], b[1], z01, afh[0], b[4], z11, afh[1], b[4]);
        x3_grp4(z00, afh[0], b[0], z10, afh[1], b[0], z01, afh[0], b[3], z11, afh[1], b[3]);
      }
    }
    POSF_STAMP(4);
    if (POS_PAIR_SYNC_FWD == 0 || (s & 1) != 0) {
      pos_dma_wait();
      __syncthreads();
    }
    POSF_STAMP(5);
  }
}

template <int C, int NA, int NPL, int NW>
__global__ __launch_bounds__(64 * NW) void cnn_pos_fwd_kernel(int nb, const float *__restrict__ theta, pqn_cnn_layout_t L, float inv_b,
                                                              float *__restrict__ wsx, pos_ws_t W, pqn_seeds_t sd,
                                                              unsigned long long *__restrict__ stamps) {
  using P = PosCfg<C, NPL>;
  using F = PosFwdCfg<C, NPL, NW>;
  constexpr int NT = 64 * NW;                        // threads of the workgroup
  constexpr bool H2 = NPL == 2;
  constexpr int CONVBLK = P::CONVBLK, NCS = P::NCS;
  constexpr int REC = CONVBLK + 384 + 128 * NA + NA + 2;
  extern __shared__ __attribute__((aligned(16))) char pos_smem[];
  constexpr size_t FRONT = F::ring_bytes + F::rows_bytes > F::tail_bytes ? F::ring_bytes + F::rows_bytes : F::tail_bytes;
  u32x4 *ring = reinterpret_cast<u32x4 *>(pos_smem);                               // [RS][NPL][8][64]
  uint32_t *s_rows = reinterpret_cast<uint32_t *>(pos_smem + F::ring_bytes);       // [NW waves][32][ROWSTRIDE * 4]
  u32x4 *s_cvw = reinterpret_cast<u32x4 *>(pos_smem + FRONT);                       // conv kernel planes [K step][plane][lane]
  float *s_wc = reinterpret_cast<float *>(s_cvw + NCS * NPL * 64);                  // conv kernel | bias | ln0 scale | ln0 bias
  float *s_hp = s_wc + ((CONVBLK + 3) & ~3);                                        // b1 | ln1 scale | ln1 bias | w2[128][A] | b2
  float *s_at = s_hp + ((384 + 128 * NA + NA + 3) & ~3);                            // [NW waves][32 act (as int) | 32 tgt]
  u32x4 *s_lut = reinterpret_cast<u32x4 *>(s_at + NW * 64);                         // pos_expand8 table
  // XCD-aware (seed, block): the blocks of a seed stream the same 768 KB of planes -- one XCD's L2 per seed when possible
  const int nblk = nb / (POS_ST * NW), nsl = gridDim.x / nblk;
  int seed_l, blk;
  {
    const int lin = blockIdx.x;
    if ((nsl & 7) == 0) {
      const int xcd = lin & 7, k = lin >> 3;
      seed_l = xcd * (nsl >> 3) + k / nblk;
      blk = k % nblk;
    } else {
      seed_l = lin / nblk;
      blk = lin % nblk;
    }
  }
  const int seed = seed_l + sd.seed_base;
  theta += seed * sd.theta_stride;
  wsx += seed * sd.ws_stride;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g = lane >> 4;          // conv phase: sample column, channels 4 g .. 4 g + 3; head: output column, rows 4 g ..
  const int st = blk * NW + wave;                    // this wave's super-tile
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const u32x4 *wf = reinterpret_cast<const u32x4 *>(theta + L.off_w1h + (H2 ? H2_PLANES_OFF : 0));   // forward-order planes [NPL][32 steps][8 cb][64]
  const uint32_t ring_lds = pos_lds_addr(ring);
  {   // the first two K steps' planes go out before anything else
    PosFwdCtx<C> c0;
    c0.wf = wf; c0.ring_lds = ring_lds; c0.lane = lane; c0.wave = wave;
    pos_fwd_dma_step<C, NPL, NW>(c0, 0);
    if constexpr (POS_PAIR_SYNC_FWD != 0) pos_fwd_dma_step<C, NPL, NW>(c0, 1);
  }
  // ---- prologue: parameters, this wave's packed rows / actions / targets ----
  for (int i = tid; i < CONVBLK; i += NT) s_wc[i] = theta[L.off_wc + i];
  for (int i = tid; i < 384; i += NT) s_hp[i] = theta[L.off_b1 + i];
  for (int i = tid; i < 128 * NA; i += NT) s_hp[384 + i] = theta[L.off_w2 + i];
  if (tid < NA) s_hp[384 + 128 * NA + tid] = theta[L.off_b2 + tid];
  pos_lut_fill(s_lut, tid, NT);
  {
    const u32x4 *g_rows = reinterpret_cast<const u32x4 *>(wsx + W.mb_bits) + (size_t)st * POS_ST * P::ROWCH;
    u32x4 *rw = reinterpret_cast<u32x4 *>(s_rows + wave * F::ROWW);
    for (int c = lane; c < POS_ST * P::ROWCH; c += 64) {
      const int row = c / P::ROWCH, cc = c - row * P::ROWCH;
      rw[row * P::ROWSTRIDE + cc] = g_rows[c];
    }
    if (lane < POS_ST) {
      s_at[wave * 64 + lane] = __int_as_float(reinterpret_cast<const int32_t *>(wsx + W.act)[st * POS_ST + lane]);
      s_at[wave * 64 + 32 + lane] = (wsx + W.tgt)[st * POS_ST + lane];
    }
  }
  pos_dma_wait();
  __syncthreads();
  const uint32_t *rowsW = s_rows + wave * F::ROWW;
  float *g_stat = wsx + W.stats + (size_t)st * (64 * POS_ST * 2);
  PosFwdCtx<C> cx;
  cx.wf = wf; cx.ring = ring; cx.ring_lds = ring_lds; cx.s_cvw = s_cvw; cx.s_lut = s_lut; cx.rowsW = rowsW; cx.g_stat = g_stat;
  cx.lane = lane; cx.wave = wave; cx.i16 = i16; cx.g = g; cx.stamps = stamps;
  {   // lane constants of the transposed conv (channels 4 g .. 4 g + 3), the operand scales, the conv kernel's planes
    const float sc = pos_fwd_consts<C, NPL>(cx, s_wc);
    if (wave == 0) pos_conv_planes<C, NPL>(s_wc, s_cvw, lane, sc);
  }
  __syncthreads();
  f32x4 zacc[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 8; ++c) zacc[t][c] = zero4;
  pos_fwd_kloop<C, NPL, NW, true>(cx, 0, zacc);
  { const int s = 0; (void)s; POSF_STAMP(8); }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int c = 0; c < 8; c += 4) x3_drain(zacc[t][c], zacc[t][c + 1], zacc[t][c + 2], zacc[t][c + 3]);

  // ================= head, on the accumulator layout: lane = columns o = 16 c + i16, rows = samples 16 t + 4 g + r =================
  float *dzt = reinterpret_cast<float *>(pos_smem) + (size_t)wave * POS_ST * F::DZS;   // this wave's dz tile (ring / rows are dead)
  float hb1[8], hs1[8], hbe1[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    hb1[c] = s_hp[16 * c + i16];
    hs1[c] = s_hp[128 + 16 * c + i16];
    hbe1[c] = s_hp[256 + 16 * c + i16];
  }
  float a_b1[8], a_sc[8], a_bi[8], a_w2[8][NA], a_b2[NA], a_loss = 0.f, a_cq = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    a_b1[c] = 0.f; a_sc[c] = 0.f; a_bi[c] = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) a_w2[c][a] = 0.f;
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) a_b2[a] = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    float zz[8][4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if constexpr (H2) {   // accumulator / (h1 scale * 128): a power of two, so the fma rounds once, like the add
        zz[c][0] = fmaf(zacc[t][c].x, cx.zscale, hb1[c]); zz[c][1] = fmaf(zacc[t][c].y, cx.zscale, hb1[c]);
        zz[c][2] = fmaf(zacc[t][c].z, cx.zscale, hb1[c]); zz[c][3] = fmaf(zacc[t][c].w, cx.zscale, hb1[c]);
      } else {
        zz[c][0] = zacc[t][c].x + hb1[c]; zz[c][1] = zacc[t][c].y + hb1[c]; zz[c][2] = zacc[t][c].z + hb1[c]; zz[c][3] = zacc[t][c].w + hb1[c];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __builtin_amdgcn_sched_barrier(0);                 // one row at a time: interleaved rows multiply the live registers (spills)
      const int smp = 16 * t + 4 * g + r;
      float sum = 0.f, sq = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) { sum += zz[c][r]; sq = fmaf(zz[c][r], zz[c][r], sq); }
      sum = group16_sum(sum);
      sq = group16_sum(sq);
      const float mean = sum * (1.0f / QN_HID);
      const float var = fmaxf(sq * (1.0f / QN_HID) - mean * mean, 0.0f);
      const float rstd = rsqrt_exact(var + QN_LN_EPS);
      float xh[8], h2[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        xh[c] = (zz[c][r] - mean) * rstd;
        h2[c] = fmaxf(fmaf(xh[c], hs1[c], hbe1[c]), 0.0f);
      }
      float qv[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        float part = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) part = fmaf(h2[c], s_hp[384 + (16 * c + i16) * NA + a], part);
        qv[a] = group16_sum(part) + s_hp[384 + 128 * NA + a];
      }
      const int act = __float_as_int(s_at[wave * 64 + smp]);
      const float tgt = s_at[wave * 64 + 32 + smp];
      float chosen = qv[0];
#pragma unroll
      for (int a = 1; a < NA; ++a)
        if (a == act) chosen = qv[a];
      const float diff = chosen - tgt;
      const float gm = diff * inv_b;                     // d loss / d q_a, loss = 0.5 * mean(diff^2)  (pqn_minatar.py:285)
      a_loss = fmaf(0.5f * diff, diff, a_loss);
      a_cq += chosen;
      float dxh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float w2a = s_hp[384 + (16 * c + i16) * NA];
#pragma unroll
        for (int a = 1; a < NA; ++a)
          if (a == act) w2a = s_hp[384 + (16 * c + i16) * NA + a];
        const float dy = h2[c] > 0.0f ? gm * w2a : 0.0f;
        a_sc[c] = fmaf(dy, xh[c], a_sc[c]);
        a_bi[c] += dy;
#pragma unroll
        for (int a = 0; a < NA; ++a) a_w2[c][a] = fmaf(h2[c], a == act ? gm : 0.0f, a_w2[c][a]);
        dxh[c] = dy * hs1[c];
        s1 += dxh[c];
        s2 = fmaf(dxh[c], xh[c], s2);
      }
#pragma unroll
      for (int a = 0; a < NA; ++a) a_b2[a] += (a == act ? gm : 0.0f);
      s1 = group16_sum(s1) * (1.0f / QN_HID);
      s2 = group16_sum(s2) * (1.0f / QN_HID);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float dz = rstd * (dxh[c] - s1 - xh[c] * s2);
        a_b1[c] += dz;
        zz[c][r] = dz;
      }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float *dp = dzt + (16 * t + 4 * g) * F::DZS + 16 * c + i16;
      dp[0] = zz[c][0]; dp[F::DZS] = zz[c][1]; dp[2 * F::DZS] = zz[c][2]; dp[3 * F::DZS] = zz[c][3];
    }
  }
  { const int s = 0; (void)s; POSF_STAMP(9); }
  // ---- dz as bf16 planes in the dgrad's operand order (dz_planes_a; the backward derives the weight gradient's operand from
  // the same image with transposing LDS reads): through the wave's LDS tile, rows = samples ----
  {
    u32x4 *g_dza = reinterpret_cast<u32x4 *>(wsx + W.dz);
    const size_t pa = (size_t)nb * 16;
    // (the tile was written by this wave only: no barrier, the compiler orders the LDS reads behind the writes)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 va[4], vb[4];       // sample row 16 t + i16: this lane's 32 of its 128 values (lanes g = 0..3 hold the rest)
#pragma unroll
      for (int sK = 0; sK < 4; ++sK) {
        const float *zr = dzt + (16 * t + i16) * F::DZS + 32 * sK + 4 * g;
        va[sK] = *reinterpret_cast<const f32x4 *>(zr);
        vb[sK] = *reinterpret_cast<const f32x4 *>(zr + 16);
      }
      if constexpr (H2) {       // the row goes out scaled into [2^14, 2^15) by its own power of two; the backward gets 2^-k / 128
        float am = 0.0f;
#pragma unroll
        for (int sK = 0; sK < 4; ++sK) {
          am = fmaxf(fmaxf(am, fmaxf(fabsf(va[sK].x), fabsf(va[sK].y))), fmaxf(fabsf(va[sK].z), fabsf(va[sK].w)));
          am = fmaxf(fmaxf(am, fmaxf(fabsf(vb[sK].x), fabsf(vb[sK].y))), fmaxf(fabsf(vb[sK].z), fabsf(vb[sK].w)));
        }
        am = fmaxf(am, __shfl_xor(am, 16, 64));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        int e = __builtin_amdgcn_frexp_expf(am);
        e = e < -40 ? -40 : (e > 60 ? 60 : e);
        const float inv = __int_as_float((127 + 15 - e) << 23);
        // (stored by all four g lanes of the sample -- the same bits: a lane-masked store here is a branch, and with it the kernel spilled 85 VGPRs)
        (wsx + W.t32 + (size_t)st * P::TW + P::F_OFF)[16 * t + i16] = __int_as_float((127 + e - 15 - H2_W_SHIFT) << 23);
#pragma unroll
        for (int sK = 0; sK < 4; ++sK) { va[sK] *= inv; vb[sK] *= inv; }
      }
#pragma unroll
      for (int sK = 0; sK < 4; ++sK) {
        const size_t e = ((size_t)st * POS_ST + 16 * t + i16) * 16 + sK * 4 + g;
        if constexpr (H2) {
          const H2Frag f = h2_split8(va[sK], vb[sK]);
          g_dza[e] = f.h;
          g_dza[pa + e] = f.l;
        } else {
          const X3Frag f = x3_split8(va[sK], vb[sK]);
          g_dza[e] = f.h;
          g_dza[pa + e] = f.m;
          g_dza[2 * pa + e] = f.l;
        }
      }
    }
  }
  // ---- the workgroup's record of head-parameter gradient sums: rows of the lane -> the 4 row groups -> the 8 waves ----
  __syncthreads();                                       // every wave is done with its dz tile: the front of the LDS becomes the record scratch
  float *recw = reinterpret_cast<float *>(pos_smem) + (size_t)wave * REC;
  static_assert((size_t)NW * REC * sizeof(float) <= FRONT, "records must fit the front region");
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    pos_quad_sum2(a_b1[c], a_sc[c]);
    a_bi[c] = pos_quad_sum1(a_bi[c]);
#pragma unroll
    for (int a = 0; a < NA; ++a) a_w2[c][a] = pos_quad_sum1(a_w2[c][a]);
  }
#pragma unroll
  for (int a = 0; a < NA; ++a) a_b2[a] = pos_quad_sum1(a_b2[a]);
  pos_quad_sum2(a_loss, a_cq);
  if (lane < 16) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int o = 16 * c + lane;
      recw[CONVBLK + o] = a_b1[c];
      recw[CONVBLK + 128 + o] = a_sc[c];
      recw[CONVBLK + 256 + o] = a_bi[c];
#pragma unroll
      for (int a = 0; a < NA; ++a) recw[CONVBLK + 384 + o * NA + a] = a_w2[c][a];
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < NA; ++a) recw[CONVBLK + 384 + 128 * NA + a] = a_b2[a];
      recw[REC - 2] = a_loss;
      recw[REC - 1] = a_cq;
    }
  }
  __syncthreads();
  float *rec = wsx + W.recs + (size_t)blk * REC;
  const float *r0 = reinterpret_cast<const float *>(pos_smem);
  for (int e = tid; e < REC; e += NT) {
    float v = 0.0f;
    if (e >= CONVBLK) {
      if constexpr (NW == 8) v = ((r0[e] + r0[REC + e]) + (r0[2 * REC + e] + r0[3 * REC + e])) + ((r0[4 * REC + e] + r0[5 * REC + e]) + (r0[6 * REC + e] + r0[7 * REC + e]));
      else if constexpr (NW == 4) v = (r0[e] + r0[REC + e]) + (r0[2 * REC + e] + r0[3 * REC + e]);
      else v = r0[e] + r0[REC + e];
    }
    rec[e] = v;
  }
  { const int s = 0; (void)s; POSF_STAMP(11); }
}

// ---------------------------------------------------------------------------
// Persistent rollout in the same structure (the `_step_env` scan + bootstrap forward, pqn_minatar.py:181-235; with
// eps = EPS_TEST and store_obs = 0 the evaluation scan :380-401): one workgroup owns 256 envs for all T steps, wave = 32 envs.
// Per env step a wave runs the K loop of the training forward (pos_fwd_kloop: conv on the fly from its own packed rows in
// LDS, W1 planes through the workgroup's LDS ring, z in registers), the head's forward on the accumulator layout, and -- lane
// = env, the lower half of the wave -- the eps-greedy draw, the transition rule, LogWrapper and the record, exactly as
// qnet_cnn_rollout_kernel does; the new observation bits go straight back into the wave's rows.  Nothing but the K loop's
// barriers synchronises the waves.  q is summed in the K order of the training forward kernel (one chain over the 32 K steps),
// not in the order of the 16-env kernels of pqn_qnet.hip: the two agree to f32 rounding, not bit for bit.
// ---------------------------------------------------------------------------
template <int C, int NA, int NPL, int NW>
struct PosRollCfg {
  using P = PosCfg<C, NPL>;
  using F = PosFwdCfg<C, NPL, NW>;
  static constexpr size_t fixed_floats = ((P::CONVBLK + 3) & ~3) + ((384 + 128 * NA + NA + 3) & ~3) + NW * POS_ST * 8 + 1024;
  static constexpr size_t lds_bytes = F::ring_bytes + F::rows_bytes + (size_t)P::NCS * NPL * 64 * 16 + sizeof(float) * fixed_floats;
};

template <int C, class Env, int NA, int NPL, int NW>
__global__ __launch_bounds__(64 * NW) void cnn_pos_rollout_kernel(
    int n, int t_len, uint32_t *__restrict__ state, uint32_t *__restrict__ bits_all, const float *__restrict__ theta,
    pqn_cnn_layout_t L, int32_t *__restrict__ action, float *__restrict__ qmax, float *__restrict__ reward,
    uint8_t *__restrict__ done, float *__restrict__ discount, float *__restrict__ rer, int32_t *__restrict__ rel,
    int32_t *__restrict__ ts, float *__restrict__ last_q, const float *__restrict__ eps_dev,
    const uint64_t *__restrict__ keys, float rscale, int store_obs, int n_per_seed, long long theta_stride, int keys_stride) {
  using P = PosCfg<C, NPL>;
  using F = PosFwdCfg<C, NPL, NW>;
  using Cfg = CnnCfg<C>;
  constexpr bool H2 = NPL == 2;
  constexpr int NT = 64 * NW, WGE = POS_ST * NW;     // threads / envs of the workgroup
  static_assert(Cfg::OW == Env::OBS_WORDS, "packed observation width");
  static_assert(NA <= 8, "q exchange buffer");
  constexpr int CONVBLK = P::CONVBLK, NCS = P::NCS;
  extern __shared__ __attribute__((aligned(16))) char pos_smem[];
  u32x4 *ring = reinterpret_cast<u32x4 *>(pos_smem);
  uint32_t *s_rows = reinterpret_cast<uint32_t *>(pos_smem + F::ring_bytes);
  u32x4 *s_cvw = reinterpret_cast<u32x4 *>(pos_smem + F::ring_bytes + F::rows_bytes);
  float *s_wc = reinterpret_cast<float *>(s_cvw + NCS * NPL * 64);
  float *s_hp = s_wc + ((CONVBLK + 3) & ~3);
  float *s_q = s_hp + ((384 + 128 * NA + NA + 3) & ~3);                              // [NW waves][32 envs][8]
  u32x4 *s_lut = reinterpret_cast<u32x4 *>(s_q + NW * POS_ST * 8);
  // (seed, block) of this workgroup, XCD-aware as in the training forward: the blocks of a seed share its W1 planes
  const int nps = n_per_seed > 0 ? n_per_seed : n, nblk = nps / WGE, nsl = gridDim.x / nblk;
  int seed_l, blk;
  {
    const int lin = blockIdx.x;
    if ((nsl & 7) == 0) {
      const int xcd = lin & 7, k = lin >> 3;
      seed_l = xcd * (nsl >> 3) + k / nblk;
      blk = k % nblk;
    } else {
      seed_l = lin / nblk;
      blk = lin % nblk;
    }
  }
  int e_off = 0;           // first env of this seed: the RNG counters use the env index inside the seed
  if (n_per_seed > 0) {
    theta += seed_l * theta_stride;
    keys += (size_t)seed_l * keys_stride;
    e_off = seed_l * n_per_seed;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g = lane >> 4;
  const int e0 = e_off + blk * WGE + POS_ST * wave;   // first env of this wave
  const int e = e0 + (lane & 31), e_rng = e - e_off;
  const bool owner = lane < POS_ST;
  const size_t bstride = (size_t)n * Cfg::OW;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  PosFwdCtx<C> cx;
  cx.wf = reinterpret_cast<const u32x4 *>(theta + L.off_w1h + (H2 ? H2_PLANES_OFF : 0));
  cx.ring = ring; cx.ring_lds = pos_lds_addr(ring); cx.s_cvw = s_cvw; cx.s_lut = s_lut;
  cx.rowsW = s_rows + wave * F::ROWW; cx.g_stat = nullptr;
  cx.lane = lane; cx.wave = wave; cx.i16 = i16; cx.g = g; cx.stamps = nullptr;
  pos_fwd_dma_step<C, NPL, NW>(cx, 0);
  if constexpr (POS_PAIR_SYNC_FWD != 0) pos_fwd_dma_step<C, NPL, NW>(cx, 1);
  for (int i = tid; i < CONVBLK; i += NT) s_wc[i] = theta[L.off_wc + i];
  for (int i = tid; i < 384; i += NT) s_hp[i] = theta[L.off_b1 + i];
  for (int i = tid; i < 128 * NA; i += NT) s_hp[384 + i] = theta[L.off_w2 + i];
  if (tid < NA) s_hp[384 + 128 * NA + tid] = theta[L.off_b2 + tid];
  pos_lut_fill(s_lut, tid, NT);
  uint32_t *rowsM = s_rows + wave * F::ROWW;         // (mutable view of cx.rowsW)
  {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(bits_all) + (size_t)e0 * P::ROWCH;   // slot 0 = the current observation
    u32x4 *rw = reinterpret_cast<u32x4 *>(rowsM);
    for (int c = lane; c < POS_ST * P::ROWCH; c += 64) {
      const int row = c / P::ROWCH, cc = c - row * P::ROWCH;
      rw[row * P::ROWSTRIDE + cc] = src[c];
    }
  }
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
  pos_dma_wait();
  __syncthreads();
  {
    const float sc = pos_fwd_consts<C, NPL>(cx, s_wc);
    if (wave == 0) pos_conv_planes<C, NPL>(s_wc, s_cvw, lane, sc);
  }
  __syncthreads();
  float *qx = s_q + wave * (POS_ST * 8);
#pragma unroll 1
  for (int t = 0; t <= t_len; ++t) {
    f32x4 zacc[2][8];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int c = 0; c < 8; ++c) zacc[tt][c] = zero4;
    pos_fwd_kloop<C, NPL, NW, false>(cx, 32 * t, zacc);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int c = 0; c < 8; c += 4) x3_drain(zacc[tt][c], zacc[tt][c + 1], zacc[tt][c + 2], zacc[tt][c + 3]);
    // ---- head forward on the accumulator layout: lane = columns o = 16 c + i16, rows = envs 16 tt + 4 g + r ----
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float zz[8], sum = 0.f, sq = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          zz[c] = H2 ? fmaf(zacc[tt][c][r], cx.zscale, s_hp[16 * c + i16]) : zacc[tt][c][r] + s_hp[16 * c + i16];
          sum += zz[c];
          sq = fmaf(zz[c], zz[c], sq);
        }
        sum = group16_sum(sum);
        sq = group16_sum(sq);
        const float mean = sum * (1.0f / QN_HID);
        const float var = fmaxf(sq * (1.0f / QN_HID) - mean * mean, 0.0f);
        const float rstd = rsqrt_exact(var + QN_LN_EPS);
        float h2[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) h2[c] = fmaxf(fmaf((zz[c] - mean) * rstd, s_hp[128 + 16 * c + i16], s_hp[256 + 16 * c + i16]), 0.0f);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          float part = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c) part = fmaf(h2[c], s_hp[384 + (16 * c + i16) * NA + a], part);
          const float qa = group16_sum(part) + s_hp[384 + 128 * NA + a];
          if (i16 == 0) qx[(16 * tt + 4 * g + r) * 8 + a] = qa;
        }
      }
    // ---- lane = env: eps-greedy (pqn_minatar.py:184-196), env.step + auto-reset + LogWrapper, the transition record ----
    if (owner) {
      int best = 0;
      float bv = qx[lane * 8];
#pragma unroll
      for (int a = 1; a < NA; ++a) {
        const float qa = qx[lane * 8 + a];
        if (qa > bv) { bv = qa; best = a; }
      }
      if (t == t_len) {
        if (last_q) last_q[e] = bv;                        // bootstrap value of obs_T
      } else {
        const uint64_t key = keys[t];
        uint32_t o0, o1;
        pqn_bits(key, (uint32_t)e_rng, PQN_STREAM_ACT, o0, o1);
        const int act = (pqn_uniform(o0) < eps) ? (int)pqn_randint(o1, (uint32_t)NA) : best;
        int dn = 0;
        const float rw = env.step(act, key, (uint32_t)e_rng, dn);
        log.step(rw, dn);
        if (dn) env.reset(key, (uint32_t)e_rng);           // gymnax auto-reset
        const size_t o = (size_t)t * n + e;
        if (action) action[o] = act;
        if (qmax) qmax[o] = bv;
        if (reward) reward[o] = rw * rscale;
        if (done) done[o] = (uint8_t)dn;
        if (discount) discount[o] = dn ? 0.0f : 1.0f;
        if (rer) rer[o] = log.ret_ret;
        if (rel) rel[o] = log.ret_len;
        if (ts) ts[o] = log.timestep;
        env.obs_bits(&rowsM[lane * (P::ROWSTRIDE * 4)]);   // obs_{t+1} straight into the wave's rows
      }
    }
    if (t == t_len) break;
    // transition record: packed obs_{t+1} of the wave's envs (the training kernels gather from it); an evaluation rollout
    // (store_obs = 0) keeps only the running observation, in slot 0
    if (store_obs || t + 1 == t_len) {
      u32x4 *dst = reinterpret_cast<u32x4 *>(bits_all + (store_obs ? (size_t)(t + 1) * bstride : (size_t)0)) + (size_t)e0 * P::ROWCH;
      const u32x4 *rw = reinterpret_cast<const u32x4 *>(rowsM);
      for (int c = lane; c < POS_ST * P::ROWCH; c += 64) {
        const int row = c / P::ROWCH, cc = c - row * P::ROWCH;
        dst[c] = rw[row * P::ROWSTRIDE + cc];
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
// host side
// ---------------------------------------------------------------------------

static unsigned long long *g_pos_stamps = nullptr;   // profiling only (PQN_T1_STAMPS=1)
extern "C" int pqn_debug_pos_stamps(unsigned long long *out /* host, 32 entries */) {
  if (!g_pos_stamps) return PQN_E_INVALID;
  if (hipMemcpy(out, g_pos_stamps, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return PQN_E_HIP;
  return PQN_OK;
}

template <int C, int NA, int NPL, int NW>
static int pos_forward_launch_m(const pqn_cnn_layout_t &L, int nb, const float *theta, float inv_b, float *wsx, const pos_ws_t &W,
                                const pqn_seeds_t &sg, int nseeds, hipStream_t st) {
  using F = PosFwdCfg<C, NPL, NW>;
  static pqn_once_per_device attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&cnn_pos_fwd_kernel<C, NA, NPL, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)F::lds_bytes(NA));
  }
  if (!g_pos_stamps && getenv("PQN_T1_STAMPS") && pqn_not_capturing(st)) {   // (profiling only; an allocation is illegal under stream capture)
    if (hipMalloc(&g_pos_stamps, 32 * sizeof(unsigned long long)) != hipSuccess) g_pos_stamps = nullptr;
  }
  hipLaunchKernelGGL((cnn_pos_fwd_kernel<C, NA, NPL, NW>), dim3((nb / (POS_ST * NW)) * nseeds), dim3(64 * NW), F::lds_bytes(NA), st, nb, theta, L,
                     inv_b, wsx, W, sg, g_pos_stamps);
  return pqn_check_launch("pqn_cnn_pos_forward");
}
template <int C, int NA>
static int pos_forward_launch(const pqn_cnn_layout_t &L, int nb, const float *theta, float inv_b, float *wsx, const pos_ws_t &W,
                              const pqn_seeds_t &sg, int nseeds, hipStream_t st, int nw) {
  if (!L.pos_f16x2) return pos_forward_launch_m<C, NA, 3, 8>(L, nb, theta, inv_b, wsx, W, sg, nseeds, st);
  if (nw == 4) return pos_forward_launch_m<C, NA, 2, 4>(L, nb, theta, inv_b, wsx, W, sg, nseeds, st);
  if (nw == 2) return pos_forward_launch_m<C, NA, 2, 2>(L, nb, theta, inv_b, wsx, W, sg, nseeds, st);
  return pos_forward_launch_m<C, NA, 2, 8>(L, nb, theta, inv_b, wsx, W, sg, nseeds, st);
}

// the (channels, actions) pairs of the MinAtar games gymnax implements
bool pqn_cnn_pos_forward_supported(int c, int a) { return (c == 4 && (a == 3 || a == 5)) || (c == 6 && a == 4) || (c == 7 && a == 3); }

int pqn_cnn_pos_forward(const pqn_cnn_layout_t &L, int nb, const float *theta, float inv_b, float *wsx, const pos_ws_t &W,
                        const pqn_seeds_t &sg, int nseeds, hipStream_t st, int nw) {
  if ((nw != 8 && nw != 4 && nw != 2) || nb % (POS_ST * nw) != 0 || (nw != 8 && !L.pos_f16x2)) {
    pqn_set_error("pqn_cnn_pos_forward: %d waves per workgroup with a minibatch of %d (8 always; 4 / 2 for f16x2 layouts)", nw, nb);
    return PQN_E_INVALID;
  }
  if (L.c == 4 && L.a == 3) return pos_forward_launch<4, 3>(L, nb, theta, inv_b, wsx, W, sg, nseeds, st, nw);
  if (L.c == 4 && L.a == 5) return pos_forward_launch<4, 5>(L, nb, theta, inv_b, wsx, W, sg, nseeds, st, nw);
  if (L.c == 6 && L.a == 4) return pos_forward_launch<6, 4>(L, nb, theta, inv_b, wsx, W, sg, nseeds, st, nw);
  if (L.c == 7 && L.a == 3) return pos_forward_launch<7, 3>(L, nb, theta, inv_b, wsx, W, sg, nseeds, st, nw);
  pqn_set_error("pqn_cnn_pos_forward: unsupported (channels, actions) = (%d, %d)", L.c, L.a);
  return PQN_E_UNSUPPORTED;
}

template <int C>
static int pos_gather_launch(int nb, const int64_t *idx, const uint32_t *bits, const int32_t *action, const float *target, float *wsx,
                             const pos_ws_t &W, const pqn_seeds_t &sg, int nseeds, hipStream_t st) {
  const int nst = nb / POS_ST;
  const int gg = pqn_opt(PQN_OPT_GATHER_GROUP);
  if (nst % 4 == 0 && (gg == 4 || (gg == 0 && (long long)(nst / 4) * nseeds >= 2048)))   // four super-tiles per workgroup while that still leaves 8 workgroups per CU
    hipLaunchKernelGGL((pos_gather_kernel<C, 4>), dim3(nst / 4, nseeds), dim3(256), 0, st, nb, idx, bits, action, target, wsx, W, sg);
  else
    hipLaunchKernelGGL((pos_gather_kernel<C, 1>), dim3(nst, nseeds), dim3(256), 0, st, nb, idx, bits, action, target, wsx, W, sg);
  return pqn_check_launch("pqn_cnn_pos_gather");
}
int pqn_cnn_pos_gather(const pqn_cnn_layout_t &L, int nb, const int64_t *idx, const uint32_t *bits, const int32_t *action,
                       const float *target, float *wsx, const pos_ws_t &W, const pqn_seeds_t &sg, int nseeds, hipStream_t st) {
  switch (L.c) {
    case 4: return pos_gather_launch<4>(nb, idx, bits, action, target, wsx, W, sg, nseeds, st);
    case 6: return pos_gather_launch<6>(nb, idx, bits, action, target, wsx, W, sg, nseeds, st);
    case 7: return pos_gather_launch<7>(nb, idx, bits, action, target, wsx, W, sg, nseeds, st);
    default: pqn_set_error("pqn_cnn_pos_gather: unsupported channel count %d", L.c); return PQN_E_UNSUPPORTED;
  }
}

template <int C, int NPL>
static int pos_backward_launch_m(const pqn_cnn_layout_t &L, int nb, int nch, const float *theta, float *wsx, float *w1out,
                                 const pos_ws_t &W, const pqn_seeds_t &sg, int nseeds, hipStream_t st) {
  using P = PosCfg<C, NPL>;
  static pqn_once_per_device attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&cnn_pos_bwd_kernel<C, NPL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)P::lds_bytes());
  }
  if (!g_pos_stamps && getenv("PQN_T1_STAMPS") && pqn_not_capturing(st)) {   // (profiling only; an allocation is illegal under stream capture)
    if (hipMalloc(&g_pos_stamps, 32 * sizeof(unsigned long long)) != hipSuccess) g_pos_stamps = nullptr;
  }
  hipLaunchKernelGGL((cnn_pos_bwd_kernel<C, NPL>), dim3(8 * nch * nseeds), dim3(POS_THREADS), P::lds_bytes(), st, nb, nch, theta, L, wsx, w1out,
                     W, sg, g_pos_stamps);
  return pqn_check_launch("pqn_cnn_pos_backward");
}
template <int C>
static int pos_backward_launch(const pqn_cnn_layout_t &L, int nb, int nch, const float *theta, float *wsx, float *w1out,
                               const pos_ws_t &W, const pqn_seeds_t &sg, int nseeds, hipStream_t st) {
  return L.pos_f16x2 ? pos_backward_launch_m<C, 2>(L, nb, nch, theta, wsx, w1out, W, sg, nseeds, st)
                     : pos_backward_launch_m<C, 3>(L, nb, nch, theta, wsx, w1out, W, sg, nseeds, st);
}

int pqn_cnn_pos_backward(const pqn_cnn_layout_t &L, int nb, int nch, const float *theta, float *wsx, float *w1out, const pos_ws_t &W,
                         const pqn_seeds_t &sg, int nseeds, hipStream_t st) {
  switch (L.c) {
    case 4: return pos_backward_launch<4>(L, nb, nch, theta, wsx, w1out, W, sg, nseeds, st);
    case 6: return pos_backward_launch<6>(L, nb, nch, theta, wsx, w1out, W, sg, nseeds, st);
    case 7: return pos_backward_launch<7>(L, nb, nch, theta, wsx, w1out, W, sg, nseeds, st);
    default: pqn_set_error("pqn_cnn_pos_backward: unsupported channel count %d", L.c); return PQN_E_UNSUPPORTED;
  }
}

template <int C, class Env, int NA, int NPL, int NW>
static int pos_rollout_launch_m(const pqn_cnn_layout_t &L, int n, int t_len, uint32_t *state, uint32_t *bits, const float *theta,
                                const pqn_step_out_t &rec, int32_t *action, float *qmax, float *last_q, const float *eps_dev,
                                const uint64_t *keys, float rscale, int store_obs, hipStream_t st, int n_per_seed,
                                long long theta_stride, int keys_stride) {
  using R = PosRollCfg<C, NA, NPL, NW>;
  static pqn_once_per_device attr;
  if (attr.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&cnn_pos_rollout_kernel<C, Env, NA, NPL, NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)R::lds_bytes);
  }
  hipLaunchKernelGGL((cnn_pos_rollout_kernel<C, Env, NA, NPL, NW>), dim3(n / (POS_ST * NW)), dim3(64 * NW), R::lds_bytes, st, n, t_len, state, bits,
                     theta, L, action, qmax, rec.reward, rec.done, rec.discount, rec.returned_episode_returns,
                     rec.returned_episode_lengths, rec.timestep, last_q, eps_dev, keys, rscale, store_obs, n_per_seed, theta_stride,
                     keys_stride);
  return pqn_check_launch("pqn_cnn_pos_rollout");
}
template <int C, class Env, int NA>
static int pos_rollout_launch(const pqn_cnn_layout_t &L, int n, int t_len, uint32_t *state, uint32_t *bits, const float *theta,
                              const pqn_step_out_t &rec, int32_t *action, float *qmax, float *last_q, const float *eps_dev,
                              const uint64_t *keys, float rscale, int store_obs, hipStream_t st, int n_per_seed,
                              long long theta_stride, int keys_stride, int nw) {
#define POS_ROLL_M(NPL_, NW_) \
  return pos_rollout_launch_m<C, Env, NA, NPL_, NW_>(L, n, t_len, state, bits, theta, rec, action, qmax, last_q, eps_dev, keys, rscale, store_obs, st, \
                                                     n_per_seed, theta_stride, keys_stride)
  if (!L.pos_f16x2) POS_ROLL_M(3, 8);
  if (nw == 4) POS_ROLL_M(2, 4);
  if (nw == 2) POS_ROLL_M(2, 2);
  POS_ROLL_M(2, 8);
#undef POS_ROLL_M
}

bool pqn_cnn_pos_rollout_supported(int env_id, int c, int a, int n, int n_per_seed, int nw) {
  const int nps = n_per_seed > 0 ? n_per_seed : n;
  if (nps <= 0 || (nw != 8 && nw != 4 && nw != 2) || nps % (POS_ST * nw) != 0 || n % nps != 0) return false;
  return (env_id == PQN_ENV_BREAKOUT && c == 4 && a == 3) || (env_id == PQN_ENV_ASTERIX && c == 4 && a == 5) ||
         (env_id == PQN_ENV_FREEWAY && c == 7 && a == 3) || (env_id == PQN_ENV_SPACEINVADERS && c == 6 && a == 4);
}

int pqn_cnn_pos_rollout(int env_id, const pqn_cnn_layout_t &L, int n, int t_len, uint32_t *state, uint32_t *bits, const float *theta,
                        const pqn_step_out_t &rec, int32_t *action, float *qmax, float *last_q, const float *eps_dev,
                        const uint64_t *keys, float rscale, int store_obs, hipStream_t st, int n_per_seed, long long theta_stride,
                        int keys_stride, int nw) {
  if (nw != 8 && !L.pos_f16x2) nw = 8;
#define POS_ROLL(CH, ENV, NACT) \
  return pos_rollout_launch<CH, ENV, NACT>(L, n, t_len, state, bits, theta, rec, action, qmax, last_q, eps_dev, keys, rscale, store_obs, st, \
                                           n_per_seed, theta_stride, keys_stride, nw)
  if (env_id == PQN_ENV_BREAKOUT && L.c == 4 && L.a == 3) POS_ROLL(4, Breakout, 3);
  if (env_id == PQN_ENV_ASTERIX && L.c == 4 && L.a == 5) POS_ROLL(4, Asterix, 5);
  if (env_id == PQN_ENV_FREEWAY && L.c == 7 && L.a == 3) POS_ROLL(7, Freeway, 3);
  if (env_id == PQN_ENV_SPACEINVADERS && L.c == 6 && L.a == 4) POS_ROLL(6, SpaceInvaders, 4);
#undef POS_ROLL
  pqn_set_error("pqn_cnn_pos_rollout: env %d / %d channels / %d actions has no position-parallel rollout", env_id, L.c, L.a);
  return PQN_E_UNSUPPORTED;
}
