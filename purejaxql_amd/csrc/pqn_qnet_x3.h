// pqn_qnet_x3.h -- device helpers shared by the MinAtar CNN kernels (pqn_qnet.hip, pqn_qnet_pos.hip): tile constants,
// the exact three-way bf16 split of an f32 operand and the tied-accumulator MFMA wrappers of the bf16x3 operand mode,
// the conv-as-MFMA operand builder on packed observation bits, the fragment orders of the fc1 weight planes and the
// carve-up helpers of the training workspace.  gfx950 only.
#pragma once
#include <type_traits>

#include "pqn_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define QN_TILE 16        // samples per workgroup (= one MFMA M-tile)
#define QN_WAVES 8        // 512-thread workgroups: 2 waves per SIMD so MFMA / VALU / loads of different waves overlap
#define QN_THREADS (64 * QN_WAVES)
#define QN_SPW (QN_TILE / QN_WAVES)   // samples owned by one wave in the per-sample phases
#define QN_H1 1024        // conv features (8*8*16)
#define QN_H1S 1032       // LDS row stride of the h1 tile (floats): conflict-free ds_read_b128
#define QN_HID 128
#define QN_ZS 132         // LDS row stride of the z tile
#define QN_LN_EPS 1e-6f   // flax nn.LayerNorm default
#define QN_MAXA 8
#define QN_HP_FLOATS (384 + 128 * QN_MAXA + QN_MAXA)   // head parameters staged in LDS
#define QN_STG 20        // floats per staged point (16 + pad: conflict-free ds_read_b128 across lanes)
// bf16x3 kernels (in-place conv staging): the h1 tile keeps the quad swizzle of its staging slots -- feature i of a row
// lives at slot i ^ (((i >> 6) & 3) << 2), i.e. the four 16-B quads of position p are XORed with (p >> 2) & 3 (round 4).
// The conv phase's final ds_write_b128 of a lane's 64 B (lane = position, lane stride 64 B) was a 4-way bank conflict
// with the plain layout; swizzled it goes to the very slots the lane just read its staged values from, conflict-free.
// Readers (fc1 A fragments, the h1^T slices, the relu mask of the in-place dgrad) apply the same map.
#ifndef QN_H1_SWIZZLE
#define QN_H1_SWIZZLE 1
#endif
__device__ __forceinline__ int h1_slot(int i) { return QN_H1_SWIZZLE ? (i ^ (((i >> 6) & 3) << 2)) : i; }

template <int C>
struct CnnCfg {
  static constexpr int OW = (((100 * C + 31) / 32) + 3) / 4 * 4;  // packed obs words (16-B multiple)
  static constexpr int ROWBITS = 3 * C;                           // bits of one window row
  static constexpr int KW = 9 * C;                                // conv reduction length
};

// v_rsq_f32 (1 ulp) + one Newton step: ~1e-7 relative, a handful of VALU ops instead of the
// ~25-instruction IEEE sqrt+divide expansion.
PQN_D float rsqrt_exact(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  const float h = 0.5f * x * y;
  return fmaf(y, fmaf(-h, y, 0.5f), y);
}

// all-reduce sum over each 16-lane row with DPP row rotates (VALU speed, no LDS crossbar).
// Rotating by half the remaining period is a butterfly: every lane ends with bit-identical sums.
template <int N>
PQN_D float row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
// four such sums at once, as ONE asm block of 16 v_add_f32_dpp (the DPP rotate fused into the add).  Through
// __builtin_amdgcn_update_dpp the compiler emits v_mov_b32 0 / v_mov_b32_dpp / v_pk_add_f32 per step -- 2.5 instructions
// where one does the work -- plus the wait states between a VALU write and a DPP read of the same register; here the four
// independent chains are interleaved, so every DPP read sits three instructions behind the write it depends on (two wait
// states are required) and only the block's first instruction needs padding.  Same butterfly, same bits as group16_sum.
PQN_D void group16_sum4(float &a, float &b, float &c, float &d) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// the same four sums with the FIRST butterfly step out of place (o = a + ror8(a)): the inputs stay intact, no copies (round 6)
PQN_D void group16_sum4_from(float &o0, float &o1, float &o2, float &o3, const float a, const float b, const float c, const float d) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %2, %2 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %3, %3 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %2, %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "v_add_f32_dpp %3, %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1"
               : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
               : "v"(a), "v"(b), "v"(c), "v"(d));
}
PQN_D float group16_sum(float v) {
  v += row_ror<8>(v);
  v += row_ror<4>(v);
  v += row_ror<2>(v);
  v += row_ror<1>(v);
  return v;
}

// ---------------------------------------------------------------------------
// bf16x3 split-operand products (pqn_cnn_layout_t.matmul_f16 == 2, config MATMUL_DTYPE: bf16x3).
// gfx950 has no tf32/xf32 path and its f32-input MFMA runs at the vector rate (1/16 of the bf16 rate), so an
// f32 x f32 product is evaluated on the bf16 matrix core from EXACT three-way splits: x = hi + mid + lo with
// hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (8 + 8 + 8 significand bits; both subtractions are
// exact in f32), and a*b ~= ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh -- the three dropped terms are
// <= 2^-23 |a b|, i.e. f32 rounding level; every partial product is exact in the f32 accumulator's input and
// the accumulation is f32.  6 x v_mfma_f32_16x16x32_bf16 (K = 32) replace 8 x v_mfma_f32_16x16x4_f32: ~5x
// fewer matrix-pipe cycles.  Measured error vs an f64 dot product (K = 1024, tools/ubench/bf16x3.hip):
// 1.7e-7 * sum|a b|, against 0.9e-7 for the f32 fma chain.
// The split is 9 VALU ops per pair of values (v_cvt_pk_bf16_f32, shift / mask back to f32, v_pk_add_f32).
// ---------------------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

PQN_D void x3_split2(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
  const f32x2 x = {x0, x1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
  const f32x2 hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xFFFF0000u)};
  const f32x2 r1 = x - hf;
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_t));
  const f32x2 mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xFFFF0000u)};
  const f32x2 r2 = r1 - mf;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
}

// the same split in three stages (4 + 4 + 1 VALU ops), for kernels that interleave it with MFMAs by hand
struct X3Split {
  f32x2 x, r;
  unsigned h, m, l;
};
PQN_D void x3_stage1(X3Split &s) {
  s.h = __builtin_bit_cast(unsigned, __builtin_convertvector(s.x, bf16x2_t));
  const f32x2 hf = {__uint_as_float(s.h << 16), __uint_as_float(s.h & 0xFFFF0000u)};
  s.r = s.x - hf;
}
PQN_D void x3_stage2(X3Split &s) {
  s.m = __builtin_bit_cast(unsigned, __builtin_convertvector(s.r, bf16x2_t));
  const f32x2 mf = {__uint_as_float(s.m << 16), __uint_as_float(s.m & 0xFFFF0000u)};
  s.r = s.r - mf;
}
PQN_D void x3_stage3(X3Split &s) { s.l = __builtin_bit_cast(unsigned, __builtin_convertvector(s.r, bf16x2_t)); }

// one MFMA operand fragment (8 k-values per lane) as three bf16 planes
struct X3Frag {
  u32x4 h, m, l;
};
// k-values 0..3 = a, 4..7 = b (the two float4 halves a lane holds of a 32-wide K step)
PQN_D X3Frag x3_split8(const f32x4 a, const f32x4 b) {
  unsigned h[4], m[4], l[4];
  x3_split2(a.x, a.y, h[0], m[0], l[0]);
  x3_split2(a.z, a.w, h[1], m[1], l[1]);
  x3_split2(b.x, b.y, h[2], m[2], l[2]);
  x3_split2(b.z, b.w, h[3], m[3], l[3]);
  X3Frag f;
  f.h = u32x4{h[0], h[1], h[2], h[3]};
  f.m = u32x4{m[0], m[1], m[2], m[3]};
  f.l = u32x4{l[0], l[1], l[2], l[3]};
  return f;
}
// v_mfma_f32_16x16x32_bf16 with the accumulator TIED (D and C the same register tuple), as VOLATILE inline asm.
// Why not the builtin, and why volatile: with ROCm 7.2 the builtin form of this instruction produced run-to-run
// DIFFERENT results in the conv phase of the training kernel (only in builds whose assembly gave the MFMA a destination
// tuple partially overlapping its accumulator input -- tools/check_mfma_overlap.py scans for that), and so did the
// tied but NON-volatile asm once the compiler was free to re-order it (10-channel / 7-channel / 6-channel kernels,
// conv phase only; the 4-channel kernel never showed it).  The hardware hazards one might suspect were measured and
// ruled out (tools/ubench/mfma_war.hip: overwriting A / B right behind the MFMA is safe, a VALU-written operand needs
// ONE wait state, dependent chains at any distance are exact), so the root cause is left as "compiler scheduling of
// this new instruction"; what is relied on is what was verified: source-order issue (volatile), tied accumulators,
// the pads below, and tests/test_qnet_gpu.py::test_bf16x3_is_deterministic_and_matches_f32_mode over all channel counts.
// Inline asm carries its own wait states (the compiler pads nothing inside the string):
//   - `s_nop 1` ahead of the MFMA covers a VALU write of an operand in the preceding issue slots (1 state needed);
//   - the result is consumed only by the next tied MFMA of the chain (no wait states needed) or after x3_drain*.
PQN_D f32x4 x3_mfma_tied(const u32x4 &a, const u32x4 &b, f32x4 c) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
#define X3_MFMA(A, B, C) x3_mfma_tied(A, B, C)
// GROUPS of independent MFMAs (different accumulators) as ONE asm statement with ONE leading `s_nop 1` (round 4).  The pad
// in front of a single MFMA covers a VALU write of one of its operands in the preceding issue slots -- the compiler cannot
// see inside the asm string and schedules its own VALU instructions between the statements -- but inside a run of MFMAs
// it is pure cost: MI355X_MICROARCH.md prices one extra issue state between MFMAs at ~6 cycles (different accumulators)
// against the ~16 cycles the MFMA itself occupies the pipe.  Inside a group nothing can be scheduled between the MFMAs, so
// only the first needs the pad; instruction order, operands and therefore results are unchanged.
PQN_D void x3_grp2(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\t"
               "v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1"
               : "+v"(c0), "+v"(c1)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
PQN_D void x3_grp4(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1, f32x4 &c2,
                   const u32x4 &a2, const u32x4 &b2, f32x4 &c3, const u32x4 &a3, const u32x4 &b3) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\t"
               "v_mfma_f32_16x16x32_bf16 %1, %6, %7, %1\n\t"
               "v_mfma_f32_16x16x32_bf16 %2, %8, %9, %2\n\t"
               "v_mfma_f32_16x16x32_bf16 %3, %10, %11, %3"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3));
}
PQN_D void x3_grp6(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1, f32x4 &c2,
                   const u32x4 &a2, const u32x4 &b2, f32x4 &c3, const u32x4 &a3, const u32x4 &b3, f32x4 &c4, const u32x4 &a4,
                   const u32x4 &b4, f32x4 &c5, const u32x4 &a5, const u32x4 &b5) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_bf16 %0, %6, %7, %0\n\t"
               "v_mfma_f32_16x16x32_bf16 %1, %8, %9, %1\n\t"
               "v_mfma_f32_16x16x32_bf16 %2, %10, %11, %2\n\t"
               "v_mfma_f32_16x16x32_bf16 %3, %12, %13, %3\n\t"
               "v_mfma_f32_16x16x32_bf16 %4, %14, %15, %4\n\t"
               "v_mfma_f32_16x16x32_bf16 %5, %16, %17, %5"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5));
}
// first MFMA of a chain: C = 0 as the inline constant instead of a zeroed register tuple (round 6: four v_mov per accumulator less;
// 0 + a b is what the zeroed tuple gave, bit for bit)
PQN_D void x3_grp4_zero(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1, f32x4 &c2,
                        const u32x4 &a2, const u32x4 &b2, f32x4 &c3, const u32x4 &a3, const u32x4 &b3) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_bf16 %0, %4, %5, 0\n\t"
               "v_mfma_f32_16x16x32_bf16 %1, %6, %7, 0\n\t"
               "v_mfma_f32_16x16x32_bf16 %2, %8, %9, 0\n\t"
               "v_mfma_f32_16x16x32_bf16 %3, %10, %11, 0"
               : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3));
}
PQN_D void x3_grp2_zero(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_bf16 %0, %2, %3, 0\n\t"
               "v_mfma_f32_16x16x32_bf16 %1, %4, %5, 0"
               : "=&v"(c0), "=&v"(c1)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
PQN_D void x3_grp3(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1, f32x4 &c2, const u32x4 &a2,
                   const u32x4 &b2) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n\t"
               "v_mfma_f32_16x16x32_bf16 %1, %5, %6, %1\n\t"
               "v_mfma_f32_16x16x32_bf16 %2, %7, %8, %2"
               : "+v"(c0), "+v"(c1), "+v"(c2)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2));
}
// N MFMAs sharing the B operand (N row blocks against one fragment): grouped for the N the kernels use
template <int N>
PQN_D void x3_grp_sameb(f32x4 (&c)[N], const u32x4 (&a)[N], const u32x4 &b) {
  if constexpr (N == 2) x3_grp2(c[0], a[0], b, c[1], a[1], b);
  else if constexpr (N == 3) x3_grp3(c[0], a[0], b, c[1], a[1], b, c[2], a[2], b);
  else if constexpr (N == 4) x3_grp4(c[0], a[0], b, c[1], a[1], b, c[2], a[2], b, c[3], a[3], b);
  else if constexpr (N == 6) x3_grp6(c[0], a[0], b, c[1], a[1], b, c[2], a[2], b, c[3], a[3], b, c[4], a[4], b, c[5], a[5], b);
  else {
#pragma unroll
    for (int j = 0; j < N; ++j) c[j] = x3_mfma_tied(a[j], b, c[j]);
  }
}
// end of an accumulation chain: 16 wait states (an MFMA result may not be read by anything but a tied MFMA earlier)
PQN_D void x3_drain(f32x4 &a) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a)); }
PQN_D void x3_drain(f32x4 &a, f32x4 &b) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b)); }
PQN_D void x3_drain(f32x4 &a, f32x4 &b, f32x4 &c) { asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c)); }
PQN_D void x3_drain(f32x4 &a, f32x4 &b, f32x4 &c, f32x4 &d) {
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// acc_s takes the three small cross terms, acc_b the three leading ones: two independent dependency chains, and the
// small terms are summed among themselves before they meet the large ones
PQN_D void x3_mfma6(const X3Frag &a, const X3Frag &b, f32x4 &acc_b, f32x4 &acc_s) {
  acc_s = X3_MFMA(a.l, b.h, acc_s);
  acc_b = X3_MFMA(a.m, b.h, acc_b);
  acc_s = X3_MFMA(a.h, b.l, acc_s);
  acc_b = X3_MFMA(a.h, b.m, acc_b);
  acc_s = X3_MFMA(a.m, b.m, acc_s);
  acc_b = X3_MFMA(a.h, b.h, acc_b);
}

// ---------------------------------------------------------------------------
// f16x2 split-operand products (round 6; pqn_cnn_layout_t.pos_f16x2, config MATMUL_DTYPE: f16x2 -- position-parallel kernels only).
// The same idea on fp16 pieces: s x = hi + lo with hi = f16(s x), lo = f16(s x - hi) -- 11 + 11 significand bits, the subtraction
// exact in f32 -- and a b ~= (ah bl + al bh) + ah bh: THREE v_mfma_f32_16x16x32_f16 per 32-wide K step instead of six bf16 ones,
// five VALU instructions per pair of values for the split instead of nine.  What is given up: the representation keeps 22 bits
// (error <= 2^-22 |x| per operand, the dropped al bl <= 2^-22 |a b|) where bf16x3 keeps 24; measured against float64 on the whole
// learn phase the mode stays inside the f32 fma-chain kernels' own distance (profiles/r06_v7_f16x2_accuracy.txt).
// fp16 has 5 exponent bits, so every operand is brought into range by an EXACT power-of-two scale s chosen from a bound that holds by
// construction -- nothing can overflow, and what falls below fp16's normal range is below 2^-40 of the operand's largest element:
//   fc1 kernel W1          s = 2^7 (static; |w| < 511 -- beyond that fp16 overflows to inf and the loss goes NaN, loudly)
//   conv kernel            s from max |w| of the kernel, per workgroup (h2_pow2_below)
//   h1 (LayerNorm_0, relu) s from 4 max|scale| + max|bias| of LayerNorm_0 (|xhat| <= sqrt(15) < 4), folded into scale / bias
//   dz                     per SAMPLE, from the row's max |dz| (forward kernel); the backward undoes it per row in the input gradient
//                          and carries the row factors into the h1 operand of the weight gradient, normalised by the chunk's largest
// ---------------------------------------------------------------------------
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
// (H2_W_SCALE = 2^H2_W_SHIFT = 128, the fc1 planes' scale, and the plane writer live in pqn_common.h: the optimizer kernel keeps them)
PQN_D void h2_split2(float x0, float x1, unsigned &h, unsigned &l) {
  const f32x2 x = {x0, x1};
  const f16x2_t hh = __builtin_convertvector(x, f16x2_t);          // v_cvt_pk_f16_f32 (round to nearest even)
  h = __builtin_bit_cast(unsigned, hh);
  const f32x2 r = x - __builtin_convertvector(hh, f32x2);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}
struct H2Frag {
  u32x4 h, l;
};
PQN_D H2Frag h2_split8(const f32x4 a, const f32x4 b) {   // (already scaled) k-values 0..3 = a, 4..7 = b
  unsigned h[4], l[4];
  h2_split2(a.x, a.y, h[0], l[0]);
  h2_split2(a.z, a.w, h[1], l[1]);
  h2_split2(b.x, b.y, h[2], l[2]);
  h2_split2(b.z, b.w, h[3], l[3]);
  H2Frag f;
  f.h = u32x4{h[0], h[1], h[2], h[3]};
  f.l = u32x4{l[0], l[1], l[2], l[3]};
  return f;
}
// 2^(15 - e) for bound = m 2^e, m in [0.5, 1): bound * result < 2^15 < 65504, whatever the bound (exponent clamped to +-60)
PQN_D float h2_pow2_below(float bound) {
  int e = __builtin_amdgcn_frexp_expf(bound);
  e = e < -45 ? -45 : (e > 75 ? 75 : e);
  return __int_as_float((127 + 15 - e) << 23);
}
// the f16 twins of the grouped MFMA statements above (same operand layout, same pads)
PQN_D void h2_grp2(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\t"
               "v_mfma_f32_16x16x32_f16 %1, %4, %5, %1"
               : "+v"(c0), "+v"(c1)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
PQN_D void h2_grp2_zero(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_f16 %0, %2, %3, 0\n\t"
               "v_mfma_f32_16x16x32_f16 %1, %4, %5, 0"
               : "=&v"(c0), "=&v"(c1)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
PQN_D void h2_grp4(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1, f32x4 &c2,
                   const u32x4 &a2, const u32x4 &b2, f32x4 &c3, const u32x4 &a3, const u32x4 &b3) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_f16 %0, %4, %5, %0\n\t"
               "v_mfma_f32_16x16x32_f16 %1, %6, %7, %1\n\t"
               "v_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n\t"
               "v_mfma_f32_16x16x32_f16 %3, %10, %11, %3"
               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3));
}
PQN_D void h2_grp4_zero(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1, f32x4 &c2,
                        const u32x4 &a2, const u32x4 &b2, f32x4 &c3, const u32x4 &a3, const u32x4 &b3) {
  asm volatile("s_nop 1\n\t"
               "v_mfma_f32_16x16x32_f16 %0, %4, %5, 0\n\t"
               "v_mfma_f32_16x16x32_f16 %1, %6, %7, 0\n\t"
               "v_mfma_f32_16x16x32_f16 %2, %8, %9, 0\n\t"
               "v_mfma_f32_16x16x32_f16 %3, %10, %11, 0"
               : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3)
               : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3));
}
// operand mode of the position-parallel kernels as a type: NPL = planes per operand (3: bf16x3, 2: f16x2)
template <int NPL>
struct PosMM {
  static constexpr bool H2 = NPL == 2;
  static PQN_D void grp2(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1) {
    if constexpr (H2) h2_grp2(c0, a0, b0, c1, a1, b1);
    else x3_grp2(c0, a0, b0, c1, a1, b1);
  }
  static PQN_D void grp2_zero(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1) {
    if constexpr (H2) h2_grp2_zero(c0, a0, b0, c1, a1, b1);
    else x3_grp2_zero(c0, a0, b0, c1, a1, b1);
  }
  static PQN_D void grp4(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1, f32x4 &c2,
                         const u32x4 &a2, const u32x4 &b2, f32x4 &c3, const u32x4 &a3, const u32x4 &b3) {
    if constexpr (H2) h2_grp4(c0, a0, b0, c1, a1, b1, c2, a2, b2, c3, a3, b3);
    else x3_grp4(c0, a0, b0, c1, a1, b1, c2, a2, b2, c3, a3, b3);
  }
  static PQN_D void grp4_zero(f32x4 &c0, const u32x4 &a0, const u32x4 &b0, f32x4 &c1, const u32x4 &a1, const u32x4 &b1, f32x4 &c2,
                              const u32x4 &a2, const u32x4 &b2, f32x4 &c3, const u32x4 &a3, const u32x4 &b3) {
    if constexpr (H2) h2_grp4_zero(c0, a0, b0, c1, a1, b1, c2, a2, b2, c3, a3, b3);
    else x3_grp4_zero(c0, a0, b0, c1, a1, b1, c2, a2, b2, c3, a3, b3);
  }
};

// The same conv on the bf16 matrix core (operand mode 2).  The observation bits are exact in bf16, so the A operand is
// ONE plane; the kernel Wc is split exactly into three bf16 planes (x3_split2) when the workgroup starts, and
//   out = sum_k bit_k * (Wh + Wm + Wl)[k]   -- 3 x v_mfma_f32_16x16x32_bf16 per 32 window bits, f32 accumulate
// has no rounding beyond the f32 accumulation itself (the f32-operand path rounds bit/255 * w per product instead).
// K slot (kq = lane>>4, j) of step s stands for window element k = 32 s + 8 kq + j.  A "set" bit is written as the
// bf16 value 2.0 (0x4000: a single bit, so a pair of slots is two shifts and two masks); the 1/255 input scale and
// the 1/2 are applied once to the accumulator.
template <int C>
struct ConvX3 {
  static constexpr int NK = 9 * C, NS = (NK + 31) / 32, RB = 3 * C;
  static constexpr float OUT_SCALE = 0.5f / 255.0f;
  X3Frag w[NS];
  PQN_D void init(const float *wc, int lane) {
    const int kq = lane >> 4, o = lane & 15;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 32 * s + 8 * kq + j;
        v[j] = (k < NK) ? wc[k * 16 + o] : 0.0f;
      }
      w[s] = x3_split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
    }
  }
  // bits [32 s, 32 s + 32) of the 9C-bit window string m0 | m1 << RB | m2 << 2 RB (compile-time shifts)
  static PQN_D uint32_t word(const uint32_t (&m)[3], int s) {
    uint32_t wv = 0u;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int off = r * RB - 32 * s;
      if (off >= 0 && off < 32) wv |= m[r] << off;
      else if (off < 0 && off > -32) wv |= m[r] >> (-off);
    }
    return wv;
  }
  // 8 bits -> 8 bf16 slots (2.0 or 0): y carries bit 2jj at 2jj and bit 2jj+1 at 2jj+16; one shift + mask per pair
  static PQN_D u32x4 expand8(uint32_t byte) {
    const uint32_t y = (byte << 15) | byte;
    return u32x4{(y << 14) & 0x40004000u, (y << 12) & 0x40004000u, (y << 10) & 0x40004000u, (y << 8) & 0x40004000u};
  }
  // the four 16-point tiles of one sample at once: eight independent accumulators ({h plane, m + l planes} per tile),
  // reuse distance >= 4 in issue order (a dependent MFMA issues ~90 counter ticks after its producer)
  PQN_D void tile4(const uint32_t *wm, int i, int kq, f32x4 (&d)[4]) const {
    u32x4 fa[4][NS];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int p = 16 * t + i;
      const uint32_t m[3] = {wm[p * 3], wm[p * 3 + 1], wm[p * 3 + 2]};
#pragma unroll
      for (int s = 0; s < NS; ++s) fa[t][s] = expand8(__builtin_amdgcn_ubfe(word(m, s), 8u * kq, 8u));
    }
    f32x4 ab[4], as[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { ab[t] = f32x4{0.f, 0.f, 0.f, 0.f}; as[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      x3_grp4(as[0], fa[0][s], w[s].l, as[1], fa[1][s], w[s].l, as[2], fa[2][s], w[s].l, as[3], fa[3][s], w[s].l);
      x3_grp4(ab[0], fa[0][s], w[s].h, ab[1], fa[1][s], w[s].h, ab[2], fa[2][s], w[s].h, ab[3], fa[3][s], w[s].h);
      x3_grp4(as[0], fa[0][s], w[s].m, as[1], fa[1][s], w[s].m, as[2], fa[2][s], w[s].m, as[3], fa[3][s], w[s].m);
    }
    x3_drain(ab[0], ab[1], ab[2], ab[3]);
    x3_drain(as[0], as[1], as[2], as[3]);
#pragma unroll
    for (int t = 0; t < 4; ++t) d[t] = (ab[t] + as[t]) * OUT_SCALE;
  }
};

// Weight operands of the bf16x3 mode: the fc1 kernel's three bf16 planes, kept in the tail of the parameter buffer
// (pqn_cnn_layout_t.off_w1h, 6 x 131072 bf16 = 393216 floats) by the optimizer kernel, in the two fragment orders the
// kernels stream:
//   forward  Wf[p][s][cb][lane][8]    lane = kk*16 + o%16, slot j: i = 32 s + 16 (j>>2) + 4 kk + (j&3), o = 16 cb + o%16
//   dgrad    Wd[p][ib][sK][lane][8]   lane = kd*16 + i%16, slot j: o = 32 sK + 16 (j>>2) + 4 kd + (j&3), i = 16 ib + i%16
// (one dwordx4 per lane = the 8 k-values of a 32-wide MFMA step; 1 KB per wave-instruction).  Splitting the weights
// once per optimizer step instead of once per use leaves the kernels with the A-operand split only.
#define X3_PLANE (QN_H1 * QN_HID)                 // bf16 elements per plane
PQN_HD int x3_fwd_index(int i, int o) {           // element offset inside one forward plane
  const int s = i >> 5, h = (i >> 4) & 1, kk = (i >> 2) & 3, sx = i & 3;
  return ((((s * 8 + (o >> 4)) * 64) + kk * 16 + (o & 15)) << 3) + 4 * h + sx;
}
PQN_HD int x3_dgrad_index(int i, int o) {         // element offset inside one dgrad plane
  const int sK = o >> 5, h = (o >> 4) & 1, kd = (o >> 2) & 3, sx = o & 3;
  return (((((i >> 4) * 4 + sK) * 64) + kd * 16 + (i & 15)) << 3) + 4 * h + sx;
}

#define QW_SLAB 256
// leading dimension of the transposed operands: nb + 32 floats, so consecutive rows (16 KB apart at
// nb = 4096) do not all land on the same L2 channel
__host__ __device__ inline int qw_ld(int nb) { return nb + 32; }
// columns reserved per h1 feature row in the workspace: the sample-major layout needs nb + 32, the slab-major layout
// of the bf16x3 mode (h1s_index) whole 256-sample slabs
__host__ __device__ inline int qw_h1_cols(int nb) { return max(nb + 32, (nb + QW_SLAB - 1) / QW_SLAB * QW_SLAB); }
// bf16x3 mode: h1 is handed from T1 to T2 slab-major, [slab ks][row block it][step u][64 rows][32 samples] -- the tile
// a T2 workgroup consumes per step is 8 KB contiguous, a workgroup's whole stream 64 KB x G contiguous (the sample-major
// layout made every 128-B line of that stream a different DRAM page: ~3.6 TB/s with no compute at all)
__host__ __device__ inline size_t h1s_index(int i, int b) {
  return ((((size_t)(b >> 8) * 16 + (i >> 6)) * 8 + ((b >> 5) & 7)) * 64 + (i & 63)) * 32 + (b & 31);
}

__host__ __device__ inline int small_record_floats(int c, int a) { return 9 * c * 16 + 48 + 384 + 128 * a + a + 2; }
// bf16x3 mode: dz is handed from T1 to T2 as three bf16 planes, already split, in the B-fragment order T2's MFMAs read:
//   dzw[plane][slab ks][column block cb][step u][lane = kg * 16 + (o & 15)][8]  -- slot j of the 8 = sample
//   256 ks + 32 u + 16 (j >> 2) + 4 kg + (j & 3), output o = 16 cb + (o & 15); one dwordx4 per lane = one operand.
// Plane stride = slabs * 256 * 128 elements.  The region sits behind the split-K slabs of the workspace; every kernel
// derives it from the dz^T pointer (workspace carve-up of launch_train).
__host__ __device__ inline int qw_slabs(int nb) { return (nb + 255) / 256; }
__host__ __device__ inline size_t qw_dzw_offset(int nb, int c, int a) {   // floats from dzT to the planes
  return (size_t)QN_HID * qw_ld(nb) + (size_t)QN_H1 * qw_h1_cols(nb) + (size_t)(nb / QN_TILE) * small_record_floats(c, a) +
         (size_t)qw_slabs(nb) * QN_H1 * QN_HID;
}
__host__ __device__ inline size_t qw_dzw_floats(int nb) { return (size_t)qw_slabs(nb) * 256 * QN_HID * 3 / 2; }
PQN_HD size_t dzw_index(int b, int o) {   // element offset of (sample b, output o) inside one plane
  const int ks = b >> 8, u = (b >> 5) & 7, hf = (b >> 4) & 1, kg = (b >> 2) & 3;
  return (((((size_t)ks * 8 + (o >> 4)) * 8 + u) * 64 + kg * 16 + (o & 15)) << 3) + 4 * hf + (b & 3);
}

// dz as bf16 planes for the position-parallel backward (element offsets in bf16 units inside one plane set):
//   dzA[plane][sample][sK][kq][8]: the 8 values are outputs 32 sK + 16 (j >> 2) + 4 kq + (j & 3) -- one dwordx4 per lane is
//       the A fragment (row = sample) of a K = 32 dgrad step, in the K-slot order of the optimizer's dgrad planes;
//   (the weight gradient's B fragment -- column = output, K slots = samples -- is read from the same image by the backward with
//   transposing LDS reads, pqn_qnet_pos.hip.)  Plane stride: nb * 128 elements.
PQN_HD size_t dz_planes_a(int nb, int sample, int sK, int kq) { (void)nb; return ((size_t)sample * 16 + sK * 4 + kq) * 8; }

