// pqn_qnet_pos.h -- interface between pqn_qnet.hip (kernel selection, workspace, reduction) and pqn_qnet_pos.hip (the
// position-parallel training kernels).  Internal to libpqn_hip.so.
#pragma once
#include "pqn_common.h"

// words of the bit-transposed observation table of one 32-sample super-tile: 100 C bit words + the zero word the padding
// rows of the conv weight gradient read, rounded up to whole 1-KB LDS-DMA instructions
__host__ __device__ constexpr int pos_t32_words(int c) { return (100 * c + 1 + 255) / 256 * 256; }

// carve-up of the h1^T region of a seed's training workspace in the position-parallel form (float offsets from the region's
// start; every offset a multiple of 4 floats = 16 B).  dz comes first: the DMA plan addresses everything relative to it.
#define POS_MAX_CHUNKS 8       // sample chunks of the backward (one partial dW1 slab each)
struct pos_ws_t {
  long long dz;        // dz as three bf16 planes [3][nb][128] in the dgrad's operand order (dz_planes_a)
  long long mb_bits;   // [nb][OW] packed observation rows in minibatch order
  long long t32;       // [nb / 32][pos_t32_words] bit-transposed rows
  long long stats;     // [nb / 32][64 positions][2][32 samples]: LayerNorm_0 mean | 1/std (forward kernel; round 5: [..][32][2])
  long long gpos;      // [8 nch][9C*16 + 48] conv-block records of the backward workgroups (nch <= POS_MAX_CHUNKS)
  long long act;       // [nb] i32 (minibatch order)
  long long tgt;       // [nb] f32
  long long recs;      // [nb / (32 NW)][record] head-block records of the forward workgroups
  long long end;       // floats used
};
inline pos_ws_t pos_ws_layout(int nb, int c, int a) {
  auto al = [](long long x) { return (x + 3) & ~3ll; };
  const int ow = ((((100 * c + 31) / 32) + 3) / 4) * 4;
  const long long rec = 9 * c * 16 + 48 + 384 + 128 * a + a + 2;
  pos_ws_t w;
  w.dz = 0;
  w.mb_bits = al((long long)nb * 192);
  w.t32 = al(w.mb_bits + (long long)nb * ow);
  w.stats = al(w.t32 + (long long)(nb / 32) * pos_t32_words(c));
  w.gpos = al(w.stats + (long long)nb * 128);
  w.act = al(w.gpos + 8ll * POS_MAX_CHUNKS * (9 * c * 16 + 48));
  w.tgt = al(w.act + nb);
  w.recs = al(w.tgt + nb);
  w.end = al(w.recs + (long long)((nb + 63) / 64) * rec);     // one record per forward workgroup: 256 samples, or 128 / 64 (f16x2, round 6)
  return w;
}

// the epoch region (round 6): the gather outputs of ALL minibatches of an epoch, each array [num_minibatches][per minibatch] so that
// minibatch mb's slice is contiguous (float offsets from the region's start, multiples of 4)
struct pos_epoch_t {
  long long bits, t32, act, tgt, end;
};
inline pos_epoch_t pos_epoch_layout(int nb, int nmb, int c) {
  auto al = [](long long x) { return (x + 3) & ~3ll; };
  const int ow = ((((100 * c + 31) / 32) + 3) / 4) * 4;
  pos_epoch_t e;
  e.bits = 0;
  e.t32 = al(e.bits + (long long)nmb * nb * ow);
  e.act = al(e.t32 + (long long)nmb * (nb / 32) * pos_t32_words(c));
  e.tgt = al(e.act + (long long)nmb * nb);
  e.end = al(e.tgt + (long long)nmb * nb);
  return e;
}

// number of sample chunks of the backward (one partial dW1 slab each): a function of the minibatch size alone, so that a
// seed's summation order does not depend on how many seeds share the launch
inline int pos_chunks(int nb) { return nb >= 1024 ? 2 : 1; }
inline bool pos_shape_ok(int nb) { return nb % (64 * pos_chunks(nb)) == 0; }
// Round 6 (f16x2 layouts, not under pin_form): launches that would leave CUs idle get finer workgroups -- `nw` waves (32 samples each) per
// forward / rollout workgroup instead of 8, more sample chunks in the backward -- chosen from the launch so that both kernels put >= 160
// workgroups on the chip.  nw = 0: the launch does not fill the chip even with the finest cut (the form is not taken in auto mode).
struct pos_plan_t {
  int nw, nch;
};
inline pos_plan_t pos_plan(bool f16x2, int nb, int nseeds, bool pinned) {
  pos_plan_t p = {0, pos_chunks(nb)};
  if (nb % 256 == 0 && (nb / 256) * nseeds >= 160) p.nw = 8;
  else if (f16x2 && !pinned) {
    if (nb % 128 == 0 && (nb / 128) * nseeds >= 160) p.nw = 4;
    else if (nb % 64 == 0 && (nb / 64) * nseeds >= 160) p.nw = 2;
  }
  if (f16x2 && !pinned && p.nw != 0)
    while (8 * p.nch * nseeds < 160 && p.nch < POS_MAX_CHUNKS && nb % (128 * p.nch) == 0 && nb / (64 * p.nch) >= 4 && 2 * p.nch <= (nb + 255) / 256) p.nch *= 2;
  return p;
}

// The three launches of the position-parallel form.  wsx = seed 0's h1^T region (carved up by pos_ws_layout), w1out = seed 0's
// split-K slab region (nch slabs are written); sg.seed_base / nseeds select the seeds of this launch.
//   gather    minibatch rows, actions, targets in minibatch order + the bit-transpose per super-tile
//   forward   conv .. loss and the head's backward: dz planes, LayerNorm_0 statistics, one head record per 256 samples
//             (nb % 256 == 0; (channels, actions) as pqn_cnn_pos_forward_supported says)
//   backward  dgrad, LayerNorm_0 / conv backward, dW1 rows in registers (dz planes and LayerNorm_0 statistics of `forward`)
int pqn_cnn_pos_gather(const pqn_cnn_layout_t &L, int nb, const int64_t *idx, const uint32_t *bits, const int32_t *action,
                       const float *target, float *wsx, const pos_ws_t &W, const pqn_seeds_t &sg, int nseeds, hipStream_t st);
bool pqn_cnn_pos_forward_supported(int c, int a);
int pqn_cnn_pos_forward(const pqn_cnn_layout_t &L, int nb, const float *theta, float inv_b, float *wsx, const pos_ws_t &W,
                        const pqn_seeds_t &sg, int nseeds, hipStream_t st, int nw = 8);
int pqn_cnn_pos_backward(const pqn_cnn_layout_t &L, int nb, int nch, const float *theta, float *wsx, float *w1out, const pos_ws_t &W,
                         const pqn_seeds_t &sg, int nseeds, hipStream_t st);

// persistent rollout in the structure of the forward kernel (one workgroup per 256 envs, wave = 32 envs): the scan of
// pqn_qnet_cnn_rollout for launches whose envs (per seed) come in multiples of 256; same arguments
bool pqn_cnn_pos_rollout_supported(int env_id, int c, int a, int n, int n_per_seed, int nw = 8);
int pqn_cnn_pos_rollout(int env_id, const pqn_cnn_layout_t &L, int n, int t_len, uint32_t *state, uint32_t *bits, const float *theta,
                        const pqn_step_out_t &rec, int32_t *action, float *qmax, float *last_q, const float *eps_dev,
                        const uint64_t *keys, float rscale, int store_obs, hipStream_t st, int n_per_seed, long long theta_stride,
                        int keys_stride, int nw = 8);
