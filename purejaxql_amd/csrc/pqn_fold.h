// Fold of the training step's gradient partials into the flat gradient (T3a), as a device function shared by
//   qnet_grad_reduce_kernel (pqn_qnet.hip): fold -> a->grad + per-block sums of squares, then radam_apply_kernel in a second launch
//   radam_apply_kernel<true> (pqn_algo.hip, round 6): fold + clip_by_global_norm + RAdam in ONE launch -- the folded gradient stays in
//     registers, the per-block sums of squares cross the workgroups of a seed through tagged 8-byte slots (agent-scope stores /
//     loads, no cache-wide fence, no read-modify-write), every workgroup then applies the optimizer step to the elements it folded.
// Both take the same blocks, the same loads and the same order of additions: a seed's gradient, norm and parameters are bit-identical
// between the two (tests/test_qnet_gpu.py).  The reference has one optax.chain(clip_by_global_norm, radam) update per minibatch
// (pqn_minatar.py:159-162,293-297).
#pragma once
#include "pqn_common.h"

#define PQN_FC1_ELEMS (1024 * 128)                 // fc1 kernel: 8*8*16 conv features x 128 hidden units
#define QR_W1_BLOCKS (PQN_FC1_ELEMS / 1024)        // blocks [0, QR_W1_BLOCKS): fc1 region, one float4 per lane
__host__ __device__ inline int grad_reduce_blocks(int total) { return QR_W1_BLOCKS + (total - PQN_FC1_ELEMS + 63) / 64; }

// what the fold reads: partial records / slabs of the training kernels of ONE launch (all seeds; per-seed slices at sd strides)
struct pqn_fold_args_t {
  pqn_cnn_layout_t L;
  int ntiles, nks, rec;          // small records per seed, fc1 slabs per seed, floats per small record
  const float *gpart, *wpart;    // small records, fc1 weight-gradient slabs
  float *loss_out, *qv_out;      // metrics td_loss / qvals of this minibatch (may be NULL)
  float inv_b;
  const float *gpos;             // conv-block partials of the position-parallel backward (npos records of 9C*16+48 floats), or NULL
  int npos;
  long long ws_stride, lq_stride;   // seed strides of gpart / wpart / gpos, and of loss_out / qv_out
  int valid;                     // set by launch_train when it deferred the fold to the optimizer kernel
};

// Slots of the one-launch form, inside the 1024-float optimizer scratch of a seed (partials of the two-launch form: floats [0, nparts),
// count snapshot: [1023]): 8-byte slot b = {sum of squares of block b, tag} at floats [512 + 2b, 512 + 2b + 2), launch serial at [1022].
#define PQN_FOLD_SLOT0 512
#define PQN_FOLD_SERIAL 1022
#define PQN_FOLD_MAX_BLOCKS 254

typedef float pqn_f4 __attribute__((ext_vector_type(4)));

// One block of the fold for seed `s`.  fc1 blocks (bx < QR_W1_BLOCKS): g4 = the folded float4 of lane (bx, tid).  Other blocks: wave 0's
// lane holds the folded element `i_small` (g_small; i_small < 0: nothing).  ss = this WAVE's sum of squares (lane 0 valid), exactly
// as qnet_grad_reduce_kernel summed it.  grad != NULL: the folded values are also stored (seed slice of the flat gradient).
__device__ inline void pqn_fold_block(const pqn_fold_args_t &fa, int bx, long long s, float *__restrict__ grad,
                                      float (*s_red)[64], pqn_f4 &g4, float &g_small, int &i_small, float &ss) {
  const pqn_cnn_layout_t &L = fa.L;
  const float *gpart = fa.gpart + s * fa.ws_stride;
  const float *gpos = fa.gpos ? fa.gpos + s * fa.ws_stride : nullptr;
  const float *wpart = fa.wpart + s * fa.ws_stride;
  const int ntiles = fa.ntiles, nks = fa.nks, rec = fa.rec, npos = fa.npos;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  ss = 0.0f;
  g4 = pqn_f4{0.f, 0.f, 0.f, 0.f};
  g_small = 0.0f;
  i_small = -1;
  if (bx < QR_W1_BLOCKS) {
    const int j4 = bx * 256 + threadIdx.x;  // float4 index inside the fc1 region
    pqn_f4 g = {0.f, 0.f, 0.f, 0.f};
    // 16 slab loads in flight at a time, UNCONDITIONAL (slab index clamped, the surplus masked in the add: a load under
    // a condition makes the compiler wait for the whole queue), added in slab order
    if (nks <= 2) {   // the position-parallel backward leaves one or two chunk slabs: two loads, not sixteen (the same sums: 0 + a + b)
      const pqn_f4 t0 = __builtin_nontemporal_load(reinterpret_cast<const pqn_f4 *>(wpart) + j4);
      const pqn_f4 t1 = __builtin_nontemporal_load(reinterpret_cast<const pqn_f4 *>(wpart + (size_t)(nks - 1) * PQN_FC1_ELEMS) + j4);
      g += t0;
      g += t1 * (nks > 1 ? 1.0f : 0.0f);
    } else
    for (int k0 = 0; k0 < nks; k0 += 16) {
      pqn_f4 t[16];
#pragma unroll
      for (int q = 0; q < 16; ++q)
        t[q] = __builtin_nontemporal_load(reinterpret_cast<const pqn_f4 *>(wpart + (size_t)min(k0 + q, nks - 1) * PQN_FC1_ELEMS) + j4);   // read once
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float keep = (k0 + q < nks) ? 1.0f : 0.0f;
        g += t[q] * keep;
      }
    }
    if (grad) reinterpret_cast<pqn_f4 *>(grad + L.off_w1)[j4] = g;
    g4 = g;
    ss = fmaf(g.x, g.x, fmaf(g.y, g.y, fmaf(g.z, g.z, g.w * g.w)));
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
  } else {
    // lane = element (64 consecutive non-fc1 elements per block), so that the loads of a wave run along the records
    // (256 B contiguous) instead of across them; wave w adds records w, w + 4, ... in order, 8 loads in flight
    // (unconditional: index clamped, surplus masked in the add), and the four waves are folded in fixed order.
    const int sidx = (bx - QR_W1_BLOCKS) * 64 + lane;  // index among the non-fc1 elements
    const int i = sidx < L.off_w1 ? sidx : sidx + PQN_FC1_ELEMS;
    const int convblk = 9 * L.c * 16 + 48;
    int r = -1;  // index into the small record (-1: dummy BatchNorm / padding -> zero gradient)
    if (i >= L.off_wc && i < L.off_wc + convblk) r = i - L.off_wc;
    else if (i >= L.off_b1 && i < L.off_b1 + 384) r = convblk + (i - L.off_b1);
    else if (i >= L.off_w2 && i < L.off_w2 + 128 * L.a) r = convblk + 384 + (i - L.off_w2);
    else if (i >= L.off_b2 && i < L.off_b2 + L.a) r = convblk + 384 + 128 * L.a + (i - L.off_b2);
    const bool from_pos = gpos && r >= 0 && r < convblk;
    const float *src = (from_pos ? gpos : gpart) + (r >= 0 ? r : 0);
    const int stride = from_pos ? convblk : rec, n = r < 0 ? 0 : (from_pos ? npos : ntiles);
    const int n_all = (gpos && npos > ntiles) ? npos : ntiles;   // uniform loop bound
    // 16 loads in flight per lane (round 4; 8 before: the fold of 256 records per seed was 8 dependent HBM round trips per
    // wave); the order of the additions -- records wave, wave + 4, wave + 8, ... -- is unchanged
    float g = 0.0f;
    if (n_all <= 16) {   // at most four records per wave (position-parallel form at the bench shape: 16 + 16): four loads in flight, same order
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = src[(size_t)max(min(wave + 4 * q, n - 1), 0) * stride];
#pragma unroll
      for (int q = 0; q < 4; ++q) g += v[q] * ((wave + 4 * q < n) ? 1.0f : 0.0f);
    } else
    for (int t0 = wave; t0 < n_all; t0 += 64) {
      float v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = src[(size_t)max(min(t0 + 4 * q, n - 1), 0) * stride];
#pragma unroll
      for (int q = 0; q < 16; ++q) g += v[q] * ((t0 + 4 * q < n) ? 1.0f : 0.0f);
    }
    s_red[wave][lane] = g;
    __syncthreads();
    if (wave == 0) {
      g = (s_red[0][lane] + s_red[1][lane]) + (s_red[2][lane] + s_red[3][lane]);
      if (i < L.total) {
        if (grad) grad[i] = g;
        g_small = g;
        i_small = i;
        ss = g * g;
      }
      for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
    } else ss = 0.0f;
    if (bx == QR_W1_BLOCKS && wave == 0) {  // metrics td_loss / qvals (pqn_minatar.py:334-335)
      float l = 0.f, qv = 0.f;
      for (int t = lane; t < ntiles; t += 64) {
        l += gpart[(size_t)t * rec + rec - 2];
        qv += gpart[(size_t)t * rec + rec - 1];
      }
      for (int off = 32; off > 0; off >>= 1) { l += __shfl_down(l, off, 64); qv += __shfl_down(qv, off, 64); }
      if (lane == 0) {
        if (fa.loss_out) fa.loss_out[s * fa.lq_stride] = l * fa.inv_b;
        if (fa.qv_out) fa.qv_out[s * fa.lq_stride] = qv * fa.inv_b;
      }
    }
  }
}

// pqn_algo.hip: the one-launch form.  Returns PQN_OK after enqueueing radam_apply_kernel<true> on `st`.
int pqn_launch_radam_fold(const pqn_fold_args_t &fa, float *p, float *g_or_null, float *m, float *v, int32_t *count, float lr_init,
                          float lr_end, double lr_steps, float max_norm, float *scratch, float *w1b, hipStream_t st, int nseeds,
                          long long pstride, long long sstride, long long w1bstride, int half_off, int copy_mode);
// pqn_qnet.hip: a deferred fold in a launch of its own (qnet_grad_reduce_kernel)
int pqn_cnn_fold_launch(const pqn_fold_args_t &fa, float *grad, const int32_t *count, float *scratch, int nseeds, long long theta_stride,
                        hipStream_t st);
