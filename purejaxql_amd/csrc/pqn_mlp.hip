// pqn_mlp.hip -- fused MLP Q-network kernels (gfx950, f32) for the gymnax classic-control path.
//
// Network = QNetwork of purejaxql/pqn_gymnax.py:29-58 with NORM_TYPE=layer_norm, NORM_INPUT=False:
//   x -> [Dense(H) -> LayerNorm(H) -> relu] x NUM_LAYERS -> Dense(A)          (+ the dummy input BatchNorm
//   whose 2*D parameters exist and never receive gradient, :38-42)
// One 256-thread workgroup owns a tile of 16 samples (= one MFMA M-tile); activations live in LDS, weights
// (flax (in,out) layout, out contiguous) stream from L2.  Hidden x hidden products (forward, input gradient
// on the transposed copy, weight gradient over the 16 samples) run as v_mfma_f32_16x16x4_f32; the narrow
// input layer (CartPole: 4 features) and the LayerNorm / head work stay on the VALU.  Per-tile partial
// gradients are folded in fixed order (deterministic).
#include <string.h>

#include "pqn_common.h"

#define ML_TILE 16
#define ML_EPS 1e-6f
#define ML_MAXL PQN_MLP_MAX_LAYERS

template <int N>
PQN_D float ml_row_ror(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
PQN_D float ml_group16_sum(float v) {
  v += ml_row_ror<8>(v);
  v += ml_row_ror<4>(v);
  v += ml_row_ror<2>(v);
  v += ml_row_ror<1>(v);
  return v;
}
PQN_D float ml_rsqrt(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  const float h = 0.5f * x * y;
  return fmaf(y, fmaf(-h, y, 0.5f), y);
}

// y[m][o] = b[o] + sum_k x[m][k] W[k][o]   for the 16 samples of the tile; lanes along o.
PQN_D void ml_dense(const float *__restrict__ w, const float *__restrict__ b, const float *x, int xs, int in, int out,
                    float *y, int ys, int tid) {
  for (int o = tid; o < out; o += 256) {
    float acc[ML_TILE];
    const float bo = b[o];
#pragma unroll
    for (int m = 0; m < ML_TILE; ++m) acc[m] = bo;
    for (int k = 0; k < in; ++k) {
      const float wv = w[(size_t)k * out + o];
#pragma unroll
      for (int m = 0; m < ML_TILE; ++m) acc[m] = fmaf(x[m * xs + k], wv, acc[m]);
    }
#pragma unroll
    for (int m = 0; m < ML_TILE; ++m) y[m * ys + o] = acc[m];
  }
}

typedef float ml_f32x4 __attribute__((ext_vector_type(4)));

// Same product on the matrix core (in % 16 == 0, out % 16 == 0, xs % 4 == 0): the 16-sample tile is exactly one
// MFMA M-tile.  v_mfma_f32_16x16x4_f32 operand maps (cdna guide): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D: col = l&15, rows 4*(l>>4)+reg.  K is walked in groups of 16 with lane kk owning k = 16g + 4kk + q, so one
// ds_read_b128 of the activation row feeds 4 MFMAs; B comes straight from the flax (in,out) kernel: row k,
// 16 consecutive outputs = one 64-B run per kk.  Wave w owns output blocks w, w+4, ... four at a time.
PQN_D void ml_dense_mfma(const float *__restrict__ w, const float *__restrict__ b, const float *x, int xs, int in, int out,
                         float *y, int ys, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  const float *arow = x + j * xs + 4 * kk;
  const int nblk = out >> 4;
  for (int nb0 = wave; nb0 < nblk; nb0 += 16) {      // blocks nb0, nb0+4, nb0+8, nb0+12
    ml_f32x4 acc[4];
    const float *wp[4];
    bool on[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc[c] = ml_f32x4{0.f, 0.f, 0.f, 0.f};
      on[c] = nb0 + 4 * c < nblk;
      wp[c] = w + (size_t)(4 * kk) * out + 16 * (on[c] ? nb0 + 4 * c : nb0) + j;
    }
    // weight fragments of K group g+1 are loaded (in place, after their consumers -- see phase2_fc1 in
    // pqn_qnet.hip) while the 16 MFMAs of group g run; the four accumulators are interleaved so that
    // consecutive MFMAs are independent
    float bv[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) bv[c][qd] = wp[c][(size_t)qd * out];
    ml_f32x4 a_next = *reinterpret_cast<const ml_f32x4 *>(arow);
    for (int g = 0; g < in; g += 16) {
      const ml_f32x4 a = a_next;
      const int gn = (g + 16 < in) ? g + 16 : g;      // last group: harmless reload of itself
      a_next = *reinterpret_cast<const ml_f32x4 *>(arow + gn);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const float aq = qd == 0 ? a.x : (qd == 1 ? a.y : (qd == 2 ? a.z : a.w));
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq, bv[c][qd], acc[c], 0, 0, 0);
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c][qd] = wp[c][(size_t)(gn + qd) * out];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (!on[c]) continue;
      const int o = 16 * (nb0 + 4 * c) + j;
      const float bo = b ? b[o] : 0.0f;
      float *yp = y + (4 * kk) * ys + o;
      yp[0] = acc[c].x + bo;
      yp[ys] = acc[c].y + bo;
      yp[2 * ys] = acc[c].z + bo;
      yp[3 * ys] = acc[c].w + bo;
    }
  }
}

PQN_D void ml_dense_any(const float *__restrict__ w, const float *__restrict__ b, const float *x, int xs, int in, int out,
                        float *y, int ys, int tid) {
  if ((in & 15) == 0 && (out & 15) == 0 && (xs & 3) == 0) ml_dense_mfma(w, b, x, xs, in, out, y, ys, tid);
  else ml_dense(w, b, x, xs, in, out, y, ys, tid);
}

// dW[k][o] = sum_m in[m][k] dz[m][o] over the 16 samples of the tile, on the matrix core (ind % 16 == 0):
// A[i = k][kk = m] = in[m][16kb + i], B[kk = m][j = o] = dz[m][16nb + j]; 4 MFMAs (K = 16 samples) per 16x16
// output block, D row k = 4*(l>>4)+reg, col o.  Wave w owns row blocks w, w+4, ...
PQN_D void ml_wgrad_mfma(const float *in, int ins, int ind, const float *dz, int dzs, int h, float *__restrict__ gw,
                         int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kk = lane >> 4;
  for (int kb = wave; kb < (ind >> 4); kb += 4) {
    float a[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) a[s] = in[(4 * s + kk) * ins + 16 * kb + j];
    for (int nb = 0; nb < (h >> 4); ++nb) {
      ml_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], dz[(4 * s + kk) * dzs + 16 * nb + j], acc, 0, 0, 0);
      float *gp = gw + (size_t)(16 * kb + 4 * kk) * h + 16 * nb + j;
      gp[0] = acc.x;
      gp[h] = acc.y;
      gp[2 * (size_t)h] = acc.z;
      gp[3 * (size_t)h] = acc.w;
    }
  }
}

// LayerNorm(H) + relu per sample, 16 lanes per sample; writes h = relu(LN(y)) and (optionally) xhat / rstd.
PQN_D void ml_ln_relu(const float *y, int ys, const float *__restrict__ g, const float *__restrict__ b, int h,
                      float *hout, float *xhat_out, float *rstd_out, int tid) {
  const int m = tid >> 4, sub = tid & 15;
  float s = 0.f, sq = 0.f;
  for (int o = sub; o < h; o += 16) {
    const float v = y[m * ys + o];
    s += v;
    sq = fmaf(v, v, sq);
  }
  s = ml_group16_sum(s);
  sq = ml_group16_sum(sq);
  const float mean = s / (float)h;
  const float var = fmaxf(sq / (float)h - mean * mean, 0.0f);
  const float rstd = ml_rsqrt(var + ML_EPS);
  for (int o = sub; o < h; o += 16) {
    const float xh = (y[m * ys + o] - mean) * rstd;
    if (xhat_out) xhat_out[m * ys + o] = xh;
    hout[m * ys + o] = fmaxf(fmaf(xh, g[o], b[o]), 0.0f);
  }
  if (rstd_out && sub == 0) rstd_out[m] = rstd;
}

// ---------------------------------------------------------------------------
// forward (+ eps-greedy epilogue): network.apply(train=False) + eps_greedy_exploration
// (pqn_gymnax.py:178-190 == pqn_minatar.py:184-196).
// LDS: x[16][DS] | a[16][HS] | y[16][HS]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_fwd_kernel(int n, const float *__restrict__ obs, const float *__restrict__ theta,
                                                      pqn_mlp_layout_t L, float *__restrict__ q_out,
                                                      int32_t *__restrict__ action, float *__restrict__ qmax, float eps,
                                                      uint64_t key, const float *__restrict__ eps_dev,
                                                      const uint64_t *__restrict__ key_dev, int n_per_seed,
                                                      long long theta_stride, int key_stride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, e0 = blockIdx.x * ML_TILE;
  int e_off = 0;   // seed batching: tile -> seed (n_per_seed % 16 == 0): that seed's parameters, key and env numbering
  if (n_per_seed > 0) {
    const int seed = e0 / n_per_seed;
    theta += seed * theta_stride;
    if (key_dev) key_dev += (size_t)seed * key_stride;
    e_off = seed * n_per_seed;
  }
  const int DS = L.d + 1, HS = L.h + 4;
  float *x = sm, *a = x + ML_TILE * DS, *y = a + ML_TILE * HS;
  for (int i = tid; i < ML_TILE * L.d; i += 256) {
    const int m = i / L.d, k = i - m * L.d;
    x[m * DS + k] = (e0 + m < n) ? obs[(size_t)(e0 + m) * L.d + k] : 0.0f;
  }
  __syncthreads();
  const float *in = x;
  int ins = DS, ind = L.d;
  for (int l = 0; l < L.layers; ++l) {
    ml_dense_any(theta + L.off_w[l], theta + L.off_b[l], in, ins, ind, L.h, y, HS, tid);
    __syncthreads();
    ml_ln_relu(y, HS, theta + L.off_lns[l], theta + L.off_lnb[l], L.h, a, nullptr, nullptr, tid);
    __syncthreads();
    in = a; ins = HS; ind = L.h;
  }
  // head: 16 lanes per sample, every lane ends with all q
  const int m = tid >> 4, sub = tid & 15, e = e0 + m;
  float q[8];
#pragma unroll
  for (int aa = 0; aa < 8; ++aa) {
    float part = 0.f;
    if (aa < L.a)
      for (int k = sub; k < ind; k += 16) part = fmaf(in[m * ins + k], theta[L.off_wout + k * L.a + aa], part);
    q[aa] = ml_group16_sum(part) + (aa < L.a ? theta[L.off_bout + aa] : 0.0f);
  }
  if (sub == 0 && e < n) {
    int best = 0;
    float bv = q[0];
#pragma unroll
    for (int aa = 1; aa < 8; ++aa)
      if (aa < L.a && q[aa] > bv) { bv = q[aa]; best = aa; }
    if (q_out)
      for (int aa = 0; aa < L.a; ++aa) q_out[(size_t)e * L.a + aa] = q[aa];
    if (qmax) qmax[e] = bv;
    if (action) {
      if (key_dev) key = *key_dev;
      if (eps_dev) eps = *eps_dev;
      uint32_t o0, o1;
      pqn_bits(key, (uint32_t)(e - e_off), PQN_STREAM_ACT, o0, o1);
      action[e] = (pqn_uniform(o0) < eps) ? (int)pqn_randint(o1, (uint32_t)L.a) : best;
    }
  }
}

// ---------------------------------------------------------------------------
// training: forward + backward of one 16-sample tile; writes the tile's partial gradient (whole
// parameter vector, kernel layout) to gpart[tile][total] and loss / chosen-q partials to lq[tile][2].
// `wt` = transposed copies of the hidden kernels of layers >= 1 ([out][in], refreshed by
// mlp_transpose_kernel after every optimizer step) so the input-gradient GEMM also reads with lanes
// along the contiguous axis.
// LDS: x[16][DS] | act[l][16][HS] | xhat[l][16][HS] | rstd[l][16] | y[16][HS] | dz[16][HS] | gs[16] | act_i[16]
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_train_kernel(int nb, const int64_t *__restrict__ idx,
                                                        const float *__restrict__ obs, const int32_t *__restrict__ action,
                                                        const float *__restrict__ target, const float *__restrict__ theta,
                                                        const float *__restrict__ wt, pqn_mlp_layout_t L, float inv_b,
                                                        float *__restrict__ gpart, float *__restrict__ lq, pqn_seeds_t sd,
                                                        long long wt_stride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, b0 = blockIdx.x * ML_TILE;
  const int seed = blockIdx.y;           // seed slice of the stacked buffers (all strides 0 for a single seed)
  idx += seed * sd.idx_stride;
  theta += seed * sd.theta_stride;
  if (wt) wt += seed * wt_stride;
  gpart += seed * sd.ws_stride;
  lq += seed * sd.ws_stride;
  auto row_of = [&](int64_t key) -> int64_t {   // transition t*N+e of this seed -> row of the stacked [T][S*N] record
    const uint32_t j = (uint32_t)(key & sd.idx_mask);
    if (sd.n_env_total == sd.n_env) return (int64_t)j;
    const uint32_t t = j / (uint32_t)sd.n_env;
    return (int64_t)t * sd.n_env_total + (int64_t)seed * sd.n_env + (int64_t)(j - t * (uint32_t)sd.n_env);
  };
  const int DS = L.d + 1, HS = L.h + 4, H = L.h, NL = L.layers;
  float *x = sm;
  float *act = x + ML_TILE * DS;                   // [NL][16][HS]
  float *xhat = act + NL * ML_TILE * HS;           // [NL][16][HS]
  float *rstd = xhat + NL * ML_TILE * HS;          // [NL][16]
  float *y = rstd + NL * ML_TILE;                  // [16][HS]
  float *dz = y + ML_TILE * HS;                    // [16][HS]
  float *gs = dz + ML_TILE * HS;                   // [16]
  int *acts = reinterpret_cast<int *>(gs + ML_TILE);
  float *gp = gpart + (size_t)blockIdx.x * L.total;
  for (int i = tid; i < L.total; i += 256) gp[i] = 0.0f;   // dummy BatchNorm + padding keep zero gradient
  for (int i = tid; i < ML_TILE * L.d; i += 256) {
    const int m = i / L.d, k = i - m * L.d;
    x[m * DS + k] = (b0 + m < nb) ? obs[(size_t)row_of(idx[b0 + m]) * L.d + k] : 0.0f;
  }
  __syncthreads();
  // ---- forward, keeping every layer's input, xhat and rstd ------------------------------------------------
  {
    const float *in = x;
    int ins = DS, ind = L.d;
    for (int l = 0; l < NL; ++l) {
      ml_dense_any(theta + L.off_w[l], theta + L.off_b[l], in, ins, ind, H, y, HS, tid);
      __syncthreads();
      ml_ln_relu(y, HS, theta + L.off_lns[l], theta + L.off_lnb[l], H, act + l * ML_TILE * HS,
                 xhat + l * ML_TILE * HS, rstd + l * ML_TILE, tid);
      __syncthreads();
      in = act + l * ML_TILE * HS; ins = HS; ind = H;
    }
  }
  const float *hl = act + (NL - 1) * ML_TILE * HS;
  // ---- head + loss gradient (pqn_gymnax.py:262-278 == pqn_minatar.py:271-287) --------------------------------
  {
    const int m = tid >> 4, sub = tid & 15;
    const bool valid = (b0 + m) < nb;
    const int64_t src = valid ? row_of(idx[b0 + m]) : 0;
    const int am = valid ? action[src] : 0;
    float part = 0.f;
    for (int k = sub; k < H; k += 16) part = fmaf(hl[m * HS + k], theta[L.off_wout + k * L.a + am], part);
    const float chosen = ml_group16_sum(part) + theta[L.off_bout + am];
    const float diff = valid ? chosen - target[src] : 0.0f;
    if (sub == 0) {
      gs[m] = diff * inv_b;
      acts[m] = am;
      y[m * HS] = 0.5f * diff * diff;        // per-sample loss / chosen q, folded below
      y[m * HS + 1] = valid ? chosen : 0.0f;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float l = 0.f, c = 0.f;
    for (int m = 0; m < ML_TILE; ++m) { l += y[m * HS]; c += y[m * HS + 1]; }
    lq[2 * blockIdx.x] = l;
    lq[2 * blockIdx.x + 1] = c;
  }
  // d head kernel / bias; d h_L -> dz buffer
  for (int k = tid; k < H; k += 256) {
    for (int a = 0; a < L.a; ++a) {
      float acc = 0.f;
#pragma unroll
      for (int m = 0; m < ML_TILE; ++m) acc += (acts[m] == a) ? gs[m] * hl[m * HS + k] : 0.0f;
      gp[L.off_wout + k * L.a + a] = acc;
    }
#pragma unroll
    for (int m = 0; m < ML_TILE; ++m) dz[m * HS + k] = gs[m] * theta[L.off_wout + k * L.a + acts[m]];
  }
  if (tid < L.a) {
    float acc = 0.f;
    for (int m = 0; m < ML_TILE; ++m) acc += (acts[m] == tid) ? gs[m] : 0.0f;
    gp[L.off_bout + tid] = acc;
  }
  __syncthreads();
  // ---- hidden layers, last to first --------------------------------------------------------------------------
  for (int l = NL - 1; l >= 0; --l) {
    const float *xh = xhat + l * ML_TILE * HS;
    const float *g = theta + L.off_lns[l];
    // relu mask + LayerNorm backward, 16 lanes per sample; dz <- d(pre-LN);  y <- d(relu input) for the LN grads
    {
      const int m = tid >> 4, sub = tid & 15;
      float s1 = 0.f, s2 = 0.f;
      for (int o = sub; o < H; o += 16) {
        const float hv = act[l * ML_TILE * HS + m * HS + o];
        const float dy = hv > 0.0f ? dz[m * HS + o] : 0.0f;
        y[m * HS + o] = dy;
        const float dxh = dy * g[o];
        s1 += dxh;
        s2 = fmaf(dxh, xh[m * HS + o], s2);
      }
      s1 = ml_group16_sum(s1) / (float)H;
      s2 = ml_group16_sum(s2) / (float)H;
      const float rs = rstd[l * ML_TILE + m];
      for (int o = sub; o < H; o += 16) {
        const float dxh = y[m * HS + o] * g[o];
        dz[m * HS + o] = rs * (dxh - s1 - xh[m * HS + o] * s2);
      }
    }
    __syncthreads();
    const float *in = l == 0 ? x : act + (l - 1) * ML_TILE * HS;
    const int ins = l == 0 ? DS : HS, ind = l == 0 ? L.d : H;
    // lanes along o: LN scale/bias grads, dense bias grad, dense kernel grad (fixed order over the 16 samples)
    for (int o = tid; o < H; o += 256) {
      float dzo[ML_TILE], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int m = 0; m < ML_TILE; ++m) {
        dzo[m] = dz[m * HS + o];
        a1 += dzo[m];
        a2 = fmaf(y[m * HS + o], xh[m * HS + o], a2);
        a3 += y[m * HS + o];
      }
      gp[L.off_b[l] + o] = a1;
      gp[L.off_lns[l] + o] = a2;
      gp[L.off_lnb[l] + o] = a3;
      if (ind & 15) {   // narrow input layer (CartPole: 4 features): VALU
        for (int k = 0; k < ind; ++k) {
          float acc = 0.f;
#pragma unroll
          for (int m = 0; m < ML_TILE; ++m) acc = fmaf(in[m * ins + k], dzo[m], acc);
          gp[L.off_w[l] + (size_t)k * H + o] = acc;
        }
      }
    }
    if ((ind & 15) == 0) ml_wgrad_mfma(in, ins, ind, dz, HS, H, gp + L.off_w[l], tid);
    if (l > 0) {
      // d input[m][k] = sum_o dz[m][o] W[k][o]  ->  lanes along k on the transposed copy Wt[o][k]
      const float *wtl = wt + (size_t)(l - 1) * H * H;
      __syncthreads();   // everyone is done reading y (relu-input grads) before it is overwritten
      ml_dense_mfma(wtl, nullptr, dz, HS, H, H, y, HS, tid);   // y[m][k] = sum_o dz[m][o] Wt[o][k]
      __syncthreads();
      for (int i = tid; i < ML_TILE * H; i += 256) {
        const int m = i / H, k = i - m * H;
        dz[m * HS + k] = y[m * HS + k];
      }
    }
    __syncthreads();
  }
}

// fold tile partials into the flat gradient + block sums of squares (radam_apply scratch protocol);
// block 0 also folds loss / chosen-q and snapshots *count.
__global__ __launch_bounds__(256) void mlp_grad_reduce_kernel(int total, int ntiles, const float *__restrict__ gpart,
                                                              const float *__restrict__ lq, float *__restrict__ grad,
                                                              const int32_t *__restrict__ count,
                                                              float *__restrict__ scratch, float *__restrict__ loss_out,
                                                              float *__restrict__ qv_out, float inv_b, pqn_seeds_t sd) {
  __shared__ float s_part[4];
  {  // seed slice
    const long long s = blockIdx.y;
    gpart += s * sd.ws_stride;
    lq += s * sd.ws_stride;
    scratch += s * sd.ws_stride;
    grad += s * sd.theta_stride;
    count += s;
    if (loss_out) loss_out += s * sd.lq_stride;
    if (qv_out) qv_out += s * sd.lq_stride;
  }
  float ss = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    float g = 0.f;
    for (int t = 0; t < ntiles; ++t) g += gpart[(size_t)t * total + i];
    grad[i] = g;
    ss = fmaf(g, g, ss);
  }
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    scratch[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    if (blockIdx.x == 0) {
      reinterpret_cast<int32_t *>(scratch)[1023] = *count;
      float l = 0.f, c = 0.f;
      for (int t = 0; t < ntiles; ++t) { l += lq[2 * t]; c += lq[2 * t + 1]; }
      if (loss_out) *loss_out = l * inv_b;
      if (qv_out) *qv_out = c * inv_b;
    }
  }
}

// wt[l-1][o][k] = W_l[k][o] for hidden layers l >= 1
__global__ void mlp_transpose_kernel(const float *__restrict__ theta, pqn_mlp_layout_t L, float *__restrict__ wt,
                                     long long theta_stride, long long wt_stride) {
  theta += blockIdx.z * theta_stride;   // seed slice
  wt += blockIdx.z * wt_stride;
  const int H = L.h;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int l = blockIdx.y + 1;
  if (i >= H * H) return;
  const int o = i / H, k = i - o * H;
  wt[(size_t)(l - 1) * H * H + i] = theta[L.off_w[l] + (size_t)k * H + o];
}

// ===========================================================================
// host side
// ===========================================================================
static int ml_align4(int x) { return (x + 3) & ~3; }

extern "C" int pqn_mlp_layout(int32_t d, int32_t h, int32_t layers, int32_t a, pqn_mlp_layout_t *L) {
  PQN_REQUIRE(L, "pqn_mlp_layout: layout is NULL");
  PQN_REQUIRE(d >= 1 && d <= 1024, "pqn_mlp_layout: obs dim %d out of range", d);
  PQN_REQUIRE(h >= 16 && h <= 1024 && h % 16 == 0, "pqn_mlp_layout: hidden size %d must be a multiple of 16 in [16,1024]", h);
  PQN_REQUIRE(layers >= 1 && layers <= ML_MAXL, "pqn_mlp_layout: NUM_LAYERS %d out of range [1,%d]", layers, ML_MAXL);
  PQN_REQUIRE(a >= 1 && a <= 8, "pqn_mlp_layout: num_actions %d out of range [1,8]", a);
  memset(L, 0, sizeof(*L));
  L->d = d; L->h = h; L->layers = layers; L->a = a;
  int off = 0;
  L->off_bn = off; off = ml_align4(off + 2 * d);
  int in = d;
  for (int l = 0; l < layers; ++l) {
    L->off_w[l] = off; off = ml_align4(off + in * h);
    L->off_b[l] = off; off += h;
    L->off_lns[l] = off; off += h;
    L->off_lnb[l] = off; off += h;
    in = h;
  }
  L->off_wout = off; off = ml_align4(off + h * a);
  L->off_bout = off; off = ml_align4(off + a);
  L->total = off;
  return PQN_OK;
}

static size_t ml_fwd_smem(const pqn_mlp_layout_t &L) { return sizeof(float) * ML_TILE * ((L.d + 1) + 2 * (L.h + 4)); }
static size_t ml_train_smem(const pqn_mlp_layout_t &L) {
  return sizeof(float) * (ML_TILE * (L.d + 1) + (2 * L.layers + 2) * ML_TILE * (L.h + 4) + L.layers * ML_TILE + 2 * ML_TILE);
}

int pqn_mlp_forward_dyn(const pqn_mlp_layout_t &L, int n, const float *obs, const float *theta, float *q, int32_t *action,
                        float *qmax, float eps, uint64_t key, const float *eps_dev, const uint64_t *key_dev,
                        hipStream_t st, int n_per_seed, long long theta_stride, int key_stride) {
  PQN_REQUIRE(n_per_seed == 0 || n_per_seed % ML_TILE == 0, "pqn_mlp_forward: seed batching needs NUM_ENVS %% %d == 0", ML_TILE);
  const size_t smem = ml_fwd_smem(L);
  PQN_REQUIRE(smem <= 160 * 1024, "pqn_mlp_forward: layer width %d needs %zu B of LDS", L.h, smem);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
  hipLaunchKernelGGL(mlp_fwd_kernel, dim3((n + ML_TILE - 1) / ML_TILE), dim3(256), smem, st, n, obs, theta, L, q, action,
                     qmax, eps, key, eps_dev, key_dev, n_per_seed, theta_stride, key_stride);
  return pqn_check_launch("pqn_mlp_forward");
}

extern "C" int pqn_mlp_forward(const pqn_mlp_layout_t *L, int32_t n, const float *obs, const float *theta, float *q,
                               int32_t *action, float *qmax, float eps, uint64_t key, void *stream) {
  PQN_REQUIRE(L && obs && theta, "pqn_mlp_forward: NULL argument");
  PQN_REQUIRE(n > 0 && (q || action || qmax), "pqn_mlp_forward: n must be > 0 and an output requested");
  return pqn_mlp_forward_dyn(*L, n, obs, theta, q, action, qmax, eps, key, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int64_t pqn_mlp_workspace_floats(const pqn_mlp_layout_t *L, int32_t nb) {
  if (!L || nb <= 0) return -1;
  const int64_t ntiles = (nb + ML_TILE - 1) / ML_TILE;
  return 1024 + ntiles * ((int64_t)L->total + 2);
}

extern "C" int pqn_mlp_grad(const pqn_mlp_layout_t *L, int32_t nb, const int64_t *idx, const float *obs,
                            const int32_t *action, const float *target, const float *theta, const float *wt, float *grad,
                            const int32_t *count, float *workspace, float *loss_out, float *qv_out, void *stream) {
  PQN_REQUIRE(L && idx && obs && action && target && theta && grad && count && workspace, "pqn_mlp_grad: NULL argument");
  return pqn_mlp_grad_seeds(*L, nb, idx, obs, action, target, theta, wt, grad, count, workspace, loss_out, qv_out,
                            pqn_one_seed(), 0, (hipStream_t)stream);
}

// internal (pqn_update.hip): S seeds per launch, buffers = slices of stacked allocations (pqn_seeds_t)
int pqn_mlp_grad_seeds(const pqn_mlp_layout_t &L, int nb, const int64_t *idx, const float *obs, const int32_t *action,
                       const float *target, const float *theta, const float *wt, float *grad, const int32_t *count,
                       float *workspace, float *loss_out, float *qv_out, const pqn_seeds_t &sd, long long wt_stride,
                       hipStream_t st) {
  PQN_REQUIRE(nb > 0, "pqn_mlp_grad: minibatch size must be > 0");
  PQN_REQUIRE(L.layers == 1 || wt, "pqn_mlp_grad: transposed hidden kernels (wt) required for NUM_LAYERS > 1");
  const size_t smem = ml_train_smem(L);
  PQN_REQUIRE(smem <= 160 * 1024, "pqn_mlp_grad: %d layers of width %d need %zu B of LDS (limit 160 KB)", L.layers, L.h,
              smem);
  const int ntiles = (nb + ML_TILE - 1) / ML_TILE;
  float *scratch = workspace, *gpart = workspace + 1024, *lq = gpart + (size_t)ntiles * L.total;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_train_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
  const float inv_b = 1.0f / (float)nb;
  hipLaunchKernelGGL(mlp_train_kernel, dim3(ntiles, sd.nseeds), dim3(256), smem, st, nb, idx, obs, action, target, theta, wt,
                     L, inv_b, gpart, lq, sd, wt_stride);
  hipLaunchKernelGGL(mlp_grad_reduce_kernel, dim3(pqn_radam_blocks(L.total), sd.nseeds), dim3(256), 0, st, L.total, ntiles,
                     gpart, lq, grad, count, scratch, loss_out, qv_out, inv_b, sd);
  return pqn_check_launch("pqn_mlp_grad");
}

extern "C" int pqn_mlp_refresh_transposed(const pqn_mlp_layout_t *L, const float *theta, float *wt, void *stream) {
  PQN_REQUIRE(L && theta, "pqn_mlp_refresh_transposed: NULL argument");
  return pqn_mlp_refresh_transposed_seeds(*L, theta, wt, 1, 0, 0, (hipStream_t)stream);
}

int pqn_mlp_refresh_transposed_seeds(const pqn_mlp_layout_t &L, const float *theta, float *wt, int nseeds,
                                     long long theta_stride, long long wt_stride, hipStream_t st) {
  if (L.layers <= 1) return PQN_OK;
  PQN_REQUIRE(wt, "pqn_mlp_refresh_transposed: wt is NULL");
  hipLaunchKernelGGL(mlp_transpose_kernel, dim3((L.h * L.h + 255) / 256, L.layers - 1, nseeds), dim3(256), 0, st, theta, L, wt,
                     theta_stride, wt_stride);
  return pqn_check_launch("pqn_mlp_refresh_transposed");
}

extern "C" int pqn_mlp_apply(const pqn_mlp_layout_t *L, float *theta, float *wt, const float *grad, float *m, float *v,
                             int32_t *count, float lr_init, float lr_end, double lr_steps, float max_norm, float *workspace,
                             float *gnorm_out, int32_t recompute_norm, void *stream) {
  PQN_REQUIRE(L && theta && grad && m && v && count && workspace, "pqn_mlp_apply: NULL argument");
  const int rc = pqn_launch_radam(theta, grad, m, v, L->total, count, lr_init, lr_end, lr_steps, max_norm, workspace,
                                  gnorm_out, 0, nullptr, recompute_norm, pqn_radam_blocks(L->total), (hipStream_t)stream);
  if (rc != PQN_OK) return rc;
  return pqn_mlp_refresh_transposed(L, theta, wt, stream);
}
