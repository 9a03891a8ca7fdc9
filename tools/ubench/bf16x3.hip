// micro-benchmarks behind the bf16x3 split-operand design:
//  (1) issue rate of v_mfma_f32_16x16x32_bf16 with 8 independent accumulators, 1 and 2 waves per SIMD
//  (2) L2 -> CU streaming rate when EVERY workgroup streams the SAME buffer (the fc1 weight stream of T1):
//      256 workgroups x 512 threads, dwordx4 loads, 12 / 24 in flight per wave, 512 KB and 768 KB buffers
//  (3) VALU cost of splitting 8 f32 into 3 x 8 bf16 (hi, mid, lo planes), per wave-instruction group
//  (4) numerics: the 6-product split dot product vs f64 and vs an f32 fma chain, K = 1024
//  (5) (round 6) the same for the f16x2 split: two fp16 pieces of 2^s x, three v_mfma_f32_16x16x32_f16 per K step
// build: hipcc -O3 --offload-arch=gfx950 bf16x3.hip -o bf16x3 ; run on an MI355X
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_rne(float x) {   // f32 -> bf16 bits (round to nearest even), no NaN handling
  unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// x = hi + mid + lo exactly (each a bf16)
__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
  h = bf16_rne(x);
  const float r1 = x - __uint_as_float(h << 16);
  m = bf16_rne(r1);
  const float r2 = r1 - __uint_as_float(m << 16);
  l = bf16_rne(r2);
}

__global__ void mfma_rate(float *out, int iters, unsigned long long *cyc) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int INFLIGHT>
__global__ __launch_bounds__(512) void l2_stream(const f32x4 *__restrict__ buf, int n16, int reps, float *out,
                                                 unsigned long long *cyc) {
  // every workgroup reads the whole buffer `reps` times, starting at a different offset
  const int tid = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int start = (blockIdx.x * 977) % (n16 / 512);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    for (int i = 0; i < n16 / 512; i += INFLIGHT) {
      f32x4 v[INFLIGHT];
#pragma unroll
      for (int j = 0; j < INFLIGHT; ++j) v[j] = buf[(((i + j + start) % (n16 / 512)) * 512) + tid];
#pragma unroll
      for (int j = 0; j < INFLIGHT; ++j) acc += v[j];
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + tid] = acc.x + acc.y + acc.z + acc.w;
  if (tid == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

__global__ void split_cost(const float *in, unsigned *out, int iters, unsigned long long *cyc) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = in[threadIdx.x * 8 + i];
  unsigned acc = 0;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    unsigned hp[4], mp[4], lp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned h0, m0, l0, h1, m1, l1;
      split3(x[2 * i], h0, m0, l0);
      split3(x[2 * i + 1], h1, m1, l1);
      hp[i] = h0 | (h1 << 16); mp[i] = m0 | (m1 << 16); lp[i] = l0 | (l1 << 16);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc ^= hp[i] + mp[i] * 3u + lp[i] * 5u; x[2 * i] += 1e-3f; x[2 * i + 1] *= 1.0001f; }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// numerics: one wave computes a 16x16 tile, K = 1024, from f32 A/B via the 6-product split; reference on the host
__global__ void split_gemm(const float *A /*[16][K]*/, const float *B /*[K][16]*/, int K, float *D, int order) {
  const int lane = threadIdx.x, i = lane & 15, kg = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < K / 32; ++s) {
    s16x8 ah, am, al, bh, bm, bl;
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * s + 8 * kg + j;
      unsigned h, m, l;
      split3(A[i * K + k], h, m, l);
      ah[j] = (short)h; am[j] = (short)m; al[j] = (short)l;
      split3(B[k * 16 + i], h, m, l);
      bh[j] = (short)h; bm[j] = (short)m; bl[j] = (short)l;
    }
#define MF(x, y) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc, 0, 0, 0)
    if (order == 0) { MF(al, bh); MF(ah, bl); MF(am, bm); MF(am, bh); MF(ah, bm); MF(ah, bh); }   // small terms first
    else { MF(ah, bh); MF(ah, bm); MF(am, bh); MF(am, bm); MF(ah, bl); MF(al, bh); }
    if (order == 2) { MF(am, bl); MF(al, bm); MF(al, bl); }                                        // all 9 products
  }
  const int col = lane & 15, r0 = 4 * (lane >> 4);
  D[(r0 + 0) * 16 + col] = acc.x; D[(r0 + 1) * 16 + col] = acc.y; D[(r0 + 2) * 16 + col] = acc.z; D[(r0 + 3) * 16 + col] = acc.w;
}

// f16x2: s x = hi + lo (fp16 each, round to nearest even), products l h + h l + h h; sa / sb = the power-of-two operand scales
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void split_gemm_h2(const float *A, const float *B, int K, float *D, float sa, float sb) {
  const int lane = threadIdx.x, i = lane & 15, kg = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < K / 32; ++s) {
    f16x8 ah, al, bh, bl;
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * s + 8 * kg + j;
      const float x = A[i * K + k] * sa, y = B[k * 16 + i] * sb;
      ah[j] = (_Float16)x; al[j] = (_Float16)(x - (float)ah[j]);
      bh[j] = (_Float16)y; bl[j] = (_Float16)(y - (float)bh[j]);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
  }
  const int col = lane & 15, r0 = 4 * (lane >> 4);
  const float inv = 1.0f / (sa * sb);
  D[(r0 + 0) * 16 + col] = acc.x * inv; D[(r0 + 1) * 16 + col] = acc.y * inv; D[(r0 + 2) * 16 + col] = acc.z * inv; D[(r0 + 3) * 16 + col] = acc.w * inv;
}

int main() {
  float *out; unsigned long long *cyc, h;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  printf("== (1) v_mfma_f32_16x16x32_bf16 issue rate, 8 accumulators\n");
  for (int threads : {256, 512}) {
    const int iters = 256;   // 2048 MFMAs per wave
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_rate, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per_simd = 2048.0 * (threads / 256);
    printf("threads=%d: %.1f us, %llu cycles -> %.1f cyc per MFMA per SIMD, %.0f TFLOP/s\n", threads, ms * 1e3, h,
           (double)h / per_simd, 256.0 * 4 * per_simd * 16384.0 / (ms * 1e-3) / 1e12);
  }
  printf("== (2) L2 -> CU streaming, all 256 workgroups read the same buffer\n");
  for (int kb : {512, 768, 1536}) {
    const int n16 = kb * 1024 / 16;
    f32x4 *buf; hipMalloc(&buf, (size_t)n16 * 16); hipMemset(buf, 0, (size_t)n16 * 16);
    for (int infl : {4, 8, 16}) {
      const int reps = 4;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (infl == 4) hipLaunchKernelGGL(l2_stream<4>, dim3(256), dim3(512), 0, 0, buf, n16, reps, out, cyc);
        else if (infl == 8) hipLaunchKernelGGL(l2_stream<8>, dim3(256), dim3(512), 0, 0, buf, n16, reps, out, cyc);
        else hipLaunchKernelGGL(l2_stream<16>, dim3(256), dim3(512), 0, 0, buf, n16, reps, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      const double bytes = (double)kb * 1024 * reps;
      printf("%4d KB, %2d dwordx4 in flight per lane: %.1f us, %llu cycles -> %.1f B/clk/CU, %.1f us per pass, %.2f TB/s chip\n", kb, infl,
             ms * 1e3, h, bytes / (double)h, ms * 1e3 / reps, 256.0 * bytes / (ms * 1e-3) / 1e12);
    }
    hipFree(buf);
  }
  printf("== (3) split of 8 f32 into 3 x 8 bf16 (per lane), VALU only\n");
  {
    float *in; unsigned *o2; hipMalloc(&in, 512 * 8 * 4); hipMalloc(&o2, 256 * 512 * 4);
    hipMemset(in, 0x3f, 512 * 8 * 4);
    for (int threads : {256, 512}) {
      hipLaunchKernelGGL(split_cost, dim3(256), dim3(threads), 0, 0, in, o2, 256, cyc);
      hipDeviceSynchronize();
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("threads=%d: %.1f cycles per 8-element split (incl. loop filler)\n", threads, (double)h / 256.0);
    }
  }
  printf("== (4) numerics of the 6-product split, K = 1024, |a|,|b| ~ N(0,1)\n");
  {
    const int K = 1024;
    float *hA = (float *)malloc(16 * K * 4), *hB = (float *)malloc(K * 16 * 4), hD[256];
    srand(1);
    auto rn = []() { float u = 0; for (int i = 0; i < 12; ++i) u += rand() / (float)RAND_MAX; return u - 6.0f; };
    for (int i = 0; i < 16 * K; ++i) { hA[i] = rn(); hB[i] = rn(); }
    float *dA, *dB, *dD; hipMalloc(&dA, 16 * K * 4); hipMalloc(&dB, 16 * K * 4); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, hA, 16 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 16 * K * 4, hipMemcpyHostToDevice);
    for (int order = 0; order < 3; ++order) {
      hipLaunchKernelGGL(split_gemm, dim3(1), dim3(64), 0, 0, dA, dB, K, dD, order);
      hipMemcpy(hD, dD, 256 * 4, hipMemcpyDeviceToHost);
      double e_split = 0, e_f32 = 0, scale = 0;
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double ref = 0, sabs = 0; float f = 0.f;
          for (int k = 0; k < K; ++k) { ref += (double)hA[i * K + k] * hB[k * 16 + j]; sabs += fabs((double)hA[i * K + k] * hB[k * 16 + j]); f = fmaf(hA[i * K + k], hB[k * 16 + j], f); }
          e_split = fmax(e_split, fabs(hD[i * 16 + j] - ref) / sabs);
          e_f32 = fmax(e_f32, fabs((double)f - ref) / sabs);
          scale = fmax(scale, sabs);
        }
      printf("order %d (%s): max |err| / sum|a b| = %.3e   (f32 fma chain: %.3e)\n", order,
             order == 0 ? "6 products, small first" : order == 1 ? "6 products, large first" : "9 products", e_split, e_f32);
    }
    printf("== (5) numerics of the f16x2 split (3 products), same data; operand scales 2^11 x 2^11 (max |x| ~ 6 -> < 2^15) and a 2^6 x smaller one\n");
    for (float sc : {2048.0f, 32.0f}) {
      hipLaunchKernelGGL(split_gemm_h2, dim3(1), dim3(64), 0, 0, dA, dB, K, dD, sc, sc);
      hipMemcpy(hD, dD, 256 * 4, hipMemcpyDeviceToHost);
      double e_split = 0, rms = 0, rms32 = 0;
      for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
          double ref = 0, sabs = 0; float f = 0.f;
          for (int k = 0; k < K; ++k) { ref += (double)hA[i * K + k] * hB[k * 16 + j]; sabs += fabs((double)hA[i * K + k] * hB[k * 16 + j]); f = fmaf(hA[i * K + k], hB[k * 16 + j], f); }
          e_split = fmax(e_split, fabs(hD[i * 16 + j] - ref) / sabs);
          rms += (hD[i * 16 + j] - ref) * (hD[i * 16 + j] - ref) / (sabs * sabs);
          rms32 += ((double)f - ref) * ((double)f - ref) / (sabs * sabs);
        }
      printf("scale %.0f: max |err| / sum|a b| = %.3e, rms %.3e   (f32 fma chain rms %.3e)\n", sc, e_split, sqrt(rms / 256), sqrt(rms32 / 256));
    }
  }
  return 0;
}
