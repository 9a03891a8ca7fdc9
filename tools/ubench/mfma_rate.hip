// micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 with W waves per workgroup (1 WG per CU)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(float *out, int iters, unsigned long long *cyc) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *out; unsigned long long *cyc, h;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  for (int threads : {64, 256, 512, 1024}) {
    for (int nacc : {1, 4}) {
      int iters = 1024 / nacc;  // 1024 MFMAs per wave
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (nacc == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
        else hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      int waves_per_simd = threads / 256 > 0 ? threads / 256 : 1;
      double mf = 1024.0 * (threads / 64);  // MFMAs per CU
      printf("threads=%4d nacc=%d: %.1f us, %llu cycles for 1024 MFMA/wave -> %.1f cyc per MFMA per SIMD (waves/SIMD=%d), %.1f TFLOP/s\n",
             threads, nacc, ms * 1e3, h, (double)h / (1024.0 * (threads >= 256 ? threads / 256 : 1)), waves_per_simd,
             256.0 * mf * 2048.0 / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
