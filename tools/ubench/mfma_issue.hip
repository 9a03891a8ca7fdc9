// How the bf16 matrix pipe and the vector ALU of one SIMD share issue slots (behind the T2 pipeline design):
//   v_mfma_f32_16x16x32_bf16 with NACC accumulators round-robin, as builtin / as tied inline asm (with the `s_nop 1`
//   the product uses), with V independent VALU ops after every MFMA; 1 and 2 waves per SIMD.  Prints cycles per MFMA.
// build: hipcc -O3 --offload-arch=gfx950 mfma_issue.hip -o mfma_issue ; run on an MI355X
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC, int V>
__global__ void k(float *out, int iters, unsigned long long *cyc) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a = {threadIdx.x * 3u + 0x3f800000u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  float v[4] = {1.0f + threadIdx.x, 2.0f, 3.0f, 4.0f};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
      else if (MODE == 1) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < V; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 3]) : "v"(v[(j + 1) & 3]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// distinct B operand per MFMA (8 register tuples, like a weight ring) and a distinct A per pair
template <int NACC>
__global__ void k_ops(float *out, int iters, unsigned long long *cyc) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a[4], b[8];
  for (int i = 0; i < 4; ++i) a[i] = u32x4{threadIdx.x * 3u + 0x3f800000u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  for (int i = 0; i < 8; ++i) b[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[i & 7]));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run_ops(float *out, unsigned long long *cyc) {
  const int iters = 2000;
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL((k_ops<NACC>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("distinct A/B operands acc=%d waves/SIMD=%d: %.1f ticks per MFMA of one wave (%.1f per SIMD-MFMA)\n", NACC, threads / 256,
           (double)c / (iters * NACC), (double)c / (iters * NACC) / (threads / 256));
  }
}

template <int MODE, int NACC, int V>
void run(const char *name, float *out, unsigned long long *cyc) {
  const int iters = 2000;
  for (int threads : {256, 512}) {   // 1 and 2 waves per SIMD
    hipLaunchKernelGGL((k<MODE, NACC, V>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-10s acc=%d valu/mfma=%d waves/SIMD=%d: %.1f cycles per MFMA of one wave (%.1f per SIMD-MFMA)\n", name, NACC, V,
           threads / 256, (double)c / (iters * NACC), (double)c / (iters * NACC) / (threads / 256));
  }
}

int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
  run<0, 8, 0>("builtin", out, cyc);
  run<2, 8, 0>("asm", out, cyc);
  run<1, 8, 0>("asm+nop1", out, cyc);
  run<1, 4, 0>("asm+nop1", out, cyc);
  run<1, 8, 2>("asm+nop1", out, cyc);
  run<1, 8, 4>("asm+nop1", out, cyc);
  run<1, 4, 4>("asm+nop1", out, cyc);
  run<2, 8, 4>("asm", out, cyc);
  run<0, 8, 4>("builtin", out, cyc);
  run<1, 8, 6>("asm+nop1", out, cyc);
  run_ops<8>(out, cyc);
  run_ops<4>(out, cyc);
  return 0;
}
