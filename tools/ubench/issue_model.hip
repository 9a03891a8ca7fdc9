// Issue model of one gfx950 SIMD for the position kernels (round 6): how v_mfma_f32_16x16x32_bf16, independent VALU work,
// s_nop and ds_read_b128 share a SIMD at 1 / 2 waves per SIMD.  Every number is WALL time (hipEvents over the whole launch,
// all 256 CUs busy) converted to "ns per SIMD-MFMA"; 16384 flop / that = the rate the matrix pipe ran at (dense bf16 peak
// 2.5 PF = 6.7 ns per SIMD-MFMA at 2.4 GHz = 16 cycles).  s_memtime deltas are printed beside it to calibrate the counter.
// build: hipcc -O3 --offload-arch=gfx950 issue_model.hip -o issue_model
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// NACC accumulators round-robin; after every GRP MFMAs: V independent VALU ops, NOP x `s_nop 1`, LDS x ds_read_b128
template <int NACC, int GRP, int V, int NOP, int LDS>
__global__ __launch_bounds__(1024) void k(float *out, int iters, unsigned long long *cyc) {
  __shared__ u32x4 sm[1024];
  sm[threadIdx.x] = u32x4{threadIdx.x, 1u, 2u, 3u};
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = u32x4{threadIdx.x * 3u + 0x3f800000u + i, 0x3f803f80u, 0x3f813f80u, 0x3f803f82u};
    b[i] = u32x4{0x3f803f80u + i, 0x3f803f81u, 0x3f823f80u, 0x3f803f80u};
  }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + threadIdx.x * 0.001f + i;
  u32x4 ld[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 1) & 3]));
      if ((i % GRP) == GRP - 1) {
#pragma unroll
        for (int j = 0; j < NOP; ++j) asm volatile("s_nop 1");
#pragma unroll
        for (int j = 0; j < V; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(v[(j + 4) & 7]));
#pragma unroll
        for (int j = 0; j < LDS; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[j & 1]) : "v"((unsigned)(threadIdx.x * 16)));
      }
    }
    if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  s += __uint_as_float(ld[0].x + ld[1].y);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// one MFMA followed by N fillers of kind KIND (0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_mul_f32, 3 v_pk_add_f32, 4 v_add_f32_dpp row_ror,
// 5 v_mov_b32, 6 v_cvt_pk_bf16_f32, 7 v_and_b32, 8 v_lshrrev_b64, 9 s_add_u32, 10 s_nop 0, 11 s_nop 7, 12 v_permlane32_swap, 13 v_max_f32,
// 14 s_mov_b32 m0 (the LDS-DMA destination register))
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int KIND, int N>
__global__ __launch_bounds__(1024) void kf(float *out, int iters, unsigned long long *cyc) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = u32x4{threadIdx.x * 3u + 0x3f800000u + i, 0x3f803f80u, 0x3f813f80u, 0x3f803f82u};
    b[i] = u32x4{0x3f803f80u + i, 0x3f803f81u, 0x3f823f80u, 0x3f803f80u};
  }
  f32x2 p[8];
  float v[8];
  unsigned u[8];
  unsigned long long w[8];
  for (int i = 0; i < 8; ++i) { v[i] = 1.0f + threadIdx.x * 0.001f + i; p[i] = f32x2{v[i], v[i] + 1.f}; u[i] = threadIdx.x + i; w[i] = threadIdx.x * 77ull + i; }
  unsigned sreg = 1;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i >> 1) & 3]));
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int d = (i * N + j) & 7, e = (d + 4) & 7;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[d]) : "v"(v[e]));
        else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[d]) : "v"(p[e]));
        else if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[d]) : "v"(p[e]));
        else if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[d]) : "v"(p[e]));
        else if (KIND == 4) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(v[d]));
        else if (KIND == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(v[d]) : "v"(v[e]));
        else if (KIND == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[d]) : "v"(v[e]), "v"(v[(e + 1) & 7]));
        else if (KIND == 7) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[d]) : "v"(u[e]));
        else if (KIND == 8) asm volatile("v_lshrrev_b64 %0, 3, %1" : "=v"(w[d]) : "v"(w[e]));
        else if (KIND == 9) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sreg));
        else if (KIND == 10) asm volatile("s_nop 0");
        else if (KIND == 11) asm volatile("s_nop 7");
        else if (KIND == 12) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[d]), "+v"(u[e]));
        else if (KIND == 13) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[d]) : "v"(v[e]));
        else if (KIND == 14) asm volatile("s_mov_b32 m0, %0" : : "s"(sreg) : "memory");
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = (float)sreg;
  for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y + (float)u[i] + (float)w[i];
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// VALU only: V independent fma per iteration
template <int V>
__global__ __launch_bounds__(1024) void kv(float *out, int iters, unsigned long long *cyc) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + threadIdx.x * 0.001f + i;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < V; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j & 7]) : "v"(v[(j + 4) & 7]));
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static float *g_out;
static unsigned long long *g_cyc;

template <class F>
static void timeit(const char *name, int wps, int per_iter_mfma, int per_iter_valu, F launch) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(iters);   // warm
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c;
  hipMemcpy(&c, g_cyc, 8, hipMemcpyDeviceToHost);
  const double ns = ms * 1e6;
  printf("%-44s waves/SIMD=%d  wall %8.1f us  ticks/us %7.1f", name, wps, ns / 1e3, (double)c / (ns / 1e3));
  if (per_iter_mfma) printf("  ns per SIMD-MFMA %6.2f (%.0f%% of 2.5 PF)  ticks per wave-MFMA %5.1f", ns / ((double)iters * per_iter_mfma * wps),
                            100.0 * 16384.0 / (ns / ((double)iters * per_iter_mfma * wps)) * 1024 / 2.5e6, (double)c / ((double)iters * per_iter_mfma));
  if (per_iter_valu && !per_iter_mfma) printf("  ns per SIMD-VALU %6.2f  ticks per wave-VALU %5.1f", ns / ((double)iters * per_iter_valu * wps), (double)c / ((double)iters * per_iter_valu));
  printf("\n");
}

template <int NACC, int GRP, int V, int NOP, int LDS>
static void run(const char *name) {
  for (int wps : {1, 2, 4}) {
    timeit(name, wps, NACC, V * (NACC / GRP), [&](int iters) { hipLaunchKernelGGL((k<NACC, GRP, V, NOP, LDS>), dim3(256), dim3(256 * wps), 0, 0, g_out, iters, g_cyc); });
  }
}
template <int V>
static void runv(const char *name) {
  for (int wps : {1, 2, 4})
    timeit(name, wps, 0, V, [&](int iters) { hipLaunchKernelGGL((kv<V>), dim3(256), dim3(256 * wps), 0, 0, g_out, iters, g_cyc); });
}

template <int KIND, int N>
static void runf(const char *name) {
  for (int wps : {1, 2})
    timeit(name, wps, 8, 0, [&](int iters) { hipLaunchKernelGGL((kf<KIND, N>), dim3(256), dim3(256 * wps), 0, 0, g_out, iters, g_cyc); });
}
#define RUNF(K, NAME) runf<K, 2>("mfma + 2 x " NAME); runf<K, 4>("mfma + 4 x " NAME);

int main(int argc, char **argv) {
  hipMalloc(&g_out, 256 * 1024 * 4);
  hipMalloc(&g_cyc, 8);
  if (argc > 1) {
    RUNF(0, "v_fma_f32") RUNF(1, "v_pk_fma_f32") RUNF(2, "v_pk_mul_f32") RUNF(3, "v_pk_add_f32") RUNF(4, "v_add_f32_dpp")
    RUNF(5, "v_mov_b32") RUNF(6, "v_cvt_pk_bf16_f32") RUNF(7, "v_and_b32") RUNF(8, "v_lshrrev_b64") RUNF(9, "s_add_u32")
    RUNF(10, "s_nop 0") RUNF(11, "s_nop 7") RUNF(12, "v_permlane32_swap") RUNF(13, "v_max_f32") RUNF(14, "s_mov_b32 m0")
    return 0;
  }
  run<8, 1, 0, 0, 0>("mfma only, 8 acc");
  run<4, 1, 0, 0, 0>("mfma only, 4 acc");
  run<2, 1, 0, 0, 0>("mfma only, 2 acc");
  run<8, 1, 0, 1, 0>("mfma + s_nop 1 each");
  run<8, 4, 0, 1, 0>("mfma x4 + s_nop 1");
  run<8, 1, 1, 0, 0>("mfma + 1 valu each");
  run<8, 1, 2, 0, 0>("mfma + 2 valu each");
  run<8, 1, 3, 0, 0>("mfma + 3 valu each");
  run<8, 1, 4, 0, 0>("mfma + 4 valu each");
  run<8, 1, 6, 0, 0>("mfma + 6 valu each");
  run<8, 2, 6, 0, 0>("mfma x2 + 6 valu (3 per mfma)");
  run<8, 4, 12, 0, 0>("mfma x4 + 12 valu (3 per mfma)");
  run<8, 8, 24, 0, 0>("mfma x8 + 24 valu (3 per mfma)");
  run<8, 8, 48, 0, 0>("mfma x8 + 48 valu (6 per mfma)");
  run<8, 1, 3, 1, 0>("mfma + s_nop 1 + 3 valu each");
  run<8, 4, 12, 1, 0>("mfma x4 + s_nop 1 + 12 valu");
  run<8, 4, 0, 0, 1>("mfma x4 + 1 ds_read_b128");
  run<8, 4, 0, 0, 2>("mfma x4 + 2 ds_read_b128");
  run<8, 4, 12, 0, 1>("mfma x4 + 12 valu + 1 ds_read_b128");
  runv<32>("valu only (32 fma / iter)");
  return 0;
}
