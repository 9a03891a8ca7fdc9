// Feasibility microbenchmark for the wide-MLP GEMM: C[M][N] (+)= A[M][K] B[N][K]^T as six bf16 MFMA
// products over three operand planes, 64 x 64 tile, four waves as 2 x 2 -- the shape of bm_gemm_kernel<64, 64> -- but with the
// planes stored FRAGMENT-MAJOR (one contiguous 1 KB block per (16 rows, 32 K): slot 16 kb + (r ^ 2 kb) of 16 B) and moved
// global -> LDS by LDS-DMA (global_load_lds_dwordx4 as inline asm, fully contiguous reads, no VGPR staging, no ds_write).
// STAGES-deep ring, one barrier per K step.  No result check beyond a checksum: this only answers "how fast would it be".
// build: hipcc -O3 --offload-arch=gfx950 gemm_dma.hip -o gemm_dma ; run on an MI355X: ./gemm_dma [M N K nsplit]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char_t;
#define D __device__ __forceinline__
D uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(lds_char_t *)p; }
D void dma16(uint32_t voff, const void *sbase, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
template <int N_> D void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
D f32x4 mfma(const u32x4 &a, const u32x4 &b, f32x4 c) {
  asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  return c;
}

// A, B: [3 planes][rows / 16][Kp / 32][64] u32x4 ; C: [nsplit][M][N]
template <int STAGES>
__global__ __launch_bounds__(256) void gemm_dma(int M, int N, int Kp, int klen, const u32x4 *__restrict__ A, const u32x4 *__restrict__ B,
                                                float *__restrict__ C) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4 *ring = reinterpret_cast<u32x4 *>(smem);   // [STAGES][24 blocks][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int KB = Kp / 32;
  const long long a_ps = (long long)(M / 16) * KB * 64, b_ps = (long long)(N / 16) * KB * 64;
  const int kb0 = bz * (klen / 32), nk = min(KB, kb0 + klen / 32) - kb0;
  const uint32_t ring_lds = lds_addr(ring);
  // this wave's six blocks of a K step: q = 6 wave .. 6 wave + 5; q < 12: A (plane q / 4, row block q % 4), else B
  const u32x4 *src[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = 6 * wave + i, isb = q >= 12, qq = isb ? q - 12 : q, pl = qq >> 2, rb = qq & 3;
    src[i] = isb ? B + pl * b_ps + ((long long)(bx * 4 + rb) * KB + kb0) * 64 : A + pl * a_ps + ((long long)(by * 4 + rb) * KB + kb0) * 64;
  }
  auto dma_step = [&](int ks) {
    const uint32_t dst = ring_lds + (uint32_t)(((ks % STAGES) * 24 + 6 * wave) * 1024);
#pragma unroll
    for (int i = 0; i < 6; ++i) dma16(lane * 16u, src[i] + (long long)ks * 64, dst + i * 1024u);
  };
  f32x4 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fl = (lane & 48) + ((lane & 15) ^ (2 * (lane >> 4)));
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) dma_step(s);
  dma_wait<0>();
  __syncthreads();
#pragma unroll 1
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + STAGES - 1 < nk) dma_step(ks + STAGES - 1);   // its slot was last read in iteration ks - 1: every wave passed that barrier
    const u32x4 *st = ring + (ks % STAGES) * 24 * 64;
    u32x4 a[2][3], b[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi][pl] = st[(pl * 4 + wm * 2 + mi) * 64 + fl];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) b[ni][pl] = st[(12 + pl * 4 + wn * 2 + ni) * 64 + fl];
    }
#define PROD(PA_, PB_)                                                   \
    _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                       \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = mfma(a[mi][PA_], b[ni][PB_], acc[mi][ni]);
    PROD(2, 0) PROD(1, 0) PROD(0, 2) PROD(0, 1) PROD(1, 1) PROD(0, 0)
#undef PROD
    // the DMAs of step ks + 1 (issued an iteration or more ago) must have landed; those of later steps may stay in flight
    if (STAGES == 2) dma_wait<0>();
    else if (STAGES == 3) { if (ks + 2 < nk) dma_wait<6>(); else dma_wait<0>(); }
    else { if (ks + 3 < nk) dma_wait<12>(); else if (ks + 2 < nk) dma_wait<6>(); else dma_wait<0>(); }
    __syncthreads();
  }
  float *out = C + (long long)bz * M * N;
  const int col_l = lane & 15, rq = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[mi][ni]));
      const float tv[4] = {acc[mi][ni].x, acc[mi][ni].y, acc[mi][ni].z, acc[mi][ni].w};
      const int col = bx * 64 + (wn * 2 + ni) * 16 + col_l;
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long long)(by * 64 + (wm * 2 + mi) * 16 + 4 * rq + r) * N + col] = tv[r];
    }
}

template <int STAGES>
static float run(int M, int N, int Kp, int nsplit, const u32x4 *A, const u32x4 *B, float *C, int iters) {
  const int klen = ((Kp + nsplit - 1) / nsplit + 31) / 32 * 32;
  const int ns = (Kp + klen - 1) / klen;
  const int lds = STAGES * 24 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_dma<STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  dim3 grid(N / 64, M / 64, ns);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gemm_dma<STAGES>, grid, dim3(256), lds, 0, M, N, Kp, klen, A, B, C);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_dma<STAGES>, grid, dim3(256), lds, 0, M, N, Kp, klen, A, B, C);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / iters;
}

int main(int argc, char **argv) {
  int M = 1024, N = 1024, K = 1024, nsplit = 3;
  if (argc > 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); nsplit = atoi(argv[4]); }
  const int Kp = (K + 31) / 32 * 32;
  const size_t na = (size_t)3 * M * Kp / 8, nb = (size_t)3 * N * Kp / 8;   // u32x4 = 8 bf16
  u32x4 *A, *B;
  float *C;
  hipMalloc(&A, na * 16); hipMalloc(&B, nb * 16); hipMalloc(&C, (size_t)8 * M * N * 4);
  unsigned *h = (unsigned *)malloc((na > nb ? na : nb) * 16);
  for (size_t i = 0; i < (na > nb ? na : nb) * 4; ++i) h[i] = 0x3c003c00u + (unsigned)((i * 2654435761u) >> 28) * 0x00010001u;   // small bf16 values
  hipMemcpy(A, h, na * 16, hipMemcpyHostToDevice);
  hipMemcpy(B, h, nb * 16, hipMemcpyHostToDevice);
  const double flop = 2.0 * M * N * Kp;
  const float t2 = run<2>(M, N, Kp, nsplit, A, B, C, 200);
  const float t3 = run<3>(M, N, Kp, nsplit, A, B, C, 200);
  const float t4 = run<4>(M, N, Kp, nsplit, A, B, C, 200);
  float c0 = 0.f;
  hipMemcpy(&c0, C, 4, hipMemcpyDeviceToHost);
  printf("gemm_dma %d x %d x %d, %d K split(s): 2 stages %.2f us (%.1f TFLOP/s algorithmic, %.2f of 157.3), 3 stages %.2f us (%.2f), 4 stages %.2f us (%.2f)  c[0]=%g\n",
         M, N, K, nsplit, t2, flop / t2 * 1e-6, flop / t2 * 1e-6 / 157.3, t3, flop / t3 * 1e-6 / 157.3, t4, flop / t4 * 1e-6 / 157.3, c0);
  return 0;
}
