// round 6: what does a dependent kernel cost inside a replayed hipGraph on this machine?  N kernels back to back on one captured stream:
// (a) empty, (b) one global load -> one store (a dependent round trip), (c) load -> load (pointer chase of 2) -> store, (d) 166 blocks x 256
// threads each streaming 16 B (the shape of the fold / optimizer kernels at one seed).  Prints us per kernel.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/launch_floor.hip -o /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct big_t { int w[96]; };
// ~6 KB of straight-line code per instantiation (768 dependent FMAs), distinct per ID: same kernel repeated vs four alternating
template <int ID> __global__ void k_code(float *b, float x) { float v = x + ID;
#pragma unroll
  for (int i = 0; i < 768; ++i) v = __builtin_fmaf(v, 1.0001f + 0.001f * (i + ID), 0.5f + i);
  if (v == 12345.f) b[0] = v; }
__global__ void k_empty() {}
__global__ void k_bigarg(big_t a, int *b) { if (threadIdx.x == 0) b[0] = a.w[95] + a.w[3]; }
__global__ void k_bigarg_load(big_t a, const int *src, int *b) { if (threadIdx.x == 0) b[0] = src[a.w[95]] + a.w[3]; }
__global__ void k_load1(const int *a, int *b) { if (threadIdx.x == 0) b[0] = a[0] + 1; }
__global__ void k_load2(const int *a, const int *idx, int *b) { if (threadIdx.x == 0) b[0] = a[idx[0]] + 1; }
__global__ void k_stream(const float4 *a, float4 *b) { const int i = blockIdx.x * 256 + threadIdx.x; float4 v = a[i]; v.x += 1.f; b[i] = v; }
__global__ void k_read(const float4 *a, float4 *b) { const int i = blockIdx.x * 256 + threadIdx.x; float4 v = a[i]; if (v.x == 12345.f) b[i] = v; }
__global__ void k_write(float4 *b, float x) { const int i = blockIdx.x * 256 + threadIdx.x; b[i] = float4{x, x, x, x}; }
__global__ void k_write_nt(float4 *b, float x) { const int i = blockIdx.x * 256 + threadIdx.x; float *p = reinterpret_cast<float *>(b + i);
  __builtin_nontemporal_store(x, p); __builtin_nontemporal_store(x, p + 1); __builtin_nontemporal_store(x, p + 2); __builtin_nontemporal_store(x, p + 3); }
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void k_stream_nt(const f4v *a, f4v *b) { const int i = blockIdx.x * 256 + threadIdx.x; f4v v = __builtin_nontemporal_load(a + i); v.x += 1.f; __builtin_nontemporal_store(v, b + i); }
__global__ void k_stream_ntst(const f4v *a, f4v *b) { const int i = blockIdx.x * 256 + threadIdx.x; f4v v = a[i]; v.x += 1.f; __builtin_nontemporal_store(v, b + i); }
int main() {
  const int N = 2000;
  int *a, *b, *idx; float4 *fa, *fb;
  CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096)); CK(hipMalloc(&idx, 4096)); CK(hipMemset(a, 0, 4096)); CK(hipMemset(idx, 0, 4096));
  CK(hipMalloc(&fa, 16 * 166 * 256 * 16)); CK(hipMalloc(&fb, 16 * 166 * 256 * 16)); CK(hipMemset(fa, 0, 16 * 166 * 256 * 16));
  hipStream_t st; CK(hipStreamCreate(&st));
  big_t big = {}; big.w[95] = 0; big.w[3] = 7;
  for (int mode = 0; mode < 18; ++mode) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) {
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); break;
        case 1: hipLaunchKernelGGL(k_empty, dim3(166), dim3(256), 0, st); break;
        case 2: hipLaunchKernelGGL(k_load1, dim3(1), dim3(64), 0, st, a, b); break;
        case 3: hipLaunchKernelGGL(k_load2, dim3(1), dim3(64), 0, st, a, idx, b); break;
        case 4: hipLaunchKernelGGL(k_stream, dim3(166), dim3(256), 0, st, (const float4 *)((i & 1) ? fb : fa), (i & 1) ? fa : fb); break;
        case 5: hipLaunchKernelGGL(k_stream, dim3(166 * 16), dim3(256), 0, st, (const float4 *)fa, fb); break;
        case 6: hipLaunchKernelGGL(k_read, dim3(166), dim3(256), 0, st, (const float4 *)fa, fb); break;
        case 7: hipLaunchKernelGGL(k_write, dim3(166), dim3(256), 0, st, fb, (float)i); break;
        case 8: hipLaunchKernelGGL(k_write_nt, dim3(166), dim3(256), 0, st, fb, (float)i); break;
        case 9: hipLaunchKernelGGL(k_stream_nt, dim3(166), dim3(256), 0, st, (const f4v *)((i & 1) ? fb : fa), (f4v *)((i & 1) ? fa : fb)); break;
        case 10: hipLaunchKernelGGL(k_stream, dim3(166), dim3(256), 0, st, (const float4 *)fa, fb); break;
        case 11: hipLaunchKernelGGL(k_stream_ntst, dim3(166), dim3(256), 0, st, (const f4v *)((i & 1) ? fb : fa), (f4v *)((i & 1) ? fa : fb)); break;
        case 12: hipLaunchKernelGGL(k_stream, dim3(8), dim3(256), 0, st, (const float4 *)((i & 1) ? fb : fa), (i & 1) ? fa : fb); break;
        case 16: hipLaunchKernelGGL(k_code<0>, dim3(166), dim3(256), 0, st, (float *)b, (float)i); break;
        case 17: switch (i & 3) { case 0: hipLaunchKernelGGL(k_code<0>, dim3(166), dim3(256), 0, st, (float *)b, (float)i); break; case 1: hipLaunchKernelGGL(k_code<1>, dim3(166), dim3(256), 0, st, (float *)b, (float)i); break;
                   case 2: hipLaunchKernelGGL(k_code<2>, dim3(166), dim3(256), 0, st, (float *)b, (float)i); break; default: hipLaunchKernelGGL(k_code<3>, dim3(166), dim3(256), 0, st, (float *)b, (float)i); break; } break;
        case 14: hipLaunchKernelGGL(k_bigarg, dim3(1), dim3(64), 0, st, big, b); break;
        case 15: hipLaunchKernelGGL(k_bigarg_load, dim3(1), dim3(64), 0, st, big, (const int *)a, b); break;
        case 13: hipLaunchKernelGGL(k_stream, dim3(1), dim3(64), 0, st, (const float4 *)((i & 1) ? fb : fa), (i & 1) ? fa : fb); break;
      }
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5.0 * N);
    const char *names[] = {"empty 1x64", "empty 166x256", "load->store 1x64", "load->load->store 1x64", "stream 166x256 (ping-pong, dependent)", "stream 2656x256", "read-only 166x256", "write-only 166x256", "write-only nontemporal 166x256",
                           "ping-pong nontemporal load+store 166x256", "stream 166x256 same direction (fa -> fb)", "ping-pong, nontemporal store only 166x256",
                           "ping-pong 8x256", "ping-pong 1x64", "384-byte by-value argument -> store 1x64", "384-byte argument -> load -> store 1x64", "6 KB of code, same kernel 166x256", "6 KB of code, four kernels alternating 166x256"};
    printf("%-40s %.2f us per kernel", names[mode], us);
    if (mode == 4 || mode == 9 || mode == 11 || mode == 12 || mode == 13) {   // ping-pong chains: every kernel adds 1 to what its predecessor wrote
      float x = -1.f;
      CK(hipMemcpy(&x, (N & 1) ? (void *)fb : (void *)fa, 4, hipMemcpyDeviceToHost));
      printf("   (element 0 after 6 replays = %.0f, expected a multiple of %d: %s)", x, N, ((long long)x % N) == 0 ? "serialised" : "RACED");
      CK(hipMemset(fa, 0, 16 * 166 * 256 * 16)); CK(hipMemset(fb, 0, 16 * 166 * 256 * 16));
    }
    printf("\n");
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  // the same empty kernels launched eagerly on the stream (no graph)
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
  CK(hipStreamSynchronize(st));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
  CK(hipStreamSynchronize(st));
  printf("%-40s %.2f us per kernel\n", "empty 1x64, eager", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N);
  return 0;
}
