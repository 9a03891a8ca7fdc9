// ds_read_b64_tr_b16 on gfx950: which LDS element lands in which (lane, slot)?  LDS holds u16 element e = its own index;
// lane l supplies byte address addr[l]; the result (4 x u16 per lane) is printed for two addressing patterns.
//   hipcc --offload-arch=gfx950 tools/ubench/tr_b16.hip -o /tmp/tr_b16 && /tmp/tr_b16
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const uint32_t *addr, uint32_t *out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)lds + addr[threadIdx.x];
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[2 * threadIdx.x] = r.x;
  out[2 * threadIdx.x + 1] = r.y;
}
int main() {
  uint32_t h_addr[64], h_out[128], *d_addr, *d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 3; ++pat) {
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, i = l & 15;
      if (pat == 0) h_addr[l] = (uint32_t)(l * 8);                                    // contiguous 8 B per lane
      else if (pat == 1) h_addr[l] = (uint32_t)(((4 * g + (i >> 2)) * 128 + 4 * (i & 3)) * 2);   // row-major [rows][128], 4 rows x 16 cols per group
      else h_addr[l] = (uint32_t)(((4 * g + (i >> 2)) * 128 + 32 + 8 * (i & 3)) * 2);            // strided column groups (every other quad)
    }
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    for (int l = 0; l < 64; ++l)
      printf("  lane %2d addr %5u(elem %4u): %5u %5u %5u %5u\n", l, h_addr[l], h_addr[l] / 2, h_out[2 * l] & 0xFFFF, h_out[2 * l] >> 16,
             h_out[2 * l + 1] & 0xFFFF, h_out[2 * l + 1] >> 16);
  }
  return 0;
}
