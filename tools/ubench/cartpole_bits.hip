// which intermediate of the CartPole step differs between device and host arithmetic? (debug aid)
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off cartpole_bits.hip -o cartpole_bits
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define HD __host__ __device__ inline
HD void sc(float x, float &s, float &c) {
  const float k = __builtin_rintf(x * 0.636619772367581343f);
  float r = x - k * 1.5703125f;
  r = r - k * 4.837512969970703125e-4f;
  r = r - k * 7.54978995489188216e-8f;
  const float z = r * r;
  const float sp = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
  const float cp = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
  const int q = ((int)k) & 3;
  s = (q == 0) ? sp : (q == 1) ? cp : (q == 2) ? -sp : -cp;
  c = (q == 0) ? cp : (q == 1) ? -sp : (q == 2) ? -cp : sp;
}
HD void step(float theta, float theta_dot, int action, float *o) {
  const float gravity = 9.8f, masspole = 0.1f, total_mass = 1.1f, length = 0.5f, polemass_length = 0.05f, force_mag = 10.0f;
  const float force = force_mag * (float)action - force_mag * (float)(1 - action);
  float sn, cs;
  sc(theta, sn, cs);
  const float t1 = polemass_length * (theta_dot * theta_dot) * sn;
  const float temp = (force + t1) / total_mass;
  const float d1 = masspole * (cs * cs) / total_mass;
  const float den = length * (4.0f / 3.0f - d1);
  const float num = gravity * sn - cs * temp;
  const float thetaacc = num / den;
  const float xacc = temp - polemass_length * thetaacc * cs / total_mass;
  o[0] = sn; o[1] = cs; o[2] = t1; o[3] = temp; o[4] = d1; o[5] = den; o[6] = num; o[7] = thetaacc; o[8] = xacc;
}
__global__ void k(const float *th, const float *thd, const int *a, float *out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) step(th[i], thd[i], a[i], out + 9 * i);
}
int main() {
  const int n = 1 << 16;
  float *th = (float *)malloc(n * 4), *thd = (float *)malloc(n * 4), *ho = (float *)malloc(n * 36), *go = (float *)malloc(n * 36);
  int *a = (int *)malloc(n * 4);
  srand(3);
  for (int i = 0; i < n; ++i) { th[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.4f; thd[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.0f; a[i] = rand() & 1; }
  float *dth, *dthd, *dout; int *da;
  hipMalloc(&dth, n * 4); hipMalloc(&dthd, n * 4); hipMalloc(&da, n * 4); hipMalloc(&dout, n * 36);
  hipMemcpy(dth, th, n * 4, hipMemcpyHostToDevice); hipMemcpy(dthd, thd, n * 4, hipMemcpyHostToDevice); hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dth, dthd, da, dout, n);
  hipMemcpy(go, dout, n * 36, hipMemcpyDeviceToHost);
  const char *names[9] = {"sin", "cos", "t1", "temp", "d1", "den", "num", "thetaacc", "xacc"};
  int bad[9] = {0};
  for (int i = 0; i < n; ++i) { step(th[i], thd[i], a[i], ho + 9 * i); for (int j = 0; j < 9; ++j) bad[j] += memcmp(&ho[9 * i + j], &go[9 * i + j], 4) != 0; }
  for (int j = 0; j < 9; ++j) printf("%-9s mismatching: %d of %d\n", names[j], bad[j], n);
  return 0;
}
