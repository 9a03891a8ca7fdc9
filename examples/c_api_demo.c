/* c_api_demo.c -- the drop-in boundary used from plain C: no Python, no PyTorch.
 *
 * What a maintainer of another runtime would write against include/pqn_hotpath.h: own the HBM buffers
 * (hipMalloc), hand raw pointers + a stream to the library.  Here: gymnax.make("Breakout-MinAtar"), vmap_reset over
 * N envs, then T x (uniform-random actions from the host, vmap_step with auto-reset + LogWrapper), and the mean of the
 * last completed episode returns -- the same surface purejaxql/pqn_minatar.py:103-112 drives.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I include examples/c_api_demo.c \
 *       -L purejaxql_amd/csrc -lpqn_hip -L /opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/purejaxql_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/c_api_demo && /tmp/c_api_demo
 * (plain gcc: the header needs no C++ and no HIP compiler.  Compile- and link-checked on the build box
 *  (tests/test_host_cpu.py), where it runs up to the first device call; the rest needs an MI355X)
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pqn_hotpath.h"

#define CHECK_HIP(x)                                                      \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      return 1;                                                           \
    }                                                                     \
  } while (0)
#define CHECK_PQN(x)                                                      \
  do {                                                                    \
    if ((x) != PQN_OK) {                                                  \
      fprintf(stderr, "%s: %s\n", #x, pqn_last_error());                  \
      return 1;                                                           \
    }                                                                     \
  } while (0)

int main(void) {
  const int32_t n = 4096, steps = 200;
  const int env = pqn_env_id("Breakout-MinAtar");
  if (env < 0) {
    fprintf(stderr, "%s\n", pqn_last_error());
    return 1;
  }
  pqn_env_spec_t spec;
  CHECK_PQN(pqn_env_spec(env, &spec));
  printf("obs %dx%dx%d, %d actions, %d state words, %d packed obs words, library version %d\n", spec.obs_dim[0],
         spec.obs_dim[1], spec.obs_dim[2], spec.num_actions, spec.state_words, spec.obs_words, pqn_version());

  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  uint32_t *state, *bits;
  int32_t *action, *rel;
  float *reward, *rer;
  uint8_t *done;
  CHECK_HIP(hipMalloc((void **)&state, sizeof(uint32_t) * spec.state_words * n));
  CHECK_HIP(hipMalloc((void **)&bits, sizeof(uint32_t) * spec.obs_words * n));
  CHECK_HIP(hipMalloc((void **)&action, sizeof(int32_t) * n));
  CHECK_HIP(hipMalloc((void **)&reward, sizeof(float) * n));
  CHECK_HIP(hipMalloc((void **)&rer, sizeof(float) * n));
  CHECK_HIP(hipMalloc((void **)&rel, sizeof(int32_t) * n));
  CHECK_HIP(hipMalloc((void **)&done, n));

  const uint64_t key = 0x123456789abcdefull;
  /* vmap_reset(n)(key): packed observations only (the f32 [n,10,10,C] surface is optional: obs = NULL) */
  CHECK_PQN(pqn_env_reset(env, n, pqn_fold_in(key, 0), state, NULL, bits, st));

  int32_t *h_action = (int32_t *)malloc(sizeof(int32_t) * n);
  float *h_rer = (float *)malloc(sizeof(float) * n);
  int32_t *h_rel = (int32_t *)malloc(sizeof(int32_t) * n);
  uint32_t lcg = 12345u;
  for (int t = 0; t < steps; ++t) {
    for (int i = 0; i < n; ++i) {
      lcg = lcg * 1664525u + 1013904223u;
      h_action[i] = (int32_t)((lcg >> 16) % (uint32_t)spec.num_actions);
    }
    CHECK_HIP(hipMemcpyAsync(action, h_action, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    pqn_step_out_t out;
    memset(&out, 0, sizeof(out));
    out.obs_bits = bits;
    out.reward = reward;
    out.done = done;
    out.returned_episode_returns = rer;
    out.returned_episode_lengths = rel;
    /* vmap_step(n)(key_t, state, action): in place (state_out == state_in), auto-reset and LogWrapper inside */
    CHECK_PQN(pqn_env_step(env, n, pqn_fold_in(key, 1 + (uint32_t)t), state, state, action, &out, st));
  }
  CHECK_HIP(hipMemcpyAsync(h_rer, rer, sizeof(float) * n, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipMemcpyAsync(h_rel, rel, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
  CHECK_HIP(hipStreamSynchronize(st));
  double sr = 0.0, sl = 0.0;
  for (int i = 0; i < n; ++i) {
    sr += h_rer[i];
    sl += h_rel[i];
  }
  printf("%d envs x %d random steps: mean returned_episode_returns %.3f, mean returned_episode_lengths %.1f\n", n, steps,
         sr / n, sl / n);
  free(h_action);
  free(h_rer);
  free(h_rel);
  return 0;
}
