// pqn_peer.hip -- one-shot all-reduce of the flat gradient bucket over peer-mapped buffers (env-sharded mode).
//
// With the envs of ONE seed sharded over the GPUs of a node, clip + RAdam must see the gradient averaged over the global
// minibatch (purejaxql/pqn_minatar.py:159-162,285-292): one all-reduce of 0.53 MB per optimizer step, 64 of them per
// update -- latency-bound.  Going through torch.distributed puts the host between the gradient and the optimizer kernels
// 64 times per update; here every rank exposes a staging region through hipIpc, maps its peers' regions, and the
// collective is two small kernels on the training stream (capturable in the update's hipGraph):
//
//   publish   copy the local gradient into this rank's staging buffer (double-buffered by step parity); the last block to
//             finish makes it visible (system-scope fence) and raises this rank's step flag
//   reduce    wait until every peer's flag has reached this step (bounded spin), then every rank sums the W staging
//             buffers IN RANK ORDER -- all ranks get the same bits -- scales by 1 / W and writes its own gradient
//
// xGMI is point-to-point, W <= 8 on a node and the bucket is small, so "everyone reads everyone" is one hop per peer and
// 7 x 0.53 MB per rank and step; no ring, no intermediate buffers.  Reuse of a staging buffer two steps later is safe:
// a rank reaches the publish of step s + 2 only after its reduce of step s + 1 saw every peer's flag at s + 2, i.e. after
// every peer finished reading step s.  Peer data and flags are read with system-scope loads (no stale L2 lines for
// memory another device writes); the regions are fine-grained allocations.
#include <string.h>

#include "pqn_common.h"

// each of the two staging buffers holds PEER_STRIDE(n) floats: a multiple of 4, so both start 16-byte aligned
#define PEER_STRIDE(n) (((size_t)(n) + 3) & ~(size_t)3)
#define PEER_FLAG_OFFSET_BYTES(n) (((PEER_STRIDE(n) * 2 * sizeof(float)) + 255) & ~(size_t)255)
// A missing peer ends in an error word, not in a hung GPU -- but only after a WALL-CLOCK time-out (round 4; rounds 1-3
// counted polls, ~1-5 s depending on the xGMI load latency): s_memrealtime ticks of the constant 100 MHz clock, option
// "peer_timeout_s" / PQN_PEER_TIMEOUT_S, default 60 s -- rank skew of a few seconds (a slow graph instantiate, a host
// stall, logging on one rank) is normal and must not turn into silently unsynchronised gradients.
#define PEER_TICKS_PER_S 100000000ull

extern "C" int64_t pqn_peer_region_bytes(int64_t n) { return n > 0 ? (int64_t)PEER_FLAG_OFFSET_BYTES(n) + 256 : -1; }

extern "C" int pqn_peer_alloc(int64_t bytes, void **ptr, uint8_t *handle64) {
  PQN_REQUIRE(bytes > 0 && ptr && handle64, "pqn_peer_alloc: bad argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
  void *p = nullptr;
  if (hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    pqn_set_error("pqn_peer_alloc: hipExtMallocWithFlags(%lld bytes, fine-grained) failed", (long long)bytes);
    return PQN_E_HIP;
  }
  hipIpcMemHandle_t h;
  if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipIpcGetMemHandle(&h, p) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(p);
    pqn_set_error("pqn_peer_alloc: hipIpcGetMemHandle failed (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
    return PQN_E_HIP;
  }
  memcpy(handle64, &h, 64);
  *ptr = p;
  return PQN_OK;
}

extern "C" int pqn_peer_open(const uint8_t *handle64, void **ptr) {
  PQN_REQUIRE(handle64 && ptr, "pqn_peer_open: NULL argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, 64);
  if (hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
    (void)hipGetLastError();
    pqn_set_error("pqn_peer_open: hipIpcOpenMemHandle failed");
    return PQN_E_HIP;
  }
  return PQN_OK;
}

extern "C" int pqn_peer_close(void *ptr) { return ptr && hipIpcCloseMemHandle(ptr) != hipSuccess ? PQN_E_HIP : PQN_OK; }
extern "C" int pqn_peer_free(void *ptr) { return ptr && hipFree(ptr) != hipSuccess ? PQN_E_HIP : PQN_OK; }

namespace {

struct PeerPtrs {
  float *send[PQN_PEER_MAX];
  unsigned *flag[PQN_PEER_MAX];
};

// local_state: [0] steps published so far, [1] ticket of the publish kernel, [2] error word (1 = a peer never arrived)
__global__ __launch_bounds__(256) void peer_publish_kernel(const float *__restrict__ grad, long long n, float *send_mine,
                                                           unsigned *flag_mine, unsigned *local_state) {
  const unsigned seq = local_state[0];   // every block reads it before the last block (below) advances it
  float *dst = send_mine + (size_t)(seq & 1u) * PEER_STRIDE(n);
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
    reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(grad)[i];
  for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = grad[i];
  __threadfence_system();   // this block's part of the staging buffer is visible to the other devices
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(&local_state[1], 1u);
    if (t == gridDim.x - 1) {   // last block: everything is out
      local_state[1] = 0u;
      local_state[0] = seq + 1u;
      __threadfence_system();
      __hip_atomic_store(flag_mine, seq + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(256) void peer_reduce_kernel(float *__restrict__ grad, long long n, PeerPtrs P, int rank, int world,
                                                          unsigned *local_state, unsigned long long timeout_ticks) {
  const unsigned target = local_state[0];   // the publish kernel of this step ran before us on the stream
  __shared__ unsigned s_bad;                // this block saw a time-out (now or at an earlier step)
  if (threadIdx.x == 0) s_bad = local_state[2];
  __syncthreads();
  if (threadIdx.x < world && (int)threadIdx.x != rank && s_bad == 0u) {   // after one time-out: fail fast
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(P.flag[threadIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
      if (wall_clock64() - t0 > timeout_ticks) {
        local_state[2] = 1u;                       // sticky error word: pqn_peer_status / PeerAllReduce.check()
        local_state[3] = threadIdx.x + 1u;         // which peer never arrived (1-based)
        s_bad = 1u;
        break;
      }
      __builtin_amdgcn_s_sleep(32);
    }
  }
  __syncthreads();
  if (s_bad != 0u) {
    // a peer never arrived: the staging buffers hold some older step.  The bucket is poisoned instead of averaged, so that
    // nothing downstream (optimizer step, metrics row, checkpoint) can pass for a synchronised result before the host
    // has read the error word
    const float qnan = __uint_as_float(0x7fc00000u);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) grad[i] = qnan;
    return;
  }
  const size_t off = (size_t)((target - 1u) & 1u) * PEER_STRIDE(n);
  const float scale = 1.0f / (float)world;
  const long long n2 = n >> 1;
  const long long stride = (long long)gridDim.x * 256;
  // 8-byte system-scope loads: never served from a stale line of this device's caches
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n2; i += stride) {
    float a0 = 0.0f, a1 = 0.0f;
    for (int p = 0; p < world; ++p) {
      const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(P.send[p] + off) + i,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      a0 += __uint_as_float((unsigned)v);
      a1 += __uint_as_float((unsigned)(v >> 32));
    }
    reinterpret_cast<float2 *>(grad)[i] = make_float2(a0 * scale, a1 * scale);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    float a = 0.0f;
    for (int p = 0; p < world; ++p)
      a += __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(P.send[p] + off) + (n - 1), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_SYSTEM));
    grad[n - 1] = a * scale;
  }
}

}  // namespace

extern "C" int pqn_peer_allreduce_mean(const pqn_peers_t *P, float *grad, void *stream) {
  PQN_REQUIRE(P && grad, "pqn_peer_allreduce_mean: NULL argument");
  PQN_REQUIRE(P->world >= 1 && P->world <= PQN_PEER_MAX && P->rank >= 0 && P->rank < P->world && P->n > 0 && P->local_state,
              "pqn_peer_allreduce_mean: bad peer table (rank %d of %d, n %lld)", P->rank, P->world, (long long)P->n);
  PQN_REQUIRE(((uintptr_t)grad & 15) == 0, "pqn_peer_allreduce_mean: the bucket must be 16-byte aligned");
  PeerPtrs pp = {};
  for (int r = 0; r < P->world; ++r) {
    PQN_REQUIRE(P->region[r], "pqn_peer_allreduce_mean: region of rank %d is not mapped", r);
    pp.send[r] = reinterpret_cast<float *>(P->region[r]);
    pp.flag[r] = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(P->region[r]) + PEER_FLAG_OFFSET_BYTES(P->n));
  }
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)min((long long)64, (P->n / 4 + 255) / 256 + 1);   // small grids: always co-resident, never in the way
  hipLaunchKernelGGL(peer_publish_kernel, dim3(blocks), dim3(256), 0, st, grad, (long long)P->n, pp.send[P->rank], pp.flag[P->rank],
                     P->local_state);
  const int tsec = pqn_opt(PQN_OPT_PEER_TIMEOUT_S);
  const unsigned long long ticks = (unsigned long long)(tsec > 0 ? tsec : 60) * PEER_TICKS_PER_S;
  hipLaunchKernelGGL(peer_reduce_kernel, dim3(blocks), dim3(256), 0, st, grad, (long long)P->n, pp, P->rank, P->world, P->local_state,
                     ticks);
  return pqn_check_launch("pqn_peer_allreduce_mean");
}

extern "C" int pqn_peer_status(const pqn_peers_t *P, int32_t *error_out) {
  PQN_REQUIRE(P && P->local_state && error_out, "pqn_peer_status: NULL argument");
  unsigned st[4] = {0, 0, 0, 0};
  if (hipMemcpy(st, P->local_state, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    pqn_set_error("pqn_peer_status: hipMemcpy failed");
    return PQN_E_HIP;
  }
  *error_out = st[2] ? (int32_t)(st[3] ? st[3] : 1u) : 0;   // 0 = fine, r + 1 = rank r never published in time
  return PQN_OK;
}
