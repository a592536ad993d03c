// ipc_probe.hip -- probes the mechanisms of the peer-store exchange transport (csrc/mrf_comm.hip) on one box:
//   * does hipIpcGetMemHandle / hipIpcOpenMemHandle work for fine-grained / uncached / plain device memory,
//   * can two processes' kernels run concurrently on one GPU and hand data to each other through flags,
//   * what one flag round trip costs.
// Two processes (fork before any HIP call), each allocates a buffer, maps the other's, and runs a kernel that
// K times: stores a payload into the peer's buffer, fences, sets the peer's flag, then polls its own flag with a
// bounded spin (a missed hand-over ends the kernel with an error code instead of hanging the GPU).
//   hipcc --offload-arch=gfx950 -O2 -o ipc_probe tools/ipc_probe.hip && ./ipc_probe [mode 0|1|2] [K] [payload]
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "[%d] %s -> %s\n", g_rank, #x, hipGetErrorString(e_));       \
      exit(3);                                                                     \
    }                                                                              \
  } while (0)
static int g_rank = -1;

struct Buf {
  unsigned long long flag[64];  // flag[0] used; own cache lines
  double payload[1 << 16];
};

__global__ void k_pingpong(Buf* own, Buf* peer, int rank, int K, int npay, long long timeout_ticks, int* result) {
  // one wave; lane l moves payload[l], lane 0 handles the flag
  const int lane = threadIdx.x;
  int bad = 0;
  for (int k = 1; k <= K; ++k) {
    for (int i = lane; i < npay; i += 64) peer->payload[i] = (double)(k * 1000 + rank * 100) + i;
    __threadfence_system();
    __syncthreads();
    if (lane == 0) __hip_atomic_store(&peer->flag[0], (unsigned long long)k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    bool ok = true;
    if (lane == 0) {
      while (__hip_atomic_load(&own->flag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)k) {
        if (wall_clock64() - t0 > timeout_ticks) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    ok = __shfl((int)ok, 0) != 0;
    if (!ok) {
      if (lane == 0) result[0] = -k;
      return;
    }
    __syncthreads();
    const int other = 1 - rank;
    for (int i = lane; i < npay; i += 64) {
      // the next iteration's payload may already have landed (the peer runs ahead by at most one step)
      const double v = __hip_atomic_load(&own->payload[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const double want = (double)(k * 1000 + other * 100) + i, want2 = (double)((k + 1) * 1000 + other * 100) + i;
      if (v != want && v != want2) ++bad;
    }
    __syncthreads();
  }
  atomicAdd(&result[1], bad);
  if (lane == 0) result[0] = K;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;  // 0 fine-grained, 1 uncached, 2 plain hipMalloc
  const int K = argc > 2 ? atoi(argv[2]) : 1000;
  const int npay = argc > 3 ? atoi(argv[3]) : 54;
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 2;
  pid_t pid = fork();  // before any HIP call
  g_rank = pid == 0 ? 1 : 0;
  const int rfd = g_rank == 0 ? c2p[0] : p2c[0], wfd = g_rank == 0 ? p2c[1] : c2p[1];
  CK(hipSetDevice(0));
  int can_wait = -1;
  (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
  Buf* own = nullptr;
  if (mode == 2)
    CK(hipMalloc((void**)&own, sizeof(Buf)));
  else
    CK(hipExtMallocWithFlags((void**)&own, sizeof(Buf), mode == 0 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
  CK(hipMemset(own, 0, sizeof(Buf)));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t mine, theirs;
  CK(hipIpcGetMemHandle(&mine, own));
  if (write(wfd, &mine, sizeof(mine)) != (ssize_t)sizeof(mine)) return 2;
  if (read(rfd, &theirs, sizeof(theirs)) != (ssize_t)sizeof(theirs)) return 2;
  Buf* peer = nullptr;
  CK(hipIpcOpenMemHandle((void**)&peer, theirs, hipIpcMemLazyEnablePeerAccess));
  int* result = nullptr;
  CK(hipMalloc((void**)&result, 2 * sizeof(int)));
  CK(hipMemset(result, 0, 2 * sizeof(int)));
  CK(hipDeviceSynchronize());
  char go = 'g';  // both sides mapped: start together
  if (write(wfd, &go, 1) != 1 || read(rfd, &go, 1) != 1) return 2;
  int rate_khz = 100000;
  (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
  const long long timeout_ticks = (long long)rate_khz * 2000;  // 2 s
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(64), 0, 0, own, peer, g_rank, K, npay, timeout_ticks, result);
  CK(hipEventRecord(b, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  int res[2];
  CK(hipMemcpy(res, result, sizeof(res), hipMemcpyDeviceToHost));
  printf("[rank %d] mode %d can_stream_wait %d wallclock_khz %d: completed %d of %d hand-overs, %d bad payload words, %.3f ms total, %.2f us per round\n",
         g_rank, mode, can_wait, rate_khz, res[0], K, res[1], ms, 1e3 * ms / K);
  if (write(wfd, &go, 1) != 1 || read(rfd, &go, 1) != 1) return 2;  // keep the mapping alive until both are done
  CK(hipIpcCloseMemHandle(peer));
  CK(hipFree(own));
  int status = 0;
  if (g_rank == 0) waitpid(pid, &status, 0);
  return (res[0] == K && res[1] == 0 && status == 0) ? 0 : 1;
}
