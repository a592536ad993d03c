// clock_chain.hip -- what s_memtime counts on this device (the unit behind mrf_rollout_clock / bench.py's
// effective_clock_ghz): a chain of N DEPENDENT v_fma_f64 in one wave, stamped with s_memtime (shader-cycle counter) and
// s_memrealtime (constant-rate wall clock) before and after.  Prints, for an idle chip (1 workgroup) and a full one
// (4 waves on every CU), shader cycles per dependent FMA -- an integer if s_memtime counts shader cycles -- and the shader
// clock = cycles / wall ticks x wall rate.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/clock_chain tools/clock_chain.hip && /tmp/clock_chain
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int N = 1 << 16;

__global__ __launch_bounds__(64) void chain(double* out, long long* stamps, double a, double b) {
  double x = a + threadIdx.x * 1e-9;
  const long long c0 = (long long)__builtin_readcyclecounter(), w0 = (long long)wall_clock64();
#pragma unroll 16
  for (int i = 0; i < N; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(a));
  const long long c1 = (long long)__builtin_readcyclecounter(), w1 = (long long)wall_clock64();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = c1 - c0;
    stamps[2 * blockIdx.x + 1] = w1 - w0;
  }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

int main() {
  int cus = 0, wall_khz = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  const int full = cus * 4;
  double* out;
  long long* st;
  hipMalloc(&out, sizeof(double) * 64 * full);
  hipMalloc(&st, sizeof(long long) * 2 * full);
  for (int blocks : {1, full}) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chain, dim3(blocks), dim3(64), 0, 0, out, st, 1.0, 0.999999);
    hipDeviceSynchronize();
    long long h[2];
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"workgroups\": %d, \"dependent_fma_f64\": %d, \"s_memtime_ticks\": %lld, \"wall_ticks\": %lld, \"wall_khz\": %d, "
           "\"s_memtime_ticks_per_fma\": %.4f, \"shader_ghz\": %.4f}\n",
           blocks, N, h[0], h[1], wall_khz, (double)h[0] / N, (double)h[0] / (double)h[1] * wall_khz * 1e-6);
  }
  return 0;
}
