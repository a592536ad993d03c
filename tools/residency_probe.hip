// residency_probe.hip -- how many single-wave workgroups with the footprint of the persistent peer kernel are CO-RESIDENT on
// the device?  Every workgroup counts itself in and spins until the count reaches the grid size or a timeout: a grid that is
// not fully resident never gets there (its first workgroups give up, which is counted).  Variants: REGS = the workgroup
// really holds 512 registers per lane (one wave per SIMD; otherwise only the 37 KB of LDS limit a CU to four), SCR = it
// also needs scratch memory.  Used in round 6 to separate "the device cannot place the grid" from "several processes
// sharing one device cannot" (DESIGN.md section 6).
//   hipcc --offload-arch=gfx950 -O2 -o tools/residency_probe.out tools/residency_probe.hip
//   tools/residency_probe.out [repeats] [grid ...]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

template <bool REGS, bool SCR>
__global__ __launch_bounds__(64) void k_probe(unsigned* count, unsigned* seen, long long timeout_ticks, int idx) {
  __shared__ double lds[37 * 128];
  lds[threadIdx.x] = (double)blockIdx.x;
  if constexpr (REGS) {
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
  }
  double spill[24];
  if constexpr (SCR) {
#pragma unroll 1
    for (int i = 0; i < 24; ++i) spill[i] = lds[threadIdx.x] + i;
    lds[64 + threadIdx.x] = spill[idx & 15];  // runtime index: the array lives in scratch memory
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(count, 1u);
    const long long t0 = wall_clock64();
    bool timed_out = true;
    do {
      const unsigned c = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c >= gridDim.x) {
        timed_out = false;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    } while (wall_clock64() - t0 < timeout_ticks);
    if (timed_out) atomicAdd(seen, 1u);  // a workgroup that gave up: the grid was not co-resident
    if (lds[1] < 0) count[1] = 1;
  }
}

template <bool REGS, bool SCR>
void run(const char* name, const std::vector<int>& grids, int repeats, int rate, unsigned* d) {
  int per_cu = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_probe<REGS, SCR>, 64, 0);
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_probe<REGS, SCR>));
  printf("%s: %d registers, %zu B scratch per lane, %zu B LDS; occupancy API %d per CU\n", name, fa.numRegs, fa.localSizeBytes,
         fa.sharedSizeBytes, per_cu);
  for (int g : grids) {
    unsigned bad = 0;
    for (int r = 0; r < repeats; ++r) {
      hipMemset(d, 0, 16);
      hipLaunchKernelGGL((k_probe<REGS, SCR>), dim3(g), dim3(64), 0, 0, d, d + 2, (long long)rate * 100, r);  // 100 ms
      hipDeviceSynchronize();
      unsigned h[4];
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      bad += h[2] != 0;
    }
    printf("  grid %5d: %d of %d launches NOT co-resident\n", g, bad, repeats);
  }
}

int main(int argc, char** argv) {
  int cus = 0, rate = 100000;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  const int repeats = argc > 1 ? atoi(argv[1]) : 3;
  std::vector<int> grids;
  for (int i = 2; i < argc; ++i) grids.push_back(atoi(argv[i]));
  if (grids.empty()) grids = {4 * cus, 4 * cus - 8, 3 * cus, 2 * cus, cus, 4 * cus + 8};
  unsigned* d;
  hipMalloc(&d, 16);
  printf("CUs %d\n", cus);
  run<false, false>("LDS only", grids, repeats, rate, d);
  run<true, false>("512 registers", grids, repeats, rate, d);
  run<true, true>("512 registers + scratch", grids, repeats, rate, d);
  return 0;
}
