// mrf_rollout_wp.hip -- translation unit of k_rollout_panda_wp (mrf_rollout_wp.hpp): the coupled joint-space rollout
// (FPJ:190-249) as a pair of waves per row.  Its own unit because it builds the device header in the SGPR-literal flavour
// (two waves per SIMD, no spare VGPRs for hoisted float64 literals), the other kernels in the VGPR-literal one.
// Opt-in since round 6 (-DMRF_WITH_WP, MRF_WITH_WP=1 for __graft_entry__.build()): the kernel measures 7-9 % slower than the
// row-per-lane kernel and is never auto-selected, so the default library does not carry it (mrf_build_has_wp()).
#ifdef MRF_WITH_WP
#define MRF_SGPR_CONST 1
#include <hip/hip_runtime.h>

#include "mrf_host.hpp"
#include "mrf_rollout_wp.hpp"

#ifdef MRF_WP_CLOCKS
extern "C" int mrf_debug_wp_clocks(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mrf::mrf_wp_clocks), sizeof(long long) * (n < 32 ? n : 32)) == hipSuccess ? 0 : -1;
}
#endif

int mrf_host::rollout_wave_pair(mrf_handle* h, int64_t n_scen, const void* q0, const void* qdot0, const void* params,
                                void* avg_out, void* traj_q, void* traj_qd, void* stream) {
  const int spw = 64 / h->cfg.n_robots;
  const dim3 grid((unsigned)((n_scen + spw - 1) / spw)), block(128);
  return launch(h, mrf::k_rollout_panda_wp<double, LeafSetPanda>, grid, block, (hipStream_t)stream,
                (const mrf::DevCfg<double>*)h->dcfg, n_scen, (const double*)q0, (const double*)qdot0, (const double*)params,
                (double*)avg_out, (double*)traj_q, (double*)traj_qd, (long long*)h->clock_probe, h->rollout_serial);
}
#endif  // MRF_WITH_WP
