// mrf_kernels.hip -- gfx950 kernels and the C ABI of include/mrf.h.
//
// Kernels (all templated on the scalar type; one thread per (scenario, robot) row unless noted):
//   k_action_panda / k_action_planar   batched compute_action, obstacles in HBM   (EXJ:441,444; pointmass :191-199)
//   k_action_coupled                   compute_action of all robots, obstacle assembly on chip (EXJ:394-448)
//   k_rollout_panda                    coupled joint-space Rollout Fabrics         (FPJ:190-249)
//   k_coop_panda                       the two coupled kernels for small batches: one WAVE per scenario
//   k_rollout_cart_panda               Cartesian constant-velocity rollout         (FPC:421-458)
//   k_fk_spheres_panda                 sphere x, v, jac_dot*qd                     (utils.py:16-54,87-119)
//   k_step_predict / k_step_action     the two halves of one robot-sharded rollout step
// The control-step glue (hand FK, deadlock logic, integration, episodes) lives in mrf_control.hip.
//
// Layout: every array is component-major over rows (a[c*rows + row]) so that consecutive lanes touch
// consecutive addresses on every load and store.
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "mrf_device.hpp"
#include "mrf_host.hpp"
#include "mrf_tile.hpp"

namespace mrf {

// -DMRF_COOP_CLOCKS: the cooperative kernel stamps s_memtime at its phase boundaries (block 0, lane 0, horizon step 5)
// into mrf_dbg_clocks, read back by mrf_debug_clocks() -- a development aid (tools/coop_phases.py), not built by default.
#ifdef MRF_COOP_CLOCKS
__device__ long long mrf_dbg_clocks[32];
#define MRF_STAMP(slot)                                                                      \
  do {                                                                                       \
    if (blockIdx.x == 0 && threadIdx.x == 0 && k == 5) mrf_dbg_clocks[slot] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define MRF_STAMP(slot)
#endif

// The thread-per-row kernels whose obstacle loop streams from HBM / L2 can be built for two waves per SIMD
// (<= 256 registers, two-phase walk) so that one wave's loads overlap the other's arithmetic: experiment switches
// -DMRF_OCC2_ACTION / _CART / _STEP (tools/build_variant.sh), measured by tools/prof_kernels.py.
#define MRF_WPE2 __attribute__((amdgpu_waves_per_eu(2, 2)))
#ifdef MRF_OCC2_ACTION
#define MRF_ATTR_ACTION MRF_WPE2
constexpr bool kActionSingleWalk = false;
#else
#define MRF_ATTR_ACTION
constexpr bool kActionSingleWalk = true;
#endif
#ifdef MRF_OCC2_STEP
#define MRF_ATTR_STEP MRF_WPE2
constexpr bool kStepSingleWalk = false;
#else
#define MRF_ATTR_STEP
constexpr bool kStepSingleWalk = true;
#endif
#ifdef MRF_OCC2_CART
#define MRF_ATTR_CART MRF_WPE2
#else
#define MRF_ATTR_CART
#endif
#ifdef MRF_CART_TWO_WALKS
constexpr bool kCartSingleWalk = false;
#else
constexpr bool kCartSingleWalk = true;
#endif

// obstacle loop over HBM arrays [n_obst][3][rows] (compute_action / Cartesian rollout); tk = elapsed obstacle time
// ACC = false: the caller passes no obstacle accelerations (oa == NULL -- what the reference's drivers do,
// FPC:33, EXJ:411): the three acceleration loads per obstacle and the n.a_o term of every leaf are compiled out.
// ADDR: how a lane finds its element of component `comp`: LaneAddr (a[comp*rows + row], per-lane 64-bit addresses) or
// RowAddr (wave-uniform base + 32-bit lane offset, mrf_device.hpp).
template <typename T>
struct LaneAddr {
  int64_t rows, r;
  __device__ __forceinline__ T load(const T* __restrict__ a, int64_t comp) const { return a[comp * rows + r]; }
};

template <class CL, bool ACC = true, typename T, class ADDR>
__device__ __forceinline__ void obstacles_from_arrays(const DevCfg<T>& cfg, const ADDR& addr, int n_obst,
                                                      int n_static, const T* __restrict__ ox, const T* __restrict__ ov,
                                                      const T* __restrict__ oa, const T* __restrict__ orad, T tk,
                                                      bool allow_planar, const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc,
                                                      int m_first = 0) {  // obstacles [m_first, n_obst)
  // buf: x[3], v[3], a[3], radius.  Missing arrays (NULL) and static obstacles read x in their place and are zeroed in
  // the fold, so that the fetch is the same ten loads for every obstacle (no divergent address arithmetic).
  const T* pv = ov ? ov : ox;
  const T* pa = oa ? oa : ox;
  constexpr int NV = ACC ? 10 : 7;  // x[3], v[3], (a[3],) radius
  pipelined_pairs<T, NV>(
      n_obst - m_first,
      [&](int mi, T (&buf)[NV]) {
        const int m = mi + m_first;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          buf[c] = addr.load(ox, m * 3 + c);
          buf[3 + c] = addr.load(pv, m * 3 + c);
          if constexpr (ACC) buf[6 + c] = addr.load(pa, m * 3 + c);
        }
        buf[NV - 1] = addr.load(orad, m);
      },
      [&](int mi, T (&buf)[NV]) {
        const int m = mi + m_first;
        const bool is_static = m < n_static;  // static leaves: full 3-D distance, no reference motion
        const bool has_v = ov && !is_static, has_a = ACC && oa && !is_static;
        T xo[3], vo[3], ao[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          vo[c] = has_v ? buf[3 + c] : T(0);
          xo[c] = buf[c] + tk * vo[c];  // Cartesian rollout: x += dt*v per step (FPC:448-453); tk = 0 otherwise
          if constexpr (ACC)
            ao[c] = has_a ? buf[6 + c] : T(0);
          else
            ao[c] = T(0);
        }
#ifdef MRF_OBST_NOFOLD  // development switch (tools/build_variant.sh): the obstacle stream alone, results meaningless
        acc.b[0][0] += xo[0] + xo[1] + xo[2] + vo[0] + vo[1] + vo[2] + ao[0] + ao[1] + ao[2] + buf[NV - 1];
#else
        accumulate_obstacle<CL>(cfg, E, xo, vo, ao, buf[NV - 1], allow_planar && !is_static && cfg.obst_dim == 2, acc);
#endif
      });
}

// The Cartesian rollout passes over the same obstacle set H times (x0, v, a, r are constants of the rollout; only the
// elapsed time changes, FPC:448-453).  The set does not fit on chip (16 obstacles = 82 KB per wave in f64), but a prefix
// does: the first CART_RESIDENT<T> obstacles of every row are copied once into a per-wave LDS tile [m][10][64] and folded
// from there in every step, the rest streams from HBM / the Infinity Cache as before.
// An obstacle is 10 scalars per row (x, v, a, radius) -- 7 when no accelerations are passed (ACC = false: FPC:33 presets
// them to zero and the example drivers never change that), so the 35 KB tile holds 7 (ACC) or 10 of a row's obstacles in
// f64 (14 / 20 in f32).
template <typename T, bool ACC>
constexpr int CART_RESIDENT = 35840 / ((ACC ? 10 : 7) * 64 * (int)sizeof(T));

template <bool ACC, typename T>
__device__ __forceinline__ void stage_resident_obstacles(T* __restrict__ res, int lane, int nres, int64_t rows, int64_t r,
                                                         const T* __restrict__ ox, const T* __restrict__ ov,
                                                         const T* __restrict__ oa, const T* __restrict__ orad) {
  constexpr int NV = ACC ? 10 : 7;
  const T* pv = ov ? ov : ox;  // missing arrays: any finite value, zeroed in the fold
  const T* pa = oa ? oa : ox;
#pragma unroll 1
  for (int m = 0; m < nres; ++m) {
    const int64_t base = (int64_t)(m * 3) * rows + r;
    T* dst = res + m * (NV * 64) + lane;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dst[c * 64] = ox[base + c * rows];
      dst[(3 + c) * 64] = pv[base + c * rows];
      if constexpr (ACC) dst[(6 + c) * 64] = pa[base + c * rows];
    }
    dst[(NV - 1) * 64] = orad[(int64_t)m * rows + r];
  }
}

template <class CL, bool ACC, typename T>
__device__ __forceinline__ void obstacles_resident(const DevCfg<T>& cfg, const T* __restrict__ res, int lane, int nres,
                                                   int n_static, bool any_v, bool any_a, T tk, const EgoPts<T, NG>& E,
                                                   EgoAcc<T, NG>& acc) {
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  constexpr int NV = ACC ? 10 : 7;
  pipelined_pairs<T, NV>(
      nres,
      [&](int m, T (&buf)[NV]) {
        lds_ptr src = (lds_ptr)(res + m * (NV * 64) + lane);
#pragma unroll
        for (int c = 0; c < NV; ++c) buf[c] = src[c * 64];
      },
      [&](int m, T (&buf)[NV]) {
        const bool is_static = m < n_static;
        const bool has_v = any_v && !is_static, has_a = ACC && any_a && !is_static;
        T xo[3], vo[3], ao[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          vo[c] = has_v ? buf[3 + c] : T(0);
          xo[c] = buf[c] + tk * vo[c];
          if constexpr (ACC)
            ao[c] = has_a ? buf[6 + c] : T(0);
          else
            ao[c] = T(0);
        }
        accumulate_obstacle<CL>(cfg, E, xo, vo, ao, buf[NV - 1], false, acc);
      });
}

// One pipelined loop over ALL obstacles of a row: the first nres come from the resident LDS tile, the rest from the HBM
// arrays (wave-uniform switch inside the fetch).  A single loop keeps one pair of ping-pong buffers and two inlined copies
// of the five-point fold alive instead of two loops with four.  Streamed loads use RowAddr (uniform base + lane offset).
template <class CL, bool ACC, typename T, class ADDR>
__device__ __forceinline__ void obstacles_cart(const DevCfg<T>& cfg, const T* __restrict__ res, int lane, int nres,
                                               const ADDR& ra, int n_obst, int n_static,
                                               const T* __restrict__ ox, const T* __restrict__ ov,
                                               const T* __restrict__ oa, const T* __restrict__ orad, T tk,
                                               const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  constexpr int NV = ACC ? 10 : 7;
  const T* pv = ov ? ov : ox;
  const T* pa = oa ? oa : ox;
  const bool any_v = ov != nullptr, any_a = ACC && oa != nullptr;
  pipelined_pairs<T, NV>(
      n_obst,
      [&](int m, T (&buf)[NV]) {
        if (m < nres) {
          lds_ptr src = (lds_ptr)(res + m * (NV * 64) + lane);
#pragma unroll
          for (int c = 0; c < NV; ++c) buf[c] = src[c * 64];
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            buf[c] = ra.load(ox, m * 3 + c);
            buf[3 + c] = ra.load(pv, m * 3 + c);
            if constexpr (ACC) buf[6 + c] = ra.load(pa, m * 3 + c);
          }
          buf[NV - 1] = ra.load(orad, m);
        }
      },
      [&](int m, T (&buf)[NV]) {
        const bool is_static = m < n_static;
        const bool has_v = any_v && !is_static, has_a = any_a && !is_static;
        T xo[3], vo[3], ao[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          vo[c] = has_v ? buf[3 + c] : T(0);
          xo[c] = buf[c] + tk * vo[c];  // x += dt*v per step (FPC:448-453)
          if constexpr (ACC)
            ao[c] = has_a ? buf[6 + c] : T(0);
          else
            ao[c] = T(0);
        }
        accumulate_obstacle<CL>(cfg, E, xo, vo, ao, buf[NV - 1], false, acc);
      });
}

// ---------------------------------------------------------------------------- compute_action
template <typename T, class LS, bool ACC>
__global__ __launch_bounds__(256) MRF_ATTR_ACTION void k_action_panda(const DevCfg<T>* __restrict__ cfgp, int64_t rows,
                                                       const T* __restrict__ q, const T* __restrict__ qd,
                                                       const T* __restrict__ prm, int n_obst, int n_static,
                                                       const T* __restrict__ ox, const T* __restrict__ ov,
                                                       const T* __restrict__ oa, const T* __restrict__ orad,
                                                       T* __restrict__ qdd_out, T* __restrict__ act_out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const DevCfg<T>& cfg = *cfgp;
  PandaState<T> R;
  load_state(rows, r, q, qd, R);
  PrmView<T> P{prm, rows, r, {T(0), T(0), T(0)}, false};
  T qdd[7], act[7];
  panda_solve_row<LS, kActionSingleWalk>(
      cfg, cfg.mount[(int)(r % cfg.n_robots)], R, P,
      [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
        obstacles_from_arrays<typename LS::Collision, ACC>(cfg, LaneAddr<T>{rows, r}, n_obst, n_static, ox, ov, oa, orad, T(0),
                                                           false, E, acc);
      },
      qdd, act);
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    if (qdd_out) qdd_out[j * rows + r] = qdd[j];
    act_out[j * rows + r] = act[j];
  }
}


template <typename T>
__global__ __launch_bounds__(256) void k_action_planar(const DevCfg<T>* __restrict__ cfgp, int64_t rows,
                                                        const T* __restrict__ q, const T* __restrict__ qd,
                                                        const T* __restrict__ prm, int n_obst, int n_static,
                                                        const T* __restrict__ ox, const T* __restrict__ ov,
                                                        const T* __restrict__ oa, const T* __restrict__ orad,
                                                        T* __restrict__ qdd_out, T* __restrict__ act_out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const DevCfg<T>& cfg = *cfgp;
  PlanarRow<T> R;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    R.q[j] = q[j * rows + r];
    R.qd[j] = qd[j * rows + r];
  }
#pragma unroll
  for (int p = 0; p < MRF_NPARAM; ++p) R.prm[p] = prm[p * rows + r];
  EgoPts<T, 1> E;
  planar_ego(R, E);
  EgoAcc<T, 1> acc;
  acc.zero();
  if (cfg.n_ego > 0) {
#pragma unroll 1
    for (int m = 0; m < n_obst; ++m) {
      const bool is_static = m < n_static;
      T xo[3], vo[3], ao[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int64_t idx = (int64_t)(m * 3 + c) * rows + r;
        xo[c] = ox[idx];
        vo[c] = (ov && !is_static) ? ov[idx] : T(0);
        ao[c] = (oa && !is_static) ? oa[idx] : T(0);
      }
      accumulate_obstacle<LeafGeneric>(cfg, E, xo, vo, ao, orad[(int64_t)m * rows + r],
                                       !is_static && cfg.obst_dim == 2, acc);
    }
  }
  T qdd[3], act[3];
  planar_finish_row(cfg, R, acc, qdd, act);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (qdd_out) qdd_out[j * rows + r] = qdd[j];
    act_out[j * rows + r] = act[j];
  }
}


// ---------------------------------------------------------------------------- coupled joint-space rollout
// One wave per block.  Lanes are (scenario, robot) pairs with the N robots of a scenario adjacent, so the
// exchange step of the recurrence (FPJ:211-225: every robot needs every other robot's spheres at step k)
// stays inside the wave.  LO = true (link-origin sphere table): lanes exchange their link states through the
// [72][64] tile above.  LO = false (any other table): each lane publishes cos q, sin q, qdot of its 7 joints to a
// 21 x 64 LDS tile and re-walks the other robots' chains from it, streaming their spheres into its leaf sums.
template <typename T, class LS, bool LO>
__global__ __launch_bounds__(64) void k_rollout_panda(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen,
                                                       const T* __restrict__ q0, const T* __restrict__ qd0,
                                                       const T* __restrict__ prm, T* __restrict__ avg_out,
                                                       T* __restrict__ traj_q, T* __restrict__ traj_qd,
                                                       long long* __restrict__ probe, long long serial) {
  __shared__ T xch[LO ? TILE_SCALARS : GEN_SCALARS];
  const DevCfg<T>& cfg = *cfgp;
  // clock probe (mrf_rollout_clock): the first and the last workgroup stamp the shader-cycle counter (s_memtime) and the
  // constant-rate wall clock on entry and on exit -- stored at once, nothing is carried through the step loop
  const bool probing = probe && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && threadIdx.x == 0;
  long long* const stamp = probe + (blockIdx.x == 0 ? 0 : 4);
  if (probing) {
    stamp[0] = (long long)__builtin_readcyclecounter();
    stamp[1] = (long long)wall_clock64();
    if (blockIdx.x == 0) probe[8] = serial;  // which mrf_rollout call these stamps belong to (mrf_rollout_clock checks it)
    if (gridDim.x == 1) {                    // one workgroup is first and last: both slots carry its stamps
      probe[4] = stamp[0];
      probe[5] = stamp[1];
    }
  }
  if constexpr (LO) stage_sphere_radii(cfg, xch, threadIdx.x);  // visible after the first publish barrier
  const int N = cfg.n_robots;
  const int spw = 64 / N;  // scenarios per wave
  const int lane = threadIdx.x;
  int ls = lane / N;
  const int li = lane - ls * N;
  int64_t scen = (int64_t)blockIdx.x * spw + ls;
  const bool active = ls < spw && scen < n_scen;
  if (ls >= spw) ls = 0;  // idle tail lanes shadow the wave's first scenario (no stores)
  if (scen >= n_scen || !active) scen = (int64_t)blockIdx.x * spw + ls;
  if (scen >= n_scen) scen = n_scen - 1;
  const int64_t rows = n_scen * N;
  const int64_t row = scen * N + li;

  PandaState<T> R;
  load_state(rows, row, q0, qd0, R);
  const T* mount_own = cfg.mount[li];
  PrmView<T> P{prm, rows, row, {T(0), T(0), T(0)}, false};

  if ((cfg.goal_mask >> li) & 1) {
    // RF-CV: the goal of this robot is not communicated; use x_ee + T * v_ee (EXC:355-357)
    PandaKin<T> K0;
    panda_walk_own<T>(mount_own, R.cq, R.sq, R.qd, K0);
#pragma unroll
    for (int c = 0; c < 3; ++c) P.g0[c] = K0.p8[c] + cfg.goal_T * K0.v8[c];
    P.own_goal = true;
  }

  T sumsq = T(0);
  const int H = cfg.horizon;
#pragma unroll 1
  for (int k = 0; k < H; ++k) {
    // system_step 'vel' (FPJ:77-80): q += dt*qdot.  cos q / sin q advance by the angle-sum formula while every
    // |dq| in the wave is small; a full sincos otherwise.
    T dq[7];
    bool small = true;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      dq[j] = cfg.dt * R.qd[j];
      small = small && (m_abs(dq[j]) < T(0.125));
      R.q[j] += dq[j];
    }
    if (__all(small)) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        T sd, cd;
        small_sincos(dq[j], sd, cd);
        const T c = R.cq[j] * cd - R.sq[j] * sd;
        const T s = R.sq[j] * cd + R.cq[j] * sd;
        R.cq[j] = c;
        R.sq[j] = s;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 7; ++j) m_sincos(R.q[j], &R.sq[j], &R.cq[j]);
    }
    T qdd[7], act[7];
    if constexpr (LO) {
      panda_solve_row<LS, kSingleWalk<LS>>(
          cfg, mount_own, R, P,
          [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
            obstacles_from_tile<typename LS::Collision>(cfg, xch, ls, li, N, E, acc);
          },
          qdd, act,
          [&](const PandaKin<T>& K1) {
            __syncthreads();
            publish_link_spheres(xch, lane, K1, cfg.dynamic != 0, cfg.dynamic != 0, cfg.jsign, cfg.lo_merge01, cfg.lo_merge45);  // FPJ:97-99,215-220
            __syncthreads();
          });
    } else {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        xch[(3 * j + 0) * 64 + lane] = R.cq[j];
        xch[(3 * j + 1) * 64 + lane] = R.sq[j];
        xch[(3 * j + 2) * 64 + lane] = R.qd[j];
      }
      __syncthreads();
      panda_solve_row<LS, false>(
          cfg, mount_own, R, P,
          [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
            obstacles_generic_chunked<typename LS::Collision>(cfg, xch, lane, ls, li, N, mount_own, cfg.dynamic != 0, cfg.jsign, E, acc);
          },
          qdd, act);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      R.qd[j] = act[j];  // FPJ:233
      sumsq += act[j] * act[j];
    }
    if (active && traj_q) {
#pragma unroll
      for (int j = 0; j < 7; ++j) traj_q[((int64_t)k * 7 + j) * rows + row] = R.q[j];
    }
    if (active && traj_qd) {
#pragma unroll
      for (int j = 0; j < 7; ++j) traj_qd[((int64_t)k * 7 + j) * rows + row] = R.qd[j];
    }
  }
  if (active) avg_out[row] = sumsq / (T)(H * 7);  // FPJ:102-116
  if (probing) {
    stamp[2] = (long long)__builtin_readcyclecounter();
    stamp[3] = (long long)wall_clock64();
    if (gridDim.x == 1) {
      probe[6] = stamp[2];
      probe[7] = stamp[3];
    }
  }
}

// Fold for the coupled Cartesian rollout: the tile holds the other robots' spheres as they are at the START of the horizon
// (x0 in rows s*9 + 0..2, v in rows s*9 + 3..5); a sphere is folded at x0 + tk v with zero acceleration (FPC:33,448-453) --
// the acceleration rows are not read (the generic publish parks the joint state there), the n.a_o terms compile out.
// nsp distinct spheres per robot; radius and multiplicity per slot follow the tile (TILE_RADII / TILE_MULT).
template <class CL, typename T>
__device__ __forceinline__ void obstacles_from_tile_drift(const DevCfg<T>& cfg, const T* __restrict__ tile, int ls, int li, int N,
                                                          int nsp, T tk, const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
  const int M = (N - 1) * nsp;
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  typedef const volatile __attribute__((address_space(3))) T* lds_vptr;
  auto address = [&](int d, int sp) {
    int jr = li + 1 + d;
    if (jr >= N) jr -= N;
    return (sp * 9) * 64 + ls * N + jr;
  };
  T bufA[8], bufB[8];  // x0[3], v[3], radius, multiplicity: ping-pong, the loop is unrolled by two
  {
    lds_vptr src = (lds_vptr)(tile + address(0, 0));  // volatile: keeps the first fetch out of the loop (obstacles_from_tile)
#pragma unroll
    for (int k = 0; k < 6; ++k) bufA[k] = src[k * 64];
    bufA[6] = ((lds_vptr)tile)[TILE_RADII];
    bufA[7] = ((lds_vptr)tile)[TILE_MULT];
  }
  if constexpr (!CL::generic) asm volatile("" ::"s"(cfg.jsign), "s"(cfg.cf.k), "s"(cfg.cg.k));
  int dn = 0, sn = 0;  // (other robot, slot) of the sphere fetched last
  auto fetch_next = [&](T (&buf)[8], bool advance) {
    if (advance) {
      if (++sn == nsp) {
        sn = 0;
        ++dn;
      }
    }
    lds_ptr src = (lds_ptr)(tile + address(dn, sn));
#pragma unroll
    for (int k = 0; k < 6; ++k) buf[k] = src[k * 64];
    buf[6] = ((lds_ptr)tile)[TILE_RADII + sn];
    buf[7] = ((lds_ptr)tile)[TILE_MULT + sn];
  };
  auto fold = [&](T (&buf)[8]) {
    T x[3] = {buf[0] + tk * buf[3], buf[1] + tk * buf[4], buf[2] + tk * buf[5]};
    const T zero[3] = {T(0), T(0), T(0)};
    accumulate_obstacle<CL>(cfg, E, x, buf + 3, zero, buf[6], false, acc, buf[7]);
  };
  int m = 0;
#pragma unroll 1
  for (; m + 1 < M; m += 2) {
    fetch_next(bufB, true);
    fold(bufA);
    fetch_next(bufA, m + 2 < M);  // past the end: re-reads the last sphere, never used
    fold(bufB);
  }
  if (m < M) fold(bufA);  // odd count
}

// Tables of more than eight spheres per robot (the reference's default, n_obst_per_link: 4 -> 32, panda_config.yaml:8,
// EXC:184): 32 x (x0, v) of 64 lanes are 98 KB, more than a wave's LDS, so the spheres are RE-DERIVED in every step from the
// robots' START joint states (cos q0, sin q0, qdot0: [21][64], staged once) -- every lane walks its own start chain, emitting
// its spheres in table order at x0 + tk v; they are exchanged CART_CH at a time through a [CART_CH][6][64] tile and folded by
// the other lanes of the scenario before the walk moves on (the pattern of obstacles_generic_chunked, without accelerations).
// ~40 instructions per sphere and step instead of seven loads from the obstacle arrays of k_publish_obstacles.
constexpr int CART_CH = 8;
constexpr int CART_ST0 = 21 * 64;
static_assert(CART_ST0 + CART_CH * 6 * 64 <= TILE_SCALARS, "start states + chunk tile must fit the [72][64] tile's storage");

template <class CL, typename T>
__device__ __forceinline__ void obstacles_start_chunked(const DevCfg<T>& cfg, T* __restrict__ st0, int lane, int ls, int li, int N,
                                                        const T* __restrict__ mount_own, bool dyn, T tk, const EgoPts<T, NG>& E,
                                                        EgoAcc<T, NG>& acc) {
  T* chunk = st0 + CART_ST0;
  const int S = cfg.n_spheres;
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  panda_walk_spheres<false, T>(
      cfg, mount_own,
      [&](int j, T& c, T& s, T& qdj) {
        c = st0[(3 * j + 0) * 64 + lane];
        s = st0[(3 * j + 1) * 64 + lane];
        qdj = st0[(3 * j + 2) * 64 + lane];
      },
      [&](int s, const T* x, const T* v, const T*) {
        const int k = s % CART_CH;
        if (k == 0) __syncthreads();  // the previous chunk has been folded by every lane
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const T vc = dyn ? v[c] : T(0);
          chunk[((k * 6) + c) * 64 + lane] = x[c] + tk * vc;   // constant-velocity obstacle (FPC:448-453)
          chunk[((k * 6) + 3 + c) * 64 + lane] = vc;
        }
        if (k != CART_CH - 1 && s != S - 1) return;
        __syncthreads();
        const int n = k + 1, s0 = s - k;  // spheres in this chunk, first sphere of the chunk
        pipelined_pairs<T, 6>(
            (N - 1) * n,
            [&](int m, T (&buf)[6]) {
              const int d = m / n, kk = m - d * n;
              int jr = li + 1 + d;
              if (jr >= N) jr -= N;
              lds_ptr src = (lds_ptr)(chunk + (kk * 6) * 64 + ls * N + jr);
#pragma unroll
              for (int c = 0; c < 6; ++c) buf[c] = src[c * 64];
            },
            [&](int m, T (&buf)[6]) {
              const int kk = m % n;
              const T zero[3] = {T(0), T(0), T(0)};
              accumulate_obstacle<CL>(cfg, E, buf, buf + 3, zero, cfg.sphere_r[s0 + kk], false, acc);
            });
      });
}

// ---------------------------------------------------------------------------- coupled Cartesian rollout (up to 8 spheres per robot)
// mrf_rollout_cartesian_coupled at throughput batch sizes: every robot rolls out its OWN fabric against the other robots'
// spheres as they are at the START of the horizon, moving on at constant velocity (FPC:421-458, EXC:335-357).  Those spheres
// belong to the robots of the same scenario, i.e. to neighbouring lanes: each lane walks its chain once, leaves its spheres
// (x0, v) in the [72][64] tile of the joint-space kernel -- LO: the eight link origins, coincident ones merged; otherwise any
// table of up to eight spheres, one per link with its offset being the evaluation scripts' (evaluate_horizon.py:45) -- and for
// the whole horizon every fold reads the tile at x0 + k dt v.  No obstacle arrays in HBM (k_publish_obstacles + k_rollout_cart_panda stream 16 obstacles x 7
// scalars per row and step), no per-step exchange either.  A step is action, then system_step (FPC:77-92); mode 'vel' (the
// Panda drivers' mode, parameters_manipulators.py:12) -- 'acc' stays with the obstacle-array kernel.
// MODE: 0 = link-origin table (LO), 1 = any table of up to eight spheres (published once), 2 = any table, re-derived from the
// start states chunk by chunk in every step (obstacles_start_chunked)
template <typename T, class LS, int MODE>
__global__ __launch_bounds__(64) void k_rollout_cartc_panda(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen,
                                                             const T* __restrict__ q0, const T* __restrict__ qd0,
                                                             const T* __restrict__ prm, T* __restrict__ avg_out,
                                                             T* __restrict__ traj_q, T* __restrict__ traj_qd) {
  __shared__ T xch[TILE_SCALARS];
  constexpr bool LO = MODE == 0;
  const DevCfg<T>& cfg = *cfgp;
  const int lane = threadIdx.x;
  const int nsp = LO ? 8 - cfg.lo_merge01 - cfg.lo_merge45 : cfg.n_spheres;  // distinct spheres per robot
  if constexpr (LO) {
    stage_sphere_radii(cfg, xch, lane);  // visible after the publish barrier
  } else if (MODE == 1 && lane < nsp) {
    xch[TILE_RADII + lane] = cfg.sphere_r[lane];
    xch[TILE_MULT + lane] = T(1);
  }
  const int N = cfg.n_robots;
  const int spw = 64 / N;  // scenarios per wave
  int ls = lane / N;
  const int li = lane - ls * N;
  int64_t scen = (int64_t)blockIdx.x * spw + ls;
  const bool active = ls < spw && scen < n_scen;
  if (ls >= spw) ls = 0;  // idle tail lanes shadow the wave's first scenario (no stores)
  if (scen >= n_scen || !active) scen = (int64_t)blockIdx.x * spw + ls;
  if (scen >= n_scen) scen = n_scen - 1;
  const int64_t rows = n_scen * N;
  const int64_t row = scen * N + li;

  PandaState<T> R;
  load_state(rows, row, q0, qd0, R);
  const T* mount_own = cfg.mount[li];
  PrmView<T> P{prm, rows, row, {T(0), T(0), T(0)}, false};
  if constexpr (MODE == 2) {
    // start states of every lane, read by joint index in every step's sphere walk (own lane's entries only)
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      xch[(3 * j + 0) * 64 + lane] = R.cq[j];
      xch[(3 * j + 1) * 64 + lane] = R.sq[j];
      xch[(3 * j + 2) * 64 + lane] = R.qd[j];
    }
  }
  if constexpr (MODE == 1) {
    // Any table: the rolled sphere walk reads the joint state by joint index, so cos q, sin q, qdot are parked in the
    // tile's acceleration rows (j*9 + 6..8, which the Cartesian fold never reads) and the spheres go to the x / v rows.
    // Own lane's entries only until the barrier.
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      xch[(j * 9 + 6) * 64 + lane] = R.cq[j];
      xch[(j * 9 + 7) * 64 + lane] = R.sq[j];
      xch[(j * 9 + 8) * 64 + lane] = R.qd[j];
    }
    const bool dyn = cfg.dynamic != 0;
    panda_walk_spheres<false, T>(
        cfg, mount_own,
        [&](int j, T& c, T& sn, T& qdj) {
          c = xch[(j * 9 + 6) * 64 + lane];
          sn = xch[(j * 9 + 7) * 64 + lane];
          qdj = xch[(j * 9 + 8) * 64 + lane];
        },
        [&](int sp, const T* x, const T* v, const T*) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            xch[(sp * 9 + c) * 64 + lane] = x[c];
            xch[(sp * 9 + 3 + c) * 64 + lane] = dyn ? v[c] : T(0);
          }
        });
    __syncthreads();
  }
  T sumsq = T(0);
  const int H = cfg.horizon;
  // system_step 'vel' (FPC:77-92): qdot = action, q += dt*qdot; cos q / sin q advance by the angle-sum formula while every
  // |dq| of the wave is small.  Written at the TOP of the following iteration (and once after the last one), where neither
  // the action nor the solve's temporaries are live -- the order the joint-space kernel keeps its registers with.
  auto system_step = [&](int k_done) {
    T dq[7];
    bool small = true;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      dq[j] = cfg.dt * R.qd[j];
      small = small && (m_abs(dq[j]) < T(0.125));
      R.q[j] += dq[j];
    }
    if (__all(small)) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        T sd, cd;
        small_sincos(dq[j], sd, cd);
        const T c = R.cq[j] * cd - R.sq[j] * sd;
        const T sn = R.sq[j] * cd + R.cq[j] * sd;
        R.cq[j] = c;
        R.sq[j] = sn;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 7; ++j) m_sincos(R.q[j], &R.sq[j], &R.cq[j]);
    }
    if (active && traj_q) {
#pragma unroll
      for (int j = 0; j < 7; ++j) traj_q[((int64_t)k_done * 7 + j) * rows + row] = R.q[j];
    }
    if (active && traj_qd) {
#pragma unroll
      for (int j = 0; j < 7; ++j) traj_qd[((int64_t)k_done * 7 + j) * rows + row] = R.qd[j];
    }
  };
#pragma unroll 1
  for (int k = 0; k < H; ++k) {
    if (k > 0) system_step(k - 1);
    const T tk = to_uniform((T)k * cfg.dt);  // elapsed obstacle time
    T qdd[7], act[7];
    panda_solve_row<LS, kSingleWalk<LS> && MODE != 2>(
        cfg, mount_own, R, P,
        [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
          if constexpr (MODE == 2)
            obstacles_start_chunked<typename LS::Collision>(cfg, xch, lane, ls, li, N, mount_own, cfg.dynamic != 0, tk, E, acc);
          else
            obstacles_from_tile_drift<typename LS::Collision>(cfg, xch, ls, li, N, nsp, tk, E, acc);
        },
        qdd, act,
        [&](const PandaKin<T>& K1) {
          if (!LO || k != 0) return;  // the tile holds the start states for the whole horizon
          publish_link_spheres(xch, lane, K1, cfg.dynamic != 0, false, cfg.jsign, cfg.lo_merge01, cfg.lo_merge45);
          __syncthreads();
        });
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      R.qd[j] = act[j];
      sumsq += act[j] * act[j];
    }
  }
  if (H > 0) system_step(H - 1);
  if (active) avg_out[row] = sumsq / (T)(H * 7);
}

// ---------------------------------------------------------------------------- coupled compute_action
// One control step's compute_action for every robot of every scenario with the host-side obstacle assembly of the
// reference's loop (EXJ:394-412) done on chip: the dynamic obstacles of robot i are the configured spheres of all
// other robots of its scenario, x from FK, v = J qdot, a = 0 ("currently no acceleration", EXJ:411) or
// jac_dot*qdot (use_accel).  Same wave layout and LDS exchange as the rollout kernel, no time stepping.
// Round 6: PERSISTENT over the resident grid with a one-block software prefetch.  The kernel is one
// solve per row with nothing of its own to hide memory latency behind (one wave per SIMD): the 14 state loads, then the
// row's parameters read one group at a time where the solve uses them, were 4-6 exposed HBM round trips per wave
// (SQ_WAIT_ANY 0.23 of the wave cycles, VALU-busy 0.59; profiles/r05_configs_pmc.json).  A workgroup now walks blocks
// blockIdx.x, blockIdx.x + gridDim.x, ... ; at the top of a block it issues the 29 parameter loads of THIS block (they land
// during the seven sincos calls, ~900 instructions) and the 14 state loads of its NEXT block (they land during the solve),
// so that no load is waited for with nothing else to do.  (Prefetching the next block's parameters as well -- 43 values held
// across the solve -- does not fit: 196-248 B of scratch per lane, 0.204 ms against 0.167 ms on C2; gpurun_out/act_ab.txt.)
// MEASURED RESULT (C2, 196 608 scenarios, same box, alternating runs; profiles/r06_experiments.json): one-shot kernel of
// round 5 0.1650 / 0.1641 ms; persistent without any prefetch 0.1621 / 0.1619; this form 0.1642 / 0.1644; this form with the
// parameters read lazily 0.1681 / 0.1675; + single chain walk 0.1625 / 0.1621; exchange chunks of 6 instead of 4 spheres
// 0.1662 / 0.1674.  The loads were NOT what the kernel waits for: with every one of them prefetched a block ahead nothing
// moves.  C2's table (10 offset spheres per robot) takes the generic exchange path, whose rolled sphere walk and chunked
// folds cost ~13 k instructions per block against ~7 k of a link-origin rollout step; the waits are the scalar-load /
// LDS / barrier waits inside that path, as in every other kernel of this family.  Kept: persistence (no per-block
// dispatch), the top-of-block loads, the single walk -- together -2 %.
template <typename T>
struct PrmRegs {
  T v[MRF_NPARAM];
  __device__ __forceinline__ T operator[](int i) const { return v[i]; }  // every index is a compile-time constant after unrolling
};

// TAB: how the robots of a scenario exchange their spheres -- TAB_LO the link-origin tile, TAB_PACKED (round 6) any table of
// up to TILE_PACKED_MAX spheres without obstacle accelerations (the reference's call, EXJ:411): the spheres are derived by
// the solve's OWN unrolled chain walk (emit_link hook), published once as 6 rows each and folded in one pipelined loop -- no
// rolled sphere walk, no chunk barriers; TAB_GENERIC everything else (chunked exchange).
enum { TAB_GENERIC = 0, TAB_LO = 1, TAB_PACKED = 2 };

template <typename T, class LS, int TAB>
__global__ __launch_bounds__(64) void k_action_coupled(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen,
                                                        const T* __restrict__ q, const T* __restrict__ qd,
                                                        const T* __restrict__ prm, int use_accel,
                                                        T* __restrict__ qdd_out, T* __restrict__ act_out) {
  constexpr bool LO = TAB == TAB_LO;
  __shared__ T xch[TAB != TAB_GENERIC ? TILE_SCALARS : GEN_SCALARS];
  const DevCfg<T>& cfg = *cfgp;
  if constexpr (LO) stage_sphere_radii(cfg, xch, threadIdx.x);  // visible after the first publish barrier
  if constexpr (TAB == TAB_PACKED) {
    if ((int)threadIdx.x < cfg.n_spheres) xch[TILE_RADII + threadIdx.x] = cfg.sphere_r[threadIdx.x];
  }
  const int N = cfg.n_robots;
  const int spw = 64 / N;
  const int lane = threadIdx.x;
  const int64_t rows = n_scen * N;
  const int64_t nblk = (n_scen + spw - 1) / spw;
  const int ls0 = lane / N;
  const int li = lane - ls0 * N;
  // row of this lane in block b: idle tail lanes and the rows past the batch shadow a valid row (no stores)
  auto locate = [&](int64_t b, int& ls, int64_t& row) {
    ls = ls0;
    int64_t scen = b * spw + ls;
    const bool active = ls < spw && scen < n_scen;
    if (ls >= spw) ls = 0;
    if (scen >= n_scen || !active) scen = b * spw + ls;
    if (scen >= n_scen) scen = n_scen - 1;
    row = scen * N + li;
    return active;
  };
  T nq[7], nqd[7];
  auto fetch = [&](int64_t b) {
    int ls;
    int64_t row;
    locate(b, ls, row);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      nq[j] = q[j * rows + row];
      nqd[j] = qd[j * rows + row];
    }
  };
  fetch(blockIdx.x);
#pragma unroll 1
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    int ls;
    int64_t row;
    const bool active = locate(blk, ls, row);
    PandaState<T> R;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      R.q[j] = nq[j];
      R.qd[j] = nqd[j];
    }
#if defined(MRF_ACTION_PRM_LAZY)
    PrmView<T> P{prm, rows, row, {T(0), T(0), T(0)}, false};
#else
    PrmRegs<T> P;  // this block's parameters: issued now, in flight during the sincos calls below
#pragma unroll
    for (int c = 0; c < MRF_NPARAM; ++c) P.v[c] = prm[c * rows + row];
#endif
#ifndef MRF_ACTION_NO_PREFETCH
    if (blk + gridDim.x < nblk) fetch(blk + gridDim.x);  // the NEXT block's state: in flight during this block's solve
#endif
    state_sincos(R);
    T qdd[7], act[7];
    if constexpr (LO) {
      panda_solve_row<LS, kSingleWalk<LS>>(
          cfg, cfg.mount[li], R, P,
          [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
            obstacles_from_tile<typename LS::Collision>(cfg, xch, ls, li, N, E, acc);
          },
          qdd, act,
          [&](const PandaKin<T>& K1) {
            __syncthreads();  // the previous block's folds have finished in every lane
            publish_link_spheres(xch, lane, K1, cfg.dynamic != 0, cfg.dynamic != 0 && use_accel != 0, cfg.jsign, cfg.lo_merge01,
                                 cfg.lo_merge45);  // EXJ:336-339,411
            __syncthreads();
          });
    } else if constexpr (TAB == TAB_PACKED) {
      __syncthreads();  // the previous block's folds have finished in every lane
      panda_solve_row<LS, kSingleWalk<LS>>(
          cfg, cfg.mount[li], R, P,
          [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
            obstacles_from_tile_packed<typename LS::Collision>(cfg, xch, ls, li, N, cfg.n_spheres, E, acc);
          },
          qdd, act, [&](const PandaKin<T>&) { __syncthreads(); },  // every lane's spheres are in the tile
          PackedSphereEmit<T>(cfg, xch, lane, cfg.dynamic != 0));
    } else {
      __syncthreads();  // the previous block's chunk walks have finished in every lane
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        xch[(3 * j + 0) * 64 + lane] = R.cq[j];
        xch[(3 * j + 1) * 64 + lane] = R.sq[j];
        xch[(3 * j + 2) * 64 + lane] = R.qd[j];
      }
      __syncthreads();
      // one-shot solve: the own chain's kinematics stay alive across the chunked exchange (no scratch in this kernel; the
      // rollout kernels keep the two-phase form there).  C2: 0.1654 -> 0.1622 ms.
      constexpr bool GEN_SW = kSingleWalk<LS>;
      panda_solve_row<LS, GEN_SW>(
          cfg, cfg.mount[li], R, P,
          [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
            obstacles_generic_chunked<typename LS::Collision>(cfg, xch, lane, ls, li, N, cfg.mount[li], cfg.dynamic != 0,
                                                              use_accel ? cfg.jsign : T(0), E, acc);  // EXJ:411 passes zeros
          },
          qdd, act);
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if (qdd_out) qdd_out[j * rows + row] = qdd[j];
        act_out[j * rows + row] = act[j];
      }
    }
#ifdef MRF_ACTION_NO_PREFETCH
    if (blk + gridDim.x < nblk) fetch(blk + gridDim.x);
#endif
  }
}

// ---------------------------------------------------------------------------- cooperative (latency) kernels
// One wave per scenario for small batches, where the row-per-lane kernels above would leave the chip empty and a
// single lane would walk through ~7 k instructions per rollout step.  Lane = (robot i, ego point g, sphere chunk c):
//   every lane walks its own robot's chain (redundantly, in parallel);
//   one lane per robot stages that robot's sphere states (x, v, a) in LDS                       [LDS staging]
//   each lane folds its ego point against its chunk of the other robots' spheres;
//   the chunk partials are summed with xor-shuffles, the 5 points gathered with indexed shuffles   [wave shuffles]
//   and every lane of a robot finishes the 7x7 part redundantly.
// COOP_ROLLOUT = true is the coupled rollout (FPJ:190-249), false one coupled compute_action (EXJ:394-448).
// value of the lane whose index differs in bit 0 (X = 1) or bit 1 (X = 2) within its quad: v_mov_b32 with quad_perm
template <int X>
__device__ __forceinline__ int quad_xor_b32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, X == 1 ? 0xB1 : 0x4E, 0xF, 0xF, true);  // quad_perm [1,0,3,2] / [2,3,0,1]
}
template <int X>
__device__ __forceinline__ double quad_xor(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(quad_xor_b32<X>(hi), quad_xor_b32<X>(lo));
}
template <int X>
__device__ __forceinline__ float quad_xor(float v) {
  return __int_as_float(quad_xor_b32<X>(__float_as_int(v)));
}

__host__ __device__ inline int coop_chunks(int n_robots) {
  return 5 * n_robots * 4 <= 64 ? 4 : (5 * n_robots * 2 <= 64 ? 2 : 1);
}

// CART = true (with COOP_ROLLOUT) is the Cartesian rollout of mrf_rollout_cartesian_coupled in latency mode (FPC:421-458):
// the obstacle table is the START state of the other robots -- staged once, every fold extrapolates x0 + k dt v, zero
// accelerations -- and a step is action-then-integration.
template <typename T, class LS, bool LO, bool COOP_ROLLOUT, bool CART = false>
__global__ __launch_bounds__(64) void k_coop_panda(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen,
                                                    const T* __restrict__ q0, const T* __restrict__ qd0,
                                                    const T* __restrict__ prm, int use_accel, T* __restrict__ avg_out,
                                                    T* __restrict__ traj_q, T* __restrict__ traj_qd,
                                                    T* __restrict__ qdd_out, T* __restrict__ act_out) {
  extern __shared__ __align__(16) unsigned char coop_lds[];
  T* xch = reinterpret_cast<T*>(coop_lds);  // [21][64]  cos q, sin q, qdot of every lane (generic sphere tables)
  T* sph = xch + 21 * 64;                    // [N][S][9] sphere states of the scenario
  const DevCfg<T>& cfg = *cfgp;
  const int N = cfg.n_robots;
  const int m01 = LO ? cfg.lo_merge01 : 0, m45 = LO ? cfg.lo_merge45 : 0;
  const int S = LO ? 8 - m01 - m45 : cfg.n_spheres;  // distinct spheres per robot (coincident link origins merged)
  T* prm_lds = sph + (size_t)N * S * 9;              // [MRF_NPARAM][N] the scenario's parameters, read once
  const int C = coop_chunks(N);
  const int LPR = 5 * C;  // lanes per robot
  const int lane = threadIdx.x;
  int i = lane / LPR;
  int rem = lane - i * LPR;
  const bool idle = i >= N;  // tail lanes shadow robot 0 and never write
  if (idle) {
    i = 0;
    rem = 1 % LPR;
  }
  const int g = rem / C;
  const int c = rem - g * C;
  const bool writer = !idle && rem == 0;
  const int64_t scen = blockIdx.x;
  const int64_t rows = n_scen * N;
  const int64_t row = scen * N + i;

  PandaState<T> R;
  load_state_values(rows, row, q0, qd0, R);
  const T* mount_own = cfg.mount[i];
  // one scenario is a chain of dependent latencies: the 29 parameters per robot are fetched once into LDS instead of
  // being re-read from global memory where they are used in every horizon step -- in the same round trip as the state
  for (int idx = lane; idx < MRF_NPARAM * N; idx += 64) {
    const int cpar = idx / N, rr = idx - cpar * N;
    prm_lds[idx] = prm[(int64_t)cpar * rows + scen * N + rr];
  }
  state_sincos(R);
  __syncthreads();
  PrmView<T> P{prm_lds, N, i, {T(0), T(0), T(0)}, false};
  if (COOP_ROLLOUT && !CART && ((cfg.goal_mask >> i) & 1)) {
    PandaKin<T> K0;
    panda_walk_own<T>(mount_own, R.cq, R.sq, R.qd, K0);
#pragma unroll
    for (int k = 0; k < 3; ++k) P.g0[k] = K0.p8[k] + cfg.goal_T * K0.v8[k];
    P.own_goal = true;
  }
  const bool dyn = cfg.dynamic != 0;
  const bool acc_on = !CART && dyn && (COOP_ROLLOUT || use_accel);
  // The (other robot, sphere) pairs this lane folds are the same in every horizon step: their LDS offsets, radii and
  // multiplicities are worked out once (an integer division and a load from the constants per pair otherwise sit on
  // the single wave's critical path in every step).  Up to COOP_PRE pairs per lane; larger chunks use the loop.
  constexpr int COOP_PRE = 4;
  const int M_all = (N - 1) * S;
  const bool pre = (M_all + C - 1) / C <= COOP_PRE;
  int pre_off[COOP_PRE];
  T pre_rad[COOP_PRE], pre_mul[COOP_PRE];
#pragma unroll
  for (int t = 0; t < COOP_PRE; ++t) {
    const int m = c + t * C;
    const bool valid = pre && m < M_all;
    const int d = valid ? m / S : 0, sp = valid ? m - d * S : 0;
    int jr = i + 1 + d;
    if (jr >= N) jr -= N;
    pre_off[t] = valid ? (jr * S + sp) * 9 : -1;
    pre_rad[t] = cfg.sphere_r[LO ? lo_sphere(sp, m01, m45) : sp];
    pre_mul[t] = LO ? T(lo_count(sp, m01, m45)) : T(1);
  }
  T sumsq = T(0);
  const int H = COOP_ROLLOUT ? cfg.horizon : 1;
#pragma unroll 1
  for (int k = 0; k < H; ++k) {
    MRF_STAMP(0);
    const T tk = CART ? (T)k * cfg.dt : T(0);  // elapsed obstacle time of the Cartesian rollout
    if (COOP_ROLLOUT && !CART) {
      T dq[7];
      bool small = true;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        dq[j] = cfg.dt * R.qd[j];
        small = small && (m_abs(dq[j]) < T(0.125));
        R.q[j] += dq[j];
      }
      if (__all(small)) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          T sd, cd;
          small_sincos(dq[j], sd, cd);
          const T cc = R.cq[j] * cd - R.sq[j] * sd;
          const T ss = R.sq[j] * cd + R.cq[j] * sd;
          R.cq[j] = cc;
          R.sq[j] = ss;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 7; ++j) m_sincos(R.q[j], &R.sq[j], &R.cq[j]);
      }
    }
    MRF_STAMP(1);
    PandaKin<T> K;
    panda_walk_own<T>(mount_own, R.cq, R.sq, R.qd, K);
    MRF_STAMP(2);
    if (!CART || k == 0) {  // Cartesian rollout: the table holds the other robots' START states for the whole horizon
    __syncthreads();  // the previous step's readers are done with sph / xch
    if (LO) {
      if (writer) {
#pragma unroll
        for (int sp = 0; sp < 8; ++sp) {
          T* dst = sph + ((size_t)i * S + lo_slot(sp, m01, m45)) * 9;  // merged duplicates overwrite with equal values
#pragma unroll
          for (int k3 = 0; k3 < 3; ++k3) {
            const T x = sp < 7 ? K.o[sp < 7 ? sp : 0][k3] : K.p8[k3];
            const T v = sp < 7 ? K.vo[sp < 7 ? sp : 0][k3] : K.v8[k3];
            const T a = sp < 7 ? K.ao[sp < 7 ? sp : 0][k3] : K.a8[k3];
            dst[k3] = x;
            dst[3 + k3] = dyn ? v : T(0);
            dst[6 + k3] = acc_on ? cfg.jsign * a : T(0);
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        xch[(3 * j + 0) * 64 + lane] = R.cq[j];
        xch[(3 * j + 1) * 64 + lane] = R.sq[j];
        xch[(3 * j + 2) * 64 + lane] = R.qd[j];
      }
      __syncthreads();
      panda_walk_spheres<false, T>(
          cfg, mount_own,
          [&](int j, T& cj, T& sj, T& qdj) {
            cj = xch[(3 * j + 0) * 64 + lane];
            sj = xch[(3 * j + 1) * 64 + lane];
            qdj = xch[(3 * j + 2) * 64 + lane];
          },
          [&](int sp, const T* x, const T* v, const T* a) {
            if (!writer) return;
            T* dst = sph + ((size_t)i * S + sp) * 9;
#pragma unroll
            for (int k3 = 0; k3 < 3; ++k3) {
              dst[k3] = x[k3];
              dst[3 + k3] = dyn ? v[k3] : T(0);
              dst[6 + k3] = acc_on ? cfg.jsign * a[k3] : T(0);
            }
          });
    }
    __syncthreads();
    }

    MRF_STAMP(3);
    // ---- my ego point against my chunk of the other robots' spheres
    EgoPts<T, 1> E1;
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3) {
      E1.p[0][k3] = g == 0 ? K.o[2][k3] : (g == 1 ? K.o[3][k3] : (g == 2 ? K.o[4][k3] : (g == 3 ? K.o[6][k3] : K.p8[k3])));
      E1.v[0][k3] = g == 0 ? K.vo[2][k3] : (g == 1 ? K.vo[3][k3] : (g == 2 ? K.vo[4][k3] : (g == 3 ? K.vo[6][k3] : K.v8[k3])));
    }
    ego_point_links<LS::Collision::generic>(cfg, P, g, E1.rb[0][0], E1.rb[0][1], E1.nl[0]);
    EgoAcc<T, 1> a1;
    a1.zero();
    if (cfg.n_ego > 0 && pre) {
#pragma unroll
      for (int t = 0; t < COOP_PRE; ++t) {
        if (pre_off[t] < 0) continue;
        const T* src = sph + pre_off[t];
        T x[3] = {src[0], src[1], src[2]}, v[3] = {src[3], src[4], src[5]}, a[3] = {src[6], src[7], src[8]};
        if (CART)
          for (int k3 = 0; k3 < 3; ++k3) x[k3] += tk * v[k3];
        accumulate_obstacle<typename LS::Collision>(cfg, E1, x, v, a, pre_rad[t], false, a1, pre_mul[t]);
      }
    } else if (cfg.n_ego > 0) {
      const int M = (N - 1) * S;
#pragma unroll 1
      for (int m = c; m < M; m += C) {
        const int d = m / S;
        const int sp = m - d * S;
        int jr = i + 1 + d;
        if (jr >= N) jr -= N;
        const T* src = sph + ((size_t)jr * S + sp) * 9;
        T x[3] = {src[0], src[1], src[2]}, v[3] = {src[3], src[4], src[5]}, a[3] = {src[6], src[7], src[8]};
        if (CART)
          for (int k3 = 0; k3 < 3; ++k3) x[k3] += tk * v[k3];
        accumulate_obstacle<typename LS::Collision>(cfg, E1, x, v, a, cfg.sphere_r[LO ? lo_sphere(sp, m01, m45) : sp], false,
                                                    a1, LO ? T(lo_count(sp, m01, m45)) : T(1));
      }
    }
    // ---- the plane leaf of my point, once per point (chunk 0), instead of all 5 points redundantly in the finish
    if (cfg.n_ego > 0 && cfg.n_planes > 0 && c == 0) {
      T con[4] = {P[MRF_P_CONSTRAINT_0], P[MRF_P_CONSTRAINT_0 + 1], P[MRF_P_CONSTRAINT_0 + 2], P[MRF_P_CONSTRAINT_0 + 3]};
      accumulate_plane<typename LS::Plane>(cfg, E1, con, a1);
    }
    MRF_STAMP(4);
    // ---- sum the chunk partials, then give every lane of the robot all 5 points
    // the C (<= 4) chunk lanes of a point are neighbours inside a quad: quad-permute DPP moves, no LDS crossbar
    if (C >= 2) {
#pragma unroll
      for (int e = 0; e < 6; ++e) a1.A[0][e] += quad_xor<1>(a1.A[0][e]);
#pragma unroll
      for (int e = 0; e < 3; ++e) a1.b[0][e] += quad_xor<1>(a1.b[0][e]);
    }
    if (C >= 4) {
#pragma unroll
      for (int e = 0; e < 6; ++e) a1.A[0][e] += quad_xor<2>(a1.A[0][e]);
#pragma unroll
      for (int e = 0; e < 3; ++e) a1.b[0][e] += quad_xor<2>(a1.b[0][e]);
    }
    MRF_STAMP(5);
    T qdd[7], act[7];
    {
      // Every lane of a robot finishes the solve redundantly.  Sharing the finish between the lanes of a robot (seven
      // lanes pulling one job each with the same code, limit leaves one per lane, entry-wise sums through LDS, the
      // two LDL^T solves side by side) was built and measured in r02: correct, but 319 us instead of 310 us per H=30
      // rollout -- the selects that make the jobs uniform and the LDS round trips cost what the saved arithmetic gains.
      EgoAcc<T, NG> acc;
#pragma unroll
      for (int gg = 0; gg < NG; ++gg) {
        const int srcl = i * LPR + gg * C;
#pragma unroll
        for (int e = 0; e < 6; ++e) acc.A[gg][e] = __shfl(a1.A[0][e], srcl);
#pragma unroll
        for (int e = 0; e < 3; ++e) acc.b[gg][e] = __shfl(a1.b[0][e], srcl);
      }
      EgoPts<T, NG> E;
      panda_ego_points<LS::Collision::generic>(cfg, K, P, E);
      panda_finish_row<LS, true>(cfg, R, P, K, E, acc, qdd, act);
    }
    MRF_STAMP(6);
    if (COOP_ROLLOUT && CART) {  // system_step after the action (FPC:77-92,421-446)
      T dq[7];
      bool small = true;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if (cfg.mode == MRF_MODE_VEL) {
          R.qd[j] = act[j];
          dq[j] = cfg.dt * R.qd[j];
        } else {
          dq[j] = cfg.dt * R.qd[j] + T(0.5) * cfg.dt * cfg.dt * act[j];
          R.qd[j] += cfg.dt * act[j];
        }
        R.q[j] += dq[j];
        small = small && (m_abs(dq[j]) < T(0.125));
        sumsq += R.qd[j] * R.qd[j];
      }
      if (__all(small)) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          T sd, cd;
          small_sincos(dq[j], sd, cd);
          const T cc = R.cq[j] * cd - R.sq[j] * sd;
          const T ss = R.sq[j] * cd + R.cq[j] * sd;
          R.cq[j] = cc;
          R.sq[j] = ss;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 7; ++j) m_sincos(R.q[j], &R.sq[j], &R.cq[j]);
      }
    } else if (COOP_ROLLOUT) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        R.qd[j] = act[j];
        sumsq += act[j] * act[j];
      }
    }
    if (COOP_ROLLOUT) {
      if (writer && traj_q) {
#pragma unroll
        for (int j = 0; j < 7; ++j) traj_q[((int64_t)k * 7 + j) * rows + row] = R.q[j];
      }
      if (writer && traj_qd) {
#pragma unroll
        for (int j = 0; j < 7; ++j) traj_qd[((int64_t)k * 7 + j) * rows + row] = R.qd[j];
      }
    } else if (writer) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if (qdd_out) qdd_out[j * rows + row] = qdd[j];
        act_out[j * rows + row] = act[j];
      }
    }
  }
  if (COOP_ROLLOUT && writer) avg_out[row] = sumsq / (T)(H * 7);
}


// ---------------------------------------------------------------------------- Cartesian rollout
// RES: the first CART_RESIDENT<T> obstacles live in LDS for the whole rollout (one wave per block); RES = false is the
// plain streaming form with 256-thread blocks (switch -DMRF_CART_STREAM_ONLY, A/B by tools/prof_kernels.py).
// Every global access of the step loop goes through RowAddr: wave-uniform bases in SGPRs + one shared lane offset, and
// the elapsed obstacle time is a scalar -- the loop invariants that used to sit in VGPR pairs (and, beyond 512 registers,
// in scratch memory) are gone from the vector file.
template <typename T, class LS, bool RES, bool ACC>
__global__ __launch_bounds__(RES ? 64 : 256) MRF_ATTR_CART void k_rollout_cart_panda(const DevCfg<T>* __restrict__ cfgp, int64_t rows,
                                                             const T* __restrict__ q0, const T* __restrict__ qd0,
                                                             const T* __restrict__ prm, int n_obst, int n_static,
                                                             const T* __restrict__ ox0, const T* __restrict__ ov,
                                                             const T* __restrict__ oa, const T* __restrict__ orad,
                                                             T* __restrict__ avg_out, T* __restrict__ traj_q,
                                                             T* __restrict__ traj_qd) {
  // rows of this block: first .. first + blockDim - 1; the tail lanes shadow the last row (wave-wide votes below),
  // without stores
  RowAddr<T> ra;
  ra.rows = rows;
  ra.first = (int64_t)blockIdx.x * blockDim.x;
  const int64_t left = rows - ra.first;  // >= 1
  const bool active = (int64_t)threadIdx.x < left;
  ra.off = active ? threadIdx.x : (uint32_t)(left - 1);
  const int64_t r = ra.first + ra.off;
  const DevCfg<T>& cfg = *cfgp;
  PandaState<T> R;
  load_state(rows, r, q0, qd0, R);
  // measured (profiles/r04_experiments.json, H=30, M=16, f64): the uniform-base addressing pays without obstacle
  // accelerations (3.81 -> 3.72 ms, scratch 44 -> 20 B per lane); with them the allocator trades it for more scratch
  // (60 -> 84 B, 4.05 -> 4.19 ms), so that instantiation keeps the per-lane addresses
  constexpr bool LANE = ACC;
  using Addr = std::conditional_t<LANE, LaneAddr<T>, RowAddr<T>>;
  using Prm = std::conditional_t<LANE, PrmView<T>, PrmViewU<T>>;
  Addr addr;
  Prm P;
  if constexpr (LANE) {
    addr = LaneAddr<T>{rows, r};
    P = PrmView<T>{prm, rows, r, {T(0), T(0), T(0)}, false};
  } else {
    addr = ra;
    P = PrmViewU<T>{prm, ra, {T(0), T(0), T(0)}, false};
  }
  const int li = (int)(r % cfg.n_robots);
  const T* mount_own = cfg.mount[li];
  constexpr int NRES = CART_RESIDENT<T, ACC>;
  __shared__ T res[RES ? NRES * (ACC ? 10 : 7) * 64 : 1];
  const int nres = RES ? (n_obst < NRES ? n_obst : NRES) : 0;
  if constexpr (RES) stage_resident_obstacles<ACC>(res, (int)threadIdx.x, nres, rows, r, ox0, ov, oa, orad);
  __syncthreads();
  T sumsq = T(0);
  T tk_lane = T(0);
  const int H = cfg.horizon;
#pragma unroll 1
  for (int k = 0; k < H; ++k) {
    // elapsed obstacle time (FPC:448-453: x += dt*v per step): a scalar in the instantiation without accelerations, a
    // per-lane running sum in the other one (measured: the scalar form costs that one 4.05 -> 4.3 ms)
    const T tk = ACC ? tk_lane : to_uniform((T)k * cfg.dt);
    T qdd[7], act[7];
    panda_solve_row<LS, kCartSingleWalk && kSingleWalk<LS>>(
        cfg, mount_own, R, P,
        [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
          // one pipelined loop over resident and streamed obstacles, or one loop each (-DMRF_CART_TWO_LOOPS): which one is
          // faster has followed the register allocator from build to build (profiles/r04_experiments.json)
#if defined(MRF_CART_TWO_LOOPS)
          constexpr bool two_loops = true;
#else
          constexpr bool two_loops = false;
#endif
          if constexpr (RES && two_loops) {
            obstacles_resident<typename LS::Collision, ACC>(cfg, res, (int)threadIdx.x, nres, n_static, ov != nullptr,
                                                            oa != nullptr, tk, E, acc);
            obstacles_from_arrays<typename LS::Collision, ACC>(cfg, addr, n_obst, n_static, ox0, ov, oa, orad, tk, false, E, acc,
                                                               nres);
          } else if constexpr (RES) {
            obstacles_cart<typename LS::Collision, ACC>(cfg, res, (int)threadIdx.x, nres, addr, n_obst, n_static, ox0, ov, oa,
                                                        orad, tk, E, acc);
          } else {
            obstacles_from_arrays<typename LS::Collision, ACC>(cfg, addr, n_obst, n_static, ox0, ov, oa, orad, tk, false, E, acc,
                                                               0);
          }
        },
        qdd, act);
    // system_step (FPC:77-92); cos q / sin q advance by the angle-sum formula while every |dq| of the wave is small
    T dq[7];
    bool small = true;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if (cfg.mode == MRF_MODE_VEL) {
        R.qd[j] = act[j];
        dq[j] = cfg.dt * R.qd[j];
      } else {
        dq[j] = cfg.dt * R.qd[j] + T(0.5) * cfg.dt * cfg.dt * act[j];
        R.qd[j] += cfg.dt * act[j];
      }
      R.q[j] += dq[j];
      small = small && (m_abs(dq[j]) < T(0.125));
      sumsq += R.qd[j] * R.qd[j];
    }
    if (__all(small)) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        T sd, cd;
        small_sincos(dq[j], sd, cd);
        const T c = R.cq[j] * cd - R.sq[j] * sd;
        const T sn = R.sq[j] * cd + R.cq[j] * sd;
        R.cq[j] = c;
        R.sq[j] = sn;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 7; ++j) m_sincos(R.q[j], &R.sq[j], &R.cq[j]);
    }
    if (active && traj_q) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if constexpr (ACC)
          traj_q[((int64_t)k * 7 + j) * rows + r] = R.q[j];
        else
          ra.store(traj_q, (int64_t)k * 7 + j, R.q[j]);
      }
    }
    if (active && traj_qd) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if constexpr (ACC)
          traj_qd[((int64_t)k * 7 + j) * rows + r] = R.qd[j];
        else
          ra.store(traj_qd, (int64_t)k * 7 + j, R.qd[j]);
      }
    }
    if constexpr (ACC) tk_lane += cfg.dt;
  }
  if (active) avg_out[r] = sumsq / (T)(H * 7);
}

// ---------------------------------------------------------------------------- coupled Cartesian rollout, large sphere tables
// mrf_rollout_cartesian_coupled for tables the LDS-tile forms cannot hold (more than eight spheres per robot: the YAML's
// n_obst_per_link = 4 -> 32 spheres, EXC:184, panda_config.yaml:8) in ONE launch (round 6).  The obstacles
// of a robot are the other robots of its scenario at their START states (EXC:330-352, UFK:3-33) -- neighbouring lanes.  In the
// prologue every lane leaves cos q / sin q / qdot in LDS, re-walks the chains of its scenario's other robots and writes THEIR
// spheres (x0, v) into its OWN row of the work arrays [M][3][rows] (coalesced; no scatter into other rows, no separate
// k_publish_obstacles launch, no radius array: a sphere's radius is the table's, staged in LDS once).  The first
// CARTS_RESIDENT obstacles of the row are then copied into a per-wave LDS tile [m][6][64] and folded from there in every
// step, the rest streams through the depth-one register pipeline: 6 scalars per streamed obstacle and step instead of 7, 12
// instead of 10 obstacles resident (f64).  Step order, both modes and the incremental cos / sin as k_rollout_cart_panda.
// UR (uniform radius): every sphere of the table has the same radius -- the reference's tables do (PM:23, SIM:196) -- so it
// sits in an SGPR pair; otherwise the radius of obstacle m is a scalar load of the table entry m % S, issued with the
// obstacle's fetch.  Measured on CARTC32 (2 Pandas x 32 spheres, H = 30, 196 608 scenarios; gpurun_out/carts_ab*.txt):
// two launches 6.34 ms; this kernel with the radius from an LDS row 6.45 ms (the per-obstacle ds_read shares lgkmcnt with
// the leaf constants' scalar loads), scalar load 6.23 ms, uniform 6.08 ms; the obstacle-array kernel alone (CART32) 6.16 ms.
// 4 instead of 12 resident obstacles: +1.6 %; hoisting the leaf constants' scalar loads out of the loop: +2 % (here) and
// +10 % (k_rollout_cart_panda) -- both left as they are.
constexpr int CARTS_NV = 6;  // x0[3], v[3]
#ifndef MRF_CARTS_TILE_BYTES
#define MRF_CARTS_TILE_BYTES 36864
#endif
template <typename T>
constexpr int CARTS_RESIDENT = MRF_CARTS_TILE_BYTES / (CARTS_NV * 64 * (int)sizeof(T));  // 12 (f64), 24 (f32)

template <typename T, class LS, bool UR>
__global__ __launch_bounds__(64) void k_rollout_carts_panda(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen,
                                                             const T* __restrict__ q0, const T* __restrict__ qd0,
                                                             const T* __restrict__ prm, T* __restrict__ wx,
                                                             T* __restrict__ wv, T* __restrict__ avg_out,
                                                             T* __restrict__ traj_q, T* __restrict__ traj_qd) {
  constexpr int NRES = CARTS_RESIDENT<T>;
  __shared__ T res[NRES * CARTS_NV * 64];
  static_assert(NRES * CARTS_NV >= 21, "the prologue's joint-state rows live in the tile");
  const DevCfg<T>& cfg = *cfgp;
  const int lane = threadIdx.x;
  const int N = cfg.n_robots, S = cfg.n_spheres;
  const int spw = 64 / N;  // scenarios per wave: their rows are contiguous, first row of the block + lane
  int ls = lane / N;
  int li = lane - ls * N;
  const int64_t rows = n_scen * N;
  RowAddr<T> ra;
  ra.rows = rows;
  ra.first = (int64_t)blockIdx.x * spw * N;
  const bool active = ls < spw && ra.first + lane < rows;
  if (!active) ls = li = 0;  // idle lanes shadow the block's first row (no stores)
  ra.off = active ? (uint32_t)lane : 0u;
  const int64_t r = ra.first + ra.off;
  PandaState<T> R;
  load_state(rows, r, q0, qd0, R);
  PrmViewU<T> P{prm, ra, {T(0), T(0), T(0)}, false};
  const T* mount_own = cfg.mount[li];
  const int M = S * (N - 1);
  const int nres = M < NRES ? M : NRES;
  // ---- prologue: obstacle assembly (compute_x_obsts_dyn_0, UFK:3-33), every lane for its own row
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    res[(3 * j + 0) * 64 + lane] = R.cq[j];
    res[(3 * j + 1) * 64 + lane] = R.sq[j];
    res[(3 * j + 2) * 64 + lane] = R.qd[j];
  }
  __syncthreads();
  {
    const bool dyn = cfg.dynamic != 0;
#pragma unroll 1
    for (int d = 0; d < N - 1; ++d) {
      const int jr = d < li ? d : d + 1;  // the other robots in increasing order (EXC:336-349)
      const T* st = res + ls * N + jr;    // that robot's lane of this scenario
      panda_walk_spheres<false, T>(
          cfg, cfg.mount[jr],
          [&](int j, T& c, T& sn, T& qdj) {
            c = st[(3 * j + 0) * 64];
            sn = st[(3 * j + 1) * 64];
            qdj = st[(3 * j + 2) * 64];
          },
          [&](int sp, const T* x, const T* v, const T*) {
            if (!active) return;
            const int m = d * S + sp;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              ra.store(wx, m * 3 + c, x[c]);
              ra.store(wv, m * 3 + c, dyn ? v[c] : T(0));  // static fabrics: zero obstacle velocities (EXC:336-337)
            }
          });
    }
  }
  __syncthreads();  // every lane has finished reading the joint-state rows
#pragma unroll 1
  for (int m = 0; m < nres; ++m) {
    T* dst = res + m * (CARTS_NV * 64) + lane;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dst[c * 64] = ra.load(wx, m * 3 + c);  // the lane's own stores of a moment ago
      dst[(3 + c) * 64] = ra.load(wv, m * 3 + c);
    }
  }
  __syncthreads();
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  T sumsq = T(0);
  const int H = cfg.horizon;
#pragma unroll 1
  for (int k = 0; k < H; ++k) {
    const T tk = to_uniform((T)k * cfg.dt);  // elapsed obstacle time (FPC:448-453: x += dt*v per step)
    T qdd[7], act[7];
    panda_solve_row<LS, kCartSingleWalk && kSingleWalk<LS>>(
        cfg, mount_own, R, P,
        [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
          int sp_next = 0;  // sphere index of the obstacle fetched next (m % S without a division per obstacle)
          pipelined_pairs<T, CARTS_NV + 1>(
              M,
              [&](int m, T (&buf)[CARTS_NV + 1]) {
                if (m < nres) {
                  lds_ptr src = (lds_ptr)(res + m * (CARTS_NV * 64) + lane);
#pragma unroll
                  for (int c = 0; c < CARTS_NV; ++c) buf[c] = src[c * 64];
                } else {
#pragma unroll
                  for (int c = 0; c < 3; ++c) {
                    buf[c] = ra.load(wx, m * 3 + c);
                    buf[3 + c] = ra.load(wv, m * 3 + c);
                  }
                }
                buf[CARTS_NV] = cfg.sphere_r[UR ? 0 : sp_next];
                if (!UR && ++sp_next == S) sp_next = 0;
              },
              [&](int, T (&buf)[CARTS_NV + 1]) {
                const T xo[3] = {buf[0] + tk * buf[3], buf[1] + tk * buf[4], buf[2] + tk * buf[5]};
                const T ao[3] = {T(0), T(0), T(0)};  // FPC:33: zero obstacle accelerations (compile-time: the n.a_o terms vanish)
                accumulate_obstacle<typename LS::Collision>(cfg, E, xo, buf + 3, ao, buf[CARTS_NV], false, acc);
              });
        },
        qdd, act);
    // system_step (FPC:77-92); cos q / sin q advance by the angle-sum formula while every |dq| of the wave is small
    T dq[7];
    bool small = true;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if (cfg.mode == MRF_MODE_VEL) {
        R.qd[j] = act[j];
        dq[j] = cfg.dt * R.qd[j];
      } else {
        dq[j] = cfg.dt * R.qd[j] + T(0.5) * cfg.dt * cfg.dt * act[j];
        R.qd[j] += cfg.dt * act[j];
      }
      R.q[j] += dq[j];
      small = small && (m_abs(dq[j]) < T(0.125));
      sumsq += R.qd[j] * R.qd[j];
    }
    if (__all(small)) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        T sd, cd;
        small_sincos(dq[j], sd, cd);
        const T c = R.cq[j] * cd - R.sq[j] * sd;
        const T sn = R.sq[j] * cd + R.cq[j] * sd;
        R.cq[j] = c;
        R.sq[j] = sn;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 7; ++j) m_sincos(R.q[j], &R.sq[j], &R.cq[j]);
    }
    if (active && traj_q) {
#pragma unroll
      for (int j = 0; j < 7; ++j) ra.store(traj_q, (int64_t)k * 7 + j, R.q[j]);
    }
    if (active && traj_qd) {
#pragma unroll
      for (int j = 0; j < 7; ++j) ra.store(traj_qd, (int64_t)k * 7 + j, R.qd[j]);
    }
  }
  if (active) avg_out[r] = sumsq / (T)(H * 7);
}

// ---------------------------------------------------------------------------- sphere kinematics
template <typename T>
__global__ __launch_bounds__(64) void k_fk_spheres_panda(const DevCfg<T>* __restrict__ cfgp, int64_t rows,
                                                          const T* __restrict__ q, const T* __restrict__ qd,
                                                          T* __restrict__ x_out, T* __restrict__ v_out,
                                                          T* __restrict__ a_out) {
  __shared__ T xch[21 * 64];
  const DevCfg<T>& cfg = *cfgp;
  const int lane = threadIdx.x;
  int64_t r = (int64_t)blockIdx.x * 64 + lane;
  const bool active = r < rows;
  if (!active) r = rows - 1;
  T qj[7];  // loads first, then the sincos calls (branches the compiler keeps loads behind): one round trip, not seven
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    qj[j] = q[j * rows + r];
    xch[(3 * j + 2) * 64 + lane] = qd ? qd[j * rows + r] : T(0);
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    T s, c;
    m_sincos(qj[j], &s, &c);
    xch[(3 * j + 0) * 64 + lane] = c;
    xch[(3 * j + 1) * 64 + lane] = s;
  }
  __syncthreads();
  panda_walk_spheres<false, T>(
      cfg, cfg.mount[(int)(r % cfg.n_robots)],
      [&](int j, T& c, T& s, T& qdj) {
        c = xch[(3 * j + 0) * 64 + lane];
        s = xch[(3 * j + 1) * 64 + lane];
        qdj = xch[(3 * j + 2) * 64 + lane];
      },
      [&](int s, const T* x, const T* v, const T* a) {
        if (!active) return;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int64_t idx = (int64_t)(s * 3 + c) * rows + r;
          x_out[idx] = x[c];
          if (v_out) v_out[idx] = v[c];
          if (a_out) a_out[idx] = cfg.jsign * a[c];
        }
      });
}

// ---------------------------------------------------------------------------- robot-sharded rollout step
// predict: q += dt*qdot for the owned robots, publish their spheres as [robot][S][9][B].
template <typename T>
__global__ __launch_bounds__(64) void k_step_predict(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen, int robot_first,
                                                      int robot_count, T* __restrict__ q_io, const T* __restrict__ qd,
                                                      T* __restrict__ sph_own) {
  __shared__ T xch[21 * 64];
  const DevCfg<T>& cfg = *cfgp;
  const int lane = threadIdx.x;
  const int64_t rows = n_scen * robot_count;
  int64_t r = (int64_t)blockIdx.x * 64 + lane;
  const bool active = r < rows;
  if (!active) r = rows - 1;
  const int64_t scen = r / robot_count;
  const int lr = (int)(r - scen * robot_count);
  T qn[7];  // loads first, then the sincos calls (branches the compiler keeps loads behind): one round trip, not seven
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const T qdj = qd[j * rows + r];
    qn[j] = q_io[j * rows + r] + cfg.dt * qdj;
    xch[(3 * j + 2) * 64 + lane] = qdj;
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    if (active) q_io[j * rows + r] = qn[j];
    T s, c;
    m_sincos(qn[j], &s, &c);
    xch[(3 * j + 0) * 64 + lane] = c;
    xch[(3 * j + 1) * 64 + lane] = s;
  }
  __syncthreads();
  const int m01 = cfg.lo_merge01, m45 = cfg.lo_merge45;
  const int SX = cfg.n_spheres - m01 - m45;  // == mrf_exchange_spheres()
  panda_walk_spheres<false, T>(
      cfg, cfg.mount[robot_first + lr],
      [&](int j, T& c, T& s, T& qdj) {
        c = xch[(3 * j + 0) * 64 + lane];
        s = xch[(3 * j + 1) * 64 + lane];
        qdj = xch[(3 * j + 2) * 64 + lane];
      },
      [&](int s, const T* x, const T* v, const T* a) {
        // coincident link origins (DevCfg::lo_merge*) are exchanged once: S - merges slots per robot
        if (!active || (s == 1 && m01) || (s == 5 && m45)) return;
        const int64_t base = ((int64_t)(lr * SX + lo_slot(s, m01, m45)) * 9) * n_scen + scen;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          sph_own[base + (int64_t)c * n_scen] = x[c];
          sph_own[base + (int64_t)(3 + c) * n_scen] = v[c];
          sph_own[base + (int64_t)(6 + c) * n_scen] = cfg.jsign * a[c];
        }
      });
}

// action: fabric solve of the owned robots against every other robot's published spheres.  slots.s[j] is the
// position of robot j's block in sph_all (identity for the [n_robots][SX][9][B] layout of the C ABI; the padded
// [rank][cnt_max] layout of an all-gather with uneven robot blocks otherwise).
struct RobotSlots {
  int s[MRF_MAX_ROBOTS];
};
template <typename T, class LS>
__global__ __launch_bounds__(256) MRF_ATTR_STEP void k_step_action(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen, int robot_first,
                                                      int robot_count, const T* __restrict__ q, T* __restrict__ qd_io,
                                                      const T* __restrict__ prm, const T* __restrict__ sph_all,
                                                      RobotSlots slots, T* __restrict__ sumsq_io) {
  // robot -> block position of sph_all, staged in LDS: the lookup is per lane (each lane has its own "other robot"),
  // and an LDS read does not drain the vector-memory queue the prefetched sphere loads sit in
  __shared__ int slot_of[MRF_MAX_ROBOTS];
  if (threadIdx.x < MRF_MAX_ROBOTS) slot_of[threadIdx.x] = slots.s[threadIdx.x];
  __syncthreads();
  const DevCfg<T>& cfg = *cfgp;
  const int64_t rows = n_scen * robot_count;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int64_t scen = r / robot_count;
  const int lr = (int)(r - scen * robot_count);
  const int me = robot_first + lr;
  PandaState<T> R;
  load_state(rows, r, q, (const T*)qd_io, R);
  PrmView<T> P{prm, rows, r, {T(0), T(0), T(0)}, false};
  const int N = cfg.n_robots;
  const int m01 = cfg.lo_merge01, m45 = cfg.lo_merge45;
  const int SX = cfg.n_spheres - m01 - m45;  // == mrf_exchange_spheres()
  T qdd[7], act[7];
  panda_solve_row<LS, kStepSingleWalk>(
      cfg, cfg.mount[me], R, P,
      [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
        // (other robot, exchanged slot) pairs in one flat, software-pipelined loop: the next sphere's nine scalars
        // are in flight while the current one is folded.  Coincident link origins (DevCfg::lo_merge*) arrive once
        // and count twice.
        const bool dyn = cfg.dynamic != 0;
        pipelined_pairs<T, 9>(
            (N - 1) * SX,
            [&](int m, T (&buf)[9]) {
              const int d = m / SX, slot = m - d * SX;
              int jr = me + 1 + d;
              if (jr >= N) jr -= N;
              const T* src = sph_all + ((int64_t)(slot_of[jr] * SX + slot) * 9) * n_scen + scen;
#pragma unroll
              for (int c = 0; c < 9; ++c) buf[c] = src[(int64_t)c * n_scen];
            },
            [&](int m, T (&buf)[9]) {
              const int slot = m % SX;
              T v[3], a[3];
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                v[c] = dyn ? buf[3 + c] : T(0);
                a[c] = dyn ? buf[6 + c] : T(0);
              }
              accumulate_obstacle<typename LS::Collision>(cfg, E, buf, v, a, cfg.sphere_r[lo_sphere(slot, m01, m45)], false, acc,
                                                          T(lo_count(slot, m01, m45)));
            });
      },
      qdd, act);
  T ss = T(0);
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    qd_io[j * rows + r] = act[j];
    ss += act[j] * act[j];
  }
  sumsq_io[r] += ss;
}

}  // namespace mrf

// ================================================================================ host side / C ABI

namespace {

using mrf_host::check_hip;
using mrf_host::dispatch;
using mrf_host::dispatch_scalar;
using mrf_host::is_link_origin_table;
using mrf_host::is_panda_leafset;
using mrf_host::LeafSetGeneric;
using mrf_host::LeafSetPanda;
using mrf_host::fail;
using mrf_host::launch;

template <typename T>
void to_dev_leaf(const mrf_leaf_fn& s, mrf::LeafFn<T>& d) {
  d.family = s.family;
  d.gate = s.gate;
  d.p = s.p;
  d.pad = 0;
  d.k = (T)s.k;
  d.c = (T)s.c;
  d.s = (T)s.s;
}

template <typename T>
void to_dev_cfg(const mrf_config& c, mrf::DevCfg<T>& d) {
  std::memset(&d, 0, sizeof(d));
  d.model = c.model; d.mode = c.mode; d.n_robots = c.n_robots; d.n_spheres = c.n_spheres;
  d.horizon = c.horizon; d.dynamic = c.dynamic; d.n_ego = c.n_ego; d.n_planes = c.n_planes;
  d.use_limits = c.use_limits; d.n_goals = c.n_goals; d.plane_abs = c.plane_abs;
  d.zero_small = c.zero_small_action; d.obst_dim = c.obst_dim; d.goal_mask = c.goal_estimate_mask;
  d.ego_mask = c.ego_link_mask;
  d.dt = (T)c.dt; d.eps = (T)c.eps; d.jsign = (T)c.jdot_sign; d.goal_T = (T)c.goal_estimate_T;
  d.base_mass = (T)c.base_mass;
  d.attr_k = (T)c.attr_k; d.attr_alpha = (T)c.attr_alpha; d.attr_mu = (T)c.attr_mu; d.attr_ml = (T)c.attr_ml;
  d.attr_a = (T)c.attr_a;
  d.beta_a = (T)c.beta_a; d.beta_r = (T)c.beta_r; d.beta_b = (T)c.beta_b; d.beta_s = (T)c.beta_s;
  d.eta_a = (T)c.eta_a; d.eta_s = (T)c.eta_s;
  for (int i = 0; i < MRF_MAX_ROBOTS; ++i)
    for (int k = 0; k < 12; ++k) d.mount[i][k] = (T)c.mount[i][k];
  for (int j = 0; j < MRF_DOF_MAX; ++j) {
    d.limits[j][0] = (T)c.limits[j][0];
    d.limits[j][1] = (T)c.limits[j][1];
  }
  for (int s = 0; s < MRF_MAX_SPHERES; ++s) {
    d.sphere_link[s] = c.sphere_link[s];
    for (int k = 0; k < 3; ++k) d.sphere_off[s][k] = (T)c.sphere_offset[s][k];
    d.sphere_r[s] = (T)c.sphere_radius[s];
  }
  if (is_link_origin_table(c)) {  // coincident link origins (1,2) and (5,6): one leaf with weight 2 when the radii agree
    d.lo_merge01 = c.sphere_radius[0] == c.sphere_radius[1];
    d.lo_merge45 = c.sphere_radius[4] == c.sphere_radius[5];
  }
  to_dev_leaf(c.collision_geometry, d.cg); to_dev_leaf(c.collision_finsler, d.cf);
  to_dev_leaf(c.plane_geometry, d.pg);     to_dev_leaf(c.plane_finsler, d.pf);
  to_dev_leaf(c.limit_geometry, d.lg);     to_dev_leaf(c.limit_finsler, d.lf);
}

void set_leaf(mrf_leaf_fn& f, int family, int gate, int p, double k, double c, double s) {
  f.family = family; f.gate = gate; f.p = p; f.reserved = 0; f.k = k; f.c = c; f.s = s;
}

void common_defaults(mrf_config* c) {
  std::memset(c, 0, sizeof(*c));
  c->abi_version = MRF_ABI_VERSION;
  c->scalar = MRF_F64;
  c->eps = 1e-6;
  c->jdot_sign = -1.0;
  c->goal_estimate_T = 20 * 0.01;
  c->base_mass = 0.2;
  c->attr_k = 5.0; c->attr_alpha = 10.0; c->attr_mu = 2.0; c->attr_ml = 0.3; c->attr_a = 0.75;
  c->beta_a = 0.5; c->beta_r = 0.02; c->beta_b = 6.5; c->beta_s = 0.01;
  c->eta_a = 0.9 * (1.0 - 0.5); c->eta_s = 0.5;
  c->plane_abs = 1;
  c->zero_small_action = 1;
  c->ego_link_mask = 0x3F;
  c->dt = 0.01;
  // library defaults (recalled, overridable): limits and plane finsler
  set_leaf(c->limit_geometry, MRF_FAMILY_POW, MRF_GATE_NONE, 1, -0.1, 0, 0);
  set_leaf(c->limit_finsler, MRF_FAMILY_POW, MRF_GATE_NEG, 1, 0.1, 0, 0);
  set_leaf(c->plane_finsler, MRF_FAMILY_POW, MRF_GATE_NEG, 1, 0.1, 0, 0);
}

std::string validate(const mrf_config& c) {
  if (c.abi_version != MRF_ABI_VERSION) return "abi_version mismatch";
  if (c.model != MRF_MODEL_PANDA7 && c.model != MRF_MODEL_PLANAR3) return "unknown model";
  if (c.scalar != MRF_F64 && c.scalar != MRF_F32) return "unknown scalar type";
#ifndef MRF_WITH_F32
  if (c.scalar == MRF_F32) return "this build has no float32 kernels (build with -DMRF_WITH_F32; mrf_build_has_f32())";
#endif
  if (c.mode != MRF_MODE_ACC && c.mode != MRF_MODE_VEL) return "unknown mode";
  if (c.n_robots < 1 || c.n_robots > MRF_MAX_ROBOTS) return "n_robots out of range";
  if (c.n_spheres < 0 || c.n_spheres > MRF_MAX_SPHERES) return "n_spheres out of range";
  if (c.horizon < 1) return "horizon must be >= 1";
  if (c.model == MRF_MODEL_PANDA7 && c.n_ego != 0 && c.n_ego != MRF_N_EGO) return "panda n_ego must be 0 or 6";
  if (c.model == MRF_MODEL_PANDA7 && c.n_ego == MRF_N_EGO && (c.ego_link_mask < 1 || c.ego_link_mask > 0x3F))
    return "panda ego_link_mask must select at least one of links 3..8 (bits 0..5); use n_ego = 0 for none";
  if (c.model == MRF_MODEL_PLANAR3 && c.n_ego != 0 && c.n_ego != 1) return "planar3 n_ego must be 0 or 1";
  if (c.n_planes < 0 || c.n_planes > 1) return "n_planes must be 0 or 1";
  if (c.model == MRF_MODEL_PANDA7 && (c.n_goals < 0 || c.n_goals > 3)) return "panda n_goals must be 0..3";
  if (c.model == MRF_MODEL_PLANAR3 && (c.n_goals < 0 || c.n_goals > 1)) return "planar3 n_goals must be 0..1";
  if (c.obst_dim != 2 && c.obst_dim != 3) return "obst_dim must be 2 or 3";
  if (c.kernel_select < 0 || c.kernel_select > 3)
    return "kernel_select must be 0 (auto), 1 (row-per-lane), 2 (cooperative) or 3 (wave pair per row)";
  if (!(c.dt > 0) || !(c.eps > 0)) return "dt and eps must be positive";
  if (c.exchange != MRF_EXCHANGE_JOINTS && c.exchange != MRF_EXCHANGE_SPHERES)
    return "exchange must be MRF_EXCHANGE_JOINTS (0) or MRF_EXCHANGE_SPHERES (1)";
  int prev = 0;
  for (int s = 0; s < c.n_spheres; ++s) {
    int L = c.sphere_link[s];
    if (L < 1 || L > 8) return "sphere_link must be in 1..8";
    if (L < prev) return "sphere table must be sorted by link number";
    prev = L;
  }
  const mrf_leaf_fn* fns[6] = {&c.collision_geometry, &c.collision_finsler, &c.plane_geometry,
                               &c.plane_finsler, &c.limit_geometry, &c.limit_finsler};
  for (const mrf_leaf_fn* f : fns) {
    if (f->family != MRF_FAMILY_POW && f->family != MRF_FAMILY_LOGISTIC) return "unknown leaf family";
    if (f->gate != MRF_GATE_NONE && f->gate != MRF_GATE_NEG) return "unknown leaf gate";
    if (f->p < 0 || f->p > 16) return "leaf exponent p must be in 0..16";
  }
  return "";
}

// Cooperative (one wave per scenario) kernels pay ~5x the total work of the row-per-lane kernels but finish a
// scenario ~4x sooner; they win while the row-per-lane grid cannot fill the chip.  cfg.kernel_select overrides.
bool use_coop(const mrf_handle* h, int64_t n_scen) {
  if (5 * h->cfg.n_robots > 64) return false;
  if (h->cfg.kernel_select == 1 || h->cfg.kernel_select == 3) return false;
  if (h->cfg.kernel_select == 2) return true;
  return n_scen <= h->coop_max_scen;
}


template <bool ROLLOUT, bool CART = false>
int launch_coop(mrf_handle* h, int64_t n_scen, const void* q, const void* qd, const void* prm, int use_accel, void* avg,
                void* traj_q, void* traj_qd, void* qdd_out, void* act_out, hipStream_t st) {
  const bool lo = is_link_origin_table(h->cfg);
  const int S = lo ? 8 : h->cfg.n_spheres;
  return dispatch(h, [&](auto t, auto cl) {
    using T = decltype(t);
    using LS = decltype(cl);
    const size_t lds = sizeof(T) * (21 * 64 + (size_t)h->cfg.n_robots * (S * 9 + MRF_NPARAM));
    dim3 block(64), grid((unsigned)n_scen);
    auto k = lo ? mrf::k_coop_panda<T, LS, true, ROLLOUT, CART> : mrf::k_coop_panda<T, LS, false, ROLLOUT, CART>;
    hipLaunchKernelGGL(k, grid, block, lds, st, (const mrf::DevCfg<T>*)h->dcfg, n_scen, (const T*)q, (const T*)qd,
                       (const T*)prm, use_accel, (T*)avg, (T*)traj_q, (T*)traj_qd, (T*)qdd_out, (T*)act_out);
    return check_hip(h, hipGetLastError(), "kernel launch");
  });
}

// The wave-pair form of the joint-space rollout (mrf_rollout_wp.hpp): float64, the reference's leaf strings and full
// collision-link set, the link-origin sphere table with equal radii on the coincident origins.  Selected by
// mrf_config.kernel_select = 3 (anything it does not cover runs the row-per-lane kernel), or for every handle by the
// environment variable MRF_ROLLOUT_WP=1 (A/B on one box).  Not the default: measured 7-9 % slower than the row-per-lane
// kernel on BASELINE config 4 (profiles/r05_wp_*.json).
bool wave_pair_applies(const mrf_handle* h) {
#ifndef MRF_WITH_WP
  // default build (r6): the wave-pair kernel is not compiled in (measured 7-9 % slower, never auto-selected); kernel_select
  // = 3 then runs the row-per-lane kernel, exactly as it always has for every configuration the pair does not cover
  (void)h;
  return false;
#else
  static const bool env_on = [] {
    const char* e = getenv("MRF_ROLLOUT_WP");
    return e && e[0] == '1';
  }();
  if (h->cfg.kernel_select != 3 && !(env_on && h->cfg.kernel_select != 2)) return false;
  const mrf_config& c = h->cfg;
  if (c.scalar != MRF_F64 || !is_panda_leafset(c) || !is_link_origin_table(c)) return false;
  if (c.n_ego != MRF_N_EGO) return false;
  if (c.sphere_radius[0] != c.sphere_radius[1] || c.sphere_radius[4] != c.sphere_radius[5]) return false;
  return true;
#endif
}

}  // namespace

bool mrf_host::coop_applies(const mrf_handle* h, int64_t n_scen) { return use_coop(h, n_scen); }

bool mrf_host::cartesian_tile_applies(const mrf_handle* h) {
  static const bool off = [] {
    const char* e = getenv("MRF_CART_TILE");
    return e && e[0] == '0';
  }();
  // tables of more than eight spheres per robot keep the obstacle-array path: re-deriving them on chip in every step
  // (k_rollout_cartc_panda MODE 2) was built and measured 11 % slower at 32 spheres (profiles/r05_cartc32.json);
  // MRF_CART_CHUNKED=1 selects it for every table (A/B, parity tests)
  static const bool chunked = [] {
    const char* e = getenv("MRF_CART_CHUNKED");
    return e && e[0] == '1';
  }();
  return !off && h->cfg.n_robots >= 2 && h->cfg.n_robots <= 64 && h->cfg.mode == MRF_MODE_VEL && h->cfg.n_spheres >= 1 &&
         (is_link_origin_table(h->cfg) || h->cfg.n_spheres <= 8 || chunked);
}

// mrf_rollout_cartesian_coupled with the other robots' spheres kept on chip (mode 'vel'): start spheres in the LDS tile for
// the link-origin table and for tables of up to eight spheres (larger tables: obstacle arrays, see cartesian_tile_applies);
// 1 = this form does not apply here.  MRF_CART_TILE=0 keeps the obstacle-array path everywhere (A/B).
int mrf_host::rollout_cartesian_tile(mrf_handle* h, int64_t n_scen, const void* q0, const void* qdot0, const void* params,
                                     void* avg_out, void* traj_q, void* traj_qd, void* stream) {
  const bool lo = is_link_origin_table(h->cfg);
  if (!cartesian_tile_applies(h)) return 1;
  const int spw = 64 / h->cfg.n_robots;
  dim3 block(64), grid((unsigned)((n_scen + spw - 1) / spw));
  return dispatch(h, [&](auto t, auto cl) {
    using T = decltype(t);
    using LS = decltype(cl);
    // MRF_CART_CHUNKED=1: every non-link-origin table is re-derived from the start states in every step (MODE 2)
    static const bool chunk_all = [] {
      const char* e = getenv("MRF_CART_CHUNKED");
      return e && e[0] == '1';
    }();
    auto k = lo ? mrf::k_rollout_cartc_panda<T, LS, 0>
                : ((h->cfg.n_spheres <= 8 && !chunk_all) ? mrf::k_rollout_cartc_panda<T, LS, 1> : mrf::k_rollout_cartc_panda<T, LS, 2>);
    return launch(h, k, grid, block, (hipStream_t)stream, (const mrf::DevCfg<T>*)h->dcfg, n_scen, (const T*)q0, (const T*)qdot0,
                  (const T*)params, (T*)avg_out, (T*)traj_q, (T*)traj_qd);
  });
}

// mrf_rollout_cartesian_coupled for large sphere tables: obstacle assembly in the rollout kernel's prologue
// (k_rollout_carts_panda); wx / wv = the handle's work arrays [M][3][rows]
int mrf_host::rollout_cartesian_self(mrf_handle* h, int64_t n_scen, const void* q0, const void* qdot0, const void* params,
                                     void* wx, void* wv, void* avg_out, void* traj_q, void* traj_qd, void* stream) {
  const int spw = 64 / h->cfg.n_robots;
  const dim3 block(64), grid((unsigned)((n_scen + spw - 1) / spw));
  return dispatch(h, [&](auto t, auto cl) {
    using T = decltype(t);
    using LS = decltype(cl);
    auto go = [&](auto kernel) {
      return launch(h, kernel, grid, block, (hipStream_t)stream, (const mrf::DevCfg<T>*)h->dcfg, n_scen, (const T*)q0,
                    (const T*)qdot0, (const T*)params, (T*)wx, (T*)wv, (T*)avg_out, (T*)traj_q, (T*)traj_qd);
    };
    bool uniform = true;  // one radius for the whole table (the reference's tables: PM:23)
    for (int s = 1; s < h->cfg.n_spheres; ++s) uniform = uniform && h->cfg.sphere_radius[s] == h->cfg.sphere_radius[0];
    return uniform ? go(mrf::k_rollout_carts_panda<T, LS, true>) : go(mrf::k_rollout_carts_panda<T, LS, false>);
  });
}

// mrf_rollout_cartesian_coupled in latency mode (one wave per scenario); 1 = the cooperative form does not apply here
int mrf_host::rollout_cartesian_coop(mrf_handle* h, int64_t n_scen, const void* q0, const void* qdot0, const void* params,
                                     void* avg_out, void* traj_q, void* traj_qd, void* stream) {
  if (h->cfg.n_robots < 2 || h->cfg.n_spheres < 1 || !use_coop(h, n_scen)) return 1;
  return launch_coop<true, true>(h, n_scen, q0, qdot0, params, 0, avg_out, traj_q, traj_qd, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" {

int mrf_abi_version(void) { return MRF_ABI_VERSION; }

int mrf_build_has_f32(void) {
#ifdef MRF_WITH_F32
  return 1;
#else
  return 0;
#endif
}

int mrf_build_has_wp(void) {
#ifdef MRF_WITH_WP
  return 1;
#else
  return 0;
#endif
}

#ifdef MRF_COOP_CLOCKS
int mrf_debug_clocks(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(mrf::mrf_dbg_clocks), sizeof(long long) * (n < 32 ? n : 32)) == hipSuccess ? 0 : -1;
}
#endif
int64_t mrf_config_sizeof(void) { return (int64_t)sizeof(mrf_config); }

void mrf_default_config_panda(mrf_config* c, int32_t n_robots, int32_t horizon) {
  common_defaults(c);
  c->model = MRF_MODEL_PANDA7;
  c->mode = MRF_MODE_VEL;  // parameters_manipulators.py:12
  c->n_robots = n_robots;
  c->horizon = horizon;
  c->dynamic = 1;
  c->n_ego = MRF_N_EGO;
  c->n_planes = 1;
  c->use_limits = 1;
  c->n_goals = 3;
  c->obst_dim = 3;
  const double kPi = 3.14159265358979323846;
  for (int i = 0; i < n_robots && i < MRF_MAX_ROBOTS; ++i) {
    double px, py, yaw;
    if (n_robots <= 3) {  // parameters_manipulators.py:83-105,138-150
      const double P[3][2] = {{0.0, 0.0}, {1.0, 0.0}, {0.7, 0.6}};
      px = P[i][0]; py = P[i][1]; yaw = i == 0 ? 0.0 : kPi;
    } else {              // build-defined ring (the reference defines no layout for N > 3)
      double rad = 0.15 * n_robots > 0.75 ? 0.15 * n_robots : 0.75, ang = 2.0 * kPi * i / n_robots;
      px = 0.5 + rad * std::cos(ang); py = rad * std::sin(ang); yaw = ang + kPi;
    }
    double cy = std::cos(yaw), sy = std::sin(yaw);
    double M[12] = {cy, -sy, 0, px, sy, cy, 0, py, 0, 0, 1, 0.65};
    std::memcpy(c->mount[i], M, sizeof(M));
  }
  const double lim[7][2] = {{-2.8973, 2.8973}, {-1.7628, 1.7628}, {-2.8973, 2.8973}, {-3.0718, -0.0698},
                            {-2.8973, 2.8973}, {-0.0175, 3.7525}, {-2.8973, 2.8973}};  // EXJ:97-105
  std::memcpy(c->limits, lim, sizeof(lim));
  c->n_spheres = 8;  // link origins 1..8, radius 0.08 (PM:23-26)
  for (int s = 0; s < 8; ++s) {
    c->sphere_link[s] = s + 1;
    c->sphere_radius[s] = 0.08;
  }
  set_leaf(c->collision_geometry, MRF_FAMILY_POW, MRF_GATE_NONE, 4, -0.5, 0, 0);  // EXJ:88
  set_leaf(c->collision_finsler, MRF_FAMILY_POW, MRF_GATE_NONE, 4, 0.01, 0, 0);   // EXJ:89
  set_leaf(c->plane_geometry, MRF_FAMILY_LOGISTIC, MRF_GATE_NONE, 0, 10.0, 1.0, 10.0);  // EXJ:87
}

void mrf_default_config_planar3(mrf_config* c, int32_t n_robots) {
  common_defaults(c);
  c->model = MRF_MODEL_PLANAR3;
  c->mode = MRF_MODE_ACC;  // concretize() default, pointmass :128
  c->n_robots = n_robots;
  c->horizon = 1;
  c->dynamic = 1;
  c->n_ego = 1;
  c->n_planes = 0;
  c->use_limits = 0;
  c->n_goals = 1;
  c->obst_dim = 3;
  for (int i = 0; i < n_robots && i < MRF_MAX_ROBOTS; ++i) {
    double M[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    std::memcpy(c->mount[i], M, sizeof(M));
  }
  c->n_spheres = 1;
  c->sphere_link[0] = 1;
  c->sphere_radius[0] = 0.2;
  set_leaf(c->collision_geometry, MRF_FAMILY_POW, MRF_GATE_NONE, 1, -2.0, 0, 0);  // pointmass :106
  set_leaf(c->collision_finsler, MRF_FAMILY_POW, MRF_GATE_NEG, 2, 1.0, 0, 0);     // pointmass :107
  set_leaf(c->plane_geometry, MRF_FAMILY_POW, MRF_GATE_NEG, 5, -0.5, 0, 0);
}

int mrf_create(const mrf_config* cfg, int32_t device_id, mrf_handle** out) {
  if (!cfg || !out) return MRF_E_ARG;
  *out = nullptr;
  mrf_handle* h = new (std::nothrow) mrf_handle();
  if (!h) return MRF_E_ARG;
  static std::atomic<uint64_t> next_serial{1};
  h->cfg = *cfg;
  h->device = device_id;
  h->serial = next_serial.fetch_add(1);
  h->dcfg = nullptr;
  *out = h;  // returned even on failure so that mrf_last_error() can be read; caller destroys it
  std::string v = validate(*cfg);
  if (!v.empty()) return fail(h, MRF_E_CONFIG, "invalid mrf_config: " + v);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(h, MRF_E_DEVICE, "no HIP device available (there is no CPU path)");
  if (device_id < 0 || device_id >= ndev) return fail(h, MRF_E_DEVICE, "device_id out of range");
  mrf_host::DeviceGuard guard(device_id);  // upload on the handle's device, then restore the caller's current device
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != device_id) return fail(h, MRF_E_DEVICE, "hipSetDevice failed");
  hipError_t e;
  if (cfg->scalar == MRF_F64) {
    mrf::DevCfg<double> d;
    to_dev_cfg(*cfg, d);
    e = hipMalloc(&h->dcfg, sizeof(d));
    if (e == hipSuccess) e = hipMemcpy(h->dcfg, &d, sizeof(d), hipMemcpyHostToDevice);
  } else {
    mrf::DevCfg<float> d;
    to_dev_cfg(*cfg, d);
    e = hipMalloc(&h->dcfg, sizeof(d));
    if (e == hipSuccess) e = hipMemcpy(h->dcfg, &d, sizeof(d), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) return fail(h, MRF_E_DEVICE, std::string("config upload: ") + hipGetErrorString(e));
  if (hipMalloc(&h->clock_probe, 9 * sizeof(long long)) != hipSuccess || hipMemset(h->clock_probe, 0, 9 * sizeof(long long)) != hipSuccess)
    return fail(h, MRF_E_DEVICE, "clock probe buffer");
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id);
  // measured crossover (tools/crossover.py, 3-Panda H=30, r01 v4): cooperative 0.35-0.42 ms up to one round of 4 waves
  // per CU (1024 scenarios), 0.80 ms at 1.5 rounds, against a flat 0.49 ms of the row-per-lane kernel up to 8192
  // scenarios -> cooperative up to one round
  h->coop_max_scen = (int64_t)cus * 4;
  h->n_cus = cus;
  return MRF_OK;
}

void mrf_destroy(mrf_handle* h) {
  if (!h) return;
  mrf_host::DeviceGuard guard(h->dcfg ? h->device : -1);
  mrf_host::comm_release(h);
  mrf_host::staging_release(h);
  mrf_host::cart_work_release(h);
  if (h->graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)h->graph_exec);
  if (h->own_stream) (void)hipStreamDestroy((hipStream_t)h->own_stream);
  if (h->dcfg) (void)hipFree(h->dcfg);
  if (h->clock_probe) (void)hipFree(h->clock_probe);
  delete h;
}

const char* mrf_last_error(const mrf_handle* h) { return h ? h->err.c_str() : "null handle"; }


int mrf_compute_action(mrf_handle* h, int64_t rows, const void* q, const void* qdot, const void* params,
                       int32_t n_obst, int32_t n_obst_static, const void* ox, const void* ov, const void* oa,
                       const void* orad, void* qddot_out, void* action_out, void* stream) {
  MRF_CHECK_READY(h);
  if (rows == 0) return MRF_OK;  // empty batch: nothing to read or write, pointers may be NULL
  if (rows < 0 || n_obst < 0 || n_obst_static < 0 || n_obst_static > n_obst || !q || !qdot || !params || !action_out)
    return fail(h, MRF_E_ARG, "null/negative argument");
  if (n_obst > 0 && (!ox || !orad)) return fail(h, MRF_E_ARG, "obstacle arrays missing");
  hipStream_t st = (hipStream_t)stream;
  dim3 block(256), grid((unsigned)((rows + 255) / 256));
  if (h->cfg.model == MRF_MODEL_PLANAR3)
    return dispatch_scalar(h, [&](auto t) {
      using T = decltype(t);
      return launch(h, mrf::k_action_planar<T>, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, rows, (const T*)q,
                    (const T*)qdot, (const T*)params, (int)n_obst, (int)n_obst_static, (const T*)ox, (const T*)ov,
                    (const T*)oa, (const T*)orad, (T*)qddot_out, (T*)action_out);
    });
  return dispatch(h, [&](auto t, auto cl) {
    using T = decltype(t);
    using LS = decltype(cl);
    auto go = [&](auto kernel) {
      return launch(h, kernel, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, rows, (const T*)q, (const T*)qdot,
                    (const T*)params, (int)n_obst, (int)n_obst_static, (const T*)ox, (const T*)ov, (const T*)oa,
                    (const T*)orad, (T*)qddot_out, (T*)action_out);
    };
    // no obstacle accelerations (obst_a == NULL; the reference's drivers pass zeros, EXJ:411): 7 loads per obstacle
    if (oa) return go(mrf::k_action_panda<T, LS, true>);
    return go(mrf::k_action_panda<T, LS, false>);
  });
}

int mrf_compute_action_coupled(mrf_handle* h, int64_t n_scen, const void* q, const void* qdot, const void* params,
                               int32_t use_accel, void* qddot_out, void* action_out, void* stream) {
  MRF_CHECK_READY(h);
  if (h->cfg.model != MRF_MODEL_PANDA7) return fail(h, MRF_E_CONFIG, "coupled compute_action is defined for the panda7 model only");
  if (h->cfg.n_robots > 64) return fail(h, MRF_E_CONFIG, "n_robots > 64");
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || !q || !qdot || !params || !action_out) return fail(h, MRF_E_ARG, "null/negative argument");
  hipStream_t st = (hipStream_t)stream;
  if (use_coop(h, n_scen))
    return launch_coop<false>(h, n_scen, q, qdot, params, (int)use_accel, nullptr, nullptr, nullptr, qddot_out, action_out, st);
  const int spw = 64 / h->cfg.n_robots;
  // persistent: at most the resident workgroup count (one single-wave workgroup per SIMD, register-bound), each walking
  // its blocks with the next block's loads in flight (k_action_coupled)
  const int64_t nblk = (n_scen + spw - 1) / spw, resident = (int64_t)4 * h->n_cus;
  dim3 block(64), grid((unsigned)(nblk < resident ? nblk : resident));
  return dispatch(h, [&](auto t, auto cl) {
    using T = decltype(t);
    using LS = decltype(cl);
    auto go = [&](auto kernel) {
      return launch(h, kernel, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, n_scen, (const T*)q, (const T*)qdot,
                    (const T*)params, (int)use_accel, (T*)qddot_out, (T*)action_out);
    };
    if (is_link_origin_table(h->cfg)) return go(mrf::k_action_coupled<T, LS, mrf::TAB_LO>);
    static const bool no_packed = [] {  // A/B switch: MRF_ACTION_PACKED=0 keeps small tables on the chunked exchange
      const char* e = getenv("MRF_ACTION_PACKED");
      return e && e[0] == '0';
    }();
    if (!use_accel && !no_packed && h->cfg.n_spheres >= 1 && h->cfg.n_spheres <= mrf::TILE_PACKED_MAX)
      return go(mrf::k_action_coupled<T, LS, mrf::TAB_PACKED>);
    return go(mrf::k_action_coupled<T, LS, mrf::TAB_GENERIC>);
  });
}

int mrf_rollout(mrf_handle* h, int64_t n_scen, const void* q0, const void* qdot0, const void* params,
                void* avg_out, void* traj_q, void* traj_qd, void* stream) {
  MRF_CHECK_READY(h);
  if (h->cfg.model != MRF_MODEL_PANDA7) return fail(h, MRF_E_CONFIG, "rollouts are defined for the panda7 model only");
  if (h->cfg.mode != MRF_MODE_VEL)
    return fail(h, MRF_E_CONFIG, "joint-space rollout is defined for mode 'vel' only (reference FPJ:233)");
  if (h->cfg.n_robots > 64) return fail(h, MRF_E_CONFIG, "n_robots > 64");
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || !q0 || !qdot0 || !params || !avg_out) return fail(h, MRF_E_ARG, "null/negative argument");
  hipStream_t st = (hipStream_t)stream;
  h->rollout_serial += 1;  // the row kernels stamp it next to their clock probe; a cooperative launch leaves the old one
  if (use_coop(h, n_scen))
    return launch_coop<true>(h, n_scen, q0, qdot0, params, 1, avg_out, traj_q, traj_qd, nullptr, nullptr, st);
  const int spw = 64 / h->cfg.n_robots;
  dim3 block(64), grid((unsigned)((n_scen + spw - 1) / spw));
#ifdef MRF_WITH_WP
  if (wave_pair_applies(h)) return mrf_host::rollout_wave_pair(h, n_scen, q0, qdot0, params, avg_out, traj_q, traj_qd, stream);
#endif
  return dispatch(h, [&](auto t, auto cl) {
    using T = decltype(t);
    using LS = decltype(cl);
    if (is_link_origin_table(h->cfg))
      return launch(h, mrf::k_rollout_panda<T, LS, true>, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, n_scen,
                    (const T*)q0, (const T*)qdot0, (const T*)params, (T*)avg_out, (T*)traj_q, (T*)traj_qd,
                    (long long*)h->clock_probe, h->rollout_serial);
    return launch(h, mrf::k_rollout_panda<T, LS, false>, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, n_scen,
                  (const T*)q0, (const T*)qdot0, (const T*)params, (T*)avg_out, (T*)traj_q, (T*)traj_qd,
                  (long long*)h->clock_probe, h->rollout_serial);
  });
}

int mrf_rollout_clock(mrf_handle* h, double* out, int32_t n) {
  MRF_CHECK_READY(h);
  if (!out || n < 1) return fail(h, MRF_E_ARG, "null/negative argument");
  if (int rc = check_hip(h, hipDeviceSynchronize(), "hipDeviceSynchronize")) return rc;
  long long st[9];
  if (int rc = check_hip(h, hipMemcpy(st, h->clock_probe, sizeof(st), hipMemcpyDeviceToHost), "hipMemcpy")) return rc;
  int wall_khz = 100000;
  (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, h->device);
  double vals[MRF_ROLLOUT_CLOCK_N] = {0, 0, 0, 0, (double)wall_khz * 1e-6};
  // stamps of an OLDER call (the last mrf_rollout took a cooperative kernel, which does not stamp): report none
  const bool current = h->rollout_serial > 0 && st[8] == h->rollout_serial;
  for (int b = 0; b < 2 && current; ++b) {
    const double cycles = (double)(st[4 * b + 2] - st[4 * b + 0]), ticks = (double)(st[4 * b + 3] - st[4 * b + 1]);
    if (ticks > 0 && cycles > 0) {
      vals[b] = cycles / ticks * (double)wall_khz * 1e-6;  // shader cycles per wall-clock tick x tick rate [GHz]
      vals[2 + b] = ticks / (double)wall_khz;              // lifetime of that workgroup [ms]
    }
  }
  for (int i = 0; i < n && i < MRF_ROLLOUT_CLOCK_N; ++i) out[i] = vals[i];
  return MRF_OK;
}

int mrf_rollout_cartesian(mrf_handle* h, int64_t rows, const void* q0, const void* qdot0, const void* params,
                          int32_t n_obst, int32_t n_obst_static, const void* ox0, const void* ov, const void* oa,
                          const void* orad, void* avg_out, void* traj_q, void* traj_qd, void* stream) {
  MRF_CHECK_READY(h);
  if (h->cfg.model != MRF_MODEL_PANDA7) return fail(h, MRF_E_CONFIG, "rollouts are defined for the panda7 model only");
  if (rows == 0) return MRF_OK;
  if (rows < 0 || n_obst < 0 || n_obst_static < 0 || n_obst_static > n_obst || !q0 || !qdot0 || !params || !avg_out)
    return fail(h, MRF_E_ARG, "null/negative argument");
  if (n_obst > 0 && (!ox0 || !orad || (n_obst > n_obst_static && !ov))) return fail(h, MRF_E_ARG, "obstacle arrays missing");
  hipStream_t st = (hipStream_t)stream;
  dim3 block(256), grid((unsigned)((rows + 255) / 256));
  return dispatch(h, [&](auto t, auto cl) {
    using T = decltype(t);
    using LS = decltype(cl);
    auto go = [&](auto kernel, dim3 g, dim3 b) {
      return launch(h, kernel, g, b, st, (const mrf::DevCfg<T>*)h->dcfg, rows, (const T*)q0, (const T*)qdot0,
                    (const T*)params, (int)n_obst, (int)n_obst_static, (const T*)ox0, (const T*)ov, (const T*)oa,
                    (const T*)orad, (T*)avg_out, (T*)traj_q, (T*)traj_qd);
    };
    // oa == NULL (the reference's own use: zero obstacle accelerations, FPC:33) takes the instantiation without the
    // acceleration loads, with 10 instead of 7 obstacles of a row resident in LDS (f64)
#ifdef MRF_CART_STREAM_ONLY
    if (oa) return go(mrf::k_rollout_cart_panda<T, LS, false, true>, grid, block);
    return go(mrf::k_rollout_cart_panda<T, LS, false, false>, grid, block);
#else
    const dim3 g64((unsigned)((rows + 63) / 64)), b64(64);
    if (oa) return go(mrf::k_rollout_cart_panda<T, LS, true, true>, g64, b64);
    return go(mrf::k_rollout_cart_panda<T, LS, true, false>, g64, b64);
#endif
  });
}

int mrf_fk_spheres(mrf_handle* h, int64_t rows, const void* q, const void* qdot, void* x_out, void* v_out,
                   void* a_out, void* stream) {
  MRF_CHECK_READY(h);
  if (h->cfg.model != MRF_MODEL_PANDA7) return fail(h, MRF_E_CONFIG, "fk_spheres is defined for the panda7 model only");
  if (rows == 0) return MRF_OK;
  if (rows < 0 || !q || !x_out) return fail(h, MRF_E_ARG, "null/negative argument");
  if ((v_out || a_out) && !qdot) return fail(h, MRF_E_ARG, "qdot required for v/a");
  hipStream_t st = (hipStream_t)stream;
  dim3 block(64), grid((unsigned)((rows + 63) / 64));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_fk_spheres_panda<T>, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, rows, (const T*)q,
                  (const T*)qdot, (T*)x_out, (T*)v_out, (T*)a_out);
  });
}

int mrf_rollout_sphere_traj(mrf_handle* h, int64_t n_scen, const void* qdot0, const void* traj_q, const void* traj_qd,
                            void* x_out, void* v_out, void* a_out, void* stream) {
  MRF_CHECK_READY(h);
  if (h->cfg.model != MRF_MODEL_PANDA7) return fail(h, MRF_E_CONFIG, "rollouts are defined for the panda7 model only");
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || !qdot0 || !traj_q || !traj_qd || !x_out) return fail(h, MRF_E_ARG, "null/negative argument");
  const size_t sb = h->cfg.scalar == MRF_F64 ? 8 : 4;
  const size_t rows = (size_t)n_scen * h->cfg.n_robots;
  const size_t step_state = 7 * rows * sb, step_out = (size_t)h->cfg.n_spheres * 3 * rows * sb;
  for (int k = 0; k < h->cfg.horizon; ++k) {
    const char* qk = (const char*)traj_q + (size_t)k * step_state;
    const char* qdk = k == 0 ? (const char*)qdot0 : (const char*)traj_qd + (size_t)(k - 1) * step_state;
    char* xo = (char*)x_out + (size_t)k * step_out;
    char* vo = v_out ? (char*)v_out + (size_t)k * step_out : nullptr;
    char* ao = a_out ? (char*)a_out + (size_t)k * step_out : nullptr;
    if (int rc = mrf_fk_spheres(h, (int64_t)rows, qk, qdk, xo, vo, ao, stream)) return rc;
  }
  return MRF_OK;
}

int32_t mrf_exchange_spheres(const mrf_handle* h) {
  if (!h) return 0;
  const mrf_config& c = h->cfg;
  int n = c.n_spheres;
  if (is_link_origin_table(c)) n -= (c.sphere_radius[0] == c.sphere_radius[1]) + (c.sphere_radius[4] == c.sphere_radius[5]);
  return n;
}

int mrf_step_predict(mrf_handle* h, int64_t n_scen, int32_t robot_first, int32_t robot_count, void* q_io,
                     const void* qdot, void* sph_own, void* stream) {
  MRF_CHECK_READY(h);
  if (h->cfg.model != MRF_MODEL_PANDA7 || h->cfg.mode != MRF_MODE_VEL)
    return fail(h, MRF_E_CONFIG, "sharded rollout needs the panda7 model in mode 'vel'");
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || robot_first < 0 || robot_count < 1 || robot_first + robot_count > h->cfg.n_robots || !q_io ||
      !qdot || !sph_own)
    return fail(h, MRF_E_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = n_scen * robot_count;
  dim3 block(64), grid((unsigned)((rows + 63) / 64));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_step_predict<T>, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, n_scen, (int)robot_first,
                  (int)robot_count, (T*)q_io, (const T*)qdot, (T*)sph_own);
  });
}

int mrf_step_action(mrf_handle* h, int64_t n_scen, int32_t robot_first, int32_t robot_count, const void* q,
                    void* qdot_io, const void* params, const void* sph_all, void* sumsq_io, void* stream) {
  return mrf_host::step_action_slots(h, n_scen, robot_first, robot_count, q, qdot_io, params, sph_all, nullptr, sumsq_io, stream);
}

}  // extern "C"

int mrf_host::step_action_slots(mrf_handle* h, int64_t n_scen, int32_t robot_first, int32_t robot_count, const void* q,
                                void* qdot_io, const void* params, const void* sph_all, const int32_t* robot_slot,
                                void* sumsq_io, void* stream) {
  MRF_CHECK_READY(h);
  if (h->cfg.model != MRF_MODEL_PANDA7 || h->cfg.mode != MRF_MODE_VEL)
    return fail(h, MRF_E_CONFIG, "sharded rollout needs the panda7 model in mode 'vel'");
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || robot_first < 0 || robot_count < 1 || robot_first + robot_count > h->cfg.n_robots || !q ||
      !qdot_io || !params || !sph_all || !sumsq_io)
    return fail(h, MRF_E_ARG, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = n_scen * robot_count;
  dim3 block(256), grid((unsigned)((rows + 255) / 256));
  mrf::RobotSlots slots;
  for (int j = 0; j < MRF_MAX_ROBOTS; ++j) slots.s[j] = robot_slot ? robot_slot[j] : j;
  return dispatch(h, [&](auto t, auto cl) {
    using T = decltype(t);
    using LS = decltype(cl);
    return launch(h, mrf::k_step_action<T, LS>, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, n_scen,
                  (int)robot_first, (int)robot_count, (const T*)q, (T*)qdot_io, (const T*)params, (const T*)sph_all,
                  slots, (T*)sumsq_io);
  });
}
