// mrf_shard_step.hip -- the per-step kernels of a robot-sharded rollout with the MRF_EXCHANGE_JOINTS payload
// (include/mrf.h mrf_step_predict_joints / mrf_step_action_joints; the RCCL transport of mrf_rollout_sharded and the
// Python 'torch' transport drive them).  One rollout step (FPJ:190-249 with the exchange FPJ:211-225 across ranks):
//
//   k_step_predict_joints   q += dt*qdot for the owned robots (system_step 'vel', FPJ:77-80); cos q, sin q, qdot -> jst_own
//   <all-gather of the ranks' joint-state blocks>
//   k_step_action_joints    one wave = the owned robots of floor(64/count) scenarios, adjacent lanes.  The owned robots
//                           exchange on chip exactly as k_rollout_panda does (LDS tile / chunked generic exchange); the
//                           robots of other ranks are re-walked from the gathered joint states (mrf_shard.hpp).  The own
//                           cos q / sin q / qdot are read back from the gathered array, so the step costs no sincos.
//
//   k_step_action_joints<.., NEXT = true> (mrf_step_action_predict_joints) also does the position update of the FOLLOWING step
//                           and writes its joint state into the rank's send block: one launch per step instead of two (what
//                           the RCCL transport's loop runs; the update uses the fused kernel's incremental rotation of
//                           cos q / sin q, so it costs no sincos either).
//
// The sphere payload of the same step is k_step_predict / k_step_action in mrf_kernels.hip (everything through memory:
// the literal "all-gather of sphere centres").
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mrf_device.hpp"
#include "mrf_host.hpp"
#include "mrf_shard.hpp"

namespace mrf {

template <typename T>
__global__ __launch_bounds__(256) void k_step_predict_joints(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen, int robot_count,
                                                              T* __restrict__ q_io, const T* __restrict__ qd,
                                                              T* __restrict__ jst_own) {
  const DevCfg<T>& cfg = *cfgp;
  const int64_t rows = n_scen * robot_count;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int64_t scen = r / robot_count;
  const int lr = (int)(r - scen * robot_count);
  T qn[7], qv[7];  // loads first, then the sincos calls (branches the compiler keeps loads behind)
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    qv[j] = qd[j * rows + r];
    qn[j] = q_io[j * rows + r] + cfg.dt * qv[j];
  }
  T* dst = jst_own + ((int64_t)lr * MRF_JOINT_STATE_SCALARS) * n_scen + scen;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    q_io[j * rows + r] = qn[j];
    T s, c;
    m_sincos(qn[j], &s, &c);
    dst[(int64_t)(3 * j + 0) * n_scen] = c;
    dst[(int64_t)(3 * j + 1) * n_scen] = s;
    dst[(int64_t)(3 * j + 2) * n_scen] = qv[j];
  }
}

struct JointSlots {
  int s[MRF_MAX_ROBOTS];  // position of robot j's [21][B] block in jst_all (identity, or the padded gather layout)
};

// jst_all and jst_next are NOT restrict: a group of one rank gathers in place (its send block is the gathered array; every
// lane then reads its own robot's 21 scalars at the start and overwrites exactly those at the end).
template <typename T, class LS, bool LO, int XK, bool NEXT>
__global__ __launch_bounds__(64) void k_step_action_joints(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen, int first,
                                                            int count, T* __restrict__ q_io, T* __restrict__ qd_io,
                                                            const T* __restrict__ prm, const T* jst_all, JointSlots slots,
                                                            T* __restrict__ sumsq_io, T* jst_next) {
  __shared__ T xch[LO ? TILE_SCALARS : GEN_SCALARS];
  constexpr bool REMOTE = XK != XK_NONE;
  const DevCfg<T>& cfg = *cfgp;
  const int N = cfg.n_robots;
  const int lane = threadIdx.x;
  if constexpr (LO) stage_sphere_radii(cfg, xch, lane);  // visible after the first barrier
  const int spw = 64 / count;
  int ls = lane / count;
  const int l0 = lane - ls * count;
  int64_t scen = (int64_t)blockIdx.x * spw + ls;
  const bool active = ls < spw && scen < n_scen;
  if (!active) {  // idle lanes shadow the block's first row (no stores)
    ls = 0;
    scen = (int64_t)blockIdx.x * spw;
  }
  const int l = active ? l0 : 0;
  const int me = first + l;
  const int64_t rows = n_scen * count;
  const int64_t row = scen * count + l;
  PandaState<T> R;
  {
    const T* own = jst_all + ((int64_t)slots.s[me] * MRF_JOINT_STATE_SCALARS) * n_scen + scen;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      R.q[j] = q_io[j * rows + row];
      R.cq[j] = own[(int64_t)(3 * j + 0) * n_scen];
      R.sq[j] = own[(int64_t)(3 * j + 1) * n_scen];
      R.qd[j] = own[(int64_t)(3 * j + 2) * n_scen];
    }
  }
  PrmView<T> P{prm, rows, row, {T(0), T(0), T(0)}, false};
  T qdd[7], act[7];
  sharded_solve_row<LS, LO, REMOTE>(
      cfg, xch, lane, ls, l, count, cfg.mount[me], R, P, [&](const PandaKin<T>&) {}, [&]() {},
      [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
        if constexpr (REMOTE)
          remote_obstacles_joints<typename LS::Collision, LO>(
              cfg, xch, lane, first, count, N,
              [&](int jr, T (&v)[MRF_JOINT_STATE_SCALARS]) {
#pragma unroll
                for (int c = 0; c < MRF_JOINT_STATE_SCALARS; ++c)
                  v[c] = jst_all[((int64_t)slots.s[jr] * MRF_JOINT_STATE_SCALARS + c) * n_scen + scen];
              },
              E, acc);
      },
      qdd, act);
  if (active) {
    T ss = T(0);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      qd_io[j * rows + row] = act[j];  // FPJ:233
      ss += act[j] * act[j];
    }
    sumsq_io[row] += ss;
  }
  if constexpr (NEXT) {
    // system_step 'vel' of the following step (FPJ:77-80) as k_rollout_panda does it: q += dt * action, cos q / sin q
    // rotated by the increment where every lane's is small (a wave-wide vote), else recomputed
    T dq[7];
    bool small = true;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      dq[j] = cfg.dt * act[j];
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
    if (active) {
      T* dst = jst_next + ((int64_t)l * MRF_JOINT_STATE_SCALARS) * n_scen + scen;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        q_io[j * rows + row] = R.q[j];
        dst[(int64_t)(3 * j + 0) * n_scen] = R.cq[j];
        dst[(int64_t)(3 * j + 1) * n_scen] = R.sq[j];
        dst[(int64_t)(3 * j + 2) * n_scen] = act[j];
      }
    }
  }
}

}  // namespace mrf

using mrf_host::dispatch;
using mrf_host::dispatch_scalar;
using mrf_host::fail;
using mrf_host::is_link_origin_table;
using mrf_host::launch;

int mrf_host::step_action_joints_slots(mrf_handle* h, int64_t n_scen, int32_t robot_first, int32_t robot_count, void* q,
                                       void* qdot_io, const void* params, const void* jst_all, const int32_t* robot_slot,
                                       void* sumsq_io, void* jst_next_own, void* stream) {
  MRF_CHECK_READY(h);
  if (h->cfg.model != MRF_MODEL_PANDA7 || h->cfg.mode != MRF_MODE_VEL)
    return fail(h, MRF_E_CONFIG, "sharded rollout needs the panda7 model in mode 'vel'");
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || robot_first < 0 || robot_count < 1 || robot_first + robot_count > h->cfg.n_robots || !q || !qdot_io ||
      !params || !jst_all || !sumsq_io)
    return fail(h, MRF_E_ARG, "bad argument");
  mrf::JointSlots slots;
  for (int j = 0; j < MRF_MAX_ROBOTS; ++j) slots.s[j] = robot_slot ? robot_slot[j] : j;
  const int spw = 64 / robot_count;
  const dim3 block(64), grid((unsigned)((n_scen + spw - 1) / spw));
  const bool lo = is_link_origin_table(h->cfg), all = robot_count == h->cfg.n_robots;
  hipStream_t st = (hipStream_t)stream;
  return dispatch(h, [&](auto t, auto cl) {
    using T = decltype(t);
    using LS = decltype(cl);
    auto go = [&](auto kernel) {
      return launch(h, kernel, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, n_scen, (int)robot_first, (int)robot_count,
                    (T*)q, (T*)qdot_io, (const T*)params, (const T*)jst_all, slots, (T*)sumsq_io, (T*)jst_next_own);
    };
    auto pick = [&](auto lo_c, auto xk_c) {
      constexpr bool LO = decltype(lo_c)::value;
      constexpr int XK = decltype(xk_c)::value;
      return jst_next_own ? go(mrf::k_step_action_joints<T, LS, LO, XK, true>) : go(mrf::k_step_action_joints<T, LS, LO, XK, false>);
    };
    using std::integral_constant;
    if (all)
      return lo ? pick(integral_constant<bool, true>{}, integral_constant<int, mrf::XK_NONE>{})
                : pick(integral_constant<bool, false>{}, integral_constant<int, mrf::XK_NONE>{});
    return lo ? pick(integral_constant<bool, true>{}, integral_constant<int, mrf::XK_JOINTS>{})
              : pick(integral_constant<bool, false>{}, integral_constant<int, mrf::XK_JOINTS>{});
  });
}

extern "C" {

int32_t mrf_exchange_scalars(const mrf_handle* h) {
  if (!h) return 0;
  return h->cfg.exchange == MRF_EXCHANGE_JOINTS ? MRF_JOINT_STATE_SCALARS : 9 * mrf_exchange_spheres(h);
}

int mrf_step_predict_joints(mrf_handle* h, int64_t n_scen, int32_t robot_first, int32_t robot_count, void* q_io,
                            const void* qdot, void* jst_own, void* stream) {
  MRF_CHECK_READY(h);
  if (h->cfg.model != MRF_MODEL_PANDA7) return fail(h, MRF_E_CONFIG, "sharded rollout needs the panda7 model");
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || robot_first < 0 || robot_count < 1 || robot_first + robot_count > h->cfg.n_robots || !q_io || !qdot ||
      !jst_own)
    return fail(h, MRF_E_ARG, "bad argument");
  const int64_t rows = n_scen * robot_count;
  const dim3 block(256), grid((unsigned)((rows + 255) / 256));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_step_predict_joints<T>, grid, block, (hipStream_t)stream, (const mrf::DevCfg<T>*)h->dcfg, n_scen,
                  (int)robot_count, (T*)q_io, (const T*)qdot, (T*)jst_own);
  });
}

int mrf_step_action_joints(mrf_handle* h, int64_t n_scen, int32_t robot_first, int32_t robot_count, const void* q,
                           void* qdot_io, const void* params, const void* jst_all, void* sumsq_io, void* stream) {
  return mrf_host::step_action_joints_slots(h, n_scen, robot_first, robot_count, const_cast<void*>(q), qdot_io, params, jst_all,
                                            nullptr, sumsq_io, nullptr, stream);  // NEXT = false never writes q
}

int mrf_step_action_predict_joints(mrf_handle* h, int64_t n_scen, int32_t robot_first, int32_t robot_count, void* q_io,
                                   void* qdot_io, const void* params, const void* jst_all, void* sumsq_io, void* jst_next_own,
                                   void* stream) {
  if (h && !jst_next_own) return fail(h, MRF_E_ARG, "jst_next_own is NULL (mrf_step_action_joints is the step without the update)");
  return mrf_host::step_action_joints_slots(h, n_scen, robot_first, robot_count, q_io, qdot_io, params, jst_all, nullptr, sumsq_io,
                                            jst_next_own, stream);
}

}  // extern "C"
