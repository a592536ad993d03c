// mrf_control.hip -- device-resident control step: what the reference's driver does on the host
// between two simulator steps (examples/example_pandas_Jointspace.py:280-458) around the two hot-path calls.
//
//   k_control_prepare   hand FK + RF-CV goal estimate                        (EXJ:325-329, 346-348)
//   k_deadlock          deadlock detection / resolution, one thread/scenario (deadlock_prevention.py:50-118)
//   k_apply_action      clip + velocity integration + hard joint stops       (EXJ:452-453)
//   mrf_episode_run     n control steps back to back, optionally as one replayed HIP graph
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "mrf_device.hpp"
#include "mrf_host.hpp"

namespace mrf {

template <typename T>
__global__ __launch_bounds__(64) void k_control_prepare(const DevCfg<T>* __restrict__ cfgp, int64_t rows,
                                                         const T* __restrict__ q, const T* __restrict__ qd,
                                                         const T* prm_nom, T* prm_work, int apply_estimate,
                                                         T* __restrict__ x_ee) {
  const DevCfg<T>& cfg = *cfgp;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int li = (int)(r % cfg.n_robots);
  PandaState<T> R;
  load_state(rows, r, q, qd, R);
  PandaKin<T> K;
  panda_walk_own<T>(cfg.mount[li], R.cq, R.sq, R.qd, K);
  if (prm_work != prm_nom) {
#pragma unroll 1
    for (int c = 0; c < MRF_NPARAM; ++c) prm_work[c * rows + r] = prm_nom[c * rows + r];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    x_ee[c * rows + r] = K.p8[c];
    if (apply_estimate && ((cfg.goal_mask >> li) & 1))
      prm_work[(MRF_P_X_GOAL_0 + c) * rows + r] = K.p8[c] + cfg.goal_T * K.v8[c];  // same expression as k_rollout_panda
  }
}

template <typename T>
struct DeadlockCfg {
  T avg_vel_constant, dist_constant, w_follower, w_leader, goal_scale, ee_distance, follower_offset, min_goal_norm, z_floor;
  int time_wait, min_time_step, grasp_state, grasp_timeout;
};

// One thread per scenario; the statement order follows deadlock_checking (DP:50-118).  The reference object's
// write-only counters (deadlock_robots, deadlock_combinations, DP:69-70) are not kept.
template <typename T>
__global__ __launch_bounds__(64) void k_deadlock(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen, DeadlockCfg<T> D,
                                                  int time_step_arg, const T* __restrict__ x_ee,
                                                  const T* __restrict__ avg, const int32_t* __restrict__ sm,
                                                  T* __restrict__ prm, int32_t* __restrict__ st, T* __restrict__ dl_goal) {
#pragma clang fp contract(off)  // plain mul/add as numpy evaluates them
  const DevCfg<T>& cfg = *cfgp;
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_scen) return;
  const int N = cfg.n_robots;
  const int64_t rows = n_scen * N;
  const int64_t r0 = b * N;
  int leader = st[MRF_DL_LEADER * n_scen + b], follower = st[MRF_DL_FOLLOWER * n_scen + b];
  int dead0 = st[MRF_DL_DEAD0 * n_scen + b], dead1 = st[MRF_DL_DEAD1 * n_scen + b];
  int t_out = st[MRF_DL_TIME_DEADLOCK_OUT * n_scen + b];
  const int w = time_step_arg >= 0 ? time_step_arg : st[MRF_DL_TIME_STEP * n_scen + b];
  st[MRF_DL_TIME_STEP * n_scen + b] = w + 1;

  T avg_sum = T(0);  // vel_avg_tot = sum(vel_avg) / nr_robots  (EXJ:375)
  for (int i = 0; i < N; ++i) avg_sum += avg[r0 + i];
  avg_sum = avg_sum / T(N);
  // A non-finite rollout signal (a state the rollout could not predict, e.g. parked on a hard joint stop) makes the
  // reference's test `avg < constant` false, i.e. "no deadlock" -- kept, but counted so that the caller can see it.
  if (!(m_abs(avg_sum) <= T(1.7e308))) st[MRF_DL_NONFINITE * n_scen + b] += 1;

  T X[MRF_MAX_ROBOTS][3], dist_goal[MRF_MAX_ROBOTS];
  int state[MRF_MAX_ROBOTS];
  for (int i = 0; i < N; ++i) {
    T s2 = T(0);
    for (int c = 0; c < 3; ++c) {
      X[i][c] = x_ee[c * rows + r0 + i];
      const T d = X[i][c] - prm[(MRF_P_X_GOAL_0 + c) * rows + r0 + i];
      s2 += d * d;
    }
    dist_goal[i] = m_sqrt(s2);  // DP:57
    state[i] = sm ? sm[r0 + i] : 0;
  }
  bool deadlock = false;
  T best = T(100);  // DP:56,75: the closest pair in deadlock is the one resolved (first one on ties)
  for (int a = 0; a < N; ++a)
    for (int c2 = a + 1; c2 < N; ++c2) {  // itertools.combinations order (DP:30)
      const bool approaching = (state[a] == 0 || state[a] == 1) && (state[c2] == 0 || state[c2] == 1);  // DP:61
      T s2 = T(0);
      for (int c = 0; c < 3; ++c) {
        const T d = X[a][c] - X[c2][c];
        s2 += d * d;
      }
      const T d_ee = m_sqrt(s2);
      if (avg_sum < D.avg_vel_constant && dist_goal[a] + dist_goal[c2] > D.dist_constant && w > D.min_time_step &&
          approaching && d_ee < D.ee_distance) {  // DP:66
        deadlock = true;
        if (d_ee < best) {
          best = d_ee;
          dead0 = a;
          dead1 = c2;
        }
      }
    }
  T g0[3] = {dl_goal[0 * n_scen + b], dl_goal[1 * n_scen + b], dl_goal[2 * n_scen + b]};
  bool apply = false;
  if (deadlock && w > D.min_time_step) {  // DP:81
    if (dist_goal[dead0] > dist_goal[dead1]) {  // the robot closer to its goal leads (DP:83-88)
      leader = dead1;
      follower = dead0;
    } else {
      leader = dead0;
      follower = dead1;
    }
    T diff[3], dg[3], n2 = T(0);
    for (int c = 0; c < 3; ++c) {
      diff[c] = X[leader][c] - X[follower][c];
      dg[c] = diff[c] * D.goal_scale;
      n2 += dg[c] * dg[c];
    }
    const T nrm = m_sqrt(n2);
    if (nrm > D.min_goal_norm) {
      const T s = D.follower_offset / nrm;
      for (int c = 0; c < 3; ++c) g0[c] = X[follower][c] - s * dg[c];  // DP:95
    } else {
      for (int c = 0; c < 3; ++c) g0[c] = X[follower][c] - diff[c] * D.goal_scale;  // DP:97
    }
    if (g0[2] < T(0)) g0[2] = D.z_floor;  // DP:98-99
    apply = true;
    st[MRF_DL_TIME_IN_DEADLOCK * n_scen + b] += 1;
    t_out = 0;
  } else if (state[dead0] == D.grasp_state || state[dead1] == D.grasp_state) {  // DP:108-109
    t_out = D.grasp_timeout;
  } else if (t_out < D.time_wait) {  // DP:111-115: hold the resolution
    apply = true;
    t_out += 1;
  }
  if (apply) {
    prm[MRF_P_WEIGHT_GOAL_0 * rows + r0 + leader] = D.w_leader;
    prm[MRF_P_WEIGHT_GOAL_0 * rows + r0 + follower] = D.w_follower;
    for (int c = 0; c < 3; ++c) prm[(MRF_P_X_GOAL_0 + c) * rows + r0 + follower] = g0[c];
  }
  st[MRF_DL_LEADER * n_scen + b] = leader;
  st[MRF_DL_FOLLOWER * n_scen + b] = follower;
  st[MRF_DL_DEAD0 * n_scen + b] = dead0;
  st[MRF_DL_DEAD1 * n_scen + b] = dead1;
  st[MRF_DL_TIME_DEADLOCK_OUT * n_scen + b] = t_out;
  for (int c = 0; c < 3; ++c) dl_goal[c * n_scen + b] = g0[c];
}

template <typename T>
__global__ __launch_bounds__(256) void k_deadlock_init(int64_t n_scen, int32_t* __restrict__ st, T* __restrict__ dl_goal) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_scen) return;
  st[MRF_DL_LEADER * n_scen + b] = 0;    // DP:9-10
  st[MRF_DL_FOLLOWER * n_scen + b] = 1;
  st[MRF_DL_DEAD0 * n_scen + b] = 0;     // DP:33
  st[MRF_DL_DEAD1 * n_scen + b] = 1;
  st[MRF_DL_TIME_IN_DEADLOCK * n_scen + b] = 0;
  st[MRF_DL_TIME_DEADLOCK_OUT * n_scen + b] = 1000;  // EXJ:273
  st[MRF_DL_TIME_STEP * n_scen + b] = 0;
  st[MRF_DL_NONFINITE * n_scen + b] = 0;
  for (int c = 0; c < 3; ++c) dl_goal[c * n_scen + b] = T(0);
}

template <typename T>
struct VelLimits {
  T v[MRF_DOF_MAX];
};

template <typename T>
__global__ __launch_bounds__(256) void k_apply_action(const DevCfg<T>* __restrict__ cfgp, int64_t rows, T* __restrict__ q,
                                                       T* __restrict__ qd, T* __restrict__ act, VelLimits<T> L,
                                                       T stop_margin) {
  const DevCfg<T>& cfg = *cfgp;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    T a = act[j * rows + r];
    a = m_min(m_max(a, -L.v[j]), L.v[j]);  // np.clip(action, -limits_action, limits_action)  EXJ:452
    T qn = q[j * rows + r] + cfg.dt * a;
    if (stop_margin >= T(0)) qn = m_min(m_max(qn, cfg.limits[j][0] + stop_margin), cfg.limits[j][1] - stop_margin);
    q[j * rows + r] = qn;
    qd[j * rows + r] = a;
    act[j * rows + r] = a;
  }
}

}  // namespace mrf

namespace {
using mrf_host::check_hip;
using mrf_host::dispatch_scalar;
using mrf_host::fail;
using mrf_host::launch;

template <typename T>
mrf::DeadlockCfg<T> to_dev(const mrf_deadlock_config& c) {
  mrf::DeadlockCfg<T> d;
  d.avg_vel_constant = (T)c.avg_vel_constant; d.dist_constant = (T)c.dist_constant;
  d.w_follower = (T)c.goal_weight_follower; d.w_leader = (T)c.goal_weight_leader; d.goal_scale = (T)c.nr_goal_scale;
  d.ee_distance = (T)c.ee_distance; d.follower_offset = (T)c.follower_offset; d.min_goal_norm = (T)c.min_goal_norm;
  d.z_floor = (T)c.z_floor;
  d.time_wait = c.time_wait; d.min_time_step = c.min_time_step; d.grasp_state = c.grasp_state;
  d.grasp_timeout = c.grasp_timeout;
  return d;
}

int need_panda_vel(mrf_handle* h, const char* what) {
  if (h->cfg.model != MRF_MODEL_PANDA7) return fail(h, MRF_E_CONFIG, std::string(what) + " is defined for the panda7 model only");
  return MRF_OK;
}
}  // namespace

extern "C" {

int64_t mrf_deadlock_config_sizeof(void) { return (int64_t)sizeof(mrf_deadlock_config); }

void mrf_default_deadlock_config(mrf_deadlock_config* c, int32_t point_mass) {
  std::memset(c, 0, sizeof(*c));
  c->avg_vel_constant = point_mass ? 0.03 : 0.16;
  c->dist_constant = point_mass ? 1.0 : 0.0;
  c->goal_weight_follower = point_mass ? 10.0 : 2.0;
  c->goal_weight_leader = point_mass ? 1.0 : 3.0;
  c->nr_goal_scale = point_mass ? 100.0 : 2.0;
  c->time_wait = point_mass ? 50 : 300;
  c->ee_distance = 0.35;
  c->follower_offset = 0.3;
  c->min_goal_norm = 0.05;
  c->z_floor = 0.1;
  c->min_time_step = 10;
  c->grasp_state = 2;
  c->grasp_timeout = 400;
}

int mrf_deadlock_init(mrf_handle* h, int64_t n_scen, int32_t* dl_state, void* dl_goal, void* stream) {
  MRF_CHECK_READY(h);
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || !dl_state || !dl_goal) return fail(h, MRF_E_ARG, "null/negative argument");
  dim3 block(256), grid((unsigned)((n_scen + 255) / 256));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_deadlock_init<T>, grid, block, (hipStream_t)stream, n_scen, dl_state, (T*)dl_goal);
  });
}

int mrf_control_prepare(mrf_handle* h, int64_t n_scen, const void* q, const void* qdot, const void* params_nominal,
                        void* params_work, int32_t apply_estimate, void* x_ee_out, void* stream) {
  MRF_CHECK_READY(h);
  if (int rc = need_panda_vel(h, "control_prepare")) return rc;
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || !q || !qdot || !params_nominal || !params_work || !x_ee_out) return fail(h, MRF_E_ARG, "null/negative argument");
  const int64_t rows = n_scen * h->cfg.n_robots;
  dim3 block(64), grid((unsigned)((rows + 63) / 64));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_control_prepare<T>, grid, block, (hipStream_t)stream, (const mrf::DevCfg<T>*)h->dcfg, rows,
                  (const T*)q, (const T*)qdot, (const T*)params_nominal, (T*)params_work, (int)apply_estimate, (T*)x_ee_out);
  });
}

int mrf_deadlock_step(mrf_handle* h, int64_t n_scen, const mrf_deadlock_config* dl, int32_t time_step, const void* x_ee,
                      const void* avg_vel, const int32_t* sm_state, void* params_work, int32_t* dl_state, void* dl_goal,
                      void* stream) {
  MRF_CHECK_READY(h);
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || !dl || !x_ee || !avg_vel || !params_work || !dl_state || !dl_goal)
    return fail(h, MRF_E_ARG, "null/negative argument");
  if (h->cfg.n_robots < 2) return fail(h, MRF_E_CONFIG, "deadlock logic needs at least two robots");
  dim3 block(64), grid((unsigned)((n_scen + 63) / 64));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_deadlock<T>, grid, block, (hipStream_t)stream, (const mrf::DevCfg<T>*)h->dcfg, n_scen,
                  to_dev<T>(*dl), (int)time_step, (const T*)x_ee, (const T*)avg_vel, sm_state, (T*)params_work, dl_state,
                  (T*)dl_goal);
  });
}

int mrf_apply_action(mrf_handle* h, int64_t rows, void* q_io, void* qdot_io, void* action_io, const double* vel_limit,
                     double stop_margin, void* stream) {
  MRF_CHECK_READY(h);
  if (int rc = need_panda_vel(h, "apply_action")) return rc;
  if (h->cfg.mode != MRF_MODE_VEL) return fail(h, MRF_E_CONFIG, "apply_action integrates velocity commands (mode 'vel')");
  if (rows == 0) return MRF_OK;
  if (rows < 0 || !q_io || !qdot_io || !action_io || !vel_limit) return fail(h, MRF_E_ARG, "null/negative argument");
  dim3 block(256), grid((unsigned)((rows + 255) / 256));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    mrf::VelLimits<T> L;
    for (int j = 0; j < MRF_DOF_MAX; ++j) L.v[j] = (T)vel_limit[j];
    return launch(h, mrf::k_apply_action<T>, grid, block, (hipStream_t)stream, (const mrf::DevCfg<T>*)h->dcfg, rows,
                  (T*)q_io, (T*)qdot_io, (T*)action_io, L, (T)stop_margin);
  });
}

static int control_step(mrf_handle* hr, mrf_handle* ha, int64_t n_scen, const mrf_deadlock_config* dl, int apply_estimate,
                        const double* vel_limit, double stop_margin, void* q, void* qd, const void* prm_nom,
                        void* prm_work, const int32_t* sm, int32_t* dl_state, void* dl_goal, void* x_ee, void* avg,
                        void* act, void* st) {
  const int64_t rows = n_scen * ha->cfg.n_robots;
  int rc;
  if (hr) {
    if ((rc = mrf_control_prepare(hr, n_scen, q, qd, prm_nom, prm_work, apply_estimate, x_ee, st))) return rc;
    if ((rc = mrf_rollout(hr, n_scen, q, qd, prm_work, avg, nullptr, nullptr, st))) return rc;
    if (dl && (rc = mrf_deadlock_step(hr, n_scen, dl, -1, x_ee, avg, sm, prm_work, dl_state, dl_goal, st))) return rc;
  }
  if ((rc = mrf_compute_action_coupled(ha, n_scen, q, qd, hr ? prm_work : prm_nom, 0, nullptr, act, st))) return rc;
  return mrf_apply_action(ha, rows, q, qd, act, vel_limit, stop_margin, st);
}

int mrf_episode_run(mrf_handle* hr, mrf_handle* ha, int64_t n_scen, int32_t n_steps, const mrf_deadlock_config* dl,
                    int32_t apply_estimate, const double* vel_limit, double stop_margin, void* q_io, void* qdot_io,
                    const void* params_nominal, void* params_work, const int32_t* sm_state, int32_t* dl_state,
                    void* dl_goal, void* x_ee_work, void* avg_work, void* action_out, int32_t use_graph, void* stream) {
  MRF_CHECK_READY(ha);
  if (hr) {
    MRF_CHECK_READY(hr);
    if (hr->cfg.n_robots != ha->cfg.n_robots || hr->cfg.scalar != ha->cfg.scalar || hr->device != ha->device)
      return fail(ha, MRF_E_CONFIG, "rollout and action handles must agree in n_robots, scalar type and device");
  }
  if (n_scen == 0 || n_steps == 0) return MRF_OK;
  if (n_scen < 0 || n_steps < 0 || !vel_limit || !q_io || !qdot_io || !params_nominal || !action_out)
    return fail(ha, MRF_E_ARG, "null/negative argument");
  if (hr && (!params_work || !x_ee_work || !avg_work)) return fail(ha, MRF_E_ARG, "work buffers missing");
  if (hr && dl && (!dl_state || !dl_goal)) return fail(ha, MRF_E_ARG, "deadlock state missing");
  auto one = [&](void* st) {
    int rc = control_step(hr, ha, n_scen, dl, apply_estimate, vel_limit, stop_margin, q_io, qdot_io, params_nominal,
                          params_work, sm_state, dl_state, dl_goal, x_ee_work, avg_work, action_out, st);
    if (rc && hr && ha->err.empty()) ha->err = hr->err;
    return rc;
  };
  if (!use_graph) {
    for (int k = 0; k < n_steps; ++k)
      if (int rc = one(stream)) return rc;
    return MRF_OK;
  }
  // One control step captured once and replayed: the launch arguments do not change between steps (the step
  // counter lives in dl_state), so the graph is keyed by the argument tuple and cached in the action handle.
  hipStream_t st = (hipStream_t)stream;
  if (!st) {  // the legacy default stream cannot be captured: use an own blocking stream (implicitly ordered with it)
    if (!ha->own_stream && hipStreamCreateWithFlags((hipStream_t*)&ha->own_stream, hipStreamDefault) != hipSuccess)
      return fail(ha, MRF_E_LAUNCH, "hipStreamCreate failed");
    st = (hipStream_t)ha->own_stream;
  }
  mrf_deadlock_config dlc;
  std::memset(&dlc, 0, sizeof(dlc));
  if (dl) dlc = *dl;
  const void* key_ptrs[] = {hr, (void*)(intptr_t)n_scen, (void*)(intptr_t)apply_estimate, q_io, qdot_io, params_nominal,
                            params_work, sm_state, dl_state, dl_goal, x_ee_work, avg_work, action_out, (void*)st,
                            (void*)(intptr_t)(dl != nullptr)};
  std::string key((const char*)key_ptrs, sizeof(key_ptrs));
  // what the captured launches bake in besides the arguments: the handles' constants (a destroyed handle's address
  // can be handed to a new one: the creation serial tells them apart), and the kernel variant chosen from the config
  const uint64_t ident[] = {ha->serial, hr ? hr->serial : 0, (uint64_t)(uintptr_t)ha->dcfg,
                            (uint64_t)(uintptr_t)(hr ? hr->dcfg : nullptr)};
  key.append((const char*)ident, sizeof(ident));
  key.append((const char*)&dlc, sizeof(dlc));
  key.append((const char*)vel_limit, sizeof(double) * MRF_DOF_MAX);
  key.append((const char*)&stop_margin, sizeof(stop_margin));
  if (!ha->graph_exec || key != ha->graph_key) {
    if (ha->graph_exec) {
      (void)hipStreamSynchronize(st);
      (void)hipGraphExecDestroy((hipGraphExec_t)ha->graph_exec);
      ha->graph_exec = nullptr;
    }
    hipGraph_t g = nullptr;
    if (int rc = check_hip(ha, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture")) return rc;
    int rc = one(st);
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc) {
      if (g) (void)hipGraphDestroy(g);
      return rc;
    }
    if (int rc2 = check_hip(ha, e, "hipStreamEndCapture")) return rc2;
    hipGraphExec_t ge = nullptr;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (int rc2 = check_hip(ha, e, "hipGraphInstantiate")) return rc2;
    ha->graph_exec = ge;
    ha->graph_key = key;
  }
  for (int k = 0; k < n_steps; ++k)
    if (int rc = check_hip(ha, hipGraphLaunch((hipGraphExec_t)ha->graph_exec, st), "hipGraphLaunch")) return rc;
  return MRF_OK;
}

}  // extern "C"
