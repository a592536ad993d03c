// mrf_control.hip -- device-resident control step: what the reference's driver does on the host
// between two simulator steps (examples/example_pandas_Jointspace.py:280-458) around the two hot-path calls.
//
//   k_control_prepare   hand FK + RF-CV goal estimate                        (EXJ:325-329, 346-348)
//   k_deadlock          deadlock detection / resolution, one thread/scenario (deadlock_prevention.py:50-118)
//   k_apply_action      clip + velocity integration + hard joint stops       (EXJ:452-453)
//   mrf_episode_run     n control steps back to back, optionally as one replayed HIP graph; per step: k_step_head
//                       (recorder stamp, hand FK + goal estimate, state machine) -> rollout -> k_deadlock -> the planners
//                       -> k_step_tail (action selection, clip + integration, the step's record)
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "mrf_device.hpp"
#include "mrf_host.hpp"

namespace mrf {

// hand FK + RF-CV goal estimate of one row (mrf_control_prepare; also the head of a fused control step)
template <typename T>
__device__ __forceinline__ void prepare_row(const DevCfg<T>& cfg, int64_t rows, int64_t r, const T* __restrict__ q,
                                            const T* __restrict__ qd, const T* prm_nom, T* prm_work, int apply_estimate,
                                            T* __restrict__ x_ee) {
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
__global__ __launch_bounds__(64) void k_control_prepare(const DevCfg<T>* __restrict__ cfgp, int64_t rows,
                                                         const T* __restrict__ q, const T* __restrict__ qd,
                                                         const T* prm_nom, T* prm_work, int apply_estimate,
                                                         T* __restrict__ x_ee) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  prepare_row<T>(*cfgp, rows, r, q, qd, prm_nom, prm_work, apply_estimate, x_ee);
}

// Obstacle assembly of the Cartesian rollouts on the device (compute_x_obsts_dyn_0, utils_fabrics_kinematics.py:3-33;
// EXC:330-352): the dynamic obstacles of robot i are the configured spheres of all OTHER robots of its scenario, at
// their current positions, with their current velocities J qdot (zero for static fabrics), in increasing robot order.
// One lane walks one (scenario, robot) chain and scatters each of its spheres into the obstacle arrays
// [M][3][rows] / [M][rows] of the N - 1 other rows of the scenario (M = S (N - 1)).
template <typename T>
__global__ __launch_bounds__(64) void k_publish_obstacles(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen,
                                                           const T* __restrict__ q, const T* __restrict__ qd,
                                                           T* __restrict__ ox, T* __restrict__ ov, T* __restrict__ orad) {
  __shared__ T xch[21 * 64];
  const DevCfg<T>& cfg = *cfgp;
  const int lane = threadIdx.x;
  const int N = cfg.n_robots, S = cfg.n_spheres;
  const int64_t rows = n_scen * N;
  int64_t r = (int64_t)blockIdx.x * 64 + lane;
  const bool active = r < rows;
  if (!active) r = rows - 1;
  const int64_t scen = r / N;
  const int j = (int)(r - scen * N);
  T qk[7];  // loads first, then the sincos calls (branches the compiler keeps loads behind): one round trip, not seven
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    qk[k] = q[k * rows + r];
    xch[(3 * k + 2) * 64 + lane] = qd[k * rows + r];
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    T s, c;
    m_sincos(qk[k], &s, &c);
    xch[(3 * k + 0) * 64 + lane] = c;
    xch[(3 * k + 1) * 64 + lane] = s;
  }
  __syncthreads();
  const bool dyn = cfg.dynamic != 0;
  panda_walk_spheres<false, T>(
      cfg, cfg.mount[j],
      [&](int k, T& c, T& s, T& qdk) {
        c = xch[(3 * k + 0) * 64 + lane];
        s = xch[(3 * k + 1) * 64 + lane];
        qdk = xch[(3 * k + 2) * 64 + lane];
      },
      [&](int s, const T* x, const T* v, const T*) {
        if (!active) return;
        const T rad = cfg.sphere_r[s];
#pragma unroll 1
        for (int i = 0; i < N; ++i) {
          if (i == j) continue;
          const int m = (j < i ? j : j - 1) * S + s;  // position of robot j among the others of robot i
          const int64_t row_i = scen * N + i;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            ox[(int64_t)(m * 3 + c) * rows + row_i] = x[c];
            ov[(int64_t)(m * 3 + c) * rows + row_i] = dyn ? v[c] : T(0);
          }
          orad[(int64_t)m * rows + row_i] = rad;
        }
      });
}

// Episode recorder (mrf_episode_set_recorder): what a host loop would otherwise read back after every control step --
// joint positions, state-machine states, "who is done since when", and how long the step took -- is written by the step
// itself (k_step_head / k_step_tail below), so that n control steps can be queued back to back without the host in
// between.  k_step_begin is the head's recorder part alone, for steps that have no head (plain MRDF without pick-and-place).
__global__ void k_step_begin(int32_t* __restrict__ counter, int capacity, int64_t* __restrict__ t_begin) {
  const int i = *counter;
  *counter = i + 1;
  if (i < capacity && t_begin) t_begin[i] = (int64_t)wall_clock64();
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

// ---------------------------------------------------------------------------- pick-and-place state machine
template <typename T>
struct SmCfg {
  T reach_home, reach_pregrasp, reach_block, reach_lift, reach_drop, pregrasp_height, lift_height, grip_steps, open_tol,
      dropped_below_z, weight_high, weight_low, open0, open1, v_close, v_open, dt;
  int nr_blocks, model;
};

// One thread per row; statement order follows get_state_machine_panda (SM:133-214) and get_gripper_action_panda (:66-86).
template <typename T>
__device__ __forceinline__ void state_machine_row(const DevCfg<T>& cfg, int64_t rows, int64_t r, const SmCfg<T>& C,
                                                  const T* __restrict__ x_ee, const T* __restrict__ start,
                                                  const T* __restrict__ blocks, int n_block_arrays,
                                                  T* __restrict__ q_grip, int32_t* __restrict__ st, T* __restrict__ sg,
                                                  T* __restrict__ prm, int skip_mask, T* __restrict__ grip_act) {
#pragma clang fp contract(off)  // plain mul/add as numpy evaluates them
  int state = st[MRF_SM_STATE * rows + r], picked = st[MRF_SM_PICKED * rows + r], failed = st[MRF_SM_FAILED * rows + r];
  int t_grip = st[MRF_SM_T_GRIP * rows + r], closed = st[MRF_SM_GRIPPER * rows + r], stop = st[MRF_SM_STOP * rows + r];
  T goal[3], above_blk[3], x[3], s0[3], g[2];
  for (int c = 0; c < 3; ++c) {
    goal[c] = sg[(MRF_SM_GOAL + c) * rows + r];
    above_blk[c] = sg[(MRF_SM_GOAL_ABOVE + c) * rows + r];
    x[c] = x_ee[c * rows + r];
    s0[c] = start[c * rows + r];
  }
  T weight = sg[MRF_SM_WEIGHT * rows + r];
  g[0] = q_grip[r];
  g[1] = q_grip[rows + r];
  // goal_block of this step: an observation (model 0) or the minimal block model (model 1)
  int bi = 0;
  if (C.model == 1) bi = picked < C.nr_blocks - 1 ? picked : C.nr_blocks - 1;
  if (bi >= n_block_arrays) bi = n_block_arrays - 1;
  T block[3];
  for (int c = 0; c < 3; ++c) block[c] = blocks[((int64_t)bi * 3 + c) * rows + r];
  if (C.model == 1 && closed && (state == 12 || state == 4))
    for (int c = 0; c < 3; ++c) block[c] = x[c];  // carried: the grasp target travels with the hand
  T pre[3] = {block[0], block[1], block[2] + C.pregrasp_height};
  auto norm3 = [](T a, T b, T c) { return m_sqrt(a * a + b * b + c * c); };
  auto norm2 = [](T a, T b) { return m_sqrt(a * a + b * b); };
  const T d_home = norm3(x[0] - s0[0], x[1] - s0[1], x[2] - s0[2]);
  const T d_pre = norm2(x[0] - pre[0], x[1] - pre[1]);  // get_distance_ee_goal: horizontal only (SM:51-54)
  const T d_block = norm3(x[0] - block[0], x[1] - block[1], x[2] - block[2]);
  const T d_open = norm2(g[0] - C.open0, g[1] - C.open1);

  if (picked > C.nr_blocks - 1) {
    state = 10;
  } else if (block[2] < C.dropped_below_z) {  // "the block has been dropped!"
    picked += 1;
    failed += 1;
    state = 0;
  }
  if (state == 0) {
    for (int c = 0; c < 3; ++c) goal[c] = s0[c];
    closed = 0;
    if (d_home < C.reach_home) state = 1;
  } else if (state == 1) {
    for (int c = 0; c < 3; ++c) goal[c] = pre[c];
    if (d_pre < C.reach_pregrasp) state = 2;
  } else if (state == 2) {
    for (int c = 0; c < 3; ++c) goal[c] = block[c];
    if (d_block < C.reach_block) {
      closed = 1;
      weight = C.weight_low;
      state = 3;
    }
  } else if (state == 3) {
    for (int c = 0; c < 3; ++c) {
      goal[c] = block[c];
      above_blk[c] = block[c];
    }
    above_blk[2] += C.lift_height;
    t_grip += 1;
    if ((T)t_grip > C.grip_steps) {
      t_grip = 0;
      for (int c = 0; c < 3; ++c) goal[c] = s0[c];
      weight = C.weight_high;
      state = 12;
    }
  } else if (state == 12) {
    for (int c = 0; c < 3; ++c) goal[c] = above_blk[c];
    if (norm2(x[0] - goal[0], x[1] - goal[1]) < C.reach_lift) state = 4;
  } else if (state == 4) {
    for (int c = 0; c < 3; ++c) goal[c] = s0[c];
    if (d_home < C.reach_drop) {
      state = 5;
      closed = 0;
    }
  } else if (state == 5) {
    if (d_open < C.open_tol) {
      state = 0;
      picked += 1;
      for (int c = 0; c < 3; ++c) goal[c] = s0[c];
    }
  } else if (state == 10) {
    stop = 1;
  }
  // get_gripper_action_panda (SM:66-86), evaluated on the updated gripper status as the driver does (EXJ:448)
  T act[2] = {T(0), T(0)};
  if (closed) {
    act[0] = act[1] = C.v_close;
  } else if (d_open > C.open_tol) {
    act[0] = g[0] > C.open0 ? -C.v_open : C.v_open;
    act[1] = g[1] > C.open1 ? -C.v_open : C.v_open;
  }
  if (grip_act) {
    grip_act[r] = act[0];
    grip_act[rows + r] = act[1];
  }
  if (C.model == 1) {  // finger joints follow their velocity command between the mechanical stops
    q_grip[r] = m_min(m_max(g[0] + C.dt * act[0], T(0)), C.open0);
    q_grip[rows + r] = m_min(m_max(g[1] + C.dt * act[1], T(0)), C.open1);
  }
  st[MRF_SM_STATE * rows + r] = state;
  st[MRF_SM_PICKED * rows + r] = picked;
  st[MRF_SM_FAILED * rows + r] = failed;
  st[MRF_SM_T_GRIP * rows + r] = t_grip;
  st[MRF_SM_GRIPPER * rows + r] = closed;
  st[MRF_SM_STOP * rows + r] = stop;
  for (int c = 0; c < 3; ++c) {
    sg[(MRF_SM_GOAL + c) * rows + r] = goal[c];
    sg[(MRF_SM_GOAL_ABOVE + c) * rows + r] = above_blk[c];
  }
  sg[MRF_SM_WEIGHT * rows + r] = weight;
  const int robot = (int)(r % cfg.n_robots);
  if (prm) {  // EXJ:313-316 -> :423-424.  The RF-CV estimate replaces only the GOAL of a masked robot (EXJ:346-348); its
              // weight still comes from the state machine (get_weight_goal0 -> weight_goals0 of the rollouts, :363)
    if (!((skip_mask >> robot) & 1))
      for (int c = 0; c < 3; ++c) prm[(MRF_P_X_GOAL_0 + c) * rows + r] = goal[c];
    prm[MRF_P_WEIGHT_GOAL_0 * rows + r] = weight;
  }
}

template <typename T>
__global__ __launch_bounds__(64) void k_state_machine(const DevCfg<T>* __restrict__ cfgp, int64_t rows, SmCfg<T> C,
                                                       const T* __restrict__ x_ee, const T* __restrict__ start,
                                                       const T* __restrict__ blocks, int n_block_arrays,
                                                       T* __restrict__ q_grip, int32_t* __restrict__ st,
                                                       T* __restrict__ sg, T* __restrict__ prm, int skip_mask,
                                                       T* __restrict__ grip_act) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  state_machine_row<T>(*cfgp, rows, r, C, x_ee, start, blocks, n_block_arrays, q_grip, st, sg, prm, skip_mask, grip_act);
}

template <typename T>
__global__ __launch_bounds__(256) void k_state_machine_init(int64_t rows, const T* __restrict__ start,
                                                             int32_t* __restrict__ st, T* __restrict__ sg) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  st[MRF_SM_STATE * rows + r] = 1;  // SM:14
  for (int k = 1; k < MRF_SM_NSTATE; ++k) st[k * rows + r] = 0;
  for (int c = 0; c < 3; ++c) {
    sg[(MRF_SM_GOAL + c) * rows + r] = start[c * rows + r];
    sg[(MRF_SM_GOAL_ABOVE + c) * rows + r] = T(0);
  }
  sg[MRF_SM_WEIGHT * rows + r] = T(2);  // SM:11
}

template <typename T>
struct VelLimits {
  T v[MRF_DOF_MAX];
};

template <typename T>
__device__ __forceinline__ void apply_row(const DevCfg<T>& cfg, int64_t rows, int64_t r, T* __restrict__ q,
                                          T* __restrict__ qd, T* __restrict__ act, const VelLimits<T>& L, T stop_margin) {
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

template <typename T>
__global__ __launch_bounds__(256) void k_apply_action(const DevCfg<T>* __restrict__ cfgp, int64_t rows, T* __restrict__ q,
                                                       T* __restrict__ qd, T* __restrict__ act, VelLimits<T> L,
                                                       T stop_margin) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  apply_row<T>(*cfgp, rows, r, q, qd, act, L, stop_margin);
}

// ---------------------------------------------------------------------------- fused head and tail of a control step
// mrf_episode_run's control step is a chain of small kernels around the rollout and the two planners; for a single cell
// (one scenario) every launch costs more than its arithmetic, so the per-row stages before the rollout -- recorder begin
// stamp, hand FK + goal estimate, pick-and-place state machine -- are ONE kernel, and the per-row stages after the
// planners -- action selection by state, clip + integration, the step's record -- are another.  Same device functions
// as the stand-alone entry points (mrf_control_prepare, mrf_state_machine_step, mrf_apply_action).
struct RecView {
  int32_t* counter;   // NULL: no recorder attached
  int capacity, done_state;
  int64_t* t_begin;
  int64_t* t_end;
  void* q_hist;
  int32_t* sm_hist;
  int32_t* done_at;
};

template <typename T>
__global__ __launch_bounds__(64) void k_step_head(const DevCfg<T>* __restrict__ cfgp, int64_t rows, const T* __restrict__ q,
                                                   const T* __restrict__ qd, const T* prm_nom, T* prm_work,
                                                   int apply_estimate, T* __restrict__ x_ee, int with_sm, SmCfg<T> C,
                                                   const T* __restrict__ start, const T* __restrict__ blocks,
                                                   int n_block_arrays, T* __restrict__ q_grip, int32_t* __restrict__ st,
                                                   T* __restrict__ sg, int skip_mask, T* __restrict__ grip_act, RecView rec) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (rec.counter && r == 0) {  // the step takes its record index here; the tail of the same step reads counter - 1
    const int i = *rec.counter;
    *rec.counter = i + 1;
    if (i < rec.capacity && rec.t_begin) rec.t_begin[i] = (int64_t)wall_clock64();
  }
  if (r >= rows) return;
  prepare_row<T>(*cfgp, rows, r, q, qd, prm_nom, prm_work, apply_estimate, x_ee);
  if (with_sm)
    state_machine_row<T>(*cfgp, rows, r, C, x_ee, start, blocks, n_block_arrays, q_grip, st, sg, prm_work, skip_mask, grip_act);
}

template <typename T>
__global__ __launch_bounds__(256) void k_step_tail(const DevCfg<T>* __restrict__ cfgp, int64_t rows, T* __restrict__ q,
                                                    T* __restrict__ qd, T* __restrict__ act, VelLimits<T> L, T stop_margin,
                                                    const int32_t* __restrict__ select_state, const T* __restrict__ act_grasp,
                                                    const int32_t* __restrict__ sm_state, RecView rec) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = rec.counter ? *rec.counter - 1 : 0;
  if (r < rows) {
    if (select_state) {  // EXJ:414-445: gripping / releasing rows stand still, descending rows take the grasp planner's action
      const int s = select_state[r];
      if (s == 3 || s == 5) {
        for (int j = 0; j < 7; ++j) act[j * rows + r] = T(0);
      } else if (s == 2 && act_grasp) {
        for (int j = 0; j < 7; ++j) act[j * rows + r] = act_grasp[j * rows + r];
      }
    }
    apply_row<T>(*cfgp, rows, r, q, qd, act, L, stop_margin);
    if (rec.counter && i < rec.capacity) {
      T* q_hist = (T*)rec.q_hist;
      if (q_hist)
        for (int j = 0; j < 7; ++j) q_hist[((int64_t)i * 7 + j) * rows + r] = q[j * rows + r];
      const int s = sm_state ? sm_state[r] : 0;
      if (rec.sm_hist) rec.sm_hist[(int64_t)i * rows + r] = s;
      if (rec.done_at && sm_state && s == rec.done_state && rec.done_at[r] < 0) rec.done_at[r] = i;
    }
  }
  if (rec.counter && r == 0 && i < rec.capacity && rec.t_end) rec.t_end[i] = (int64_t)wall_clock64();
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

template <typename T>
mrf::SmCfg<T> make_smcfg(const mrf_state_machine_config* sm, double dt) {
  mrf::SmCfg<T> C;
  C.reach_home = (T)sm->reach_home; C.reach_pregrasp = (T)sm->reach_pregrasp; C.reach_block = (T)sm->reach_block;
  C.reach_lift = (T)sm->reach_lift; C.reach_drop = (T)sm->reach_drop; C.pregrasp_height = (T)sm->pregrasp_height;
  C.lift_height = (T)sm->lift_height; C.grip_steps = (T)sm->grip_steps; C.open_tol = (T)sm->open_tol;
  C.dropped_below_z = (T)sm->dropped_below_z; C.weight_high = (T)sm->weight_high; C.weight_low = (T)sm->weight_low;
  C.open0 = (T)sm->gripper_open[0]; C.open1 = (T)sm->gripper_open[1]; C.v_close = (T)sm->v_close; C.v_open = (T)sm->v_open;
  C.dt = (T)dt; C.nr_blocks = sm->nr_blocks; C.model = sm->model;
  return C;
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

int64_t mrf_state_machine_config_sizeof(void) { return (int64_t)sizeof(mrf_state_machine_config); }

void mrf_default_state_machine_config(mrf_state_machine_config* c, int32_t nr_blocks) {
  std::memset(c, 0, sizeof(*c));
  c->reach_home = 0.05; c->reach_pregrasp = 0.013; c->reach_block = 0.013; c->reach_lift = 0.04; c->reach_drop = 0.15;
  c->pregrasp_height = 0.1; c->lift_height = 0.15;
  c->grip_steps = 0.3 / 0.01;  // SM:171, evaluated in double as Python does (29.999999999999996)
  c->open_tol = 0.005; c->dropped_below_z = 0.6;
  c->weight_high = 2.0; c->weight_low = 0.0;
  c->gripper_open[0] = c->gripper_open[1] = 0.04;
  c->v_close = -0.05; c->v_open = 0.4;
  c->nr_blocks = nr_blocks;
  c->model = 0;
}

int mrf_state_machine_init(mrf_handle* h, int64_t rows, const void* start_goal, int32_t* sm_state, void* sm_goal, void* stream) {
  MRF_CHECK_READY(h);
  if (rows == 0) return MRF_OK;
  if (rows < 0 || !start_goal || !sm_state || !sm_goal) return fail(h, MRF_E_ARG, "null/negative argument");
  dim3 block(256), grid((unsigned)((rows + 255) / 256));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_state_machine_init<T>, grid, block, (hipStream_t)stream, rows, (const T*)start_goal, sm_state, (T*)sm_goal);
  });
}

int mrf_state_machine_step(mrf_handle* h, int64_t rows, const mrf_state_machine_config* sm, const void* x_ee,
                           const void* start_goal, const void* blocks, int32_t n_block_arrays, void* q_gripper_io,
                           int32_t* sm_state, void* sm_goal, void* params_work, int32_t skip_robot_mask,
                           void* gripper_action_out, void* stream) {
  MRF_CHECK_READY(h);
  if (rows == 0) return MRF_OK;
  if (rows < 0 || !sm || !x_ee || !start_goal || !blocks || n_block_arrays < 1 || !q_gripper_io || !sm_state || !sm_goal)
    return fail(h, MRF_E_ARG, "null/negative argument");
  if (sm->nr_blocks < 1 || (sm->model != 0 && sm->model != 1)) return fail(h, MRF_E_CONFIG, "nr_blocks >= 1 and model in {0,1}");
  dim3 block(64), grid((unsigned)((rows + 63) / 64));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    const mrf::SmCfg<T> C = make_smcfg<T>(sm, h->cfg.dt);
    return launch(h, mrf::k_state_machine<T>, grid, block, (hipStream_t)stream, (const mrf::DevCfg<T>*)h->dcfg, rows, C,
                  (const T*)x_ee, (const T*)start_goal, (const T*)blocks, (int)n_block_arrays, (T*)q_gripper_io, sm_state,
                  (T*)sm_goal, (T*)params_work, (int)skip_robot_mask, (T*)gripper_action_out);
  });
}

int mrf_episode_set_pick_place(mrf_handle* h, const mrf_state_machine_config* sm, const void* start_goal,
                               const void* blocks, int32_t n_block_arrays, void* q_gripper_io, int32_t* sm_state,
                               void* sm_goal, void* gripper_action_out, mrf_handle* h_grasp, void* action_grasp_work) {
  MRF_CHECK_READY(h);
  if (!sm) {
    h->pp = mrf_handle::PickPlace();
    return MRF_OK;
  }
  if (!start_goal || !blocks || n_block_arrays < 1 || !q_gripper_io || !sm_state || !sm_goal)
    return fail(h, MRF_E_ARG, "null argument");
  if (sm->nr_blocks < 1 || (sm->model != 0 && sm->model != 1)) return fail(h, MRF_E_CONFIG, "nr_blocks >= 1 and model in {0,1}");
  h->pp.on = true;
  h->pp.sm = *sm;
  h->pp.start_goal = start_goal; h->pp.blocks = blocks; h->pp.n_block_arrays = n_block_arrays;
  if (h_grasp) {
    if (!h_grasp->dcfg || !action_grasp_work) return fail(h, MRF_E_ARG, "grasp handle without device state / work buffer");
    if (h_grasp->cfg.n_robots != h->cfg.n_robots || h_grasp->cfg.scalar != h->cfg.scalar || h_grasp->device != h->device)
      return fail(h, MRF_E_CONFIG, "grasp and action handles must agree in n_robots, scalar type and device");
  }
  h->pp.q_gripper = q_gripper_io; h->pp.sm_state = sm_state; h->pp.sm_goal = sm_goal; h->pp.gripper_action = gripper_action_out;
  h->pp.h_grasp = h_grasp; h->pp.grasp_serial = h_grasp ? h_grasp->serial : 0; h->pp.action_grasp = h_grasp ? action_grasp_work : nullptr;
  return MRF_OK;
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


}  // extern "C"

namespace mrf_host {
void cart_work_release(mrf_handle* h) {
  if (h->cart_work) (void)hipFree(h->cart_work);
  h->cart_work = nullptr;
  h->cart_work_bytes = 0;
}
}  // namespace mrf_host

namespace {
size_t cart_work_need(const mrf_handle* h, int64_t n_scen) {
  const size_t sb = h->cfg.scalar == MRF_F64 ? 8 : 4;
  const size_t rows = (size_t)n_scen * h->cfg.n_robots, M = (size_t)h->cfg.n_spheres * (h->cfg.n_robots - 1);
  return sb * 7 * M * rows;  // x [M][3][rows], v [M][3][rows], radius [M][rows]
}
int cart_work_ensure(mrf_handle* h, int64_t n_scen) {
  const size_t need = cart_work_need(h, n_scen);
  if (need <= h->cart_work_bytes) return MRF_OK;
  mrf_host::cart_work_release(h);
  if (int rc = check_hip(h, hipMalloc(&h->cart_work, need), "hipMalloc (Cartesian rollout obstacles)")) return rc;
  h->cart_work_bytes = need;
  return MRF_OK;
}
}  // namespace

extern "C" {

int mrf_rollout_cartesian_coupled(mrf_handle* h, int64_t n_scen, const void* q0, const void* qdot0, const void* params,
                                  void* avg_out, void* traj_q, void* traj_qd, void* stream) {
  MRF_CHECK_READY(h);
  if (int rc = need_panda_vel(h, "rollout_cartesian_coupled")) return rc;
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || !q0 || !qdot0 || !params || !avg_out) return fail(h, MRF_E_ARG, "null/negative argument");
  const int N = h->cfg.n_robots, S = h->cfg.n_spheres;
  const int64_t rows = n_scen * N;
  const int M = S * (N - 1);
  if (M == 0) return mrf_rollout_cartesian(h, rows, q0, qdot0, params, 0, 0, nullptr, nullptr, nullptr, nullptr, avg_out, traj_q, traj_qd, stream);
  // small batches: one wave per scenario, the other robots' start states staged in LDS once (no obstacle arrays at all)
  if (int rc = mrf_host::rollout_cartesian_coop(h, n_scen, q0, qdot0, params, avg_out, traj_q, traj_qd, stream); rc != 1) return rc;
  // link-origin sphere table: the start states stay in an LDS tile for the whole horizon (no obstacle arrays either)
  if (int rc = mrf_host::rollout_cartesian_tile(h, n_scen, q0, qdot0, params, avg_out, traj_q, traj_qd, stream); rc != 1) return rc;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing((hipStream_t)stream, &cap);
  if (cart_work_need(h, n_scen) > h->cart_work_bytes) {
    if (cap != hipStreamCaptureStatusNone)
      return fail(h, MRF_E_ARG, "rollout_cartesian_coupled: work buffer too small inside a stream capture (run once outside first)");
    if (int rc = cart_work_ensure(h, n_scen)) return rc;
  }
  const size_t sb = h->cfg.scalar == MRF_F64 ? 8 : 4;
  char* ox = (char*)h->cart_work;
  char* ov = ox + sb * 3 * (size_t)M * rows;
  char* orad = ov + sb * 3 * (size_t)M * rows;
  // round 6: one launch -- the rollout kernel assembles its obstacles in its prologue (k_rollout_carts_panda).  The two-launch
  // form below (k_publish_obstacles + the obstacle-array kernel) stays selectable for A/B: MRF_CART_PUBLISH=1.
  static const bool two_launches = [] {
    const char* e = getenv("MRF_CART_PUBLISH");
    return e && e[0] == '1';
  }();
  if (!two_launches && S <= MRF_MAX_SPHERES && N <= 64)
    return mrf_host::rollout_cartesian_self(h, n_scen, q0, qdot0, params, ox, ov, avg_out, traj_q, traj_qd, stream);
  dim3 block(64), grid((unsigned)((rows + 63) / 64));
  int rc = dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_publish_obstacles<T>, grid, block, (hipStream_t)stream, (const mrf::DevCfg<T>*)h->dcfg, n_scen,
                  (const T*)q0, (const T*)qdot0, (T*)ox, (T*)ov, (T*)orad);
  });
  if (rc) return rc;
  // zero obstacle accelerations: FPC:33 presets them and the drivers never change that (obst_a = NULL kernel)
  return mrf_rollout_cartesian(h, rows, q0, qdot0, params, M, 0, ox, ov, nullptr, orad, avg_out, traj_q, traj_qd, stream);
}

int mrf_episode_set_rollout(mrf_handle* h_rollout, int32_t kind) {
  MRF_CHECK_READY(h_rollout);
  if (kind != MRF_ROLLOUT_JOINTSPACE && kind != MRF_ROLLOUT_CARTESIAN) return fail(h_rollout, MRF_E_ARG, "unknown rollout kind");
  h_rollout->episode_rollout_kind = kind;
  return MRF_OK;
}



int mrf_episode_set_recorder(mrf_handle* h, void* q_hist, int32_t* sm_hist, int64_t* t_begin, int64_t* t_end, int32_t* done_at,
                             int32_t* step_counter, int32_t capacity, int32_t done_state) {
  MRF_CHECK_READY(h);
  if (capacity <= 0 || !step_counter) {
    h->rec = mrf_handle::Recorder();
    return MRF_OK;
  }
  h->rec.on = true;
  h->rec.q_hist = q_hist; h->rec.sm_hist = sm_hist; h->rec.t_begin = t_begin; h->rec.t_end = t_end;
  h->rec.done_at = done_at; h->rec.counter = step_counter; h->rec.capacity = capacity; h->rec.done_state = done_state;
  return MRF_OK;
}

static int control_step(mrf_handle* hr, mrf_handle* ha, int64_t n_scen, const mrf_deadlock_config* dl, int apply_estimate,
                        const double* vel_limit, double stop_margin, void* q, void* qd, const void* prm_nom,
                        void* prm_work, const int32_t* sm, int32_t* dl_state, void* dl_goal, void* x_ee, void* avg,
                        void* act, void* st) {
  const int64_t rows = n_scen * ha->cfg.n_robots;
  int rc;
  const mrf_handle::PickPlace& pp = ha->pp;
  const mrf_handle::Recorder& rc_ = ha->rec;
  mrf::RecView rec;
  std::memset(&rec, 0, sizeof(rec));
  if (rc_.on) {
    rec.counter = rc_.counter; rec.capacity = rc_.capacity; rec.done_state = rc_.done_state; rec.t_begin = rc_.t_begin;
    rec.t_end = rc_.t_end; rec.q_hist = rc_.q_hist; rec.sm_hist = rc_.sm_hist; rec.done_at = rc_.done_at;
  }
  if (ha->cfg.model != MRF_MODEL_PANDA7 || ha->cfg.mode != MRF_MODE_VEL)
    return fail(ha, MRF_E_CONFIG, "the control step integrates velocity commands of the panda7 model (mode 'vel')");
  if (hr || pp.on) {
    // head: [recorder begin] + hand FK (+ RF-CV estimate) + the state machine's goals for the rows the estimate left
    // alone, one launch.  The estimate's constants are the rollout handle's.
    mrf_handle* hp = hr ? hr : ha;
    const int est = hr ? apply_estimate : 0;
    if (pp.on && (pp.sm.nr_blocks < 1 || (pp.sm.model != 0 && pp.sm.model != 1)))
      return fail(ha, MRF_E_CONFIG, "nr_blocks >= 1 and model in {0,1}");
    dim3 block(64), grid((unsigned)((rows + 63) / 64));
    rc = dispatch_scalar(ha, [&](auto t) {
      using T = decltype(t);
      mrf::SmCfg<T> C;
      std::memset(&C, 0, sizeof(C));
      if (pp.on) C = make_smcfg<T>(&pp.sm, ha->cfg.dt);
      return launch(ha, mrf::k_step_head<T>, grid, block, (hipStream_t)st, (const mrf::DevCfg<T>*)hp->dcfg, rows, (const T*)q,
                    (const T*)qd, (const T*)prm_nom, (T*)prm_work, (int)est, (T*)x_ee, (int)pp.on, C, (const T*)pp.start_goal,
                    (const T*)pp.blocks, (int)pp.n_block_arrays, (T*)pp.q_gripper, pp.sm_state, (T*)pp.sm_goal,
                    (int)(est ? hp->cfg.goal_estimate_mask : 0), (T*)pp.gripper_action, rec);
    });
    if (rc) return rc;
    if (pp.on) sm = pp.sm_state;  // row MRF_SM_STATE (the first `rows` entries)
  } else if (rec.counter) {
    if ((rc = launch(ha, mrf::k_step_begin, dim3(1), dim3(1), (hipStream_t)st, rec.counter, rec.capacity, rec.t_begin))) return rc;
  }
  if (hr) {
    rc = hr->episode_rollout_kind == MRF_ROLLOUT_CARTESIAN
             ? mrf_rollout_cartesian_coupled(hr, n_scen, q, qd, prm_work, avg, nullptr, nullptr, st)
             : mrf_rollout(hr, n_scen, q, qd, prm_work, avg, nullptr, nullptr, st);
    if (rc) return rc;
    if (dl && (rc = mrf_deadlock_step(hr, n_scen, dl, -1, x_ee, avg, sm, prm_work, dl_state, dl_goal, st))) return rc;
  }
  if ((rc = mrf_compute_action_coupled(ha, n_scen, q, qd, (hr || pp.on) ? prm_work : prm_nom, 0, nullptr, act, st))) return rc;
  if (pp.on && pp.h_grasp &&
      (rc = mrf_compute_action(pp.h_grasp, rows, q, qd, prm_work, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr,
                               pp.action_grasp, st))) {
    if (ha->err.empty()) ha->err = pp.h_grasp->err;
    return rc;
  }
  // tail: action selection by state (EXJ:414-445) + clip / integration (EXJ:452-453) + the step's record, one launch
  dim3 block(256), grid((unsigned)((rows + 255) / 256));
  return dispatch_scalar(ha, [&](auto t) {
    using T = decltype(t);
    mrf::VelLimits<T> L;
    for (int j = 0; j < MRF_DOF_MAX; ++j) L.v[j] = (T)vel_limit[j];
    return launch(ha, mrf::k_step_tail<T>, grid, block, (hipStream_t)st, (const mrf::DevCfg<T>*)ha->dcfg, rows, (T*)q, (T*)qd,
                  (T*)act, L, (T)stop_margin, (const int32_t*)(pp.on ? pp.sm_state : nullptr), (const T*)pp.action_grasp,
                  (const int32_t*)sm, rec);
  });
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
  if ((hr || ha->pp.on) && (!params_work || !x_ee_work)) return fail(ha, MRF_E_ARG, "work buffers missing");
  if (hr && !avg_work) return fail(ha, MRF_E_ARG, "work buffers missing");
  if (hr && dl && (!dl_state || !dl_goal)) return fail(ha, MRF_E_ARG, "deadlock state missing");
  auto one = [&](void* st) {
    int rc = control_step(hr, ha, n_scen, dl, apply_estimate, vel_limit, stop_margin, q_io, qdot_io, params_nominal,
                          params_work, sm_state, dl_state, dl_goal, x_ee_work, avg_work, action_out, st);
    if (rc && hr && ha->err.empty()) ha->err = hr->err;
    return rc;
  };
  if (hr && hr->episode_rollout_kind == MRF_ROLLOUT_CARTESIAN && hr->cfg.n_robots > 1 && hr->cfg.n_spheres > 0 &&
      !mrf_host::coop_applies(hr, n_scen) && !mrf_host::cartesian_tile_applies(hr)) {
    // only the obstacle-array form needs the work buffer (the cooperative and the tile forms keep the spheres on chip)
    if (int rc = cart_work_ensure(hr, n_scen)) {  // outside any capture: the captured step must not allocate
      if (ha->err.empty()) ha->err = hr->err;
      return rc;
    }
  }
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
                            (uint64_t)(uintptr_t)(hr ? hr->dcfg : nullptr), (uint64_t)(hr ? hr->episode_rollout_kind : 0),
                            (uint64_t)(uintptr_t)(hr ? hr->cart_work : nullptr)};
  key.append((const char*)ident, sizeof(ident));
  key.append((const char*)&ha->pp, sizeof(ha->pp));  // attached pick-and-place buffers and constants
  key.append((const char*)&ha->rec, sizeof(ha->rec));  // attached recorder
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
