// EXPERIMENT, not shipped (round 4; -DMRF_COOP4, then kernel_select = 3 or auto below one scenario per CU): correct -- the
// GPU parity suites pass with it, N = 1..6 -- and SLOWER than the one-wave kernel: 24.0 k cycles per horizon step against
// 22.6 k (profiles/r04_experiments.json "four waves per scenario" has the per-wave phase stamps).  A step is a chain of
// dependent phases (walk -> spheres -> fold -> pullback -> LDL^T -> action); waves shorten the wide ones, but every hand-over
// is a barrier plus an LDS round trip, the fold does not shrink with the spheres per lane (4.0-4.8 k cycles for ONE sphere:
// selects, rsqrt chain, DPP sums) and one LDL^T alone is ~4.5 k cycles of dependent f64 operations.
//
// Four waves per scenario: the latency form of the coupled kernels for batches that do not even give every CU a scenario
// (one cell in real time is n_scen = 1).  Included by mrf_kernels.hip inside namespace mrf.
//
// k_coop_panda (one wave per scenario) spreads a step's obstacle FOLD over the lanes of a wave but every lane then
// finishes the 7x7 part of its robot redundantly: 5 point pullbacks, 14 limit leaves, 3 attractors and two LDL^T solves
// one after the other, ~2.6 k instructions = 12 of the step's 22 k cycles, on ONE SIMD while the CU's other three idle.
// Lanes of a wave cannot shorten that (r02 built it: the jobs serialise), waves can: here a scenario is a 256-thread block,
// one wave per SIMD, and a step is
//   all lanes      integrate, walk the own robot's chain (every lane keeps its robot's state; duplicates stay bit-identical)
//   wave 0         publishes the robots' sphere states in LDS
//   waves 1, 2, 3  meanwhile: attractor 0 / limit leaves + attractor 2 / attractor 1 (none needs the spheres)   -- barrier 1
//   all lanes      lane = (robot i, ego point g, chunk c): fold the chunk's spheres, DPP row sum over the chunk lanes,
//                  chunk 0 leaves the point's (A, b) in LDS                                                  -- barrier 2
//   waves 0..3     pull back point 3 / point 4 (the hand) / points 0 and 1 / point 2 -> partial specs in LDS  -- barrier 3
//   wave 0 / 1     sum the partials; LDL^T of the geometry / of the forced spec, side by side -> h_g, h_f     -- barrier 4
//   all lanes      energisation, damping, action; next step
// Rows of 16 lanes (8 for 4..6 robots) are (robot, point) pairs numbered r = point * N + robot, so that every wave holds
// lanes of every robot -- the wave-wide jobs need the robot's kinematics in their own registers.
// Results differ from the other kernels in summation order only (tests: same tolerance against the oracle).

#ifdef MRF_COOP_CLOCKS  // tools/coop4_phases.py: every wave's lane 0 stamps its phase boundaries at horizon step 5
#define MRF_STAMP4(slot)                                                                                              \
  do {                                                                                                                \
    if (blockIdx.x == 0 && lane == 0 && k == 5) mrf_dbg_clocks[w * 8 + slot] = (long long)__builtin_readcyclecounter(); \
  } while (0)
#else
#define MRF_STAMP4(slot)
#endif

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// all-lanes sum over the 8 (half row) or 16 (row) lanes a value's chunk partials sit in
template <typename T>
__device__ __forceinline__ T chunk_sum(T v, int C) {
  v += quad_xor<1>(v);
  v += quad_xor<2>(v);
  v += dpp_move<0x141>(v);               // row_half_mirror: lane i of an 8-group <- lane 7 - i
  if (C == 16) v += dpp_move<0x140>(v);  // row_mirror: lane i of a row <- lane 15 - i
  return v;
}

__host__ __device__ inline int coop4_chunks(int n_robots) { return n_robots <= 3 ? 16 : 8; }
// LDS scalars: parameters, sphere table, generic-walk staging (wave 0), point accumulators, partial specs, solutions
__host__ __device__ inline size_t coop4_lds_scalars(int n_robots, int S) {
  return (size_t)n_robots * (MRF_NPARAM + S * 9 + 5 * 9 + 7 * 36 + 16) + 21 * 64;
}

template <typename T, class LS, bool LO, bool COOP_ROLLOUT>
__global__ __launch_bounds__(256) void k_coop4_panda(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen,
                                                      const T* __restrict__ q0, const T* __restrict__ qd0,
                                                      const T* __restrict__ prm, int use_accel, T* __restrict__ avg_out,
                                                      T* __restrict__ traj_q, T* __restrict__ traj_qd,
                                                      T* __restrict__ qdd_out, T* __restrict__ act_out) {
  extern __shared__ __align__(16) unsigned char coop4_lds[];
  const DevCfg<T>& cfg = *cfgp;
  const int N = cfg.n_robots;
  const int m01 = LO ? cfg.lo_merge01 : 0, m45 = LO ? cfg.lo_merge45 : 0;
  const int S = LO ? 8 - m01 - m45 : cfg.n_spheres;
  T* prm_lds = reinterpret_cast<T*>(coop4_lds);       // [MRF_NPARAM][N]
  T* sph = prm_lds + MRF_NPARAM * N;                    // [N][S][9]
  T* pacc = sph + (size_t)N * (LO ? 8 : S) * 9;         // [N][5][9]   (A, b) of every ego point
  T* part = pacc + N * 5 * 9;                           // [7][N][36]  0..3 geometry partials, 4..6 attractor partials
  T* hsol = part + 7 * N * 36;                          // [N][16]     h_g (0..6), h_f (8..14), psi norm (15)
  T* xch = hsol + N * 16;                               // [21][64]    wave 0: cos q, sin q, qdot per lane (generic tables)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: say so
  const int C = coop4_chunks(N);
  const int row = tid / C;  // (robot, point) pair: point-major so that consecutive rows walk through the robots
  const int c = tid - row * C;
  const int i = row % N;
  int g = row / N;
  const bool real = g < NG;  // spare rows shadow the hand point of their robot and leave nothing behind
  if (!real) g = NG - 1;
  // one lane per robot speaks for it: row i of wave 0 (rows 0 .. N-1 are point 0 of robots 0 .. N-1), chunk 0
  const bool writer = row < N && c == 0;
  const int64_t scen = blockIdx.x;
  const int64_t rows = n_scen * N;
  const int64_t grow = scen * N + i;

  PandaState<T> R;
  load_state_values(rows, grow, q0, qd0, R);
  const T* mount_own = cfg.mount[i];
  for (int idx = tid; idx < MRF_NPARAM * N; idx += 256) {
    const int cpar = idx / N, rr = idx - cpar * N;
    prm_lds[idx] = prm[(int64_t)cpar * rows + scen * N + rr];
  }
  state_sincos(R);
  __syncthreads();
  PrmView<T> P{prm_lds, N, i, {T(0), T(0), T(0)}, false};
  if (COOP_ROLLOUT && ((cfg.goal_mask >> i) & 1)) {  // RF-CV: the goal of this robot is estimated from its hand motion
    PandaKin<T> K0;
    panda_walk_own<T>(mount_own, R.cq, R.sq, R.qd, K0);
#pragma unroll
    for (int k = 0; k < 3; ++k) P.g0[k] = K0.p8[k] + cfg.goal_T * K0.v8[k];
    P.own_goal = true;
  }
  const bool dyn = cfg.dynamic != 0;
  const bool acc_on = dyn && (COOP_ROLLOUT || use_accel);
  const int M_all = (N - 1) * S;
  const bool forced = cfg.n_goals > 0;
  // The spheres this lane folds are the same in every step: their LDS offsets, radii and multiplicities are worked out
  // once (a division and a load from the constants per sphere otherwise sit on the step's critical path).
  constexpr int PRE = 4;
  const bool pre = (M_all + C - 1) / C <= PRE;
  int pre_off[PRE];
  T pre_rad[PRE], pre_mul[PRE];
#pragma unroll
  for (int t = 0; t < PRE; ++t) {
    const int m = c + t * C;
    const bool valid = pre && m < M_all;
    const int d = valid ? m / S : 0, sp = valid ? m - d * S : 0;
    int jr = i + 1 + d;
    if (jr >= N) jr -= N;
    pre_off[t] = valid ? (jr * S + sp) * 9 : -1;
    pre_rad[t] = cfg.sphere_r[LO ? lo_sphere(sp, m01, m45) : sp];
    pre_mul[t] = LO ? T(lo_count(sp, m01, m45)) : T(1);
  }
  const int row_in_wave = lane / C;
  const bool speaker = row_in_wave < N && c == 0;  // one lane per (wave, robot): rows 0 .. N-1 of a wave are N different robots
  T sumsq = T(0);
  const int H = COOP_ROLLOUT ? cfg.horizon : 1;
#pragma unroll 1
  for (int k = 0; k < H; ++k) {
    MRF_STAMP4(0);
    if (COOP_ROLLOUT) {
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
    PandaKin<T> K;
    panda_walk_own<T>(mount_own, R.cq, R.sq, R.qd, K);
    QSpec<T, 7> Sg, Sa;  // geometry (pullbacks, limits) and attractor contributions of this wave
    Sg.zero();
    Sa.zero();
    T xpsi = T(0);
    // ---- wave 0 publishes every robot's sphere states (the previous step's readers left at barrier 2); the other waves
    //      meanwhile evaluate the leaves that do not need the spheres
    if (w == 0) {
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
        // the generic table walk indexes the joint state at run time: staged per lane in LDS (own entries only, no barrier)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          xch[(3 * j + 0) * 64 + lane] = R.cq[j];
          xch[(3 * j + 1) * 64 + lane] = R.sq[j];
          xch[(3 * j + 2) * 64 + lane] = R.qd[j];
        }
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
    } else if (w == 1) {
      if (forced) {  // attractor 0: panda_hand position -> x_goal_0   (EXJ:32-41)
        T x0[3] = {K.p8[0] - P[MRF_P_X_GOAL_0], K.p8[1] - P[MRF_P_X_GOAL_0 + 1], K.p8[2] - P[MRF_P_X_GOAL_0 + 2]};
        T twoA, f0[3];
        attractor<T, 3>(cfg, x0, P[MRF_P_WEIGHT_GOAL_0], twoA, f0, xpsi);
        T t[3] = {f0[0] + twoA * cfg.jsign * K.a8[0], f0[1] + twoA * cfg.jsign * K.a8[1], f0[2] + twoA * cfg.jsign * K.a8[2]};
        pull_point_iso<T, 6>(Sa, K, K.p8, twoA, t);
      }
    } else if (w == 2) {
      if (cfg.use_limits) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          T m, f;
          scalar_leaf_t<typename LS::Limit>(cfg.lg, cfg.lf, R.q[j] - cfg.limits[j][0], R.qd[j], m, f);
          Sg.M[tri<7>(j, j)] += m;
          Sg.f[j] += f;
          scalar_leaf_t<typename LS::Limit>(cfg.lg, cfg.lf, cfg.limits[j][1] - R.q[j], -R.qd[j], m, f);
          Sg.M[tri<7>(j, j)] += m;
          Sg.f[j] -= f;
        }
      }
      if (cfg.n_goals > 2) {  // attractor 2: joint index 6 -> x_goal_2   (EXJ:53-60)
        T x2[1] = {R.q[6] - P[MRF_P_X_GOAL_2]};
        T twoA, f2[1], rn;
        attractor<T, 1>(cfg, x2, P[MRF_P_WEIGHT_GOAL_2], twoA, f2, rn);
        Sa.M[tri<7>(6, 6)] += twoA;
        Sa.f[6] += f2[0];
      }
    } else if (cfg.n_goals > 1) {  // wave 3, attractor 1: R (p_hand - p_link7) -> x_goal_1   (EXJ:42-52)
      T Rm[9];
#pragma unroll
      for (int e = 0; e < 9; ++e) Rm[e] = P[MRF_P_ANGLE_GOAL_1 + e];
      T d8[3] = {K.p8[0] - K.o[6][0], K.p8[1] - K.o[6][1], K.p8[2] - K.o[6][2]};
      T da[3] = {K.a8[0] - K.ao[6][0], K.a8[1] - K.ao[6][1], K.a8[2] - K.ao[6][2]};
      T x1[3], c1[3];
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        x1[e] = Rm[3 * e] * d8[0] + Rm[3 * e + 1] * d8[1] + Rm[3 * e + 2] * d8[2] - P[MRF_P_X_GOAL_1 + e];
        c1[e] = cfg.jsign * (Rm[3 * e] * da[0] + Rm[3 * e + 1] * da[1] + Rm[3 * e + 2] * da[2]);
      }
      T twoA, f1[3], rn;
      attractor<T, 3>(cfg, x1, P[MRF_P_WEIGHT_GOAL_1], twoA, f1, rn);
      T t[3] = {f1[0] + twoA * c1[0], f1[1] + twoA * c1[1], f1[2] + twoA * c1[2]};
      T J[6][3];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        T cz[3];
        cross3(K.z[j], d8, cz);  // d(p8 - o6)/dq_j
#pragma unroll
        for (int e = 0; e < 3; ++e) J[j][e] = Rm[3 * e] * cz[0] + Rm[3 * e + 1] * cz[1] + Rm[3 * e + 2] * cz[2];
        Sa.f[j] += dot3(J[j], t);
      }
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) Sa.M[tri<7>(a, b)] += twoA * dot3(J[a], J[b]);
    }
    MRF_STAMP4(1);
    __syncthreads();  // barrier 1
    MRF_STAMP4(2);
    // ---- my ego point against my chunk of the other robots' spheres
    if (cfg.n_ego > 0) {
      EgoPts<T, 1> E1;
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) {
        E1.p[0][k3] = g == 0 ? K.o[2][k3] : (g == 1 ? K.o[3][k3] : (g == 2 ? K.o[4][k3] : (g == 3 ? K.o[6][k3] : K.p8[k3])));
        E1.v[0][k3] = g == 0 ? K.vo[2][k3] : (g == 1 ? K.vo[3][k3] : (g == 2 ? K.vo[4][k3] : (g == 3 ? K.vo[6][k3] : K.v8[k3])));
      }
      ego_point_links<LS::Collision::generic>(cfg, P, g, E1.rb[0][0], E1.rb[0][1], E1.nl[0]);
      EgoAcc<T, 1> a1;
      a1.zero();
      if (pre) {
#pragma unroll
        for (int t = 0; t < PRE; ++t) {
          if (pre_off[t] < 0) continue;
          const T* src = sph + pre_off[t];
          T x[3] = {src[0], src[1], src[2]}, v[3] = {src[3], src[4], src[5]}, a[3] = {src[6], src[7], src[8]};
          accumulate_obstacle<typename LS::Collision>(cfg, E1, x, v, a, pre_rad[t], false, a1, pre_mul[t]);
        }
      } else {
#pragma unroll 1
        for (int m = c; m < M_all; m += C) {
          const int d = m / S;
          const int sp = m - d * S;
          int jr = i + 1 + d;
          if (jr >= N) jr -= N;
          const T* src = sph + ((size_t)jr * S + sp) * 9;
          T x[3] = {src[0], src[1], src[2]}, v[3] = {src[3], src[4], src[5]}, a[3] = {src[6], src[7], src[8]};
          accumulate_obstacle<typename LS::Collision>(cfg, E1, x, v, a, cfg.sphere_r[LO ? lo_sphere(sp, m01, m45) : sp], false,
                                                      a1, LO ? T(lo_count(sp, m01, m45)) : T(1));
        }
      }
      if (cfg.n_planes > 0 && c == 0) {
        T con[4] = {P[MRF_P_CONSTRAINT_0], P[MRF_P_CONSTRAINT_0 + 1], P[MRF_P_CONSTRAINT_0 + 2], P[MRF_P_CONSTRAINT_0 + 3]};
        accumulate_plane<typename LS::Plane>(cfg, E1, con, a1);
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) a1.A[0][e] = chunk_sum(a1.A[0][e], C);
#pragma unroll
      for (int e = 0; e < 3; ++e) a1.b[0][e] = chunk_sum(a1.b[0][e], C);
      if (real && c == 0) {
        T* dst = pacc + (i * 5 + g) * 9;
#pragma unroll
        for (int e = 0; e < 6; ++e) dst[e] = a1.A[0][e];
#pragma unroll
        for (int e = 0; e < 3; ++e) dst[6 + e] = a1.b[0][e];
      }
    }
    MRF_STAMP4(3);
    __syncthreads();  // barrier 2
    MRF_STAMP4(4);
    // ---- the pullbacks, one or two points per wave; every lane works for its own robot i
    if (cfg.n_ego > 0) {
      // t = b + A c with c = jsign * Jdot qd of the point (panda_finish_row)
      auto pull = [&](auto nc, int gp, const T* pp, const T* aa) {
        constexpr int NC = decltype(nc)::value;
        const T* src = pacc + (i * 5 + gp) * 9;
        T A[6] = {src[0], src[1], src[2], src[3], src[4], src[5]}, b[3] = {src[6], src[7], src[8]};
        T cc[3] = {cfg.jsign * aa[0], cfg.jsign * aa[1], cfg.jsign * aa[2]};
        T t[3] = {b[0] + A[0] * cc[0] + A[1] * cc[1] + A[2] * cc[2], b[1] + A[1] * cc[0] + A[3] * cc[1] + A[4] * cc[2],
                  b[2] + A[2] * cc[0] + A[4] * cc[1] + A[5] * cc[2]};
        pull_point<T, NC>(Sg, K, pp, A, t);
      };
      if (w == 0) {
        pull(std::integral_constant<int, 6>{}, 3, K.o[6], K.ao[6]);
      } else if (w == 1) {
        pull(std::integral_constant<int, 6>{}, 4, K.p8, K.a8);
      } else if (w == 2) {
        pull(std::integral_constant<int, 2>{}, 0, K.o[2], K.ao[2]);
        pull(std::integral_constant<int, 3>{}, 1, K.o[3], K.ao[3]);
      } else {
        pull(std::integral_constant<int, 4>{}, 2, K.o[4], K.ao[4]);
      }
    }
    if (speaker) {  // the wave's partials: geometry slot w, attractor slot 3 + w (waves 1..3)
      T* dst = part + ((size_t)w * N + i) * 36;
#pragma unroll
      for (int e = 0; e < 28; ++e) dst[e] = Sg.M[e];
#pragma unroll
      for (int e = 0; e < 7; ++e) dst[28 + e] = Sg.f[e];
      if (w > 0) {
        T* da = part + ((size_t)(3 + w) * N + i) * 36;
#pragma unroll
        for (int e = 0; e < 28; ++e) da[e] = Sa.M[e];
#pragma unroll
        for (int e = 0; e < 7; ++e) da[28 + e] = Sa.f[e];
        if (w == 1) hsol[i * 16 + 15] = xpsi;
      }
    }
    MRF_STAMP4(5);
    __syncthreads();  // barrier 3
    // ---- the two solves side by side: wave 0 the geometry, wave 1 the forced spec
    if (w < 2) {
      QSpec<T, 7> Sq;
      Sq.zero();
#pragma unroll
      for (int j = 0; j < 7; ++j) Sq.M[tri<7>(j, j)] = cfg.base_mass;
      auto add = [&](int p) {
        const T* src = part + ((size_t)p * N + i) * 36;
#pragma unroll
        for (int e = 0; e < 28; ++e) Sq.M[e] += src[e];
#pragma unroll
        for (int e = 0; e < 7; ++e) Sq.f[e] += src[28 + e];
      };
#pragma unroll
      for (int p = 0; p < 4; ++p) add(p);
      if (w == 1 && forced) {
#pragma unroll
        for (int p = 4; p < 7; ++p) add(p);
      }
      T hh[7];
      ldl_solve<T, 7>(Sq, cfg.eps, hh);
      if (speaker) {
#pragma unroll
        for (int j = 0; j < 7; ++j) hsol[i * 16 + 8 * w + j] = hh[j];
      }
    }
    MRF_STAMP4(6);
    __syncthreads();  // barrier 4
    T qdd[7], act[7];
    {
      T hg[7], hf[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        hg[j] = hsol[i * 16 + j];
        hf[j] = hsol[i * 16 + 8 + j];
      }
      const T xpsi = hsol[i * 16 + 15];
      T qq = T(0), qh = T(0);
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        qq += R.qd[j] * R.qd[j];
        qh += R.qd[j] * hg[j];
      }
      const T alpha_g = -qh * fast_rcp(cfg.eps + qq);
      finish<T, 7>(cfg, R.qd, forced, alpha_g, hg, hf, xpsi, qdd, act);
    }
    if (COOP_ROLLOUT) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        R.qd[j] = act[j];
        sumsq += act[j] * act[j];
      }
      if (writer && traj_q) {
#pragma unroll
        for (int j = 0; j < 7; ++j) traj_q[((int64_t)k * 7 + j) * rows + grow] = R.q[j];
      }
      if (writer && traj_qd) {
#pragma unroll
        for (int j = 0; j < 7; ++j) traj_qd[((int64_t)k * 7 + j) * rows + grow] = R.qd[j];
      }
    } else if (writer) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if (qdd_out) qdd_out[j * rows + grow] = qdd[j];
        act_out[j * rows + grow] = act[j];
      }
    }
    MRF_STAMP4(7);
  }
  if (COOP_ROLLOUT && writer) avg_out[grow] = sumsq / (T)(H * 7);
}
