// mrf_rollout_wp.hpp -- the coupled joint-space rollout (FPJ:190-249) with TWO resident waves per SIMD.
//
// k_rollout_panda keeps one (scenario, robot) row per lane and needs ~230 live f64 values in the pullback phase: one wave
// per SIMD, 508 of 512 registers, ~15 % of the issue slots spent on VGPR<->AGPR moves and ~19 % on scalar / LDS / wait
// slots that nothing overlaps.  A second wave per SIMD needs <= 128 live f64 values per lane AND an
// exchange tile of <= 20 KB per wave; one lane per row can do neither (the sphere loop alone holds 30 ego + 45 accumulator
// + 28 state values, and 8 waves x 64 rows x 45 exchanged scalars are 184 KB of the CU's 160 KB).
//
// Here a row is owned by a PAIR OF WAVES of one 128-thread workgroup -- lane l of wave A and lane l of wave B work on the
// same (scenario, robot) -- and the solve is split by collision point, which is also a split by joint:
//   wave A ("arm")   links 3, 4, 5/6: walks the chain up to joint 5's origin only, its pullbacks touch the 4x4 block of M;
//                    joint-limit leaves; then the geometry solve and its energization coefficient alpha_g
//   wave B ("hand")  links 7, 8: full walk, pullbacks into the 6x6 block; the three attractors, the forced solve,
//                    energization / damping and system_step (FPJ:72-80) -- the state owner
// Both waves fold THEIR points against all spheres of the other robots from one shared [45][64] LDS tile (the five moving
// link origins; the origin of links 1/2 is a per-robot constant and comes from a 3-scalar table).  What crosses between the
// waves goes through LDS: the partial specs (20 + 27 scalars per row, in the tile's storage once both sphere loops are
// done), alpha_g (and h_g for a planner without goals), and the joint state itself, which lives in a [28][64] LDS array
// between steps so that neither wave carries q, qdot, cos q, sin q across its sphere loop.  LDS: 47 + 28 rows of 512 B +
// tables (mounts, static spheres, radii) = 39.9 KB per workgroup, four workgroups = eight waves per CU.  Five workgroup
// barriers per rollout step.
//
// Applies to: float64, the reference's leaf strings (compile-time leaf policies), the link-origin sphere table with equal
// radii on the coincident origins (1,2) and (5,6) -- the reference's rollouts (PM:23-26).  Everything else runs
// k_rollout_panda.  Same results to round-off (summation order differs): the parity tests run both.
#pragma once
#include "mrf_device.hpp"

namespace mrf {

// -DMRF_WP_CLOCKS (development aid, tools/wp_phases.py): both waves of the middle workgroup stamp the shader-cycle counter
// at their phase boundaries of horizon step 5 into mrf_wp_clocks[wave][slot], read back by mrf_debug_wp_clocks().
#ifdef MRF_WP_CLOCKS
__device__ long long mrf_wp_clocks[2][16];
#define WP_STAMP(slot)                                                                                      \
  do {                                                                                                      \
    if (blockIdx.x == gridDim.x / 2 && lane == 0 && k == 5)                                                 \
      mrf_wp_clocks[wave][slot] = (long long)__builtin_readcyclecounter();                                  \
  } while (0)
#elif defined(MRF_ISA_MARKS)  // tools/isa_stats.py: static instruction counts between the same phase boundaries
#define WP_STAMP(slot) asm volatile("; MRFMARK wp" #slot)
#else
#define WP_STAMP(slot)
#endif

constexpr int WP_TILE = 47 * 64;                   // 5 slots x (x, v, a) x 64 rows during the sphere loops; then the exchange
constexpr int WP_PARK = 28 * 64;                   // q, qdot, cos q, sin q of every row, owned by wave B
constexpr int WP_STAT = MRF_MAX_ROBOTS * 3;        // origin of links 1 = 2 of every robot (a constant of the mount)
constexpr int WP_RAD = 8;                          // radius of the static sphere, of the 5 tile slots; role flag
constexpr int WP_MNT = MRF_MAX_ROBOTS * 12;        // mount transforms (read by robot index in every step)
constexpr int WP_SCALARS = WP_TILE + WP_PARK + WP_STAT + WP_RAD + WP_MNT;
static_assert(WP_SCALARS * 8 <= 40960, "four workgroups per CU need <= 40 KB of LDS each");
constexpr int WP_XB = 0;                           // tile rows of wave B's partial spec: 6x6 block of M (21) + f (6)
constexpr int WP_XA = 27;                          // tile rows of wave A's partial spec: 4x4 block (10) + f (4) and the
                                                   // joint-limit leaves' diagonal / force entries of joints 4..6 (3 + 3)
constexpr int WP_Y = 0;                            // tile rows that carry alpha_g (and h_g) to wave B once A has read XB

template <typename T, int NP>
__device__ __forceinline__ void wp_fold(const DevCfg<T>& cfg, const T* __restrict__ tile, const T* __restrict__ stat,
                                        const T* __restrict__ rad, int ls, int li, int N, const EgoPts<T, NP>& E,
                                        EgoAcc<T, NP>& acc) {
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  using CL = LeafPow<4, 4, MRF_GATE_NONE, MRF_GATE_NONE>;
  const T zero[3] = {T(0), T(0), T(0)};
  // links 1 and 2 of every other robot: one point that never moves, folded once with weight 2
#pragma unroll 1
  for (int d = 0; d + 1 < N; ++d) {
    int jr = li + 1 + d;
    if (jr >= N) jr -= N;
    lds_ptr s = (lds_ptr)(stat + jr * 3);
    const T xs[3] = {s[0], s[1], s[2]};
    accumulate_obstacle<CL>(cfg, E, xs, zero, zero, ((lds_ptr)rad)[0], false, acc, T(2));
  }
  // the five moving link origins of every other robot: one flat loop over (robot, slot), all ten operands of a sphere
  // requested at the top of its iteration (volatile: in program order).  Fetching the NEXT sphere before folding the
  // current one (the row kernel's depth-one pipeline) needs 20 more registers here and was measured slower: 212 B of scratch
  // per lane, 3.72 against 3.43 ms (profiles/r05_wp_pmc.json "_meta").
#pragma unroll 1
  for (int m = 0; m < 5 * (N - 1); ++m) {
    const int d = m / 5, slot = m - 5 * d;
    int jr = li + 1 + d;
    if (jr >= N) jr -= N;
    typedef const volatile __attribute__((address_space(3))) T* lds_vptr;
    lds_vptr src = (lds_vptr)(tile + slot * 9 * 64 + ls * N + jr);
    T buf[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) buf[k] = src[k * 64];
    const T ro = ((lds_vptr)rad)[1 + slot];
    accumulate_obstacle<CL>(cfg, E, buf, buf + 3, buf + 6, ro, false, acc, slot == 2 ? T(2) : T(1));
  }
}

// t = b + A c with c = jsign * Jdot qd of the point, then the pullback
template <typename T, int NC>
__device__ __forceinline__ void wp_pull(const DevCfg<T>& cfg, QSpec<T, 7>& S, const PandaKin<T>& K, const T* pp, const T* aa,
                                        const T* A6, const T* b) {
  T c[3] = {cfg.jsign * aa[0], cfg.jsign * aa[1], cfg.jsign * aa[2]};
  T t[3] = {b[0] + A6[0] * c[0] + A6[1] * c[1] + A6[2] * c[2], b[1] + A6[1] * c[0] + A6[3] * c[1] + A6[4] * c[2],
            b[2] + A6[2] * c[0] + A6[4] * c[1] + A6[5] * c[2]};
  pull_point<T, NC>(S, K, pp, A6, t);
}

template <typename T, class LS>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_rollout_panda_wp(
    const DevCfg<T>* __restrict__ cfgp, int64_t n_scen, const T* __restrict__ q0, const T* __restrict__ qd0,
    const T* __restrict__ prm, T* __restrict__ avg_out, T* __restrict__ traj_q, T* __restrict__ traj_qd,
    long long* __restrict__ probe, long long serial) {
  __shared__ T lds[WP_SCALARS];
  T* const tile = lds;
  T* const park = lds + WP_TILE;
  T* const stat = park + WP_PARK;
  T* const rad = stat + WP_STAT;
  T* const mnt = rad + WP_RAD;
  const DevCfg<T>& cfg = *cfgp;
  // Which wave of the pair takes which role: the two waves of a workgroup sit on neighbouring SIMDs and the next workgroup
  // on the same SIMDs gets the next hardware wave slot, so "first wave is A on even slots, B on odd slots" puts one A and
  // one B wave on every SIMD (the roles differ in length: a SIMD with two A waves would be the CU's critical path).  Only
  // a scheduling heuristic: any assignment is correct as long as the pair disagrees, and wave 1 takes what wave 0 left.
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x == 0) ((int*)rad)[2 * WP_RAD - 1] = (int)(__builtin_amdgcn_s_getreg(6148) & 1);  // HW_ID.WAVE_ID
  __syncthreads();
  const int swap = __builtin_amdgcn_readfirstlane(((const int*)rad)[2 * WP_RAD - 1]);
  const int wave = wave_in_block ^ swap;
  const int lane = threadIdx.x & 63;
  const bool probing = probe && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && wave == 0 && lane == 0;
  long long* const stamp = probe + (blockIdx.x == 0 ? 0 : 4);
  if (probing) {
    stamp[0] = (long long)__builtin_readcyclecounter();
    stamp[1] = (long long)wall_clock64();
    if (blockIdx.x == 0) probe[8] = serial;  // which mrf_rollout call these stamps belong to
    if (gridDim.x == 1) {  // one workgroup is first and last: both slots carry its stamps
      stamp[4] = stamp[0];
      stamp[5] = stamp[1];
    }
  }
  const int N = cfg.n_robots;
  const int spw = 64 / N;
  int ls = lane / N;
  const int li = lane - ls * N;
  int64_t scen = (int64_t)blockIdx.x * spw + ls;
  const bool active = ls < spw && scen < n_scen;
  if (ls >= spw) ls = 0;  // idle tail lanes shadow the block's first scenario (no stores)
  if (scen >= n_scen || !active) scen = (int64_t)blockIdx.x * spw + ls;
  if (scen >= n_scen) scen = n_scen - 1;
  const int64_t rows = n_scen * N;
  const int64_t row = scen * N + li;
  const T* const mount_lds = mnt + li * 12;
  const int H = cfg.horizon;
  const bool dyn = cfg.dynamic != 0;
  const bool forced = cfg.n_goals > 0;
  typedef const __attribute__((address_space(3))) T* lds_ptr;

  if (wave == 0) {
    // ------------------------------------------------------------------------------------------- wave A
    if (lane < N) {
      const T* m = cfg.mount[lane];
#pragma unroll
      for (int c = 0; c < 12; ++c) mnt[lane * 12 + c] = m[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) stat[lane * 3 + c] = m[4 * c + 3] + T(kPZ[0]) * m[4 * c + 2];
    }
    if (lane < 6) {
      constexpr int sph[6] = {0, 2, 3, 4, 6, 7};  // first sphere of: static point, links 3, 4, 5/6, 7, 8
      int s = sph[0];
#pragma unroll
      for (int i = 1; i < 6; ++i) s = lane == i ? sph[i] : s;
      rad[lane] = cfg.sphere_r[s];
    }
    __syncthreads();  // P0: tables staged, state of step 0 parked by wave B
#pragma unroll 1
    for (int k = 0; k < H; ++k) {
      // per-row parameters are re-read in every step (L2 hits): the opaque offset keeps the loads from being hoisted out
      // of the step loop, where they would sit in registers across the sphere loop
      int zk = 0;
      asm volatile("" : "+s"(zk));
      PrmView<T> P{prm + zk, rows, row, {T(0), T(0), T(0)}, false};
      WP_STAMP(0);
      QSpec<T, 7> S;
      S.zero();
      {
        EgoPts<T, 3> E;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          E.rb[g][0] = P[MRF_P_RADIUS_BODY + g];
          E.rb[g][1] = T(0);
          E.nl[g] = 1;
        }
        E.rb[2][1] = P[MRF_P_RADIUS_BODY + 3];
        E.nl[2] = 2;
        const T con[4] = {P[MRF_P_CONSTRAINT_0], P[MRF_P_CONSTRAINT_0 + 1], P[MRF_P_CONSTRAINT_0 + 2], P[MRF_P_CONSTRAINT_0 + 3]};
        T cq[7], sq[7], qd[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) cq[j] = sq[j] = qd[j] = T(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          qd[j] = ((lds_ptr)park)[(7 + j) * 64 + lane];
          cq[j] = ((lds_ptr)park)[(14 + j) * 64 + lane];
          sq[j] = ((lds_ptr)park)[(21 + j) * 64 + lane];
        }
        PandaKin<T> K;
        panda_walk_own<T, 4>(mount_lds, cq, sq, qd, K);
        // publish links 3, 4, 5/6 (tile slots 0..2): x, v, jsign * Jdot qd   (FPJ:97-99,215-220)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          T* dst = tile + s * 9 * 64 + lane;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            dst[c * 64] = K.o[2 + s][c];
            dst[(3 + c) * 64] = dyn ? K.vo[2 + s][c] : T(0);
            dst[(6 + c) * 64] = dyn ? cfg.jsign * K.ao[2 + s][c] : T(0);
          }
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            E.p[g][c] = K.o[2 + g][c];
            E.v[g][c] = K.vo[2 + g][c];
          }
        }
        EgoAcc<T, 3> acc;
        acc.zero();
        if (cfg.n_ego > 0 && cfg.n_planes > 0) accumulate_plane<typename LS::Plane>(cfg, E, con, acc);
        WP_STAMP(1);
        __syncthreads();  // B2: every robot's spheres of this step are in the tile
        WP_STAMP(2);
        if (cfg.n_ego > 0) {
          wp_fold<T, 3>(cfg, tile, stat, rad, ls, li, N, E, acc);
          WP_STAMP(3);
          wp_pull<T, 2>(cfg, S, K, K.o[2], K.ao[2], acc.A[0], acc.b[0]);
          wp_pull<T, 3>(cfg, S, K, K.o[3], K.ao[3], acc.A[1], acc.b[1]);
          wp_pull<T, 4>(cfg, S, K, K.o[4], K.ao[4], acc.A[2], acc.b[2]);
        }
      }
      T qq = T(0);  // qdot . qdot, for alpha_g
      {
        T q[7], qd[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          q[j] = ((lds_ptr)park)[j * 64 + lane];
          qd[j] = ((lds_ptr)park)[(7 + j) * 64 + lane];
          qq += qd[j] * qd[j];
        }
        if (cfg.use_limits) {
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            T m, f;
            scalar_leaf_t<typename LS::Limit>(cfg.lg, cfg.lf, q[j] - cfg.limits[j][0], qd[j], m, f);
            S.M[tri<7>(j, j)] += m;
            S.f[j] += f;
            scalar_leaf_t<typename LS::Limit>(cfg.lg, cfg.lf, cfg.limits[j][1] - q[j], -qd[j], m, f);
            S.M[tri<7>(j, j)] += m;
            S.f[j] -= f;
          }
        }
      }
      WP_STAMP(4);
      __syncthreads();  // Bfree: both sphere loops are done, the tile's storage becomes the exchange
      WP_STAMP(5);
      {
        int x = WP_XA;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = i; j < 4; ++j) tile[(x++) * 64 + lane] = S.M[tri<7>(i, j)];
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[(x++) * 64 + lane] = S.f[j];
#pragma unroll
        for (int j = 4; j < 7; ++j) {
          tile[(x++) * 64 + lane] = S.M[tri<7>(j, j)];
          tile[(x++) * 64 + lane] = S.f[j];
        }
      }
      WP_STAMP(6);
      __syncthreads();  // Bx: both partial specs are in the exchange
      WP_STAMP(7);
      {
        int x = WP_XB;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = i; j < 6; ++j) S.M[tri<7>(i, j)] += ((lds_ptr)tile)[(x++) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 6; ++j) S.f[j] += ((lds_ptr)tile)[(x++) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 7; ++j) S.M[tri<7>(j, j)] += cfg.base_mass;
      }
      T hg[7];
      ldl_solve<T, 7>(S, cfg.eps, hg);
      T qh = T(0);
#pragma unroll
      for (int j = 0; j < 7; ++j) qh += ((lds_ptr)park)[(7 + j) * 64 + lane] * hg[j];
      tile[WP_Y * 64 + lane] = -qh * fast_rcp(cfg.eps + qq);  // alpha_g; XB has been consumed by this very wave
      if (!forced) {
#pragma unroll
        for (int j = 0; j < 7; ++j) tile[(WP_Y + 1 + j) * 64 + lane] = hg[j];
      }
      WP_STAMP(8);
      __syncthreads();  // Bc: alpha_g (h_g) handed to wave B
      WP_STAMP(9);
      WP_STAMP(10);
      __syncthreads();  // Bd: wave B has parked the state of the next step
      WP_STAMP(11);
    }
  } else {
    // ------------------------------------------------------------------------------------------- wave B
    PandaState<T> R;
    load_state(rows, row, q0, qd0, R);
    T g0[3] = {T(0), T(0), T(0)};
    const bool own_goal = ((cfg.goal_mask >> li) & 1) != 0;
    if (own_goal) {
      // RF-CV: the goal of this robot is not communicated; use x_ee + T * v_ee of the start state (EXC:355-357)
      PandaKin<T> K0;
      panda_walk_own<T>(cfg.mount[li], R.cq, R.sq, R.qd, K0);
#pragma unroll
      for (int c = 0; c < 3; ++c) g0[c] = K0.p8[c] + cfg.goal_T * K0.v8[c];
    }
    // system_step 'vel' (FPJ:77-80): q += dt*qdot; cos q / sin q advance by the angle-sum formula while every |dq| in the
    // wave is small, a full sincos otherwise.  Step 0's here, every later one right after the action of the step before.
    auto integrate = [&](PandaState<T>& S) {
      T dq[7];
      bool small = true;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        dq[j] = cfg.dt * S.qd[j];
        small = small && (m_abs(dq[j]) < T(0.125));
        S.q[j] += dq[j];
      }
      if (__all(small)) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          T sd, cd;
          small_sincos(dq[j], sd, cd);
          const T c = S.cq[j] * cd - S.sq[j] * sd;
          const T s = S.sq[j] * cd + S.cq[j] * sd;
          S.cq[j] = c;
          S.sq[j] = s;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 7; ++j) m_sincos(S.q[j], &S.sq[j], &S.cq[j]);
      }
    };
    auto park_state = [&](const PandaState<T>& S) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        park[j * 64 + lane] = S.q[j];
        park[(7 + j) * 64 + lane] = S.qd[j];
        park[(14 + j) * 64 + lane] = S.cq[j];
        park[(21 + j) * 64 + lane] = S.sq[j];
      }
    };
    integrate(R);
    park_state(R);
    T sumsq = T(0);
    __syncthreads();  // P0
#pragma unroll 1
    for (int k = 0; k < H; ++k) {
      int zk = 0;
      asm volatile("" : "+s"(zk));
      PrmView<T> P{prm + zk, rows, row, {g0[0], g0[1], g0[2]}, own_goal};
      WP_STAMP(0);
      QSpec<T, 7> S, SA;  // geometry part (pullbacks of links 7, 8), attractor part
      S.zero();
      SA.zero();
      T xpsi = T(0);
      {
        EgoPts<T, 2> E;
        E.rb[0][0] = P[MRF_P_RADIUS_BODY + 4];
        E.rb[1][0] = P[MRF_P_RADIUS_BODY + 5];
        E.rb[0][1] = E.rb[1][1] = T(0);
        E.nl[0] = E.nl[1] = 1;
        const T con[4] = {P[MRF_P_CONSTRAINT_0], P[MRF_P_CONSTRAINT_0 + 1], P[MRF_P_CONSTRAINT_0 + 2], P[MRF_P_CONSTRAINT_0 + 3]};
        PandaKin<T> K;
        panda_walk_own<T>(mount_lds, R.cq, R.sq, R.qd, K);  // R: the state this wave integrated (and parked) last
        // publish links 7, 8 (tile slots 3, 4)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          T* dst = tile + (3 + s) * 9 * 64 + lane;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            dst[c * 64] = s ? K.p8[c] : K.o[6][c];
            dst[(3 + c) * 64] = dyn ? (s ? K.v8[c] : K.vo[6][c]) : T(0);
            dst[(6 + c) * 64] = dyn ? cfg.jsign * (s ? K.a8[c] : K.ao[6][c]) : T(0);
          }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          E.p[0][c] = K.o[6][c];
          E.v[0][c] = K.vo[6][c];
          E.p[1][c] = K.p8[c];
          E.v[1][c] = K.v8[c];
        }
        EgoAcc<T, 2> acc;
        acc.zero();
        if (cfg.n_ego > 0 && cfg.n_planes > 0) accumulate_plane<typename LS::Plane>(cfg, E, con, acc);
        WP_STAMP(1);
        __syncthreads();  // B2
        WP_STAMP(2);
        if (cfg.n_ego > 0) {
          wp_fold<T, 2>(cfg, tile, stat, rad, ls, li, N, E, acc);
          WP_STAMP(3);
          wp_pull<T, 6>(cfg, S, K, K.o[6], K.ao[6], acc.A[0], acc.b[0]);
          wp_pull<T, 6>(cfg, S, K, K.p8, K.a8, acc.A[1], acc.b[1]);
        }
        // the goal parameters in ONE batch of loads after the pullbacks, pinned by a scheduling barrier: left to itself the
        // scheduler sinks every load to its use (an L2 round trip each); before the pullbacks there is no room for them
        T gp[MRF_P_CONSTRAINT_0];
        if (forced) {
#pragma unroll
          for (int i = 0; i < MRF_P_CONSTRAINT_0; ++i) gp[i] = P[i];
          __builtin_amdgcn_sched_barrier(0);
          // attractor 0: panda_hand position -> x_goal_0   (EXJ:32-41)
          {
            T x0[3] = {K.p8[0] - gp[MRF_P_X_GOAL_0], K.p8[1] - gp[MRF_P_X_GOAL_0 + 1], K.p8[2] - gp[MRF_P_X_GOAL_0 + 2]};
            T twoA, f0[3];
            attractor<T, 3>(cfg, x0, gp[MRF_P_WEIGHT_GOAL_0], twoA, f0, xpsi);
            T t[3] = {f0[0] + twoA * cfg.jsign * K.a8[0], f0[1] + twoA * cfg.jsign * K.a8[1], f0[2] + twoA * cfg.jsign * K.a8[2]};
            pull_point_iso<T, 6>(SA, K, K.p8, twoA, t);
          }
          if (cfg.n_goals > 2) {
            // attractor 2: joint index 6 -> x_goal_2   (EXJ:53-60)
            T x2[1] = {((lds_ptr)park)[6 * 64 + lane] - gp[MRF_P_X_GOAL_2]};
            T twoA, f2[1], rn;
            attractor<T, 1>(cfg, x2, gp[MRF_P_WEIGHT_GOAL_2], twoA, f2, rn);
            SA.M[tri<7>(6, 6)] += twoA;
            SA.f[6] += f2[0];
          }
          if (cfg.n_goals > 1) {
            // attractor 1: R (p_hand - p_link7) -> x_goal_1 ; p_hand - p_link7 = 0.107 z_6   (EXJ:42-52)
            const T* Rm = gp + MRF_P_ANGLE_GOAL_1;
            T d8[3] = {K.p8[0] - K.o[6][0], K.p8[1] - K.o[6][1], K.p8[2] - K.o[6][2]};
            T da[3] = {K.a8[0] - K.ao[6][0], K.a8[1] - K.ao[6][1], K.a8[2] - K.ao[6][2]};
            T x1[3], c1[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              x1[i] = Rm[3 * i] * d8[0] + Rm[3 * i + 1] * d8[1] + Rm[3 * i + 2] * d8[2] - gp[MRF_P_X_GOAL_1 + i];
              c1[i] = cfg.jsign * (Rm[3 * i] * da[0] + Rm[3 * i + 1] * da[1] + Rm[3 * i + 2] * da[2]);
            }
            T twoA, f1[3], rn;
            attractor<T, 3>(cfg, x1, gp[MRF_P_WEIGHT_GOAL_1], twoA, f1, rn);
            T t[3] = {f1[0] + twoA * c1[0], f1[1] + twoA * c1[1], f1[2] + twoA * c1[2]};
            T J[6][3];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
              T cz[3];
              cross3(K.z[j], d8, cz);  // d(p8 - o6)/dq_j
#pragma unroll
              for (int i = 0; i < 3; ++i) J[j][i] = Rm[3 * i] * cz[0] + Rm[3 * i + 1] * cz[1] + Rm[3 * i + 2] * cz[2];
              SA.f[j] += dot3(J[j], t);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
              for (int j = i; j < 6; ++j) SA.M[tri<7>(i, j)] += twoA * dot3(J[i], J[j]);
          }
        }
      }
      WP_STAMP(4);
      __syncthreads();  // Bfree
      WP_STAMP(5);
      {
        int x = WP_XB;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = i; j < 6; ++j) tile[(x++) * 64 + lane] = S.M[tri<7>(i, j)];
#pragma unroll
        for (int j = 0; j < 6; ++j) tile[(x++) * 64 + lane] = S.f[j];
      }
      WP_STAMP(6);
      __syncthreads();  // Bx
      WP_STAMP(7);
      T hf[7];
      if (forced) {
        // forced spec = geometry (own part + wave A's part + base) + attractors
        int x = WP_XA;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = i; j < 4; ++j) S.M[tri<7>(i, j)] += ((lds_ptr)tile)[(x++) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) S.f[j] += ((lds_ptr)tile)[(x++) * 64 + lane];
#pragma unroll
        for (int j = 4; j < 7; ++j) {
          S.M[tri<7>(j, j)] += ((lds_ptr)tile)[(x++) * 64 + lane];
          S.f[j] += ((lds_ptr)tile)[(x++) * 64 + lane];
        }
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
          for (int j = i; j < 7; ++j) S.M[tri<7>(i, j)] += SA.M[tri<7>(i, j)];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          S.f[j] += SA.f[j];
          S.M[tri<7>(j, j)] += cfg.base_mass;
        }
        ldl_solve<T, 7>(S, cfg.eps, hf);
      }
      // the parked state of this step, for the action and system_step
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        R.q[j] = ((lds_ptr)park)[j * 64 + lane];
        R.qd[j] = ((lds_ptr)park)[(7 + j) * 64 + lane];
        R.cq[j] = ((lds_ptr)park)[(14 + j) * 64 + lane];
        R.sq[j] = ((lds_ptr)park)[(21 + j) * 64 + lane];
      }
      WP_STAMP(8);
      __syncthreads();  // Bc: wave A's alpha_g (h_g) is in the exchange
      WP_STAMP(9);
      const T alpha_g = ((lds_ptr)tile)[WP_Y * 64 + lane];
      T hg[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) hg[j] = T(0);
      if (!forced) {
#pragma unroll
        for (int j = 0; j < 7; ++j) hg[j] = hf[j] = ((lds_ptr)tile)[(WP_Y + 1 + j) * 64 + lane];
      }
      T qdd[7], act[7];
      finish<T, 7>(cfg, R.qd, forced, alpha_g, hg, hf, xpsi, qdd, act);
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
      if (k + 1 < H) {
        integrate(R);
        park_state(R);
      }
      WP_STAMP(10);
      __syncthreads();  // Bd: state of the next step parked
      WP_STAMP(11);
    }
    if (active) avg_out[row] = sumsq / (T)(H * 7);  // FPJ:102-116
  }
  if (probing) {
    stamp[2] = (long long)__builtin_readcyclecounter();
    stamp[3] = (long long)wall_clock64();
    if (gridDim.x == 1) {
      stamp[6] = stamp[2];
      stamp[7] = stamp[3];
    }
  }
}

}  // namespace mrf
