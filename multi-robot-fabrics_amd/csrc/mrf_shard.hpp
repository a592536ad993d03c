// mrf_shard.hpp -- device-side pieces shared by the robot-sharded kernels (mrf_comm.hip: the persistent PEER kernel;
// mrf_shard_step.hip: the per-step kernels of the RCCL / torch transports).  Round 6:
//
//   * LOCALITY.  The robots a rank owns sit in adjacent lanes of one wave, so their exchange (FPJ:211-225) stays on chip:
//     the [72][64] link-origin tile or the chunked generic exchange of mrf_tile.hpp -- exactly what k_rollout_panda does.
//     Only robots of OTHER ranks come out of an exchange buffer.
//   * PAYLOAD.  What a remote robot sends is selectable (mrf_config.exchange):
//       XK_JOINTS   cos q, sin q, qdot of its 7 joints (21 scalars).  The receiving lane stages them in LDS and re-walks
//                   the sender's chain with the rolled walk, folding every sphere as it is emitted; the spheres never
//                   exist in memory.  168 B per robot, scenario and step in float64 whatever the sphere table.
//       XK_SPHERES  the SX predicted spheres as 9 scalars each (the literal "all-gather of sphere centres"): 432 B for the
//                   reference's link-origin table, 1 440 B for BASELINE config 5's 20 spheres.
//       XK_NONE     the rank owns every robot: nothing is exchanged.
#pragma once
#include "mrf_device.hpp"
#include "mrf_tile.hpp"

namespace mrf {
inline namespace MRF_DEVICE_FLAVOUR {

enum { XK_NONE = 0, XK_JOINTS = 1, XK_SPHERES = 2, XK_JOINTS_TAGGED = 3 };  // TAGGED: mrf_comm.hip, the peer kernel only

// the d-th robot (d = 0 .. N - count - 1) that is NOT in the owned block [first, first + count)
__device__ __forceinline__ int remote_robot(int d, int first, int count) { return d < first ? d : d + count; }

// LDS rows the remote joint states are staged in.  LO: the sphere tile itself -- its 72 rows are free between the local
// fold of this step and the publish of the next one (three robots at a time); generic tables: the chunk area behind the
// own joint rows (36 rows: one robot at a time).  The radii / multiplicity rows behind the tile are never touched.
template <bool LO>
struct RemoteStage {
  static constexpr int kRobots = LO ? 3 : 1;
  static constexpr int kOffset = LO ? 0 : GEN_XCH;
};

// emit_link hook of the UNROLLED chain walk (panda_walk_own) for the reference's link-origin table: the sphere of link L sits
// at that link's origin and goes to this lane's column of the [72][64] tile the moment the walk has completed the link
// (coincident origins once, as publish_link_spheres lays them out).  Folding inside the hook instead -- eight inlined leaf
// evaluations along the unrolled walk -- was tried first: 1.5 KB of scratch per lane, 2.3x slower than the rolled walk.
template <typename T>
struct TileOriginEmit {
  T* __restrict__ col;  // tile + lane
  bool dyn;
  T jsign;
  int m01, m45;
  __device__ __forceinline__ void operator()(int link, const T*, const T*, const T*, const T* o, const T*, const T*, const T* vo,
                                             const T* ao) const {
    const int s = link - 1;
    if ((s == 1 && m01) || (s == 5 && m45)) return;
    T* d9 = col + lo_slot(s, m01, m45) * 9 * 64;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      d9[c * 64] = o[c];
      d9[(3 + c) * 64] = dyn ? vo[c] : T(0);          // FPJ:215-217
      d9[(6 + c) * 64] = dyn ? jsign * ao[c] : T(0);  // FPJ:97-99 + UT:28
    }
  }
};

// Fold the spheres of every robot of another rank, re-derived from its exchanged joint state.  load_robot(jr, v) fills v with
// the 21 scalars (cos, sin, qdot of joint c/3) of robot jr for THIS lane's scenario; all 21 loads of a robot are issued before
// the first is used (one memory round trip per robot), then go to the LDS rows the rolled walk reads by joint index (a
// runtime index into registers would put them into scratch memory).
// PREFETCH: (link-origin tables) request the next robot's scalars after this robot's walk, so that they arrive under its fold;
// false where load_robot is more than 21 loads (the tagged payload's poll loop: 42 words in flight spilled 744 B there).
template <class CL, bool LO, bool PREFETCH = true, typename T, class LoadRobot>
__device__ __forceinline__ void remote_obstacles_joints(const DevCfg<T>& cfg, T* __restrict__ xch, int lane, int first,
                                                        int count, int N, LoadRobot load_robot, const EgoPts<T, NG>& E,
                                                        EgoAcc<T, NG>& acc, [[maybe_unused]] long long* tm_stage = nullptr) {
  const int nrem = N - count;
  const bool dyn = cfg.dynamic != 0;
  const int m01 = LO ? cfg.lo_merge01 : 0, m45 = LO ? cfg.lo_merge45 : 0;
#ifndef MRF_REMOTE_WALK_ROLLED
  if constexpr (LO) {
    // Link-origin table: the remote chain is walked by the UNROLLED walk straight from the 21 loaded scalars (registers: no
    // joint-state stage, no walk loop whose every multiply-add waits for the one before in the only wave of the SIMD); its
    // spheres go to this lane's OWN column of the tile and are folded from there by the loop the local fold uses.  The
    // column is lane-private: one barrier frees the tile after the local fold, none between the remote robots.  Measured with three ranks on one die
    // (tools/peer_timing.py, round 6): the rolled form's two walks cost 7 us of a 39 us step.
    (void)tm_stage;
    T* col = xch + lane;
    const int SX = 8 - m01 - m45;
    __syncthreads();  // the tile is free: the local fold has finished in every lane
    // the next robot's 21 scalars are requested after this robot's walk (whose registers are free by then) and arrive
    // under its fold -- held across the WALK they spilled (536 B of scratch)
    T nxt[MRF_JOINT_STATE_SCALARS];
    if (PREFETCH && nrem > 0) load_robot(remote_robot(0, first, count), nxt);
#pragma unroll 1
    for (int d = 0; d < nrem; ++d) {
      const int jr = remote_robot(d, first, count);
      if (!PREFETCH) load_robot(jr, nxt);
      {
        T cq[7], sq[7], qd[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          cq[j] = nxt[3 * j + 0];
          sq[j] = nxt[3 * j + 1];
          qd[j] = nxt[3 * j + 2];
        }
        PandaKin<T> K;  // outputs nobody reads: only the hook's stores survive
        panda_walk_own<T, 7>(cfg.mount[jr], cq, sq, qd, K, TileOriginEmit<T>{col, dyn, cfg.jsign, m01, m45});
      }
      if (PREFETCH && d + 1 < nrem) load_robot(remote_robot(d + 1, first, count), nxt);
      pipelined_pairs<T, 9>(
          SX,
          [&](int m, T (&sp)[9]) {
#pragma unroll
            for (int c = 0; c < 9; ++c) sp[c] = col[(m * 9 + c) * 64];
          },
          [&](int m, T (&sp)[9]) {
            const int sph = lo_sphere(m, m01, m45);
            accumulate_obstacle<CL>(cfg, E, sp, sp + 3, sp + 6, cfg.sphere_r[sph], false, acc, T(lo_count(m, m01, m45)));
          });
    }
    return;
  }
#endif
  constexpr int RCH = RemoteStage<LO>::kRobots;
  T* stage = xch + RemoteStage<LO>::kOffset;
#pragma unroll 1
  for (int d0 = 0; d0 < nrem; d0 += RCH) {
    const int nch = nrem - d0 < RCH ? nrem - d0 : RCH;
    __syncthreads();  // the rows are free: the local fold / the previous chunk's walks have finished in every lane
#ifdef MRF_PEER_TIMING
    const long long tms_ = wall_clock64();
#endif
#pragma unroll 1
    for (int dd = 0; dd < nch; ++dd) {
      const int jr = remote_robot(d0 + dd, first, count);
      T v[MRF_JOINT_STATE_SCALARS];
      load_robot(jr, v);
#pragma unroll
      for (int c = 0; c < MRF_JOINT_STATE_SCALARS; ++c) stage[(dd * MRF_JOINT_STATE_SCALARS + c) * 64 + lane] = v[c];
    }
    __syncthreads();
#ifdef MRF_PEER_TIMING
    if (tm_stage) *tm_stage += wall_clock64() - tms_;
#endif
#pragma unroll 1
    for (int dd = 0; dd < nch; ++dd) {
      const int jr = remote_robot(d0 + dd, first, count);
      const T* st = stage + dd * MRF_JOINT_STATE_SCALARS * 64 + lane;
      panda_walk_spheres<LO, T>(
          cfg, cfg.mount[jr],
          [&](int j, T& c, T& s, T& qdj) {
            c = st[(3 * j + 0) * 64];
            s = st[(3 * j + 1) * 64];
            qdj = st[(3 * j + 2) * 64];
          },
          [&](int s, const T* x, const T* v, const T* a) {
            if (LO && ((s == 1 && m01) || (s == 5 && m45))) return;  // coincident link origins: folded once, weight 2
            const T mult = (LO && ((s == 0 && m01) || (s == 4 && m45))) ? T(2) : T(1);
            T vv[3], aa[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              vv[c] = dyn ? v[c] : T(0);              // FPJ:215-217
              aa[c] = dyn ? cfg.jsign * a[c] : T(0);  // FPJ:97-99 + UT:28
            }
            accumulate_obstacle<CL>(cfg, E, x, vv, aa, cfg.sphere_r[s], false, acc, mult);
          });
    }
  }
}

// The same for the sphere payload: load(jr, slot, c) returns scalar c (x, v, a) of exchanged sphere `slot` of robot jr for
// this lane's scenario; one flat software-pipelined loop over (remote robot, slot) pairs.
template <class CL, bool LO, typename T, class Load>
__device__ __forceinline__ void remote_obstacles_spheres(const DevCfg<T>& cfg, int first, int count, int N, Load load,
                                                         const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
  const int m01 = LO ? cfg.lo_merge01 : 0, m45 = LO ? cfg.lo_merge45 : 0;
  const int SX = cfg.n_spheres - m01 - m45;
  const bool dyn = cfg.dynamic != 0;
  pipelined_pairs<T, 9>(
      (N - count) * SX,
      [&](int m, T (&buf)[9]) {
        const int d = m / SX, slot = m - d * SX;
        const int jr = remote_robot(d, first, count);
#pragma unroll
        for (int c = 0; c < 9; ++c) buf[c] = load(jr, slot, c);
      },
      [&](int m, T (&buf)[9]) {
        const int slot = m % SX;
        const int s = LO ? lo_sphere(slot, m01, m45) : slot;
        const T mult = LO ? T(lo_count(slot, m01, m45)) : T(1);
        T v[3], a[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          v[c] = dyn ? buf[3 + c] : T(0);
          a[c] = dyn ? buf[6 + c] : T(0);
        }
        accumulate_obstacle<CL>(cfg, E, buf, v, a, cfg.sphere_r[s], false, acc, mult);
      });
}

// One fabric solve of an owned robot inside a robot-sharded step.  Lanes are (scenario ls, owned robot l) with the
// `count` owned robots of a scenario adjacent.
//   publish_remote(K1)   after the own chain walk (sphere payload: stores into the peers' buffers + flags); may be a no-op
//   before_remote()      after the LOCAL fold, before the first remote read (the PEER kernel waits for the flags here, so
//                        the flag round trip overlaps the own walk and the local fold)
//   remote(E, acc)       folds the robots of other ranks (remote_obstacles_joints / _spheres over the caller's buffers)
template <class LS, bool LO, bool REMOTE, typename T, class PRM, class PublishRemote, class BeforeRemote, class Remote>
__device__ __forceinline__ void sharded_solve_row(const DevCfg<T>& cfg, T* __restrict__ xch, int lane, int ls, int l, int count,
                                                  const T* __restrict__ mount_own, const PandaState<T>& R, const PRM& P,
                                                  PublishRemote publish_remote, BeforeRemote before_remote, Remote remote,
                                                  T (&qdd)[7], T (&act)[7]) {
  const bool dyn = cfg.dynamic != 0;
  if constexpr (!LO) {
    // rows [0, 21): the own cos q / sin q / qdot -- the rolled walks of the own chain (chunked local exchange, the sphere
    // payload's publish) read them by joint index
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      xch[(3 * j + 0) * 64 + lane] = R.cq[j];
      xch[(3 * j + 1) * 64 + lane] = R.sq[j];
      xch[(3 * j + 2) * 64 + lane] = R.qd[j];
    }
    __syncthreads();
  }
  // the single-walk form (own chain kinematics alive across the sphere loop) only where the loop is the light tile fold:
  // a remote chain walk inside it would spill
  constexpr bool SW = LO && kSingleWalk<LS> && !REMOTE;
  panda_solve_row<LS, SW>(
      cfg, mount_own, R, P,
      [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
        if (count > 1) {
          if constexpr (LO)
            obstacles_from_tile<typename LS::Collision>(cfg, xch, ls, l, count, E, acc);
          else
            obstacles_generic_chunked<typename LS::Collision>(cfg, xch, lane, ls, l, count, mount_own, dyn, cfg.jsign, E, acc);
        }
        if constexpr (REMOTE) {
          before_remote();
          remote(E, acc);
        }
      },
      qdd, act,
      [&](const PandaKin<T>& K1) {
        if constexpr (LO) {
          if (count > 1) {
            __syncthreads();
            publish_link_spheres(xch, lane, K1, dyn, dyn, cfg.jsign, cfg.lo_merge01, cfg.lo_merge45);  // FPJ:97-99,215-220
            __syncthreads();
          }
        }
        if constexpr (REMOTE) publish_remote(K1);
      });
}

}  // inline namespace
}  // namespace mrf
