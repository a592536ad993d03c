// mrf_comm.hip -- robot-sharded Rollout Fabrics inside the library (include/mrf.h "Robot-sharded rollout INSIDE the
// library").  The exchange step of the reference's rollout graph (forward_planner_Jointspace.py:211-225:
// at step k robot i reads the predicted spheres of every robot j != i) crosses GPUs here:
//
//   RCCL transport   host loop in C++: predict -> ncclAllGather -> action per horizon step, everything enqueued on the
//                    caller's stream (kernels: mrf_shard_step.hip).  librccl is dlopen'ed (torch ships its own copy under
//                    the same soname; whichever is already in the process is the one that gets used).
//   PEER transport   k_rollout_peer: ONE persistent kernel per rollout.  Workgroup = one wave = the owned robots of
//                    floor(64/cnt_max) scenarios.  Per step a lane stores what its robot sends (mrf_config.exchange: its
//                    joint state right after the position update, or its spheres after the chain walk) straight into
//                    every OTHER rank's exchange buffer (peer-mapped device memory: over xGMI between GPUs), fences,
//                    raises the per-workgroup flag on every rank, exchanges with the robots of its OWN rank on chip (LDS),
//                    folds them, and only then polls the flags the peers raised for the same scenarios and folds the
//                    remote robots from its LOCAL buffer.  Two buffer generations (step parity) are enough: a rank can
//                    publish step k+2 only after it has seen every peer's step k+1 flag, which a peer raises after it
//                    finished reading step k.  Block X only ever waits for block X of the peers; the grid is capped at
//                    the resident workgroup count and each workgroup walks its blocks in increasing order, so no wait
//                    can depend on a workgroup that is not running.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mrf_device.hpp"
#include "mrf_host.hpp"
#include "mrf_shard.hpp"

namespace mrf {

// view of the exchange buffers that the peer kernel gets by value
struct PeerView {
  unsigned char* base[MRF_MAX_ROBOTS];  // exchange allocation of every rank (own entry: the local allocation)
  int G, grank;                         // ranks in the group, own rank
  int first[MRF_MAX_ROBOTS + 1];        // robot block of rank g: [first[g], first[g+1])
  int nblk_max;                         // flag columns per rank
  int xs;                               // scalars per robot and scenario in the buffers: 21 (joints) or 9 * SX (spheres)
  long long b_max;                      // scenario capacity of the buffers
  long long off_flags, off_err, off_x;  // byte offsets inside an allocation
  long long timeout_ticks;              // bounded spin, in wall_clock64 ticks
  int tagged;                           // the payload region holds tagged words (XK_JOINTS_TAGGED), twice the bytes
};

// flags: [2 generations][G source ranks][nblk_max]   payload: [2][n_robots][xs][b_max]
__device__ __forceinline__ unsigned long long* peer_flag(const PeerView& V, int dst, int gen, int src, int blk) {
  return reinterpret_cast<unsigned long long*>(V.base[dst] + V.off_flags) + ((size_t)(gen * V.G + src) * V.nblk_max + blk);
}
template <typename T>
__device__ __forceinline__ T* peer_x(const PeerView& V, int dst, int gen, int n_robots) {
  return reinterpret_cast<T*>(V.base[dst] + V.off_x) + (size_t)gen * n_robots * V.xs * V.b_max;
}

// Payload stores into another rank's exchange buffer.  That buffer is an IPC mapping whose caching attributes on the
// writer's side are the driver's choice, so the stores are made at system scope (write-through to the owner's memory)
// instead of relying on the mapping being fine-grained.
template <typename T>
__device__ __forceinline__ void xstore(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// ... and the reads of what the peers stored into the LOCAL buffer: system-scope loads, so that no cache level of this
// device can answer with a line from the previous use of the generation (the buffer is fine-grained memory and the flags
// were acquired at system scope; the scope on the load itself makes that independent of how the allocation is mapped)
template <typename T>
__device__ __forceinline__ T xload(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// TAGGED payload (XK_JOINTS_TAGGED, opt-in MRF_PEER_TAGGED=1; round 6): every 32 bits of payload travel in ONE 8-byte store
// together with a 32-bit tag of the step they belong to (the low-latency protocol of the collective libraries): the reader
// polls the payload words themselves until all carry the tag it expects.  No store drain before a flag, no flag, no poll
// round trip in front of the payload loads -- at twice the bytes on the link, so for the latency-bound batch range.
// 8-byte stores and loads are single-copy atomic, so a word is either the old or the new (value half, tag) pair.
// tag(seq): sequence number + reset epoch * an odd constant (a late store of the previous epoch cannot pass for any step the
// new epoch will reach), never 0 (the buffer is zeroed at creation and at every reset).
using tagged_word = unsigned long long;
__device__ __forceinline__ unsigned tagged_tag(unsigned long long seq) {
  const unsigned t = (unsigned)seq + (unsigned)(seq >> 40) * 0x9E3779B1u;
  return t ? t : 0x80000000u;
}
template <typename T>
constexpr int kTaggedWords = (int)(sizeof(T) / 4);  // words per scalar
// word w of scalar c of robot `robot`, scenario `scen`:  [2 generations][n_robots][21][W][b_max]
template <typename T>
__device__ __forceinline__ tagged_word* peer_tagged(const PeerView& V, int dst, int gen, int n_robots, int robot, int c, int w,
                                                    long long scen) {
  constexpr int W = kTaggedWords<T>;
  return reinterpret_cast<tagged_word*>(V.base[dst] + V.off_x) +
         ((((size_t)gen * n_robots + robot) * MRF_JOINT_STATE_SCALARS + c) * W + w) * (size_t)V.b_max + scen;
}
template <typename T>
__device__ __forceinline__ void tagged_store(const PeerView& V, int dst, int gen, int n_robots, int robot, int c, long long scen,
                                             T v, unsigned tag) {
  if constexpr (sizeof(T) == 8) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    __hip_atomic_store(peer_tagged<T>(V, dst, gen, n_robots, robot, c, 0, scen), (b & 0xffffffffull) | ((tagged_word)tag << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(peer_tagged<T>(V, dst, gen, n_robots, robot, c, 1, scen), (b >> 32) | ((tagged_word)tag << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  } else {
    __hip_atomic_store(peer_tagged<T>(V, dst, gen, n_robots, robot, c, 0, scen),
                       (tagged_word)__builtin_bit_cast(unsigned, v) | ((tagged_word)tag << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// One attempt at the 21 scalars of `robot` in the LOCAL buffer: all words requested at once; true when every word carries `tag`.
template <typename T>
__device__ __forceinline__ bool tagged_try_load(const PeerView& V, int gen, int n_robots, int robot, long long scen, unsigned tag,
                                                T (&v)[MRF_JOINT_STATE_SCALARS]) {
  constexpr int W = kTaggedWords<T>;
  tagged_word w[MRF_JOINT_STATE_SCALARS][W];
#pragma unroll
  for (int c = 0; c < MRF_JOINT_STATE_SCALARS; ++c)
#pragma unroll
    for (int k = 0; k < W; ++k)
      w[c][k] = __hip_atomic_load(peer_tagged<T>(V, V.grank, gen, n_robots, robot, c, k, scen), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_SYSTEM);
  bool ok = true;
#pragma unroll
  for (int c = 0; c < MRF_JOINT_STATE_SCALARS; ++c) {
#pragma unroll
    for (int k = 0; k < W; ++k) ok = ok && (unsigned)(w[c][k] >> 32) == tag;
    if constexpr (W == 2)
      v[c] = __builtin_bit_cast(T, (w[c][0] & 0xffffffffull) | (w[c][1] << 32));
    else
      v[c] = __builtin_bit_cast(T, (unsigned)w[c][0]);
  }
  return ok;
}
// ... until they all do (bounded like the flag wait: the group's error word, the time-out, the post-mortem)
template <typename T>
__device__ __forceinline__ void tagged_load_robot(const PeerView& V, int gen, int n_robots, int robot, long long scen, bool active,
                                                  unsigned long long seq, int blk, const int* err, int etag, int lane,
                                                  bool& broken_seen, T (&v)[MRF_JOINT_STATE_SCALARS]) {
  const unsigned tag = tagged_tag(seq);
  const long long t0 = wall_clock64();
  int tries = 0;
#pragma unroll 1
  for (;; ++tries) {  // one copy of the 42 loads: the first attempt is the loop's first pass
    const bool ok = tagged_try_load<T>(V, gen, n_robots, robot, scen, tag, v);
    if (__all(ok || !active) || broken_seen) return;
    // the error word (an uncached round trip of its own) and the clock only every 16th miss: a miss is usually a payload
    // that is a microsecond away
    if ((tries & 15) != 15) {
      if (tries < 8)
        __builtin_amdgcn_s_sleep(4);
      else
        __builtin_amdgcn_s_sleep(32);
      continue;
    }
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == etag) break;
    if (wall_clock64() - t0 > V.timeout_ticks) {
      if (lane == 0) {
        int* dbg = reinterpret_cast<int*>(V.base[V.grank] + V.off_err);
        if (atomicCAS(dbg + 1, 0, 1) == 0) {
          dbg[2] = blk;
          dbg[3] = -1 - robot;  // tagged payload: the ROBOT whose joint state did not arrive (negative: not a rank)
          dbg[4] = (int)(seq & 0x7fffffff);
          dbg[5] = 0;
          dbg[6] = (int)blockIdx.x;
          dbg[7] = (int)gridDim.x;
        }
        for (int g = 0; g < V.G; ++g)
          __hip_atomic_store(reinterpret_cast<int*>(V.base[g] + V.off_err), etag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      break;
    }
  }
  broken_seen = true;  // the group is in error: nothing more is published by this wave, the commit pass discards the rollout
}

// the 21 scalars of robot jr for this lane's scenario out of the LOCAL buffer (flag protocol: the flags were awaited before)
template <typename T>
__device__ __forceinline__ void xload_robot(const T* xloc, int jr, long long b_max, long long scen, T (&v)[MRF_JOINT_STATE_SCALARS]) {
#pragma unroll
  for (int c = 0; c < MRF_JOINT_STATE_SCALARS; ++c) v[c] = xload(xloc + ((size_t)jr * MRF_JOINT_STATE_SCALARS + c) * b_max + scen);
}

// a quiet NaN by bit pattern (this translation unit is compiled -ffast-math, where NaN literals are undefined)
template <typename T>
__device__ __forceinline__ T quiet_nan();
template <>
__device__ __forceinline__ double quiet_nan<double>() { return __builtin_bit_cast(double, 0x7ff8000000000000ull); }
template <>
__device__ __forceinline__ float quiet_nan<float>() { return __builtin_bit_cast(float, 0x7fc00000u); }

// The wave's payload stores are performed, then the per-workgroup flag goes up on every rank (own included: unused).
// Once the group is in error (this rank timed out, or a peer did and said so in this rank's error word) no further flag
// goes up: the payload behind it may have been computed from stale data, and the peers must time out -- or see the error
// -- rather than fold it.  `broken_seen` is the wave's memory of that: read from the error word once per block, set by the
// first wait that ends without its flags (peer_wait_flags) -- a flag therefore only ever stands for a payload computed from
// data that arrived, and the publish needs no uncached read of the error word (1.5 us of every step, measured round 6).
__device__ __forceinline__ void peer_raise_flags(const PeerView& V, int gen, int blk, unsigned long long seq, const int* err,
                                                 int etag, int lane, [[maybe_unused]] bool broken_seen) {
#ifdef MRF_PEER_HEAVY_FENCE  // round 5's form: a system-scope fence (L2 write-back) and a system-scope release on top
  __threadfence_system();
  __syncthreads();
  const bool broken = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == etag;
  if (lane < V.G && lane != V.grank && !broken)
    __hip_atomic_store(peer_flag(V, lane, gen, V.grank, blk), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#else
  // The payload went out as system-scope (write-through) stores, one instruction per scalar for the whole wave; they are
  // PERFORMED once the wave's store counter has drained, which is what an agent-scope release waits for -- without the L2
  // write-back a system-scope fence adds (nothing of the payload sits dirty in L2).  The flag then follows as a relaxed
  // system-scope store.  Measured with three ranks in one process (tools/shard_local.py, round 6): see DESIGN.md section 6.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __syncthreads();
  if (lane < V.G && lane != V.grank && !broken_seen)
    __hip_atomic_store(peer_flag(V, lane, gen, V.grank, blk), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

// wait for the same workgroup of every other rank (bounded: a missing peer must not hang the GPU)
__device__ __forceinline__ void peer_wait_flags(const PeerView& V, int gen, int blk, unsigned long long seq, const int* err,
                                                int etag, int lane, bool& broken_seen) {
  bool gave_up = false;
  if (lane < V.G && lane != V.grank) {
    const unsigned long long* f = peer_flag(V, V.grank, gen, lane, blk);
    const long long t0 = wall_clock64();
    // Poll with a back-off: every poll is an uncached system-scope load, and a thousand workgroups polling every ~64 cycles
    // keep the memory fabric busy enough to delay the very stores they wait for (measured with three ranks on one die,
    // round 6: 5.2 -> see DESIGN.md section 6) -- and to starve other kernels on the device.  0.4 us between the first polls,
    // doubling to 3.4 us.
    int naps = 1;
#ifdef MRF_PEER_HEAVY_FENCE
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
#else
    // relaxed polls (an acquire per poll invalidates caches a million times per rollout); the payload is read by
    // system-scope loads after the barrier below
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
#endif
#ifndef MRF_PEER_NO_BACKOFF
      for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(16);  // 16 x 64 cycles
      if (naps < 8) naps *= 2;
#endif
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == etag) {
        gave_up = true;
        break;
      }
      if (wall_clock64() - t0 > V.timeout_ticks) {
        gave_up = true;
        // post-mortem of the FIRST wait that ran out on this rank (mrf_comm_status prints it under MRF_PEER_DEBUG): which
        // block, which peer, which sequence number was expected and what the flag held
        {
          int* dbg = reinterpret_cast<int*>(V.base[V.grank] + V.off_err);
          if (atomicCAS(dbg + 1, 0, 1) == 0) {
            const unsigned long long have = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            dbg[2] = blk;
            dbg[3] = lane;
            dbg[4] = (int)(seq & 0x7fffffff);
            dbg[5] = (int)(have & 0x7fffffff);
            dbg[6] = (int)blockIdx.x;
            dbg[7] = (int)gridDim.x;
#ifdef MRF_PEER_HEARTBEAT
            const int* pd = reinterpret_cast<const int*>(V.base[lane] + V.off_err);  // the awaited peer's heartbeat, right now
            dbg[10] = __hip_atomic_load(pd + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            dbg[11] = __hip_atomic_load(pd + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            dbg[13] = __hip_atomic_load(pd + 12, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            dbg[14] = __hip_atomic_load(pd + 15, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
          }
        }
        // the timeout is the GROUP's: raise the error word of every rank, so that a peer which went on with this
        // rank's (now missing) payload cannot return a finite result either
        for (int g = 0; g < V.G; ++g)
          __hip_atomic_store(reinterpret_cast<int*>(V.base[g] + V.off_err), etag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
#ifdef MRF_PEER_NO_BACKOFF
      __builtin_amdgcn_s_sleep(1);
#endif
    }
  }
  if (__any(gave_up)) broken_seen = true;
  __syncthreads();
#ifdef MRF_PEER_HEAVY_FENCE
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // every lane reads the peers' payload after the flags
#else
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // ... by system-scope loads, which no cache of this device answers
#endif
}

// Development aid (-DMRF_PEER_TIMING, tools/peer_timing.py): where a wave of the persistent kernels spends its time -- publish
// (payload stores, release, flags), the wait for the peers' flags, the remote fold (payload loads, re-walked chains), the
// whole block -- in wall_clock64 ticks, summed over the waves of a launch into the words behind the post-mortem area.
#ifdef MRF_PEER_TIMING
#define MRF_TM(var, ...)                  \
  {                                       \
    const long long tm0_ = wall_clock64(); \
    __VA_ARGS__;                          \
    var += wall_clock64() - tm0_;         \
  }
#define MRF_TM_BEGIN() const long long tmb_ = wall_clock64()
#define MRF_TM_END(var) var += wall_clock64() - tmb_
#else
#define MRF_TM(var, ...) \
  { __VA_ARGS__; }
#define MRF_TM_BEGIN()
#define MRF_TM_END(var)
#endif

// q_in / qd_in are only read; the advanced state and the velocity signal go to the STAGING arrays q_st / qd_st / avg_st
// (owned by the communicator) and are committed to the caller's arrays by k_peer_commit after the whole grid has
// finished -- from ONE reading of the error word, so that a timed-out exchange leaves every row where it was.
// XK: what the robots of OTHER ranks send (mrf_shard.hpp); XK_NONE = a group of one rank, which then runs the fused
// kernel's step (all robots on chip, single chain walk) inside this kernel's persistent block loop.
template <typename T, class LS, bool LO, int XK>
__global__ __launch_bounds__(64) void k_rollout_peer(const DevCfg<T>* __restrict__ cfgp, PeerView V, int64_t n_scen,
                                                      const T* __restrict__ q_in, const T* __restrict__ qd_in,
                                                      const T* __restrict__ prm, T* __restrict__ q_st,
                                                      T* __restrict__ qd_st, T* __restrict__ avg_st,
                                                      unsigned long long seq0) {
  __shared__ T xch[LO ? TILE_SCALARS : GEN_SCALARS];
  constexpr bool REMOTE = XK != XK_NONE;
  const DevCfg<T>& cfg = *cfgp;
  const int N = cfg.n_robots;
  const int first = V.first[V.grank], count = V.first[V.grank + 1] - first;
  int cnt_max = 1;
  for (int g = 0; g < V.G; ++g) cnt_max = max(cnt_max, V.first[g + 1] - V.first[g]);
  const int spw = 64 / cnt_max;  // scenarios per block: the same on every rank, so block X is the same scenarios
  const int lane = threadIdx.x;
  const int nblk = (int)((n_scen + spw - 1) / spw);
  if constexpr (LO) stage_sphere_radii(cfg, xch, lane);  // visible after the first barrier; never overwritten
  [[maybe_unused]] long long tm_pub = 0, tm_wait = 0, tm_remote = 0, tm_all = 0, tm_stage = 0;
  [[maybe_unused]] const long long tm_start = wall_clock64();
#ifdef MRF_PEER_HEARTBEAT
  if (lane == 0) {  // development aid: how many workgroups of which launch have started (read by mrf_comm_status's post-mortem)
    int* dbg = reinterpret_cast<int*>(V.base[V.grank] + V.off_err);
    const int tag = (int)(seq0 & 0x7fffffff);
    if (atomicExch(dbg + 8, tag) != tag) {
      atomicExch(dbg + 9, 0);
      atomicExch(dbg + 12, 0);
    }
    atomicAdd(dbg + 9, 1);
  }
#endif
  // The grid is capped at what is resident at once (host side); a workgroup then walks blocks blockIdx.x,
  // blockIdx.x + gridDim.x, ... in increasing order.  Block X only ever waits for block X of the peers, every workgroup
  // of every rank is resident and visits its blocks in increasing index order, so the wait graph has no cycle whatever
  // the dispatch order or the grid size of the other ranks.
#pragma unroll 1
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
  int ls = lane / count;
  const int l0 = lane - ls * count;
  int64_t scen = (int64_t)blk * spw + ls;
  const bool active = ls < spw && scen < n_scen;
  if (!active) {  // idle lanes shadow the block's first row (no stores)
    ls = 0;
    scen = (int64_t)blk * spw;
  }
  const int l = active ? l0 : 0;
  const int me = first + l;
  const int64_t rows = n_scen * count;
  const int64_t row = scen * count + l;
  const int m01 = LO ? cfg.lo_merge01 : 0, m45 = LO ? cfg.lo_merge45 : 0;
  int* err = reinterpret_cast<int*>(V.base[V.grank] + V.off_err);
  // The error word is tagged with the reset epoch (the high bits of every sequence number, mrf_comm_reset): "broken" means
  // "holds THIS epoch's tag", so a kernel of the previous epoch that times out late -- after a peer's reset has already
  // started the next epoch -- cannot break the new sequence with its store.
  const int etag = (int)(seq0 >> 40) + 1;
  // without collision leaves (the grasp planner) nobody reads anybody's spheres: no payload, no flags, on every rank
  const bool exchanging = REMOTE && V.G > 1 && cfg.n_ego > 0;
  bool broken_seen = exchanging && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == etag;

  PandaState<T> R;
  load_state(rows, row, q_in, qd_in, R);
  const T* mount_own = cfg.mount[me];
  PrmView<T> P{prm, rows, row, {T(0), T(0), T(0)}, false};
  if ((cfg.goal_mask >> me) & 1) {  // RF-CV goal estimate (EXC:355-357), as in k_rollout_panda
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
    const unsigned long long seq = seq0 + (unsigned long long)k;
    const int gen = (int)(seq & 1ull);
    // system_step 'vel' (FPJ:77-80)
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
    if constexpr (XK == XK_JOINTS_TAGGED) {
      // ---- publish: tagged words, no drain, no flag (see tagged_store)
      if (exchanging && active && !broken_seen) {
        const unsigned tag = tagged_tag(seq);
        for (int g = 0; g < V.G; ++g) {
          if (g == V.grank) continue;
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            tagged_store<T>(V, g, gen, N, me, 3 * j + 0, scen, R.cq[j], tag);
            tagged_store<T>(V, g, gen, N, me, 3 * j + 1, scen, R.sq[j], tag);
            tagged_store<T>(V, g, gen, N, me, 3 * j + 2, scen, R.qd[j], tag);
          }
        }
      }
    }
    if constexpr (XK == XK_JOINTS) {
      // ---- publish: the joint state of step k goes out BEFORE the own chain walk, so that the flag round trip runs
      // under the walk and the local fold (FPJ:211-225 across GPUs; the receivers re-walk this chain)
      if (exchanging) {
        MRF_TM_BEGIN();
        if (active) {
          for (int g = 0; g < V.G; ++g) {
            if (g == V.grank) continue;
            T* dst = peer_x<T>(V, g, gen, N) + ((size_t)me * MRF_JOINT_STATE_SCALARS) * V.b_max + scen;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
              xstore(dst + (size_t)(3 * j + 0) * V.b_max, R.cq[j]);
              xstore(dst + (size_t)(3 * j + 1) * V.b_max, R.sq[j]);
              xstore(dst + (size_t)(3 * j + 2) * V.b_max, R.qd[j]);
            }
          }
        }
        peer_raise_flags(V, gen, blk, seq, err, etag, lane, broken_seen);
        MRF_TM_END(tm_pub);
      }
    }
    const T* xloc = peer_x<T>(V, V.grank, gen, N);
    T qdd[7], act[7];
    sharded_solve_row<LS, LO, REMOTE>(
        cfg, xch, lane, ls, l, count, mount_own, R, P,
        [&](const PandaKin<T>& K1) {
          if constexpr (XK == XK_SPHERES) {
            // ---- publish: this robot's spheres of step k into every other rank's buffer
            if (!exchanging) return;
            const int SX = cfg.n_spheres - m01 - m45;
            if (active) {
              for (int g = 0; g < V.G; ++g) {
                if (g == V.grank) continue;
                T* dst = peer_x<T>(V, g, gen, N) + ((size_t)me * SX * 9) * V.b_max + scen;
                if constexpr (LO) {
#pragma unroll
                  for (int sp = 0; sp < 8; ++sp) {
                    if ((sp == 1 && m01) || (sp == 5 && m45)) continue;  // coincident link origins travel once
                    T* d9 = dst + ((size_t)lo_slot(sp, m01, m45) * 9) * V.b_max;
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                      xstore(d9 + (size_t)c * V.b_max, sp < 7 ? K1.o[sp < 7 ? sp : 0][c] : K1.p8[c]);
                      xstore(d9 + (size_t)(3 + c) * V.b_max, sp < 7 ? K1.vo[sp < 7 ? sp : 0][c] : K1.v8[c]);
                      xstore(d9 + (size_t)(6 + c) * V.b_max, cfg.jsign * (sp < 7 ? K1.ao[sp < 7 ? sp : 0][c] : K1.a8[c]));
                    }
                  }
                } else {
                  panda_walk_spheres<false, T>(
                      cfg, mount_own,
                      [&](int j, T& c, T& s, T& qdj) {
                        c = xch[(3 * j + 0) * 64 + lane];
                        s = xch[(3 * j + 1) * 64 + lane];
                        qdj = xch[(3 * j + 2) * 64 + lane];
                      },
                      [&](int s, const T* x, const T* v, const T* a) {
                        T* d9 = dst + ((size_t)s * 9) * V.b_max;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                          xstore(d9 + (size_t)c * V.b_max, x[c]);
                          xstore(d9 + (size_t)(3 + c) * V.b_max, v[c]);
                          xstore(d9 + (size_t)(6 + c) * V.b_max, cfg.jsign * a[c]);
                        }
                      });
                }
              }
            }
            peer_raise_flags(V, gen, blk, seq, err, etag, lane, broken_seen);
          }
        },
        [&]() {
          if constexpr (XK != XK_JOINTS_TAGGED)
            if (exchanging) MRF_TM(tm_wait, peer_wait_flags(V, gen, blk, seq, err, etag, lane, broken_seen))
        },
        [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
          if constexpr (XK == XK_JOINTS) {
            MRF_TM(tm_remote, remote_obstacles_joints<typename LS::Collision, LO>(
                cfg, xch, lane, first, count, N,
                [&](int jr, T (&v)[MRF_JOINT_STATE_SCALARS]) { xload_robot(xloc, jr, V.b_max, scen, v); }, E, acc, &tm_stage))
          } else if constexpr (XK == XK_JOINTS_TAGGED) {
            MRF_TM(tm_remote, remote_obstacles_joints<typename LS::Collision, LO, false>(
                cfg, xch, lane, first, count, N,
                [&](int jr, T (&v)[MRF_JOINT_STATE_SCALARS]) {
                  tagged_load_robot<T>(V, gen, N, jr, scen, active, seq, blk, err, etag, lane, broken_seen, v);
                },
                E, acc, &tm_stage))
          } else if constexpr (XK == XK_SPHERES) {
            const int SX = cfg.n_spheres - m01 - m45;
            MRF_TM(tm_remote, remote_obstacles_spheres<typename LS::Collision, LO>(
                cfg, first, count, N,
                [&](int jr, int slot, int c) { return xload(xloc + ((size_t)(jr * SX + slot) * 9 + c) * V.b_max + scen); }, E, acc))
          }
        },
        qdd, act);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      R.qd[j] = act[j];  // FPJ:233
      sumsq += act[j] * act[j];
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      q_st[j * rows + row] = R.q[j];
      qd_st[j * rows + row] = R.qd[j];
    }
    avg_st[row] = sumsq / (T)(H * 7);  // FPJ:102-116
  }
  __syncthreads();  // xch is rewritten by the next block of this workgroup
  }
#ifdef MRF_PEER_HEARTBEAT
  if (lane == 0) atomicAdd(reinterpret_cast<int*>(V.base[V.grank] + V.off_err) + 12, 1);  // workgroups that have left
#endif
#ifdef MRF_PEER_TIMING
  if (lane == 0) {
    unsigned long long* tm = reinterpret_cast<unsigned long long*>(V.base[V.grank] + V.off_err + 128);
    tm_all = wall_clock64() - tm_start;
    atomicAdd(tm + 0, (unsigned long long)tm_pub);
    atomicAdd(tm + 1, (unsigned long long)tm_wait);
    atomicAdd(tm + 2, (unsigned long long)tm_remote);
    atomicAdd(tm + 3, (unsigned long long)tm_all);
    atomicAdd(tm + 4, 1ull);
    atomicAdd(tm + 5, (unsigned long long)tm_stage);
  }
#endif
}

// k_rollout_peer_paired (round 6; OPT-IN, MRF_PEER_PAIRED=1: a parity-green experiment that does not pay with all ranks on one
// die -- see the host side and DESIGN.md section 6): the joint payload's kernel for batches of more than one block per
// workgroup.  A workgroup works on the two ADJACENT blocks 2u and 2u + 1 in turns -- step k of the first, step k of the second, step k + 1 of the
// first ... -- and publishes the joint state of step k + 1 at the END of step k (it is known as soon as the action is).  The round trip of a block's exchange -- the
// write-through of the payload, the flag, the peers' polls, over xGMI between GPUs -- then runs under the whole step of the
// OTHER block instead of stalling the only wave of the SIMD.  The state of the block that is not being worked on (q, cos q,
// sin q, qdot, the velocity sum, the goal estimate: 32 scalars per lane) sits in `swap`, an L2-resident 16 KB per workgroup,
// and is exchanged in place at each turn (the LDS is taken by the exchange tile: 37 of the 40 KB a wave has at four waves
// per CU).  The pairing is by block index, not by grid size, and every rank runs the two blocks of a pair in the same
// order, so the wait graph stays acyclic whatever the ranks' grid sizes: the lowest unfinished pair is current on every rank.
constexpr int PEER_SWAP_SCALARS = 32;
template <typename T, class LS, bool LO>
__global__ __launch_bounds__(64) void k_rollout_peer_paired(const DevCfg<T>* __restrict__ cfgp, PeerView V, int64_t n_scen,
                                                      const T* __restrict__ q_in, const T* __restrict__ qd_in,
                                                      const T* __restrict__ prm, T* __restrict__ q_st,
                                                      T* __restrict__ qd_st, T* __restrict__ avg_st, T* swap,
                                                      unsigned long long seq0) {
  __shared__ T xch[LO ? TILE_SCALARS : GEN_SCALARS];
  const DevCfg<T>& cfg = *cfgp;
  const int N = cfg.n_robots;
  const int first = V.first[V.grank], count = V.first[V.grank + 1] - first;
  int cnt_max = 1;
  for (int g = 0; g < V.G; ++g) cnt_max = max(cnt_max, V.first[g + 1] - V.first[g]);
  const int spw = 64 / cnt_max;  // scenarios per block: the same on every rank, so block X is the same scenarios
  const int lane = threadIdx.x;
  const int nblk = (int)((n_scen + spw - 1) / spw);
  if constexpr (LO) stage_sphere_radii(cfg, xch, lane);  // visible after the first barrier; never overwritten
  [[maybe_unused]] long long tm_pub = 0, tm_wait = 0, tm_remote = 0, tm_all = 0, tm_stage = 0;
  [[maybe_unused]] const long long tm_start = wall_clock64();
#ifdef MRF_PEER_HEARTBEAT
  if (lane == 0) {  // development aid: how many workgroups of which launch have started (read by mrf_comm_status's post-mortem)
    int* dbg = reinterpret_cast<int*>(V.base[V.grank] + V.off_err);
    const int tag = (int)(seq0 & 0x7fffffff);
    if (atomicExch(dbg + 8, tag) != tag) {
      atomicExch(dbg + 9, 0);
      atomicExch(dbg + 12, 0);
    }
    atomicAdd(dbg + 9, 1);
  }
#endif
  const int64_t rows = n_scen * count;
  int* err = reinterpret_cast<int*>(V.base[V.grank] + V.off_err);
  // The error word is tagged with the reset epoch (the high bits of every sequence number, mrf_comm_reset): "broken" means
  // "holds THIS epoch's tag", so a kernel of the previous epoch that times out late -- after a peer's reset has already
  // started the next epoch -- cannot break the new sequence with its store.
  const int etag = (int)(seq0 >> 40) + 1;
  // without collision leaves (the grasp planner) nobody reads anybody's spheres: no payload, no flags, on every rank
  const bool exchanging = V.G > 1 && cfg.n_ego > 0;
  const int H = cfg.horizon;
  T* sw = swap + (size_t)blockIdx.x * PEER_SWAP_SCALARS * 64 + lane;
  // The grid is capped at what is resident at once (host side); a workgroup then walks units blockIdx.x,
  // blockIdx.x + gridDim.x, ... in increasing order (a unit: one block, or a pair of adjacent blocks).  Block X only ever
  // waits for block X of the peers, every workgroup of every rank is resident and visits its units in increasing index
  // order, so the wait graph has no cycle whatever the dispatch order or the grid size of the other ranks.
  const int nunit = (nblk + 1) / 2;
#pragma unroll 1
  for (int unit = blockIdx.x; unit < nunit; unit += gridDim.x) {
  const int nctx = 2 * unit + 1 < nblk ? 2 : 1;  // an odd block count leaves the last pair with one block
  bool broken_seen = exchanging && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == etag;
  // ---- the block being worked on: lane -> (scenario, owned robot)
  int blk = 0, ls = 0, l = 0, me = first;
  int64_t scen = 0, row = 0;
  bool active = false;
  auto map_block = [&](int b) __attribute__((always_inline)) {
    blk = b;
    ls = lane / count;
    const int l0 = lane - ls * count;
    scen = (int64_t)blk * spw + ls;
    active = ls < spw && scen < n_scen;
    if (!active) {  // idle lanes shadow the block's first row (no stores)
      ls = 0;
      scen = (int64_t)blk * spw;
    }
    l = active ? l0 : 0;
    me = first + l;
    row = scen * count + l;
  };
  PandaState<T> R;
  T g0[3] = {T(0), T(0), T(0)};
  T sumsq = T(0);
  // system_step 'vel' (FPJ:77-80)
  auto integrate = [&]() __attribute__((always_inline)) {
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
  };
  // ---- publish: the joint state a step starts from goes out as soon as it is known -- before the own chain
  // walk of that step (FPJ:211-225 across GPUs; the receivers re-walk this chain)
  auto publish_joints = [&](unsigned long long seq) __attribute__((always_inline)) {
    if (!exchanging) return;
    const int gen = (int)(seq & 1ull);
    MRF_TM_BEGIN();
    if (active) {
      for (int g = 0; g < V.G; ++g) {
        if (g == V.grank) continue;
        T* dst = peer_x<T>(V, g, gen, N) + ((size_t)me * MRF_JOINT_STATE_SCALARS) * V.b_max + scen;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          xstore(dst + (size_t)(3 * j + 0) * V.b_max, R.cq[j]);
          xstore(dst + (size_t)(3 * j + 1) * V.b_max, R.sq[j]);
          xstore(dst + (size_t)(3 * j + 2) * V.b_max, R.qd[j]);
        }
      }
    }
    peer_raise_flags(V, gen, blk, seq, err, etag, lane, broken_seen);
    MRF_TM_END(tm_pub);
  };
  // the state of the block that is not being worked on <-> registers, in place (same lane, same address: in order)
  auto swap_state = [&]() __attribute__((always_inline)) {
    auto x = [&](int i, T& v) __attribute__((always_inline)) {
      const T t = sw[i * 64];
      sw[i * 64] = v;
      v = t;
    };
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      x(j, R.q[j]);
      x(7 + j, R.qd[j]);
      x(14 + j, R.cq[j]);
      x(21 + j, R.sq[j]);
    }
    x(28, sumsq);
#pragma unroll
    for (int c = 0; c < 3; ++c) x(29 + c, g0[c]);
  };
  // ---- prologue of every block of the unit: start state, RF-CV goal estimate, the position update of step 0, its publish
#pragma unroll 1
  for (int w = 0; w < nctx; ++w) {
    if (w == 1) swap_state();  // the first block's state goes to the swap area (what comes back is not used)
    map_block(2 * unit + w);
    load_state(rows, row, q_in, qd_in, R);
#pragma unroll
    for (int c = 0; c < 3; ++c) g0[c] = T(0);
    if ((cfg.goal_mask >> me) & 1) {  // RF-CV goal estimate (EXC:355-357), as in k_rollout_panda
      PandaKin<T> K0;
      panda_walk_own<T>(cfg.mount[me], R.cq, R.sq, R.qd, K0);
#pragma unroll
      for (int c = 0; c < 3; ++c) g0[c] = K0.p8[c] + cfg.goal_T * K0.v8[c];
    }
    sumsq = T(0);
    integrate();
    publish_joints(seq0);
  }
#pragma unroll 1
  for (int it = 0; it < nctx * H; ++it) {
    int k = it;
    if (nctx == 2) {
      k = it >> 1;
      swap_state();
      map_block(2 * unit + (it & 1));
    }
    const unsigned long long seq = seq0 + (unsigned long long)k;
    const int gen = (int)(seq & 1ull);
    const T* mount_own = cfg.mount[me];
    PrmView<T> P{prm, rows, row, {g0[0], g0[1], g0[2]}, ((cfg.goal_mask >> me) & 1) != 0};
    const T* xloc = peer_x<T>(V, V.grank, gen, N);
    T qdd[7], act[7];
    sharded_solve_row<LS, LO, true>(
        cfg, xch, lane, ls, l, count, mount_own, R, P,
        [&](const PandaKin<T>&) {},
        [&]() {
          if (exchanging) MRF_TM(tm_wait, peer_wait_flags(V, gen, blk, seq, err, etag, lane, broken_seen))
        },
        [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
          MRF_TM(tm_remote, remote_obstacles_joints<typename LS::Collision, LO>(
              cfg, xch, lane, first, count, N,
              [&](int jr, T (&v)[MRF_JOINT_STATE_SCALARS]) { xload_robot(xloc, jr, V.b_max, scen, v); }, E, acc, &tm_stage))
        },
        qdd, act);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      R.qd[j] = act[j];  // FPJ:233
      sumsq += act[j] * act[j];
    }
    if (k + 1 < H) {
      integrate();  // the position update of step k + 1 ...
      publish_joints(seq + 1ull);  // ... and its joint state, on its way while the rest of the unit is worked on
    } else if (active) {
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        q_st[j * rows + row] = R.q[j];
        qd_st[j * rows + row] = R.qd[j];
      }
      avg_st[row] = sumsq / (T)(H * 7);  // FPJ:102-116
    }
  }
  __syncthreads();  // xch is rewritten by the next unit of this workgroup
  }
#ifdef MRF_PEER_HEARTBEAT
  if (lane == 0) atomicAdd(reinterpret_cast<int*>(V.base[V.grank] + V.off_err) + 12, 1);  // workgroups that have left
#endif
#ifdef MRF_PEER_TIMING
  if (lane == 0) {
    unsigned long long* tm = reinterpret_cast<unsigned long long*>(V.base[V.grank] + V.off_err + 128);
    tm_all = wall_clock64() - tm_start;
    atomicAdd(tm + 0, (unsigned long long)tm_pub);
    atomicAdd(tm + 1, (unsigned long long)tm_wait);
    atomicAdd(tm + 2, (unsigned long long)tm_remote);
    atomicAdd(tm + 3, (unsigned long long)tm_all);
    atomicAdd(tm + 4, 1ull);
    atomicAdd(tm + 5, (unsigned long long)tm_stage);
  }
#endif
}

// Residency roll call (mrf_comm_peer_connect): single-wave workgroups with the footprint of k_rollout_peer -- the whole
// register file of a SIMD lane (512), the link-origin exchange tile of LDS -- count themselves in and wait, bounded, until the
// whole grid has; a workgroup that gives up says so.  The persistent kernel's grid is capped at a size for which nobody gave
// up: the cap is MEASURED on the device the communicator lives on, not taken from the occupancy API alone.
__global__ __launch_bounds__(64) void k_peer_roll_call(unsigned* __restrict__ count, unsigned* __restrict__ gave_up,
                                                        long long timeout_ticks) {
  __shared__ double lds[TILE_SCALARS];
  lds[threadIdx.x] = (double)blockIdx.x;
  asm volatile("v_mov_b32 v255, 0" ::: "v255");            // the kernel holds the architected ...
  asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");  // ... and the accumulation half of the register file
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(count, 1u);
    const long long t0 = wall_clock64();
    bool all_here = false;
    do {
      all_here = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x;
      if (all_here) break;
      __builtin_amdgcn_s_sleep(8);
    } while (wall_clock64() - t0 < timeout_ticks);
    if (!all_here) atomicAdd(gave_up, 1u);
    if (lds[1] < 0.0) count[2] = 1u;  // keeps the tile allocated
  }
}

// Do kernels on two streams of one process really run side by side?  (HIP maps streams onto a few hardware queues; two
// streams on one queue run their kernels one after the other.)  k_pair_wait spins, bounded, on a word that k_pair_set, queued
// AFTER it on the other stream, writes.
__global__ void k_pair_wait(unsigned* __restrict__ flag, unsigned* __restrict__ seen, long long timeout_ticks) {
  const long long t0 = wall_clock64();
  unsigned v = 0;
  do {
    v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v) break;
    __builtin_amdgcn_s_sleep(8);
  } while (wall_clock64() - t0 < timeout_ticks);
  *seen = v;
}
__global__ void k_pair_set(unsigned* __restrict__ flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// After k_rollout_peer (same stream): one thread latches the error word, then every row is committed from that latch --
// all rows advance, or (a timed-out exchange: some step folded stale payload) none does and the velocity signal is NaN,
// so that a caller that forgets mrf_comm_status cannot take the result for a rollout.
__global__ void k_peer_latch(const int* __restrict__ err, int* __restrict__ latch, int group, int etag) {
  *latch = group > 1 ? (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == etag) : 0;
#ifdef MRF_PEER_HEARTBEAT
  atomicAdd(const_cast<int*>(err) + 15, 1);  // latch passes that have run (development aid)
#endif
}
template <typename T>
__global__ __launch_bounds__(256) void k_peer_commit(int64_t rows, const int* __restrict__ latch, const T* __restrict__ q_st,
                                                      const T* __restrict__ qd_st, const T* __restrict__ avg_st,
                                                      T* __restrict__ q_io, T* __restrict__ qd_io, T* __restrict__ avg_out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  if (*latch != 0) {
    avg_out[r] = quiet_nan<T>();
    return;
  }
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    q_io[j * rows + r] = q_st[j * rows + r];
    qd_io[j * rows + r] = qd_st[j * rows + r];
  }
  avg_out[r] = avg_st[r];
}

// RF-CV goal estimate for the step-kernel path (the persistent kernels do it in their prologue): params copied to a
// work array with x_goal_0 := x_ee + T * v_ee for the owned robots in cfg.goal_estimate_mask (EXC:355-357)
template <typename T>
__global__ __launch_bounds__(64) void k_goal_estimate(const DevCfg<T>* __restrict__ cfgp, int64_t n_scen, int first,
                                                       int count, const T* __restrict__ q, const T* __restrict__ qd,
                                                       const T* __restrict__ prm_in, T* __restrict__ prm_out) {
  const DevCfg<T>& cfg = *cfgp;
  const int64_t rows = n_scen * count;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int me = first + (int)(r % count);
#pragma unroll 1
  for (int c = 0; c < MRF_NPARAM; ++c) prm_out[c * rows + r] = prm_in[c * rows + r];
  if ((cfg.goal_mask >> me) & 1) {
    PandaState<T> R;
    load_state(rows, r, q, qd, R);
    PandaKin<T> K;
    panda_walk_own<T>(cfg.mount[me], R.cq, R.sq, R.qd, K);
#pragma unroll
    for (int c = 0; c < 3; ++c) prm_out[(MRF_P_X_GOAL_0 + c) * rows + r] = K.p8[c] + cfg.goal_T * K.v8[c];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_avg_from_sumsq(int64_t rows, const T* __restrict__ sumsq, T scale,
                                                         T* __restrict__ avg) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) avg[r] = sumsq[r] * scale;
}

}  // namespace mrf

// ================================================================================ host side
namespace {
using mrf_host::check_hip;
using mrf_host::dispatch;
using mrf_host::dispatch_scalar;
using mrf_host::fail;
using mrf_host::is_link_origin_table;
using mrf_host::launch;

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi& rccl() {
  static RcclApi api;
  if (api.lib || !api.error.empty()) return api;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names)
    if ((api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!api.lib) {
    api.error = std::string("dlopen(librccl) failed: ") + dlerror();
    return api;
  }
  auto sym = [&](const char* n) {
    void* p = dlsym(api.lib, n);
    if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + n;
    return p;
  };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
  api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
  api.CommUserRank = (decltype(api.CommUserRank))sym("ncclCommUserRank");
  api.CommCuDevice = (decltype(api.CommCuDevice))sym("ncclCommCuDevice");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  if (!api.error.empty()) api.lib = nullptr;
  return api;
}

struct Comm {
  int transport = MRF_TRANSPORT_NONE;
  int rank = 0, world = 1;
  int first[MRF_MAX_ROBOTS + 1] = {0};  // robot blocks
  int cnt_max = 1;
  int xs = MRF_JOINT_STATE_SCALARS;  // scalars one robot sends per scenario and step (cfg.exchange at creation)
  // RCCL
  ncclComm_t nccl = nullptr;
  int nccl_count = 0, nccl_rank = -1, nccl_device = -1;  // what the communicator itself reports (mrf_comm_info)
  void* sph_own = nullptr;  // [cnt_max][xs][B]           xs = SX*9 sphere scalars or the 21 joint-state scalars
  void* sph_pad = nullptr;  // [world][cnt_max][xs][B]
  void* sumsq = nullptr;    // [B*count]
  void* prm_work = nullptr; // [MRF_NPARAM][B*count]: params with the RF-CV goal estimate applied
  int64_t cap_scen = 0;
  // PEER
  unsigned char* local = nullptr;
  unsigned char* peer[MRF_MAX_ROBOTS] = {nullptr};
  bool connected = false;
  bool in_process = false;  // the peers' buffers are plain device pointers of handles in THIS process (mrf_comm_peer_connect_local)
  int64_t b_max = 0;
  int nblk_max = 0;
  size_t off_flags = 0, off_err = 0, off_x = 0, bytes = 0;
  void* stage = nullptr;  // peer kernel outputs before the commit: q [7][rows], qdot [7][rows], avg [rows], latch
  void* swap = nullptr;   // paired blocks (joint payload): the resting block's state, [swap_wgs][32][64] scalars
  unsigned swap_wgs = 0;
  bool last_paired = false;  // the last rollout walked its blocks two at a time (MRF_PEER_DEBUG prints it)
  bool tagged = false;       // MRF_PEER_TAGGED=1 at mrf_comm_peer_open: tagged payload words (XK_JOINTS_TAGGED), twice the bytes
  unsigned grid_cap = 0;   // co-resident workgroups measured by the roll call at connect (0: not measured -- a group of one,
                           // or ranks sharing a device in tests)
  unsigned long long epoch = 0;  // mrf_comm_reset count: the high bits of every sequence number
  unsigned long long seq = 1;
  hipStream_t last_stream = nullptr;
};

void partition(Comm& c, int n_robots) {  // contiguous blocks, sizes differ by at most one (as sharded.robot_partition)
  const int base = n_robots / c.world, extra = n_robots % c.world;
  c.first[0] = 0;
  c.cnt_max = 1;
  for (int g = 0; g < c.world; ++g) {
    const int cnt = base + (g < extra ? 1 : 0);
    c.first[g + 1] = c.first[g] + cnt;
    if (cnt > c.cnt_max) c.cnt_max = cnt;
  }
}

int check_group(mrf_handle* h, int rank, int world) {
  if (h->cfg.model != MRF_MODEL_PANDA7 || h->cfg.mode != MRF_MODE_VEL)
    return fail(h, MRF_E_CONFIG, "sharded rollout needs the panda7 model in mode 'vel'");
  if (world < 1 || world > h->cfg.n_robots || world > MRF_MAX_ROBOTS)
    return fail(h, MRF_E_ARG, "world must be in 1..n_robots (further GPUs replicate the group over scenario batches)");
  if (rank < 0 || rank >= world) return fail(h, MRF_E_ARG, "rank out of range");
  if (h->comm) return fail(h, MRF_E_ARG, "the handle already has a communicator (mrf_comm_destroy first)");
  if (h->cfg.exchange != MRF_EXCHANGE_JOINTS && h->cfg.exchange != MRF_EXCHANGE_SPHERES)
    return fail(h, MRF_E_CONFIG, "cfg.exchange must be MRF_EXCHANGE_JOINTS or MRF_EXCHANGE_SPHERES");
  return MRF_OK;
}

size_t scalar_bytes(const mrf_handle* h) { return h->cfg.scalar == MRF_F64 ? 8 : 4; }

int exchange_scalars(const mrf_handle* h) { return mrf_exchange_scalars(h); }
bool joints_exchange(const mrf_handle* h) { return h->cfg.exchange == MRF_EXCHANGE_JOINTS; }

// Largest grid of k_peer_roll_call workgroups (<= want) that is co-resident on the handle's device: tried at `want`, then in
// steps of an eighth less.  ~0.1 ms when the first try holds (a device of its own: tools/residency_probe.hip).
unsigned measured_coresidency(mrf_handle* h, unsigned want) {
  unsigned* d = nullptr;
  if (want < 1u || hipMalloc((void**)&d, 16) != hipSuccess) return 0;
  int rate_khz = 100000;
  (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, h->device);
  unsigned ok = 0;
  for (unsigned g = want; g >= 1u; g = g * 7u / 8u) {
    unsigned r[4] = {0, 0, 0, 0};
    if (hipMemset(d, 0, 16) != hipSuccess) break;
    hipLaunchKernelGGL(mrf::k_peer_roll_call, dim3(g), dim3(64), 0, nullptr, d, d + 1, (long long)rate_khz * 20);  // 20 ms
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(r, d, 16, hipMemcpyDeviceToHost) != hipSuccess) break;
    if (r[1] == 0u && r[0] == g) {
      ok = g;
      break;
    }
    if (g == 1u) break;
  }
  (void)hipFree(d);
  return ok;
}

int ensure_rccl_buffers(mrf_handle* h, Comm* c, int64_t n_scen) {
  if (n_scen <= c->cap_scen) return MRF_OK;
  if (c->sph_own && c->sph_own != c->sph_pad) (void)hipFree(c->sph_own);
  if (c->sph_pad) (void)hipFree(c->sph_pad);
  if (c->sumsq) (void)hipFree(c->sumsq);
  if (c->prm_work) (void)hipFree(c->prm_work);
  c->sph_own = c->sph_pad = c->sumsq = c->prm_work = nullptr;
  c->cap_scen = 0;
  const size_t blk = (size_t)c->cnt_max * c->xs * n_scen * scalar_bytes(h);
  hipError_t e = hipMalloc(&c->sph_pad, blk * c->world);
  if (e == hipSuccess) e = hipMemset(c->sph_pad, 0, blk * c->world);
  if (e == hipSuccess && c->world > 1) e = hipMalloc(&c->sph_own, blk);
  if (e == hipSuccess && c->world > 1) e = hipMemset(c->sph_own, 0, blk);
  if (c->world == 1) c->sph_own = c->sph_pad;
  if (e == hipSuccess) e = hipMalloc(&c->sumsq, (size_t)n_scen * c->cnt_max * scalar_bytes(h));
  if (e == hipSuccess) e = hipMalloc(&c->prm_work, (size_t)MRF_NPARAM * n_scen * c->cnt_max * scalar_bytes(h));
  if (e != hipSuccess) return fail(h, MRF_E_DEVICE, std::string("exchange buffers: ") + hipGetErrorString(e));
  c->cap_scen = n_scen;
  return MRF_OK;
}

}  // namespace

void mrf_host::comm_release(mrf_handle* h) {
  if (!h || !h->comm) return;
  Comm* c = (Comm*)h->comm;
  (void)hipDeviceSynchronize();
  if (c->nccl && rccl().CommDestroy) (void)rccl().CommDestroy(c->nccl);
  if (c->sph_own && c->sph_own != c->sph_pad) (void)hipFree(c->sph_own);
  if (c->sph_pad) (void)hipFree(c->sph_pad);
  if (c->sumsq) (void)hipFree(c->sumsq);
  if (c->prm_work) (void)hipFree(c->prm_work);
  for (int g = 0; g < c->world; ++g)
    if (c->peer[g] && g != c->rank && !c->in_process) (void)hipIpcCloseMemHandle(c->peer[g]);
  if (c->local) (void)hipFree(c->local);
  if (c->stage) (void)hipFree(c->stage);
  if (c->swap) (void)hipFree(c->swap);
  delete c;
  h->comm = nullptr;
}

extern "C" {

int mrf_step_prepare(mrf_handle* h, int64_t n_scen, int32_t robot_first, int32_t robot_count, const void* q,
                     const void* qdot, const void* params, void* params_out, void* stream) {
  MRF_CHECK_READY(h);
  if (h->cfg.model != MRF_MODEL_PANDA7) return fail(h, MRF_E_CONFIG, "sharded rollout needs the panda7 model");
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || robot_first < 0 || robot_count < 1 || robot_first + robot_count > h->cfg.n_robots || !q || !qdot ||
      !params || !params_out || params == params_out)
    return fail(h, MRF_E_ARG, "bad argument");
  const int64_t rows = n_scen * robot_count;
  dim3 block(64), grid((unsigned)((rows + 63) / 64));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_goal_estimate<T>, grid, block, (hipStream_t)stream, (const mrf::DevCfg<T>*)h->dcfg, n_scen,
                  (int)robot_first, (int)robot_count, (const T*)q, (const T*)qdot, (const T*)params, (T*)params_out);
  });
}

int mrf_comm_unique_id(void* id_out) {
  if (!id_out) return MRF_E_ARG;
  static_assert(sizeof(ncclUniqueId) == MRF_COMM_ID_BYTES, "MRF_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
  RcclApi& api = rccl();
  if (!api.lib) return MRF_E_DEVICE;
  ncclUniqueId id;
  if (api.GetUniqueId(&id) != ncclSuccess) return MRF_E_DEVICE;
  std::memcpy(id_out, &id, sizeof(id));
  return MRF_OK;
}

int mrf_comm_init(mrf_handle* h, int32_t rank, int32_t world, const void* unique_id) {
  MRF_CHECK_READY(h);
  if (int rc = check_group(h, rank, world)) return rc;
  if (!unique_id && world > 1) return fail(h, MRF_E_ARG, "unique_id required for world > 1 (mrf_comm_unique_id on rank 0)");
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  c->xs = exchange_scalars(h);
  partition(*c, h->cfg.n_robots);
  if (unique_id) {
    RcclApi& api = rccl();
    if (!api.lib) {
      delete c;
      return fail(h, MRF_E_DEVICE, api.error);
    }
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    ncclResult_t r = api.CommInitRank(&c->nccl, world, id, rank);
    if (r != ncclSuccess) {
      delete c;
      return fail(h, MRF_E_DEVICE, std::string("ncclCommInitRank: ") + api.GetErrorString(r));
    }
    c->transport = MRF_TRANSPORT_RCCL;
    // ask the communicator what it is: a bench line that says "N ranks" must be able to show RCCL saw N ranks
    (void)api.CommCount(c->nccl, &c->nccl_count);
    (void)api.CommUserRank(c->nccl, &c->nccl_rank);
    (void)api.CommCuDevice(c->nccl, &c->nccl_device);
  }
  h->comm = c;
  return MRF_OK;
}

int mrf_comm_info(const mrf_handle* h, int32_t* out, int32_t n) {
  if (!h || !out || n < 1) return MRF_E_ARG;
  const Comm* c = (const Comm*)h->comm;
  int32_t one_hop = -1;
  if (c && c->transport == MRF_TRANSPORT_PEER && c->connected) {
    one_hop = 0;
    int32_t pi[MRF_MAX_ROBOTS * MRF_PEER_INFO_N];
    if (mrf_comm_peer_info(h, pi, MRF_MAX_ROBOTS * MRF_PEER_INFO_N) == MRF_OK)
      for (int g = 0; g < c->world; ++g)
        if (g != c->rank && pi[g * MRF_PEER_INFO_N + 3] == 1) one_hop += 1;
  }
  const int32_t vals[MRF_COMM_INFO_N] = {
      c ? c->transport : MRF_TRANSPORT_NONE, c ? c->rank : 0, c ? c->world : 0, c ? c->first[c->rank] : 0,
      c ? c->first[c->rank + 1] - c->first[c->rank] : 0, c ? c->nccl_count : 0, c ? c->nccl_rank : -1,
      c ? c->nccl_device : -1, h->device, c && c->transport == MRF_TRANSPORT_PEER && c->connected ? c->world - 1 : 0,
      h->cfg.exchange, c ? c->xs : mrf_exchange_scalars(h), one_hop, c ? (int32_t)c->grid_cap : 0, c && c->last_paired ? 1 : 0, c && c->tagged ? 1 : 0};
  for (int i = 0; i < n && i < MRF_COMM_INFO_N; ++i) out[i] = vals[i];
  return MRF_OK;
}

int mrf_comm_peer_open(mrf_handle* h, int32_t rank, int32_t world, int64_t max_scenarios, void* ipc_handle_out) {
  MRF_CHECK_READY(h);
  if (int rc = check_group(h, rank, world)) return rc;
  if (max_scenarios < 1 || !ipc_handle_out) return fail(h, MRF_E_ARG, "max_scenarios >= 1 and a handle buffer are required");
  static_assert(sizeof(hipIpcMemHandle_t) == MRF_IPC_HANDLE_BYTES, "MRF_IPC_HANDLE_BYTES must equal sizeof(hipIpcMemHandle_t)");
  Comm* c = new Comm();
  c->rank = rank;
  c->world = world;
  c->transport = MRF_TRANSPORT_PEER;
  c->xs = exchange_scalars(h);
  partition(*c, h->cfg.n_robots);
  const int spw = 64 / c->cnt_max;
  c->b_max = max_scenarios;
  c->nblk_max = (int)((max_scenarios + spw - 1) / spw);
  c->off_flags = 0;
  c->off_err = ((size_t)2 * world * c->nblk_max * sizeof(unsigned long long) + 255) & ~(size_t)255;
  c->off_x = c->off_err + 256;
  // opt-in (the same on every rank of a group): the joint payload travels as tagged words -- see tagged_store
  const char* tg = std::getenv("MRF_PEER_TAGGED");
  c->tagged = tg && std::atoi(tg) == 1 && world > 1 && c->xs == MRF_JOINT_STATE_SCALARS;
  c->bytes = c->off_x + (size_t)2 * h->cfg.n_robots * c->xs * c->b_max * scalar_bytes(h) * (c->tagged ? 2 : 1);
  // fine-grained device memory: coherent for the peers' stores and this GPU's loads while kernels are running
  hipError_t e = hipExtMallocWithFlags((void**)&c->local, c->bytes, hipDeviceMallocFinegrained);
  if (e == hipSuccess) e = hipMemset(c->local, 0, c->bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t mh;
  std::memset(&mh, 0, sizeof(mh));
  if (e == hipSuccess && world > 1) e = hipIpcGetMemHandle(&mh, c->local);
  if (e != hipSuccess) {
    if (c->local) (void)hipFree(c->local);
    delete c;
    return fail(h, MRF_E_DEVICE, std::string("exchange buffer / hipIpcGetMemHandle: ") + hipGetErrorString(e));
  }
  const int own_count = c->first[rank + 1] - c->first[rank];
  const size_t stage_bytes = (size_t)15 * max_scenarios * own_count * scalar_bytes(h) + 256;
  if (hipMalloc(&c->stage, stage_bytes) != hipSuccess) {
    (void)hipFree(c->local);
    delete c;
    return fail(h, MRF_E_DEVICE, "staging buffer of the peer rollout");
  }
  if (world > 1 && c->xs == MRF_JOINT_STATE_SCALARS) {
    // k_rollout_peer_paired: 16 KB (float64) per workgroup of the largest resident grid (one wave per SIMD)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    c->swap_wgs = (unsigned)cus * 4u;
    if (hipMalloc(&c->swap, (size_t)c->swap_wgs * mrf::PEER_SWAP_SCALARS * 64 * scalar_bytes(h)) != hipSuccess) {
      (void)hipFree(c->local);
      (void)hipFree(c->stage);
      delete c;
      return fail(h, MRF_E_DEVICE, "swap area of the peer rollout");
    }
  }
  std::memcpy(ipc_handle_out, &mh, sizeof(mh));
  c->peer[rank] = c->local;
  c->connected = world == 1;
  h->comm = c;
  return MRF_OK;
}

int mrf_comm_peer_connect(mrf_handle* h, const void* ipc_handles_all) {
  MRF_CHECK_READY(h);
  Comm* c = (Comm*)h->comm;
  if (!c || c->transport != MRF_TRANSPORT_PEER) return fail(h, MRF_E_ARG, "mrf_comm_peer_open first");
  if (c->connected) return MRF_OK;
  if (!ipc_handles_all) return fail(h, MRF_E_ARG, "handles missing");
  for (int g = 0; g < c->world; ++g) {
    if (g == c->rank) continue;
    hipIpcMemHandle_t mh;
    std::memcpy(&mh, (const unsigned char*)ipc_handles_all + (size_t)g * MRF_IPC_HANDLE_BYTES, sizeof(mh));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, mh, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(h, MRF_E_DEVICE, std::string("hipIpcOpenMemHandle(rank ") + std::to_string(g) + "): " + hipGetErrorString(e));
    c->peer[g] = (unsigned char*)p;
  }
  // roll call: how many workgroups of the peer kernel's footprint are co-resident HERE (one process per device; ranks that
  // share a device in tests take their fixed share instead, mrf_rollout_sharded)
  const char* share = std::getenv("MRF_PEER_DEVICE_SHARE");
  const char* force = std::getenv("MRF_PEER_ROLL_CALL");  // "1": run it in the shared-device test layout as well (reported; the
                                                          // fixed share stays the smaller cap)
  if (!share || std::atoi(share) <= 1 || (force && force[0] == '1')) {
    int cus = 256, per_cu = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mrf::k_peer_roll_call, 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    c->grid_cap = measured_coresidency(h, (unsigned)per_cu * (unsigned)cus);
    if (c->grid_cap == 0) return fail(h, MRF_E_DEVICE, "peer transport: the residency roll call could not place a single workgroup");
  }
  c->connected = true;
  return MRF_OK;
}

int mrf_streams_concurrent(int32_t device, void* stream_a, void* stream_b, int32_t* concurrent_out) {
  if (!concurrent_out || stream_a == stream_b) return MRF_E_ARG;
  mrf_host::DeviceGuard guard(device);
  unsigned* d = nullptr;
  if (hipMalloc((void**)&d, 8) != hipSuccess || hipMemset(d, 0, 8) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    if (d) (void)hipFree(d);
    return MRF_E_DEVICE;
  }
  int rate_khz = 100000;
  (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, device);
  hipLaunchKernelGGL(mrf::k_pair_wait, dim3(1), dim3(1), 0, (hipStream_t)stream_a, d, d + 1, (long long)rate_khz * 20);  // 20 ms
  hipLaunchKernelGGL(mrf::k_pair_set, dim3(1), dim3(1), 0, (hipStream_t)stream_b, d);
  unsigned r[2] = {0, 0};
  const bool ok = hipStreamSynchronize((hipStream_t)stream_a) == hipSuccess && hipStreamSynchronize((hipStream_t)stream_b) == hipSuccess &&
                  hipMemcpy(r, d, 8, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  if (!ok) return MRF_E_DEVICE;
  *concurrent_out = r[1] != 0u;
  return MRF_OK;
}

int mrf_comm_peer_local_base(const mrf_handle* h, void** base_out) {
  if (!h || !base_out) return MRF_E_ARG;
  const Comm* c = (const Comm*)h->comm;
  if (!c || c->transport != MRF_TRANSPORT_PEER || !c->local) return MRF_E_ARG;
  *base_out = c->local;
  return MRF_OK;
}

int mrf_comm_peer_connect_local(mrf_handle* h, void* const* bases_all) {
  MRF_CHECK_READY(h);
  Comm* c = (Comm*)h->comm;
  if (!c || c->transport != MRF_TRANSPORT_PEER) return fail(h, MRF_E_ARG, "mrf_comm_peer_open first");
  if (c->connected) return MRF_OK;
  if (!bases_all) return fail(h, MRF_E_ARG, "bases missing");
  for (int g = 0; g < c->world; ++g) {
    if (g == c->rank) continue;
    if (!bases_all[g]) return fail(h, MRF_E_ARG, "base of rank " + std::to_string(g) + " missing");
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, bases_all[g]) != hipSuccess) {
      (void)hipGetLastError();
      return fail(h, MRF_E_ARG, "base of rank " + std::to_string(g) + " is not a device allocation of this process");
    }
    if (at.device != h->device) {  // another device of this process: its memory must be reachable from here
      int can = 0;
      (void)hipDeviceCanAccessPeer(&can, h->device, at.device);
      if (!can) return fail(h, MRF_E_DEVICE, "no peer access from device " + std::to_string(h->device) + " to device " + std::to_string(at.device));
      hipError_t e = hipDeviceEnablePeerAccess(at.device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(h, MRF_E_DEVICE, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
      (void)hipGetLastError();
    }
    c->peer[g] = (unsigned char*)bases_all[g];
  }
  c->in_process = true;
  c->connected = true;
  return MRF_OK;
}

int mrf_comm_peer_info(const mrf_handle* h, int32_t* out, int32_t n) {
  if (!h || !out) return MRF_E_ARG;
  const Comm* c = (const Comm*)h->comm;
  if (!c || c->transport != MRF_TRANSPORT_PEER || !c->connected) return MRF_E_ARG;
  if (n < c->world * MRF_PEER_INFO_N) return MRF_E_ARG;
  for (int g = 0; g < c->world; ++g) {
    int32_t* o = out + g * MRF_PEER_INFO_N;
    if (g == c->rank) {
      o[0] = h->device; o[1] = 1; o[2] = 0; o[3] = 0;
      continue;
    }
    o[0] = o[2] = o[3] = -1;
    o[1] = 0;
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof(at));
    if (c->peer[g] && hipPointerGetAttributes(&at, c->peer[g]) == hipSuccess) o[0] = at.device;
    (void)hipGetLastError();
    if (o[0] < 0) continue;
    if (o[0] == h->device) {  // several ranks on one device (the single-GPU tests)
      o[1] = 1; o[2] = 0; o[3] = 0;
      continue;
    }
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, h->device, o[0]) == hipSuccess) o[1] = can;
    uint32_t lt = 0, hops = 0;
    if (hipExtGetLinkTypeAndHopCount(h->device, o[0], &lt, &hops) == hipSuccess) {
      o[2] = (int32_t)lt;
      o[3] = (int32_t)hops;
    }
    (void)hipGetLastError();
  }
  return MRF_OK;
}

int mrf_device_topology(int32_t* n_devices, int32_t* can_access, int32_t* link_type, int32_t* hops, int32_t cap) {
  if (!n_devices || cap < 0) return MRF_E_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    *n_devices = 0;
    return MRF_E_DEVICE;
  }
  *n_devices = n;
  const int m = n < cap ? n : cap;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      int can = i == j;
      uint32_t lt = 0, hp = 0;
      bool ok = true;
      if (i != j) {
        if (hipDeviceCanAccessPeer(&can, i, j) != hipSuccess) can = 0;
        ok = hipExtGetLinkTypeAndHopCount(i, j, &lt, &hp) == hipSuccess;
        (void)hipGetLastError();
      }
      if (can_access) can_access[i * cap + j] = can;
      if (link_type) link_type[i * cap + j] = ok ? (int32_t)lt : -1;
      if (hops) hops[i * cap + j] = ok ? (int32_t)hp : -1;
    }
  return MRF_OK;
}

int mrf_comm_partition(const mrf_handle* h, int32_t* robot_first, int32_t* robot_count) {
  if (!h || !h->comm) return MRF_E_ARG;
  const Comm* c = (const Comm*)h->comm;
  if (robot_first) *robot_first = c->first[c->rank];
  if (robot_count) *robot_count = c->first[c->rank + 1] - c->first[c->rank];
  return MRF_OK;
}

int32_t mrf_comm_transport(const mrf_handle* h) { return (h && h->comm) ? ((const Comm*)h->comm)->transport : MRF_TRANSPORT_NONE; }

int mrf_rollout_sharded(mrf_handle* h, int64_t n_scen, void* q_io, void* qdot_io, const void* params, void* avg_vel_out,
                        void* stream) {
  MRF_CHECK_READY(h);
  Comm* c = (Comm*)h->comm;
  if (!c) return fail(h, MRF_E_ARG, "no communicator: mrf_comm_init or mrf_comm_peer_open/connect first");
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || !q_io || !qdot_io || !params || !avg_vel_out) return fail(h, MRF_E_ARG, "null/negative argument");
  hipStream_t st = (hipStream_t)stream;
  c->last_stream = st;
  const int first = c->first[c->rank], count = c->first[c->rank + 1] - first;
  const int H = h->cfg.horizon;
  const int64_t rows = n_scen * count;

  if (c->transport == MRF_TRANSPORT_PEER) {
    if (!c->connected) return fail(h, MRF_E_ARG, "mrf_comm_peer_connect first");
    if (n_scen > c->b_max) return fail(h, MRF_E_ARG, "n_scen exceeds the max_scenarios of mrf_comm_peer_open");
    mrf::PeerView V;
    std::memset(&V, 0, sizeof(V));
    for (int g = 0; g < c->world; ++g) V.base[g] = c->peer[g];
    V.G = c->world;
    V.grank = c->rank;
    for (int g = 0; g <= c->world; ++g) V.first[g] = c->first[g];
    V.nblk_max = c->nblk_max;
    V.xs = c->xs;
    V.b_max = c->b_max;
    V.off_flags = (long long)c->off_flags;
    V.off_err = (long long)c->off_err;
    V.off_x = (long long)c->off_x;
    V.tagged = c->tagged ? 1 : 0;
    int rate_khz = 100000;
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, h->device);
    const char* env = std::getenv("MRF_PEER_TIMEOUT_MS");
    const long long ms = env ? std::atoll(env) : MRF_PEER_TIMEOUT_DEFAULT_MS;
    V.timeout_ticks = (long long)rate_khz * (ms > 0 ? ms : MRF_PEER_TIMEOUT_DEFAULT_MS);
    const int spw = 64 / c->cnt_max;
    const unsigned nblk = (unsigned)((n_scen + spw - 1) / spw);
    const unsigned long long seq0 = c->seq;
    c->seq += (unsigned long long)H;
    const bool lo = is_link_origin_table(h->cfg);
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
    // several ranks on ONE device share its workgroup slots (test layouts):
    //   MRF_PEER_DEVICE_SHARE = k processes whose peer kernels must be resident together: three quarters of a k-th each.
    //     What round 6 found about this layout at large batches -- a small kernel in FRONT of a rank's peer kernel is
    //     starved while the other ranks' peer kernels wait on the same device -- is handled where the rollouts are issued
    //     (sharded.py, DESIGN.md section 6 "Residency"), not by the size of the share;
    //   the ranks of an in-process group that live on this device run their persistent kernels side by side: a share each
    //     (ranks on other devices of the process do not count).
    unsigned here = 0;
    if (c->in_process && c->world > 1) {
      for (int g = 0; g < c->world; ++g) {
        hipPointerAttribute_t at;
        std::memset(&at, 0, sizeof(at));
        if (hipPointerGetAttributes(&at, c->peer[g]) == hipSuccess && at.device == h->device) here += 1;
      }
      (void)hipGetLastError();
    }
    const char* sh = std::getenv("MRF_PEER_DEVICE_SHARE");
    const int share_k = sh ? std::atoi(sh) : 0;
    auto device_share = [&](unsigned slots) {
      if (share_k > 1) slots = slots / (unsigned)share_k * 3u / 4u;
      if (here > 1) slots = slots / here;
      return slots < 1u ? 1u : slots;
    };
    // Paired blocks (k_rollout_peer_paired), OPT-IN (MRF_PEER_PAIRED=1, the same on every rank): where a workgroup has more
    // than one block to walk -- more blocks than this rank's share of one wave per SIMD -- it walks them two at a time and
    // the flag round trip of one runs under the step of the other.  Not the default: with all ranks on one die it measures
    // 5-13 % SLOWER than a block at a time (DESIGN.md section 6 "Where a sharded step's time goes": what a step loses to
    // the exchange is the wave's own store drain and poll round trips, which a second block does not hide), and the layout
    // it is meant for -- flags crossing xGMI -- cannot be measured on a one-GPU box.  EVERY rank of a group must come to the
    // same answer (a pair is worked on in turns, a single block is not), so it depends on the batch, the device model and
    // the share only -- not on the measured residency or the occupancy of the kernel; MRF_PEER_PAIRED_MIN_BLOCKS moves the
    // threshold (tests: 2).  Ranks that disagree time out with the exchange error, they do not hang.
    unsigned paired_min = device_share((unsigned)cus * 4u) + 1u;
    if (const char* pm = std::getenv("MRF_PEER_PAIRED_MIN_BLOCKS"))
      if (std::atoi(pm) >= 2) paired_min = (unsigned)std::atoi(pm);
    const char* pe = std::getenv("MRF_PEER_PAIRED");
    const bool paired = c->world > 1 && c->xs == MRF_JOINT_STATE_SCALARS && !c->tagged && c->swap && nblk >= paired_min && pe && std::atoi(pe) == 1;
    c->last_paired = paired;
    return dispatch(h, [&](auto t, auto cl) {
      using T = decltype(t);
      using LS = decltype(cl);
      auto go = [&](auto kernel, auto... swap_area) {  // swap_area: k_rollout_peer_paired's extra argument
        // every workgroup of the launch must be resident: a block waits for the same block of the peers, and a
        // workgroup that has not started cannot publish (HIP promises no dispatch order)
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        unsigned resident = (unsigned)per_cu * (unsigned)cus;
        if (c->grid_cap && c->grid_cap < resident) resident = c->grid_cap;  // what the roll call at connect found co-resident
        resident = device_share(resident);
        if (const char* mg = std::getenv("MRF_PEER_MAX_GRID")) {  // test hook: several blocks per workgroup at small batches
          const int k = std::atoi(mg);
          if (k >= 1 && (unsigned)k < resident) resident = (unsigned)k;
        }
        // a group of one waits for nobody: no residency requirement, one workgroup per block as the fused kernel launches
        // them (the dispatcher balances the tail; the block loop then runs once)
        if (c->world == 1) resident = nblk;
        if (paired && c->swap_wgs < resident) resident = c->swap_wgs;  // one slot of the swap area per workgroup
        const unsigned nunit = paired ? (nblk + 1u) / 2u : nblk;
        dim3 block(64), grid(nunit < resident ? nunit : resident);
        if (std::getenv("MRF_PEER_DEBUG")) {
          timespec ts;
          clock_gettime(CLOCK_REALTIME, &ts);
          std::fprintf(stderr, "[mrf peer %ld.%03ld] rank %d/%d: occupancy %d per CU x %d CUs -> resident %u, blocks %u%s, grid %u, seq0 %llu\n",
                       (long)(ts.tv_sec % 1000), ts.tv_nsec / 1000000, c->rank, c->world, per_cu, cus, resident, nblk,
                       paired ? " in pairs" : "", grid.x,
                       (unsigned long long)(seq0 & 0x7fffffff));
        }
        T* q_st = (T*)c->stage;
        T* qd_st = q_st + 7 * rows;
        T* avg_st = qd_st + 7 * rows;
        int* latch = (int*)((unsigned char*)c->stage + (((size_t)15 * c->b_max * count * sizeof(T)) & ~(size_t)15) + 16);
        if (int rc = launch(h, kernel, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, V, n_scen, (const T*)q_io,
                            (const T*)qdot_io, (const T*)params, q_st, qd_st, avg_st, swap_area..., seq0))
          return rc;
        if (int rc = launch(h, mrf::k_peer_latch, dim3(1), dim3(1), st, (const int*)(c->local + c->off_err), latch, c->world,
                            (int)(seq0 >> 40) + 1))
          return rc;
        const int rc3 = launch(h, mrf::k_peer_commit<T>, dim3((unsigned)((rows + 255) / 256)), dim3(256), st, rows, (const int*)latch,
                               (const T*)q_st, (const T*)qd_st, (const T*)avg_st, (T*)q_io, (T*)qdot_io, (T*)avg_vel_out);
        if (std::getenv("MRF_PEER_DEBUG")) {
          timespec ts;
          clock_gettime(CLOCK_REALTIME, &ts);
          std::fprintf(stderr, "[mrf peer %ld.%03ld] rank %d: the three launches of seq0 %llu are queued\n", (long)(ts.tv_sec % 1000),
                       ts.tv_nsec / 1000000, c->rank, (unsigned long long)(seq0 & 0x7fffffff));
        }
        return rc3;
      };
      // what the robots of other ranks send: nothing (a group of one: the fused kernel's step), joint states, spheres
      if (c->world == 1) return lo ? go(mrf::k_rollout_peer<T, LS, true, mrf::XK_NONE>) : go(mrf::k_rollout_peer<T, LS, false, mrf::XK_NONE>);
      if (paired)
        return lo ? go(mrf::k_rollout_peer_paired<T, LS, true>, (T*)c->swap) : go(mrf::k_rollout_peer_paired<T, LS, false>, (T*)c->swap);
      if (c->tagged)
        return lo ? go(mrf::k_rollout_peer<T, LS, true, mrf::XK_JOINTS_TAGGED>) : go(mrf::k_rollout_peer<T, LS, false, mrf::XK_JOINTS_TAGGED>);
      if (c->xs == MRF_JOINT_STATE_SCALARS)
        return lo ? go(mrf::k_rollout_peer<T, LS, true, mrf::XK_JOINTS>) : go(mrf::k_rollout_peer<T, LS, false, mrf::XK_JOINTS>);
      return lo ? go(mrf::k_rollout_peer<T, LS, true, mrf::XK_SPHERES>) : go(mrf::k_rollout_peer<T, LS, false, mrf::XK_SPHERES>);
    });
  }

  // ---- RCCL (or a group of one): predict -> all-gather -> action per horizon step, all on `st`
  if (int rc = ensure_rccl_buffers(h, c, n_scen)) return rc;
  if (int rc = check_hip(h, hipMemsetAsync(c->sumsq, 0, (size_t)rows * scalar_bytes(h), st), "hipMemsetAsync")) return rc;
  const bool joints = c->xs == MRF_JOINT_STATE_SCALARS;
  const size_t per_rank = (size_t)c->cnt_max * c->xs * n_scen;
  int32_t slots[MRF_MAX_ROBOTS];
  for (int j = 0; j < MRF_MAX_ROBOTS; ++j) slots[j] = 0;
  for (int g = 0; g < c->world; ++g)
    for (int r = c->first[g]; r < c->first[g + 1]; ++r) slots[r] = g * c->cnt_max + (r - c->first[g]);
  void* own = c->world == 1 ? c->sph_pad : c->sph_own;
  const int own_mask = (h->cfg.goal_estimate_mask >> first) & ((1 << count) - 1);
  if (own_mask) {  // RF-CV: the estimate is taken once, at the start state (as k_rollout_panda's prologue does)
    if (int rc = mrf_step_prepare(h, n_scen, first, count, q_io, qdot_io, params, c->prm_work, st)) return rc;
    params = c->prm_work;
  }
  // joint payload: the action kernel of step k also does the position update of step k + 1 and writes its joint state into
  // the send block (one launch per step; MRF_STEP_UNFUSED=1 keeps the two launches for A/B runs)
  const bool fuse_next = joints && !std::getenv("MRF_STEP_UNFUSED");
  for (int k = 0; k < H; ++k) {
    if (k == 0 || !fuse_next)
      if (int rc = joints ? mrf_step_predict_joints(h, n_scen, first, count, q_io, qdot_io, own, st)
                          : mrf_step_predict(h, n_scen, first, count, q_io, qdot_io, own, st))
        return rc;
    if (c->nccl) {
      ncclResult_t r = rccl().AllGather(own, c->sph_pad, per_rank, h->cfg.scalar == MRF_F64 ? ncclDouble : ncclFloat,
                                        c->nccl, st);
      if (r != ncclSuccess) return fail(h, MRF_E_LAUNCH, std::string("ncclAllGather: ") + rccl().GetErrorString(r));
    }
    if (int rc = joints ? mrf_host::step_action_joints_slots(h, n_scen, first, count, q_io, qdot_io, params, c->sph_pad, slots,
                                                             c->sumsq, fuse_next && k + 1 < H ? own : nullptr, st)
                        : mrf_host::step_action_slots(h, n_scen, first, count, q_io, qdot_io, params, c->sph_pad, slots, c->sumsq, st))
      return rc;
  }
  dim3 block(256), grid((unsigned)((rows + 255) / 256));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_avg_from_sumsq<T>, grid, block, st, rows, (const T*)c->sumsq, (T)(1.0 / (H * 7)), (T*)avg_vel_out);
  });
}

#ifdef MRF_PEER_TIMING
// development builds only: {publish, wait, remote fold, whole kernel} ticks summed over waves, waves counted, the payload
// loads + staging part of the remote fold; then zeroed
extern "C" int mrf_debug_peer_timing(mrf_handle* h, unsigned long long* out6) {
  Comm* c = h ? (Comm*)h->comm : nullptr;
  if (!c || !c->local || !out6) return MRF_E_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return MRF_E_DEVICE;
  if (hipMemcpy(out6, c->local + c->off_err + 128, 48, hipMemcpyDeviceToHost) != hipSuccess) return MRF_E_DEVICE;
  return hipMemset(c->local + c->off_err + 128, 0, 48) == hipSuccess ? MRF_OK : MRF_E_DEVICE;
}
#endif

int mrf_comm_status(mrf_handle* h) {
  MRF_CHECK_READY(h);
  Comm* c = (Comm*)h->comm;
  if (!c) return fail(h, MRF_E_ARG, "no communicator");
  if (int rc = check_hip(h, hipStreamSynchronize(c->last_stream), "hipStreamSynchronize")) return rc;
  if (c->transport == MRF_TRANSPORT_PEER) {
    int err = 0;
    if (int rc = check_hip(h, hipMemcpy(&err, c->local + c->off_err, sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy")) return rc;
    if (err == (int)c->epoch + 1) {
      int dbg[16] = {0};
      (void)hipMemcpy(dbg, c->local + c->off_err, sizeof(dbg), hipMemcpyDeviceToHost);
      std::string where;
#ifdef MRF_PEER_HEARTBEAT
      where = " [rank " + std::to_string(c->rank) + ": last launch that started had seq0 " + std::to_string(dbg[8]) + ", " +
              std::to_string(dbg[9]) + " of its workgroups started; host has issued up to seq " + std::to_string((long long)(c->seq & 0x7fffffff)) + "]";
#endif
      if (dbg[1])  // this rank's own first wait that ran out (absent when the error came from a peer)
        where += " [first wait that ran out here: block " + std::to_string(dbg[2]) + " (workgroup " + std::to_string(dbg[6]) + " of " +
                std::to_string(dbg[7]) + ") waited for rank " + std::to_string(dbg[3]) + ", expected sequence " +
                std::to_string(dbg[4]) + ", flag held " + std::to_string(dbg[5]) + "; that peer's heartbeat then: launch seq0 " +
                std::to_string(dbg[10]) + ", " + std::to_string(dbg[11]) + " workgroups started, " + std::to_string(dbg[13]) +
                " left, commit passes run " + std::to_string(dbg[14]) + "]";
      if (std::getenv("MRF_PEER_DEBUG")) {
        timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        std::fprintf(stderr, "[mrf peer %ld.%03ld] rank %d: status sees the time-out\n", (long)(ts.tv_sec % 1000), ts.tv_nsec / 1000000, c->rank);
      }
      return fail(h, MRF_E_LAUNCH, "peer exchange timed out: a rank of the group did not publish its payload "
                                   "(different call sequence, a dead peer, or kernels that cannot run concurrently)" + where);
    }
  }
  return MRF_OK;
}

int mrf_comm_reset(mrf_handle* h) {
  MRF_CHECK_READY(h);
  Comm* c = (Comm*)h->comm;
  if (!c) return fail(h, MRF_E_ARG, "no communicator");
  if (int rc = check_hip(h, hipStreamSynchronize(c->last_stream), "hipStreamSynchronize")) return rc;
  if (c->transport == MRF_TRANSPORT_PEER) {
    // error word, the flags the peers raised in THIS rank's buffer, and the sequence counter: after the reset every rank
    // starts from sequence 1 again, whatever number of rollouts each of them had issued before (a rank that missed a
    // call is the usual reason for the timeout).  The caller's barriers on both sides keep peers from writing meanwhile.
    if (int rc = check_hip(h, hipMemset(c->local + c->off_flags, 0, c->off_x - c->off_flags), "hipMemset")) return rc;
    // tagged payload: the words are their own flags (a late store of the old epoch carries a tag no step of the new one expects)
    if (c->tagged)
      if (int rc = check_hip(h, hipMemset(c->local + c->off_x, 0, c->bytes - c->off_x), "hipMemset")) return rc;
    if (int rc = check_hip(h, hipDeviceSynchronize(), "hipDeviceSynchronize")) return rc;
    // A peer whose stream was still running when this rank cleared its flags may have stored a flag of the OLD
    // sequence afterwards.  Sequence numbers carry the reset count in their high bits, so such a flag is smaller than
    // every number of the restarted sequence and can satisfy none of its waits.
    c->epoch += 1;
    c->seq = (c->epoch << 40) | 1ull;
  }
  return MRF_OK;
}

void mrf_comm_destroy(mrf_handle* h) {
  if (!h) return;
  mrf_host::DeviceGuard guard(h->dcfg ? h->device : -1);
  mrf_host::comm_release(h);
}

}  // extern "C"
