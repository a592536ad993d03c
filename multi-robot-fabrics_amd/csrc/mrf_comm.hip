// mrf_comm.hip -- robot-sharded Rollout Fabrics inside the library (include/mrf.h "Robot-sharded rollout INSIDE the
// library").  The exchange step of the reference's rollout graph (forward_planner_Jointspace.py:211-225:
// at step k robot i reads the predicted spheres of every robot j != i) crosses GPUs here:
//
//   RCCL transport   host loop in C++: k_step_predict -> ncclAllGather -> k_step_action per horizon step, everything
//                    enqueued on the caller's stream.  librccl is dlopen'ed (torch ships its own copy under the same
//                    soname; whichever is already in the process is the one that gets used).
//   PEER transport   k_rollout_peer: ONE persistent kernel per rollout.  Workgroup = one wave = the owned robots of
//                    floor(64/cnt_max) scenarios.  Per step a lane walks its chain once, stores its robot's sphere states
//                    straight into every rank's exchange buffer (peer-mapped device memory: over xGMI between GPUs),
//                    fences, raises the per-workgroup flag on every rank, polls the flags the peers raised for the
//                    same scenarios, then folds the other robots' spheres from its LOCAL buffer and finishes the
//                    solve.  Two buffer generations (step parity) are enough: a rank can publish step k+2 only after
//                    it has seen every peer's step k+1 flag, which a peer raises after it finished reading step k.
//                    Block X only ever waits for block X of the peers; the grid is capped at the resident workgroup
//                    count and each workgroup walks its blocks in increasing order, so no wait can depend on a
//                    workgroup that is not running.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mrf_device.hpp"
#include "mrf_host.hpp"

namespace mrf {

// view of the exchange buffers that the peer kernel gets by value
struct PeerView {
  unsigned char* base[MRF_MAX_ROBOTS];  // exchange allocation of every rank (own entry: the local allocation)
  int G, grank;                         // ranks in the group, own rank
  int first[MRF_MAX_ROBOTS + 1];        // robot block of rank g: [first[g], first[g+1])
  int nblk_max;                         // flag columns per rank
  long long b_max;                      // scenario capacity of the buffers
  long long off_flags, off_err, off_x;  // byte offsets inside an allocation
  long long timeout_ticks;              // bounded spin, in wall_clock64 ticks
};

// flags: [2 generations][G source ranks][nblk_max]   spheres: [2][n_robots][SX][9][b_max]
__device__ __forceinline__ unsigned long long* peer_flag(const PeerView& V, int dst, int gen, int src, int blk) {
  return reinterpret_cast<unsigned long long*>(V.base[dst] + V.off_flags) + ((size_t)(gen * V.G + src) * V.nblk_max + blk);
}
template <typename T>
__device__ __forceinline__ T* peer_x(const PeerView& V, int dst, int gen, int n_robots, int SX) {
  return reinterpret_cast<T*>(V.base[dst] + V.off_x) + (size_t)gen * n_robots * SX * 9 * V.b_max;
}

// Payload stores into an exchange buffer.  The buffer of another rank is an IPC mapping whose caching attributes on
// the writer's side are the driver's choice, so those stores are made at system scope (write-through to the owner's
// memory) instead of relying on the mapping being fine-grained; the own buffer takes plain stores.
template <typename T>
__device__ __forceinline__ void xstore(T* p, T v, bool remote) {
  if (remote)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else
    *p = v;
}

// a quiet NaN by bit pattern (this translation unit is compiled -ffast-math, where NaN literals are undefined)
template <typename T>
__device__ __forceinline__ T quiet_nan();
template <>
__device__ __forceinline__ double quiet_nan<double>() { return __builtin_bit_cast(double, 0x7ff8000000000000ull); }
template <>
__device__ __forceinline__ float quiet_nan<float>() { return __builtin_bit_cast(float, 0x7fc00000u); }

// q_in / qd_in are only read; the advanced state and the velocity signal go to the STAGING arrays q_st / qd_st / avg_st
// (owned by the communicator) and are committed to the caller's arrays by k_peer_commit after the whole grid has
// finished -- from ONE reading of the error word, so that a timed-out exchange leaves every row where it was.
template <typename T, class LS, bool LO>
__global__ __launch_bounds__(64) void k_rollout_peer(const DevCfg<T>* __restrict__ cfgp, PeerView V, int64_t n_scen,
                                                      const T* __restrict__ q_in, const T* __restrict__ qd_in,
                                                      const T* __restrict__ prm, T* __restrict__ q_st,
                                                      T* __restrict__ qd_st, T* __restrict__ avg_st,
                                                      unsigned long long seq0) {
  __shared__ T xch[21 * 64];  // cos q, sin q, qdot of every lane (sphere tables with offsets re-walk from it)
  const DevCfg<T>& cfg = *cfgp;
  const int N = cfg.n_robots;
  const int first = V.first[V.grank], count = V.first[V.grank + 1] - first;
  int cnt_max = 1;
  for (int g = 0; g < V.G; ++g) cnt_max = max(cnt_max, V.first[g + 1] - V.first[g]);
  const int spw = 64 / cnt_max;  // scenarios per block: the same on every rank, so block X is the same scenarios
  const int lane = threadIdx.x;
  const int nblk = (int)((n_scen + spw - 1) / spw);
  // The grid is capped at what is resident at once (host side); a workgroup then walks blocks blockIdx.x,
  // blockIdx.x + gridDim.x, ... in increasing order.  Block X only ever waits for block X of the peers, every workgroup
  // of every rank is resident and visits its blocks in increasing index order, so the wait graph has no cycle whatever
  // the dispatch order or the grid size of the other ranks.
#pragma unroll 1
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
  int ls = lane / count;
  const int l = lane - ls * count;
  int64_t scen = (int64_t)blk * spw + ls;
  const bool active = ls < spw && scen < n_scen;
  if (!active) {  // idle lanes shadow the block's first row (no stores)
    ls = 0;
    scen = (int64_t)blk * spw;
  }
  const int me = first + (active ? l : 0);
  const int64_t rows = n_scen * count;
  const int64_t row = scen * count + (active ? l : 0);
  const int m01 = LO ? cfg.lo_merge01 : 0, m45 = LO ? cfg.lo_merge45 : 0;
  const int SX = cfg.n_spheres - m01 - m45;
  int* err = reinterpret_cast<int*>(V.base[V.grank] + V.off_err);
  // The error word is tagged with the reset epoch (the high bits of every sequence number, mrf_comm_reset): "broken" means
  // "holds THIS epoch's tag", so a kernel of the previous epoch that times out late -- after a peer's reset has already
  // started the next epoch -- cannot break the new sequence with its store.
  const int etag = (int)(seq0 >> 40) + 1;

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
  const bool dyn = cfg.dynamic != 0;
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
    if constexpr (!LO) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        xch[(3 * j + 0) * 64 + lane] = R.cq[j];
        xch[(3 * j + 1) * 64 + lane] = R.sq[j];
        xch[(3 * j + 2) * 64 + lane] = R.qd[j];
      }
      __syncthreads();
    }
    const T* xloc = peer_x<T>(V, V.grank, gen, N, SX);
    T qdd[7], act[7];
    panda_solve_row<LS, LO && kSingleWalk<LS>>(
        cfg, mount_own, R, P,
        [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
          // fold the spheres of every other robot from the LOCAL exchange buffer (step k's generation), one flat
          // software-pipelined loop over (other robot, slot) pairs
          pipelined_pairs<T, 9>(
              (N - 1) * SX,
              [&](int m, T (&buf)[9]) {
                const int d = m / SX, slot = m - d * SX;
                int jr = me + 1 + d;
                if (jr >= N) jr -= N;
                const T* src = xloc + ((size_t)(jr * SX + slot) * 9) * V.b_max + scen;
#pragma unroll
                for (int c = 0; c < 9; ++c) buf[c] = src[(size_t)c * V.b_max];
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
                accumulate_obstacle<typename LS::Collision>(cfg, E, buf, v, a, cfg.sphere_r[s], false, acc, mult);
              });
        },
        qdd, act,
        [&](const PandaKin<T>& K1) {
          // ---- publish: this robot's spheres of step k into every rank's buffer (FPJ:211-225 across GPUs)
          if (active) {
            for (int g = 0; g < V.G; ++g) {
              const bool remote = g != V.grank;
              T* dst = peer_x<T>(V, g, gen, N, SX) + ((size_t)me * SX * 9) * V.b_max + scen;
              if constexpr (LO) {
#pragma unroll
                for (int sp = 0; sp < 8; ++sp) {
                  if ((sp == 1 && m01) || (sp == 5 && m45)) continue;  // coincident link origins travel once
                  T* d9 = dst + ((size_t)lo_slot(sp, m01, m45) * 9) * V.b_max;
#pragma unroll
                  for (int c = 0; c < 3; ++c) {
                    xstore(d9 + (size_t)c * V.b_max, sp < 7 ? K1.o[sp < 7 ? sp : 0][c] : K1.p8[c], remote);
                    xstore(d9 + (size_t)(3 + c) * V.b_max, sp < 7 ? K1.vo[sp < 7 ? sp : 0][c] : K1.v8[c], remote);
                    xstore(d9 + (size_t)(6 + c) * V.b_max, cfg.jsign * (sp < 7 ? K1.ao[sp < 7 ? sp : 0][c] : K1.a8[c]), remote);
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
                        xstore(d9 + (size_t)c * V.b_max, x[c], remote);
                        xstore(d9 + (size_t)(3 + c) * V.b_max, v[c], remote);
                        xstore(d9 + (size_t)(6 + c) * V.b_max, cfg.jsign * a[c], remote);
                      }
                    });
              }
            }
          }
          if (V.G == 1) {  // a group of one: the only reader of these stores is this very wave
            __syncthreads();
            return;
          }
          __threadfence_system();  // the wave's stores (local and remote) are performed before the flags go up
          __syncthreads();
          // Once the group is in error (this rank timed out, or a peer did and said so in this rank's error word) no
          // further flag goes up: the spheres behind it may have been computed from stale data, and the peers must
          // time out -- or see the error -- rather than fold them.
          const bool broken = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == etag;
          if (lane < V.G && !broken)
            __hip_atomic_store(peer_flag(V, lane, gen, V.grank, blk), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
          // ---- wait for the same workgroup of every other rank (bounded: a missing peer must not hang the GPU)
          if (lane < V.G && lane != V.grank) {
            const unsigned long long* f = peer_flag(V, V.grank, gen, lane, blk);
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
              if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == etag) break;
              if (wall_clock64() - t0 > V.timeout_ticks) {
                // the timeout is the GROUP's: raise the error word of every rank, so that a peer which went on with
                // this rank's (now missing) spheres cannot return a finite result either
                for (int g = 0; g < V.G; ++g)
                  __hip_atomic_store(reinterpret_cast<int*>(V.base[g] + V.off_err), etag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
              }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          __syncthreads();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // every lane reads the peers' spheres after the flags
        });
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
}

// After k_rollout_peer (same stream): one thread latches the error word, then every row is committed from that latch --
// all rows advance, or (a timed-out exchange: some step folded stale spheres) none does and the velocity signal is NaN,
// so that a caller that forgets mrf_comm_status cannot take the result for a rollout.
__global__ void k_peer_latch(const int* __restrict__ err, int* __restrict__ latch, int group, int etag) {
  *latch = group > 1 ? (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == etag) : 0;
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
  // RCCL
  ncclComm_t nccl = nullptr;
  int nccl_count = 0, nccl_rank = -1, nccl_device = -1;  // what the communicator itself reports (mrf_comm_info)
  void* sph_own = nullptr;  // [cnt_max][SX][9][B]
  void* sph_pad = nullptr;  // [world][cnt_max][SX][9][B]
  void* sumsq = nullptr;    // [B*count]
  void* prm_work = nullptr; // [MRF_NPARAM][B*count]: params with the RF-CV goal estimate applied
  int64_t cap_scen = 0;
  // PEER
  unsigned char* local = nullptr;
  unsigned char* peer[MRF_MAX_ROBOTS] = {nullptr};
  bool connected = false;
  int64_t b_max = 0;
  int nblk_max = 0;
  size_t off_flags = 0, off_err = 0, off_x = 0, bytes = 0;
  void* stage = nullptr;  // peer kernel outputs before the commit: q [7][rows], qdot [7][rows], avg [rows], latch
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
  return MRF_OK;
}

size_t scalar_bytes(const mrf_handle* h) { return h->cfg.scalar == MRF_F64 ? 8 : 4; }

int exchange_spheres(const mrf_handle* h) { return mrf_exchange_spheres(h); }

int ensure_rccl_buffers(mrf_handle* h, Comm* c, int64_t n_scen) {
  if (n_scen <= c->cap_scen) return MRF_OK;
  if (c->sph_own && c->sph_own != c->sph_pad) (void)hipFree(c->sph_own);
  if (c->sph_pad) (void)hipFree(c->sph_pad);
  if (c->sumsq) (void)hipFree(c->sumsq);
  if (c->prm_work) (void)hipFree(c->prm_work);
  c->sph_own = c->sph_pad = c->sumsq = c->prm_work = nullptr;
  c->cap_scen = 0;
  const size_t blk = (size_t)c->cnt_max * exchange_spheres(h) * 9 * n_scen * scalar_bytes(h);
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
    if (c->peer[g] && g != c->rank) (void)hipIpcCloseMemHandle(c->peer[g]);
  if (c->local) (void)hipFree(c->local);
  if (c->stage) (void)hipFree(c->stage);
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
  const int32_t vals[MRF_COMM_INFO_N] = {
      c ? c->transport : MRF_TRANSPORT_NONE, c ? c->rank : 0, c ? c->world : 0, c ? c->first[c->rank] : 0,
      c ? c->first[c->rank + 1] - c->first[c->rank] : 0, c ? c->nccl_count : 0, c ? c->nccl_rank : -1,
      c ? c->nccl_device : -1, h->device, c && c->transport == MRF_TRANSPORT_PEER && c->connected ? c->world - 1 : 0};
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
  partition(*c, h->cfg.n_robots);
  const int spw = 64 / c->cnt_max;
  c->b_max = max_scenarios;
  c->nblk_max = (int)((max_scenarios + spw - 1) / spw);
  c->off_flags = 0;
  c->off_err = ((size_t)2 * world * c->nblk_max * sizeof(unsigned long long) + 255) & ~(size_t)255;
  c->off_x = c->off_err + 256;
  c->bytes = c->off_x + (size_t)2 * h->cfg.n_robots * exchange_spheres(h) * 9 * c->b_max * scalar_bytes(h);
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
  c->connected = true;
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
    V.b_max = c->b_max;
    V.off_flags = (long long)c->off_flags;
    V.off_err = (long long)c->off_err;
    V.off_x = (long long)c->off_x;
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
    return dispatch(h, [&](auto t, auto cl) {
      using T = decltype(t);
      using LS = decltype(cl);
      auto go = [&](auto kernel) {
        // every workgroup of the launch must be resident: a block waits for the same block of the peers, and a
        // workgroup that has not started cannot publish (HIP promises no dispatch order)
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        unsigned resident = (unsigned)per_cu * (unsigned)cus;
        // several ranks on ONE device (the single-GPU test hook) share its workgroup slots: MRF_PEER_DEVICE_SHARE = number
        // of processes whose peer kernels must be resident together
        if (const char* sh = std::getenv("MRF_PEER_DEVICE_SHARE")) {
          const int k = std::atoi(sh);
          if (k > 1) resident = resident / (unsigned)k ? resident / (unsigned)k : 1u;
        }
        dim3 block(64), grid(nblk < resident ? nblk : resident);
        T* q_st = (T*)c->stage;
        T* qd_st = q_st + 7 * rows;
        T* avg_st = qd_st + 7 * rows;
        int* latch = (int*)((unsigned char*)c->stage + (((size_t)15 * c->b_max * count * sizeof(T)) & ~(size_t)15) + 16);
        if (int rc = launch(h, kernel, grid, block, st, (const mrf::DevCfg<T>*)h->dcfg, V, n_scen, (const T*)q_io,
                            (const T*)qdot_io, (const T*)params, q_st, qd_st, avg_st, seq0))
          return rc;
        if (int rc = launch(h, mrf::k_peer_latch, dim3(1), dim3(1), st, (const int*)(c->local + c->off_err), latch, c->world,
                            (int)(seq0 >> 40) + 1))
          return rc;
        return launch(h, mrf::k_peer_commit<T>, dim3((unsigned)((rows + 255) / 256)), dim3(256), st, rows, (const int*)latch,
                      (const T*)q_st, (const T*)qd_st, (const T*)avg_st, (T*)q_io, (T*)qdot_io, (T*)avg_vel_out);
      };
      if (lo) return go(mrf::k_rollout_peer<T, LS, true>);
      return go(mrf::k_rollout_peer<T, LS, false>);
    });
  }

  // ---- RCCL (or a group of one): predict -> all-gather -> action per horizon step, all on `st`
  if (int rc = ensure_rccl_buffers(h, c, n_scen)) return rc;
  if (int rc = check_hip(h, hipMemsetAsync(c->sumsq, 0, (size_t)rows * scalar_bytes(h), st), "hipMemsetAsync")) return rc;
  const int SX = exchange_spheres(h);
  const size_t per_rank = (size_t)c->cnt_max * SX * 9 * n_scen;
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
  for (int k = 0; k < H; ++k) {
    if (int rc = mrf_step_predict(h, n_scen, first, count, q_io, qdot_io, own, st)) return rc;
    if (c->nccl) {
      ncclResult_t r = rccl().AllGather(own, c->sph_pad, per_rank, h->cfg.scalar == MRF_F64 ? ncclDouble : ncclFloat,
                                        c->nccl, st);
      if (r != ncclSuccess) return fail(h, MRF_E_LAUNCH, std::string("ncclAllGather: ") + rccl().GetErrorString(r));
    }
    if (int rc = mrf_host::step_action_slots(h, n_scen, first, count, q_io, qdot_io, params, c->sph_pad, slots, c->sumsq, st))
      return rc;
  }
  dim3 block(256), grid((unsigned)((rows + 255) / 256));
  return dispatch_scalar(h, [&](auto t) {
    using T = decltype(t);
    return launch(h, mrf::k_avg_from_sumsq<T>, grid, block, st, rows, (const T*)c->sumsq, (T)(1.0 / (H * 7)), (T*)avg_vel_out);
  });
}

int mrf_comm_status(mrf_handle* h) {
  MRF_CHECK_READY(h);
  Comm* c = (Comm*)h->comm;
  if (!c) return fail(h, MRF_E_ARG, "no communicator");
  if (int rc = check_hip(h, hipStreamSynchronize(c->last_stream), "hipStreamSynchronize")) return rc;
  if (c->transport == MRF_TRANSPORT_PEER) {
    int err = 0;
    if (int rc = check_hip(h, hipMemcpy(&err, c->local + c->off_err, sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy")) return rc;
    if (err == (int)c->epoch + 1) return fail(h, MRF_E_LAUNCH, "peer exchange timed out: a rank of the group did not publish its spheres "
                                          "(different call sequence, a dead peer, or kernels that cannot run concurrently)");
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
