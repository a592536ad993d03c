// mrf_tile.hpp -- the per-wave LDS exchange tiles of the coupled kernels (moved out of mrf_kernels.hip in round 6 so that
// the robot-sharded kernels of mrf_comm.hip exchange the robots that live on the SAME rank exactly as the fused kernel
// does): the [72][64] link-origin sphere tile with its software-pipelined fold, and the chunked exchange of generic
// sphere tables.
#pragma once
#include "mrf_device.hpp"

namespace mrf {
inline namespace MRF_DEVICE_FLAVOUR {

// Link-origin sphere table (the reference's rollouts, PM:25-26): every lane already has its own robot's link
// origins, velocities and Jdot*qdot from its own chain walk, so it stages them once in a per-wave LDS tile
// [72][64] (8 links x (x, v, a)) and reads the other robots' entries from there -- no lane re-walks another
// robot's chain.  4 resident waves x 36.9 KB (f64) fit the CU's 160 KB.  The origins of links 1/2 and 5/6
// coincide (zero joint offsets): with equal radii such a pair occupies ONE slot of the tile and is folded once
// with weight 2 (DevCfg::lo_merge01 / lo_merge45, set by the host).
template <typename T>
__device__ __forceinline__ void publish_link_spheres(T* __restrict__ tile, int lane, const PandaKin<T>& K, bool dyn,
                                                      bool acc_on, T jsign, int m01, int m45) {
  // dyn / acc_on are wave-uniform: branch once instead of selecting per value.  Merged duplicates write the same
  // values into the same slot.
#pragma unroll
  for (int sp = 0; sp < 8; ++sp) {
    T* dst = tile + lo_slot(sp, m01, m45) * 9 * 64 + lane;
#pragma unroll
    for (int k3 = 0; k3 < 3; ++k3) dst[k3 * 64] = sp < 7 ? K.o[sp < 7 ? sp : 0][k3] : K.p8[k3];
  }
  if (dyn) {
#pragma unroll
    for (int sp = 0; sp < 8; ++sp) {
      T* dst = tile + lo_slot(sp, m01, m45) * 9 * 64 + lane;
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) dst[(3 + k3) * 64] = sp < 7 ? K.vo[sp < 7 ? sp : 0][k3] : K.v8[k3];
    }
  } else {
#pragma unroll
    for (int sp = 0; sp < 8; ++sp)
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) tile[(sp * 9 + 3 + k3) * 64 + lane] = T(0);
  }
  if (acc_on) {
#pragma unroll
    for (int sp = 0; sp < 8; ++sp) {
      T* dst = tile + lo_slot(sp, m01, m45) * 9 * 64 + lane;
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) dst[(6 + k3) * 64] = jsign * (sp < 7 ? K.ao[sp < 7 ? sp : 0][k3] : K.a8[k3]);
    }
  } else {
#pragma unroll
    for (int sp = 0; sp < 8; ++sp)
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) tile[(sp * 9 + 6 + k3) * 64 + lane] = T(0);
  }
}

constexpr int TILE_RADII = 72 * 64;  // per slot: sphere radius and multiplicity follow the [72][64] tile
constexpr int TILE_MULT = TILE_RADII + 8;
constexpr int TILE_SCALARS = TILE_MULT + 8;

template <typename T>
__device__ __forceinline__ void stage_sphere_radii(const DevCfg<T>& cfg, T* __restrict__ tile, int lane) {
  const int m01 = cfg.lo_merge01, m45 = cfg.lo_merge45;
  if (lane < 8 && !(lane == 1 && m01) && !(lane == 5 && m45)) {
    const int slot = lo_slot(lane, m01, m45);
    tile[TILE_RADII + slot] = cfg.sphere_r[lane];
    tile[TILE_MULT + slot] = ((lane == 0 && m01) || (lane == 4 && m45)) ? T(2) : T(1);
  }
}

template <class CL, typename T>
__device__ __forceinline__ void obstacles_from_tile(const DevCfg<T>& cfg, const T* __restrict__ tile, int ls, int li, int N,
                                                    const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
  // One wave per SIMD: nothing else hides the LDS / scalar-load latency, so the next sphere's nine scalars, radius
  // and multiplicity are fetched before the current sphere's leaves are evaluated (software pipeline, depth 1).
  const int nsp = 8 - cfg.lo_merge01 - cfg.lo_merge45;  // distinct spheres per robot
  const int M = (N - 1) * nsp;
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  typedef const volatile __attribute__((address_space(3))) T* lds_vptr;
  auto address = [&](int d, int sp) {
    int jr = li + 1 + d;
    if (jr >= N) jr -= N;
    return (sp * 9) * 64 + ls * N + jr;
  };
  // Two register buffers in ping-pong (the loop is unrolled by two) so that no buffer-to-buffer copies are needed.
  T bufA[11], bufB[11];  // x[3], v[3], a[3], radius, multiplicity
  {
    // The first fetch is volatile so that the optimizer cannot merge it with the in-loop fetches into one load of
    // a loop-carried address at the top of the loop (which would undo the pipeline).
    lds_vptr src = (lds_vptr)(tile + address(0, 0));
#pragma unroll
    for (int k = 0; k < 9; ++k) bufA[k] = src[k * 64];
    bufA[9] = ((lds_vptr)tile)[TILE_RADII];
    bufA[10] = ((lds_vptr)tile)[TILE_MULT];
  }
  if constexpr (!CL::generic) {
    // Have the scalar loads of the leaf constants complete before the loop: the wait-count pass is static, so a
    // scalar load still pending at loop entry would put an lgkmcnt(0) -- and with it a wait for the prefetch
    // just issued -- into every iteration.
    asm volatile("" ::"s"(cfg.jsign), "s"(cfg.cf.k), "s"(cfg.cg.k));
  }
  int dn = 0, sn = 0;  // (other robot, slot) of the sphere fetched last
  auto fetch_next = [&](T (&buf)[11], bool advance) {
    if (advance) {
      if (++sn == nsp) {
        sn = 0;
        ++dn;
      }
    }
    lds_ptr src = (lds_ptr)(tile + address(dn, sn));
#pragma unroll
    for (int k = 0; k < 9; ++k) buf[k] = src[k * 64];
    buf[9] = ((lds_ptr)tile)[TILE_RADII + sn];  // staged once per kernel by stage_sphere_radii
    buf[10] = ((lds_ptr)tile)[TILE_MULT + sn];
  };
  int m = 0;
#pragma unroll 1
  for (; m + 1 < M; m += 2) {
    fetch_next(bufB, true);
    accumulate_obstacle<CL>(cfg, E, bufA, bufA + 3, bufA + 6, bufA[9], false, acc, bufA[10]);
    fetch_next(bufA, m + 2 < M);  // past the end: re-reads the last sphere, never used
    accumulate_obstacle<CL>(cfg, E, bufB, bufB + 3, bufB + 6, bufB[9], false, acc, bufB[10]);
  }
  if (m < M) accumulate_obstacle<CL>(cfg, E, bufA, bufA + 3, bufA + 6, bufA[9], false, acc, bufA[10]);  // odd count
}


// Small generic tables without obstacle accelerations (round 6): up to TILE_PACKED_MAX spheres per robot as 6 rows each (x, v)
// in the same [72][64] tile, one radius per sphere behind it -- published ONCE by the solve's own unrolled chain walk
// (panda_walk_own's emit_link hook), folded in one software-pipelined loop with the accelerations as compile-time zeros.
constexpr int TILE_PACKED_MAX = 12;  // 12 x 6 rows = the tile's 72 rows; 12 <= the 16 radius / multiplicity slots behind it

template <class CL, typename T>
__device__ __forceinline__ void obstacles_from_tile_packed(const DevCfg<T>& cfg, const T* __restrict__ tile, int ls, int li,
                                                           int N, int nsp, const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
  const int M = (N - 1) * nsp;
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  typedef const volatile __attribute__((address_space(3))) T* lds_vptr;
  auto address = [&](int d, int sp) {
    int jr = li + 1 + d;
    if (jr >= N) jr -= N;
    return (sp * 6) * 64 + ls * N + jr;
  };
  T bufA[7], bufB[7];  // x[3], v[3], radius: ping-pong, the loop is unrolled by two
  {
    lds_vptr src = (lds_vptr)(tile + address(0, 0));  // volatile: keeps the first fetch out of the loop (obstacles_from_tile)
#pragma unroll
    for (int k = 0; k < 6; ++k) bufA[k] = src[k * 64];
    bufA[6] = ((lds_vptr)tile)[TILE_RADII];
  }
  int dn = 0, sn = 0;  // (other robot, sphere) of the sphere fetched last
  auto fetch_next = [&](T (&buf)[7], bool advance) {
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
  };
  auto fold = [&](T (&buf)[7]) {
    const T zero[3] = {T(0), T(0), T(0)};  // EXJ:411 "currently no acceleration": the n.a_o terms compile out
    accumulate_obstacle<CL>(cfg, E, buf, buf + 3, zero, buf[6], false, acc);
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

// The emit_link hook that goes with it: writes the spheres attached to panda_link<link> (table order) into the packed tile
// rows of this lane.  State (next table entry, prefetched one sphere ahead) lives in the caller.
template <typename T>
struct PackedSphereEmit {
  const DevCfg<T>& cfg;
  T* __restrict__ tile;
  int lane, S;
  bool dyn;
  int s, link_s;
  T off[3];
  __device__ __forceinline__ PackedSphereEmit(const DevCfg<T>& c, T* t, int ln, bool d)
      : cfg(c), tile(t), lane(ln), S(c.n_spheres), dyn(d), s(0), link_s(c.n_spheres > 0 ? c.sphere_link[0] : 0) {
    off[0] = c.sphere_off[0][0];
    off[1] = c.sphere_off[0][1];
    off[2] = c.sphere_off[0][2];
  }
  __device__ __forceinline__ void operator()(int link, const T* X, const T* Y, const T* Z, const T* o, const T* w, const T* al,
                                             const T* vo, const T* ao) {
    (void)al;
    (void)ao;
    while (s < S && link_s == link) {
      const T ox = off[0], oy = off[1], oz = off[2];
      const int s_next = s + 1 < S ? s + 1 : s;
      link_s = s + 1 < S ? cfg.sphere_link[s_next] : 0;
      off[0] = cfg.sphere_off[s_next][0];
      off[1] = cfg.sphere_off[s_next][1];
      off[2] = cfg.sphere_off[s_next][2];
      T rr[3], wr[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) rr[k] = ox * X[k] + oy * Y[k] + oz * Z[k];
      cross3(w, rr, wr);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        tile[(s * 6 + k) * 64 + lane] = o[k] + rr[k];
        tile[(s * 6 + 3 + k) * 64 + lane] = dyn ? vo[k] + wr[k] : T(0);  // EXJ:336-337
      }
      ++s;
    }
  }
};

// Generic sphere tables (offset spheres, any count): every lane walks ITS OWN chain once, emitting its spheres in
// table order; they are exchanged CH at a time through a [CH][9][64] LDS tile (18 KB in f64) and each lane folds the
// chunk's spheres of the other robots of its scenario before the walk moves on -- instead of every lane re-walking all
// N-1 other chains (r01).  The full sphere set of 64 lanes would not fit the LDS at four waves per CU (20 spheres: 92 KB
// per wave), a chunk does.  xch holds cos q, sin q, qdot of every lane ([21][64], the rolled walk reads them by joint
// index); chunk is the exchange tile.  acc_scale: jsign for x-dot-dot = jac_dot*qdot (rollouts, FPJ:97-99), 0 where the
// reference passes zero accelerations (EXJ:411).
#ifndef MRF_GEN_CH
#define MRF_GEN_CH 4  // experiment switch (tools/build_variant.sh): spheres per exchange chunk
#endif
constexpr int GEN_CH = MRF_GEN_CH;
constexpr int GEN_XCH = 21 * 64;
constexpr int GEN_SCALARS = GEN_XCH + GEN_CH * 9 * 64;

template <class CL, typename T>
__device__ __forceinline__ void obstacles_generic_chunked(const DevCfg<T>& cfg, T* __restrict__ xch, int lane, int ls, int li,
                                                          int N, const T* __restrict__ mount_own, bool dyn, T acc_scale,
                                                          const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
  T* chunk = xch + GEN_XCH;
  const int S = cfg.n_spheres;
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  panda_walk_spheres<false, T>(
      cfg, mount_own,
      [&](int j, T& c, T& s, T& qdj) {
        c = xch[(3 * j + 0) * 64 + lane];
        s = xch[(3 * j + 1) * 64 + lane];
        qdj = xch[(3 * j + 2) * 64 + lane];
      },
      [&](int s, const T* x, const T* v, const T* a) {
        const int k = s % GEN_CH;
        if (k == 0) __syncthreads();  // the previous chunk has been folded by every lane
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          chunk[((k * 9) + c) * 64 + lane] = x[c];
          chunk[((k * 9) + 3 + c) * 64 + lane] = dyn ? v[c] : T(0);          // FPJ:215-220 / EXJ:336-339
          chunk[((k * 9) + 6 + c) * 64 + lane] = dyn ? acc_scale * a[c] : T(0);
        }
        if (k != GEN_CH - 1 && s != S - 1) return;
        __syncthreads();
        const int n = k + 1, s0 = s - k;  // spheres in this chunk, first sphere of the chunk
        pipelined_pairs<T, 9>(
            (N - 1) * n,
            [&](int m, T (&buf)[9]) {
              const int d = m / n, kk = m - d * n;
              int jr = li + 1 + d;
              if (jr >= N) jr -= N;
              lds_ptr src = (lds_ptr)(chunk + (kk * 9) * 64 + ls * N + jr);
#pragma unroll
              for (int c = 0; c < 9; ++c) buf[c] = src[c * 64];
            },
            [&](int m, T (&buf)[9]) {
              const int kk = m % n;
              accumulate_obstacle<CL>(cfg, E, buf, buf + 3, buf + 6, cfg.sphere_r[s0 + kk], false, acc);
            });
      });
}

}  // inline namespace
}  // namespace mrf
