// mrf_device.hpp -- device-side building blocks of the fabric solve for gfx950 (CDNA4).
//
// One thread owns one (scenario, robot) row.  Everything a row needs between the first load and the
// last store lives in registers; the only memory traffic inside a rollout is the exchange of joint
// state between the robots of one scenario (a per-wave LDS tile), so the kernels are VALU-bound by
// construction (DESIGN.md "kernels").  No MFMA: the contractions are 3x3 / 3x7 / 7x7.
//
// What is computed (DESIGN.md "spec"):
//   chain walk      : Panda forward kinematics with the URDF constants folded in (all joint axes local z,
//                     roll in {0, +-pi/2}), plus the velocity and (qddot = 0) acceleration of every joint
//                     origin by the classic outward recursion  -> x, v = J qd, Jdot qd
//                     (replaces fk_fun / jac_fun / jac_dot_fun, reference utils.py:16-54)
//   leaf folding    : every spherical-obstacle leaf of one ego point is rank-1 in that point's task space:
//                     A += (m/R^2) n n^T,  b += n [ f/R + (m/R^2)(sign*kappa - n.a_o) ]
//                     (the 3-stage pull geometry-map / dynamic-map / fk of `fabrics`, folded)
//   pullback        : M_q += J^T A J, f_q += J^T (b + A c) once per ego point, J built column by column
//                     as z_j x (p - o_j)
//   solve           : LDL^T of the 7x7 (SPD: base 0.2 I + PSD leaf metrics), energization, damping
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/mrf.h"

// MRF_SGPR_CONST (set by the translation units whose kernels run at two waves per SIMD, mrf_rollout_wp.hip): float64
// literals of the math helpers are built in SGPR pairs, see MRF_SC below.  The two flavours of this header live in
// different inline namespaces, so a library that links both holds two distinct sets of inline functions.
#ifdef MRF_SGPR_CONST
#define MRF_DEVICE_FLAVOUR sgpr_literals
#else
#define MRF_DEVICE_FLAVOUR vgpr_literals
#endif

namespace mrf {
inline namespace MRF_DEVICE_FLAVOUR {

constexpr int NG = 5;  // distinct ego collision points of a Panda: links 3, 4, 5(=6), 7, 8

template <typename T>
struct LeafFn {
  int family, gate, p, pad;
  T k, c, s;
};

template <typename T>
struct DevCfg {
  int model, mode, n_robots, n_spheres, horizon, dynamic, n_ego, n_planes;
  int use_limits, n_goals, plane_abs, zero_small, obst_dim, goal_mask;
  // Link-origin sphere table: panda_joint2 and panda_joint6 have zero origin offsets (URDF), so the origins of links
  // 1/2 and of links 5/6 coincide with identical velocity and Jdot*qdot.  When their radii are equal too, the two
  // spheres produce the same leaf twice: the coupled kernels evaluate it once with weight 2 (set by the host).
  int lo_merge01, lo_merge45;
  int ego_mask;  // bit (l-3): panda_link l carries collision + plane leaves (mrf_config.ego_link_mask)
  T dt, eps, jsign, goal_T, base_mass;
  T attr_k, attr_alpha, attr_mu, attr_ml, attr_a;
  T beta_a, beta_r, beta_b, beta_s, eta_a, eta_s;
  T mount[MRF_MAX_ROBOTS][12];
  T limits[MRF_DOF_MAX][2];
  int sphere_link[MRF_MAX_SPHERES];
  T sphere_off[MRF_MAX_SPHERES][3];
  T sphere_r[MRF_MAX_SPHERES];
  LeafFn<T> cg, cf, pg, pf, lg, lf;
};

// ------------------------------------------------------------------------------------ math wrappers
__device__ __forceinline__ double m_sqrt(double x) { return ::sqrt(x); }
__device__ __forceinline__ float m_sqrt(float x) { return ::sqrtf(x); }
__device__ __forceinline__ double m_exp(double x) { return ::exp(x); }
__device__ __forceinline__ float m_exp(float x) { return ::expf(x); }
__device__ __forceinline__ double m_tanh(double x) { return ::tanh(x); }
__device__ __forceinline__ float m_tanh(float x) { return ::tanhf(x); }
__device__ __forceinline__ double m_abs(double x) { return ::fabs(x); }
__device__ __forceinline__ float m_abs(float x) { return ::fabsf(x); }
__device__ __forceinline__ double m_max(double a, double b) { return ::fmax(a, b); }
__device__ __forceinline__ float m_max(float a, float b) { return ::fmaxf(a, b); }
__device__ __forceinline__ double m_min(double a, double b) { return ::fmin(a, b); }
__device__ __forceinline__ float m_min(float a, float b) { return ::fminf(a, b); }
__device__ __forceinline__ void m_sincos(double x, double* s, double* c) { ::sincos(x, s, c); }
__device__ __forceinline__ void m_sincos(float x, float* s, float* c) { ::sincosf(x, s, c); }

// Reciprocal and reciprocal square root without the IEEE scaling / fix-up sequences of `1/x` and sqrt():
// hardware seed + Newton steps in fma form, ~1 ulp, for arguments well inside the normal range (all call sites:
// distances, radii, metrics -- O(1e-6 .. 1e12)).  The parity tolerance is 1e-9; tests/test_gpu_parity.py
// checks both against correctly rounded division.
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);  // 24-bit seed (4.5e-8 measured, tools/seed_accuracy.hip)
  double e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);       // 2e-15
}
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }  // 9e-8 measured
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);  // 5.2e-8 measured
  double e = __builtin_fma(-x * y, y, 1.0);
  return __builtin_fma(y, 0.5 * e, y);  // 4e-15
}
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }  // 9.3e-8 measured

// exp for |x| < 700 without the overflow / subnormal paths of the library version (all arguments here are
// bounded: -(a r)^2, -s x with x a barrier distance, tanh arguments): x = n ln2 + r, Taylor to r^12 (|r| <= 0.35).
// A float64 literal cannot be an operand of a vector instruction: the compiler builds it in a VGPR pair, hoists that out of
// every loop and, in a kernel that is short of registers, keeps it in scratch memory (the exp polynomial alone is 13 of
// them; k_rollout_panda_wp reloaded ~60 per step).  MRF_SC(v) builds the literal in an SGPR pair instead (two s_mov_b32,
// opaque to the optimizer), which a VOP3 instruction reads directly as its one scalar operand.
template <long long BITS>
__device__ __forceinline__ double sgpr_const() {
  int lo, hi;
  asm("s_mov_b32 %0, %1" : "=s"(lo) : "n"((int)(BITS & 0xffffffffLL)));
  asm("s_mov_b32 %0, %1" : "=s"(hi) : "n"((int)(BITS >> 32)));
  return __hiloint2double(hi, lo);
}
#ifdef MRF_SGPR_CONST
#define MRF_SC(v) (::mrf::sgpr_const<__builtin_bit_cast(long long, (double)(v))>())
#else
#define MRF_SC(v) (v)  // one wave per SIMD: the literals sit in AGPRs, and every s_mov would cost that wave an issue slot
#endif

__device__ __forceinline__ double fast_exp(double x) {
  const double n = __builtin_rint(x * MRF_SC(1.4426950408889634));
  double r = __builtin_fma(-n, MRF_SC(0.6931471803691238), x);
  r = __builtin_fma(-n, MRF_SC(1.9082149292705877e-10), r);
  double p = MRF_SC(1.0 / 479001600.0);
  p = __builtin_fma(p, r, MRF_SC(1.0 / 39916800.0));
  p = __builtin_fma(p, r, MRF_SC(1.0 / 3628800.0));
  p = __builtin_fma(p, r, MRF_SC(1.0 / 362880.0));
  p = __builtin_fma(p, r, MRF_SC(1.0 / 40320.0));
  p = __builtin_fma(p, r, MRF_SC(1.0 / 5040.0));
  p = __builtin_fma(p, r, MRF_SC(1.0 / 720.0));
  p = __builtin_fma(p, r, MRF_SC(1.0 / 120.0));
  p = __builtin_fma(p, r, MRF_SC(1.0 / 24.0));
  p = __builtin_fma(p, r, MRF_SC(1.0 / 6.0));
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)n);
}
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
// tanh(x) = sign(x) (1 - t)/(1 + t), t = exp(-2|x|) <= 1: absolute error ~1e-16 (the relative error near 0 does
// not matter at the call sites: attractor force ~ tanh(a r) x/r, damper switches ~ tanh(O(1)))
template <typename T>
__device__ __forceinline__ T fast_tanh(T x) {
  const T t = fast_exp(T(-2) * (x < T(0) ? -x : x));
  const T v = (T(1) - t) * fast_rcp(T(1) + t);
  return x < T(0) ? -v : v;
}

// sin/cos of a small angle |d| < 0.125 by Taylor series (truncation < 3e-18): used to advance cos q, sin q by
// dq = dt*qdot inside a rollout instead of a full-range sincos per joint per step.
template <typename T>
__device__ __forceinline__ void small_sincos(T d, T& s, T& c) {
  const T u = d * d;
  T ps = T(-1.0 / 39916800.0);
  ps = ps * u + T(1.0 / 362880.0);
  ps = ps * u + T(-1.0 / 5040.0);
  ps = ps * u + T(1.0 / 120.0);
  ps = ps * u + T(-1.0 / 6.0);
  ps = ps * u + T(1.0);
  T pc = T(-1.0 / 3628800.0);
  pc = pc * u + T(1.0 / 40320.0);
  pc = pc * u + T(-1.0 / 720.0);
  pc = pc * u + T(1.0 / 24.0);
  pc = pc * u + T(-0.5);
  pc = pc * u + T(1.0);
  s = ps * d;
  c = pc;
}

// keeps the compiler from merging a recomputation with an earlier, identical one (register lifetime control)
// -DMRF_ISA_MARKS plants comment markers in the assembly so that tools/isa_stats.py can attribute instructions
// to phases of the solve (the markers emit no instruction)
#ifdef MRF_ISA_MARKS
#define MRF_MARK(name) asm volatile("; MRFMARK " name)
#else
#define MRF_MARK(name)
#endif

__device__ __forceinline__ void opaque(double& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void opaque(float& x) { asm volatile("" : "+v"(x)); }

template <typename T>
__device__ __forceinline__ void cross3(const T* a, const T* b, T* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T>
__device__ __forceinline__ T dot3(const T* a, const T* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

// x^p for a wave-uniform integer 0 <= p <= 16
template <typename T>
__device__ __forceinline__ T powi(T x, int p) {
  T x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, r = T(1);
  if (p & 1) r *= x;
  if (p & 2) r *= x2;
  if (p & 4) r *= x4;
  if (p & 8) r *= x8;
  if (p & 16) r *= x8 * x8;
  return r;
}

template <typename T>
__device__ __forceinline__ T gate_value(int gate, T xd) {
  return gate == MRF_GATE_NONE ? T(1) : (xd < T(0) ? T(1) : (xd > T(0) ? T(0) : T(0.5)));
}

// coefficient in front of xdot^2 of a leaf string; ix = 1/x is shared between the two strings of a leaf
template <typename T>
__device__ __forceinline__ T leaf_coeff(const LeafFn<T>& f, T x, T ix, T xd) {
  T g = gate_value<T>(f.gate, xd);
  T v;
  if (f.family == MRF_FAMILY_POW)
    v = f.k * powi(ix, f.p);
  else
    v = f.k * (fast_rcp(T(1) + f.c * fast_exp(-f.s * x)) - T(1));
  return v * g;
}

// metric m = d2L/dxd2 and force f = m*h of a scalar barrier leaf at (x, xd)
template <typename T>
__device__ __forceinline__ void scalar_leaf(const LeafFn<T>& geo, const LeafFn<T>& fin, T x, T xd, T& m, T& f) {
  T ix = fast_rcp(x);
  m = T(2) * leaf_coeff(fin, x, ix, xd);
  f = m * leaf_coeff(geo, x, ix, xd) * xd * xd;
}

// Compile-time scalar leaves (plane and joint-limit leaves).  SLeaf<FAMG,PG,GG,PL,GL>: geometry of family FAMG
// (POW exponent PG or LOGISTIC) with gate GG, Finsler k/x^PL with gate GL; constants k, c, s stay runtime.
struct SLeafGeneric {
  static constexpr bool generic = true;
};
template <int FAMG, int PG, int GG, int PL, int GL>
struct SLeaf {
  static constexpr bool generic = false;
  static constexpr int famg = FAMG, pg = PG, gg = GG, pl = PL, gl = GL;
};

template <int P, typename T>
__device__ __forceinline__ T cpow(T x) {  // x^P, P compile-time
  if constexpr (P == 0) return T(1);
  else if constexpr (P == 1) return x;
  else if constexpr (P % 2 == 0) { T h = cpow<P / 2>(x); return h * h; }
  else return x * cpow<P - 1>(x);
}

template <class SL, typename T>
__device__ __forceinline__ void scalar_leaf_t(const LeafFn<T>& geo, const LeafFn<T>& fin, T x, T xd, T& m, T& f) {
  if constexpr (SL::generic) {
    scalar_leaf(geo, fin, x, xd, m, f);
  } else {
    const T ix = fast_rcp(x);
    m = T(2) * fin.k * gate_value<T>(SL::gl, xd) * cpow<SL::pl>(ix);
    T hc;
    if constexpr (SL::famg == MRF_FAMILY_POW)
      hc = geo.k * cpow<SL::pg>(ix);
    else
      hc = geo.k * (fast_rcp(T(1) + geo.c * fast_exp(-geo.s * x)) - T(1));
    f = m * (hc * gate_value<T>(SL::gg, xd)) * xd * xd;
  }
}

// A leaf set = the compile-time shape of the three barrier leaf classes of a planner.
template <class C, class P, class L>
struct LeafSet {
  using Collision = C;
  using Plane = P;
  using Limit = L;
};

// ------------------------------------------------------------------------------------ Panda chain
// panda_joint1..7 origins (URDF panda_with_finger.urdf:98-107,150-158,201-209,253-261,326-334,378-386,451-459)
// and the fixed panda_joint8 (:461-465) as an eighth, non-moving "joint".
__device__ constexpr double kPX[8] = {0.0, 0.0, 0.0, 0.0825, -0.0825, 0.0, 0.088, 0.0};
__device__ constexpr double kPY[8] = {0.0, 0.0, -0.316, 0.0, 0.384, 0.0, 0.0, 0.0};
__device__ constexpr double kPZ[8] = {0.333, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.107};
__device__ constexpr int kROLL[8] = {0, -1, 1, 1, -1, 1, 1, 0};  // roll = k * pi/2

template <typename T>
struct PandaKin {
  T o[7][3];   // origin of joint j == origin of panda_link(j+1)
  T z[7][3];   // axis of joint j (world)
  T vo[7][3];  // velocity of o[j]
  T ao[7][3];  // d(J qd)/dq qd of o[j]  (true Jdot qd, no sign convention)
  T p8[3], v8[3], a8[3];  // panda_link8 == panda_hand origin
};

// Unrolled walk of the own robot; all tables are compile-time so zero offsets and quarter-turn rolls fold away.
// JSTOP < 7: the walk ends at the origin of joint JSTOP (o, vo, ao of that joint are set, its axis is not): the part of
// the chain a wave that owns only the proximal collision points needs (k_rollout_panda_wp); cq / sq / qd of joints
// >= JSTOP are not read.
// emit_link(link, X, Y, Z, o, w, al, vo, ao), link = 1..8 (a compile-time constant at every call after unrolling): called
// once the frame, origin, angular velocity / acceleration terms of panda_link<link> are complete -- the point at which the
// rolled walk below emits the spheres attached to that link.  Lets a kernel derive its configured spheres from the SAME
// walk that feeds its pullbacks (round 6: k_action_coupled on small generic tables) instead of a second, rolled walk.
struct NoLinkEmit {
  template <typename T>
  __device__ __forceinline__ void operator()(int, const T*, const T*, const T*, const T*, const T*, const T*, const T*, const T*) const {}
};

template <typename T, int JSTOP = 7, class EmitLink = NoLinkEmit>
__device__ __forceinline__ void panda_walk_own(const T* __restrict__ mount, const T (&cq)[7], const T (&sq)[7],
                                               const T (&qd)[7], PandaKin<T>& K, EmitLink emit_link = EmitLink()) {
  T X[3] = {mount[0], mount[4], mount[8]}, Y[3] = {mount[1], mount[5], mount[9]}, Z[3] = {mount[2], mount[6], mount[10]};
  T o[3] = {mount[3], mount[7], mount[11]};
  T w[3] = {T(0), T(0), T(0)}, al[3] = {T(0), T(0), T(0)}, vo[3] = {T(0), T(0), T(0)}, ao[3] = {T(0), T(0), T(0)};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (kPX[j] != 0.0 || kPY[j] != 0.0 || kPZ[j] != 0.0) {
      T r[3] = {T(0), T(0), T(0)};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (kPX[j] != 0.0) r[k] += T(kPX[j]) * X[k];
        if (kPY[j] != 0.0) r[k] += T(kPY[j]) * Y[k];
        if (kPZ[j] != 0.0) r[k] += T(kPZ[j]) * Z[k];
      }
      T wr[3], wwr[3], ar[3];
      cross3(w, r, wr);
      cross3(w, wr, wwr);
      cross3(al, r, ar);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        o[k] += r[k];
        vo[k] += wr[k];
        ao[k] += ar[k] + wwr[k];
      }
    }
    if (j == 7) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        K.p8[k] = o[k];
        K.v8[k] = vo[k];
        K.a8[k] = ao[k];
      }
      emit_link(8, X, Y, Z, o, w, al, vo, ao);  // panda_link8: fixed joint, the frame of link 7 moved along its z
      break;
    }
    if (JSTOP < 7 && j == JSTOP) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        K.o[j][k] = o[k];
        K.vo[j][k] = vo[k];
        K.ao[j][k] = ao[k];
      }
      break;
    }
    if (kROLL[j] == 1) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        T t = Y[k];
        Y[k] = Z[k];
        Z[k] = -t;
      }
    } else if (kROLL[j] == -1) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        T t = Y[k];
        Y[k] = -Z[k];
        Z[k] = t;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      T xn = cq[j] * X[k] + sq[j] * Y[k];
      T yn = cq[j] * Y[k] - sq[j] * X[k];
      X[k] = xn;
      Y[k] = yn;
      K.o[j][k] = o[k];
      K.z[j][k] = Z[k];
      K.vo[j][k] = vo[k];
      K.ao[j][k] = ao[k];
    }
    T wz[3];
    cross3(w, Z, wz);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      al[k] += qd[j] * wz[k];
      w[k] += qd[j] * Z[k];
    }
    emit_link(j + 1, X, Y, Z, o, w, al, vo, ao);
  }
}

// Rolled walk of a robot whose spheres are wanted: emits (x, v, Jdot qd) of every configured sphere, in
// table order, to `emit(s, x, v, a)`.  State is a handful of named vectors, so the loop over joints stays
// rolled and the consumer is instantiated once.  get(j, c, s, qd) supplies cos q_j, sin q_j, qdot_j.
// LINK_ORIGINS = true is the reference's rollout table (one sphere at the origin of each of the 8 links,
// PM:25-26): the sphere of link j+1 is emitted right after joint j, with no inner loop and no table reads.
template <bool LINK_ORIGINS, typename T, typename Get, typename Emit>
__device__ __forceinline__ void panda_walk_spheres(const DevCfg<T>& cfg, const T* __restrict__ mount, Get get,
                                                   Emit emit) {
  T X[3] = {mount[0], mount[4], mount[8]}, Y[3] = {mount[1], mount[5], mount[9]}, Z[3] = {mount[2], mount[6], mount[10]};
  T o[3] = {mount[3], mount[7], mount[11]};
  T w[3] = {T(0), T(0), T(0)}, al[3] = {T(0), T(0), T(0)}, vo[3] = {T(0), T(0), T(0)}, ao[3] = {T(0), T(0), T(0)};
  int s = 0;
  const int S = LINK_ORIGINS ? 8 : cfg.n_spheres;
  int link_s = (!LINK_ORIGINS && S > 0) ? cfg.sphere_link[0] : 0;
  T off_s[3] = {T(0), T(0), T(0)};
  if (!LINK_ORIGINS) {
    off_s[0] = cfg.sphere_off[0][0];
    off_s[1] = cfg.sphere_off[0][1];
    off_s[2] = cfg.sphere_off[0][2];
  }
#pragma unroll 1
  for (int j = 0; j < 8; ++j) {
    // wave-uniform joint constants by scalar selects (no table load on the critical path of the single wave)
    const T px = j == 3 ? T(0.0825) : (j == 4 ? T(-0.0825) : (j == 6 ? T(0.088) : T(0)));
    const T py = j == 2 ? T(-0.316) : (j == 4 ? T(0.384) : T(0));
    const T pz = j == 0 ? T(0.333) : (j == 7 ? T(0.107) : T(0));
    if (j != 1 && j != 5) {  // joints 2 and 6 have a zero origin offset
      T r[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) r[k] = px * X[k] + py * Y[k] + pz * Z[k];
      T wr[3], wwr[3], ar[3];
      cross3(w, r, wr);
      cross3(w, wr, wwr);
      cross3(al, r, ar);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        o[k] += r[k];
        vo[k] += wr[k];
        ao[k] += ar[k] + wwr[k];
      }
    }
    if (j < 7) {
      const int roll = ((0x6C >> j) & 1) - ((0x12 >> j) & 1);  // +1: j = 2,3,5,6   -1: j = 1,4
      if (roll != 0) {
        const T sg = T(roll);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          T t = Y[k];
          Y[k] = sg * Z[k];
          Z[k] = -sg * t;
        }
      }
      T c, sn, qdj;
      get(j, c, sn, qdj);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        T xn = c * X[k] + sn * Y[k];
        T yn = c * Y[k] - sn * X[k];
        X[k] = xn;
        Y[k] = yn;
      }
      T wz[3];
      cross3(w, Z, wz);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        al[k] += qdj * wz[k];
        w[k] += qdj * Z[k];
      }
    }
    if (LINK_ORIGINS) {
      emit(j, o, vo, ao);
      continue;
    }
    // spheres attached to panda_link(j+1): frame (X,Y,Z,o), angular state (w, al) of that link.
    // The table entry of the *next* sphere is fetched before the current one is consumed, so the scalar-load
    // latency hides behind the leaf arithmetic instead of stalling the single resident wave.
    while (s < S && link_s == j + 1) {
      const T ox = off_s[0], oy = off_s[1], oz = off_s[2];
      const int s_next = s + 1 < S ? s + 1 : s;
      link_s = s + 1 < S ? cfg.sphere_link[s_next] : 0;
      off_s[0] = cfg.sphere_off[s_next][0];
      off_s[1] = cfg.sphere_off[s_next][1];
      off_s[2] = cfg.sphere_off[s_next][2];
      T x[3], v[3], a[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        x[k] = o[k];
        v[k] = vo[k];
        a[k] = ao[k];
      }
      if (!(ox == T(0) && oy == T(0) && oz == T(0))) {  // wave-uniform: the reference's rollout spheres are link origins
        T rr[3], wr[3], wwr[3], ar[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) rr[k] = ox * X[k] + oy * Y[k] + oz * Z[k];
        cross3(w, rr, wr);
        cross3(w, wr, wwr);
        cross3(al, rr, ar);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          x[k] += rr[k];
          v[k] += wr[k];
          a[k] += ar[k] + wwr[k];
        }
      }
      emit(s, x, v, a);
      ++s;
    }
  }
}

// ------------------------------------------------------------------------------------ leaf folding
template <typename T, int NP>
struct EgoAcc {
  T A[NP][6];  // xx xy xz yy yz zz
  T b[NP][3];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int g = 0; g < NP; ++g) {
#pragma unroll
      for (int k = 0; k < 6; ++k) A[g][k] = T(0);
#pragma unroll
      for (int k = 0; k < 3; ++k) b[g][k] = T(0);
    }
  }
};

template <typename T, int NP>
struct EgoPts {
  T p[NP][3];
  T v[NP][3];
  T rb[NP][2];  // body radius of the (up to two) links sharing this point
  int nl[NP];   // number of links sharing the point (Panda: link5 and link6 share an origin)
};

// Collision-leaf policies.  LeafGeneric evaluates the two runtime families of mrf_leaf_fn.  LeafPow<PG,PL,GG,GL>
// is the compile-time form of  h = kg/x^PG * gate * xd^2,  L = kl/x^PL * gate * xd^2  (the reference's Panda
// strings are LeafPow<4,4,NONE,NONE>, EXJ:88-89) written so that no 1/R is needed: with t = 1/(d - R),
//   1/x = R t,   m/R^2 = 2 kl R^(PL-2) t^PL,   f/R = (m/R^2) kg R^(PG-1) t^PG (n.v_rel)^2
struct LeafGeneric {
  static constexpr bool generic = true;
};
template <int PG, int PL, int GG, int GL>
struct LeafPow {
  static constexpr bool generic = false;
  static constexpr int pg = PG, pl = PL, gg = GG, gl = GL;
  static_assert(PL >= 2 && PG >= 1, "LeafPow needs PL >= 2 and PG >= 1");
};

// One spherical-obstacle leaf (d = distance, nv = n.v_rel, R = r_o + r_b, curv = sign*kappa - n.a_o):
//   wm = m/R^2   and   wf = f/R + wm*curv,   the weights of n n^T and of n in the pulled-back (A, b).
// Compile-time leaves: with t = 1/(d - R), u = R t = 1/x and cgnv2 = kg*gate_g*nv^2 (shared by the leaves of a point)
//   wm = (2 kl gate_l) u^(PL-2) t^2 ,   wf = wm * (u^(PG-1) t * cgnv2 + curv)
template <class CL, typename T>
__device__ __forceinline__ void collision_leaf(const DevCfg<T>& cfg, T d, T nv, T R, T curv, T cl, T cgnv2, T& wm, T& wf) {
  if constexpr (CL::generic) {
    T iR = fast_rcp(R);
    T x = d * iR - T(1);
    T xd = nv * iR;
    T m, f;
    scalar_leaf(cfg.cg, cfg.cf, x, xd, m, f);
    wm = m * iR * iR;
    wf = f * iR + wm * curv;
  } else {
    const T t = fast_rcp(d - R);
    const T u = R * t;
    T ug;  // u^(PG-1) t
    if constexpr (CL::pl == 4 && CL::pg == 4) {
      const T ut = u * t;
      wm = cl * (ut * ut);
      ug = (u * u) * ut;
    } else {
      wm = cl * cpow<CL::pl - 2>(u) * (t * t);
      ug = cpow<CL::pg - 1>(u) * t;
    }
    wf = wm * (ug * cgnv2 + curv);
  }
}

// Add the spherical-obstacle leaves of all ego points against one obstacle sphere.
// a_o is the obstacle's reference acceleration as the reference passes it (already sign-carrying).
template <class CL, typename T, int NP>
__device__ __forceinline__ void accumulate_obstacle(const DevCfg<T>& cfg, const EgoPts<T, NP>& E, const T* xo,
                                                    const T* vo, const T* a_o, T ro, bool planar, EgoAcc<T, NP>& acc,
                                                    T mult = T(1)) {  // mult: multiplicity of coincident spheres
#pragma unroll
  for (int g = 0; g < NP; ++g) {
    if (E.nl[g] < 1) continue;  // point without a collision link (ego_link_mask); a compile-time fact on the default set
    T dx[3] = {E.p[g][0] - xo[0], E.p[g][1] - xo[1], planar ? T(0) : E.p[g][2] - xo[2]};
    T vr[3] = {E.v[g][0] - vo[0], E.v[g][1] - vo[1], planar ? T(0) : E.v[g][2] - vo[2]};
    T d2 = dot3(dx, dx);
    T id = fast_rsqrt(d2);
    T d = d2 * id;
    T n[3] = {dx[0] * id, dx[1] * id, dx[2] * id};
    T nv = dot3(n, vr);
    T kap = (dot3(vr, vr) - nv * nv) * id;
    T na = planar ? n[0] * a_o[0] + n[1] * a_o[1] : dot3(n, a_o);
    T curv = cfg.jsign * kap - na;
    T cl = T(0), cgnv2 = T(0);
    if constexpr (!CL::generic) {
      cl = (T(2) * mult) * cfg.cf.k * gate_value<T>(CL::gl, nv);  // sign(xd) == sign(nv), R > 0
      cgnv2 = cfg.cg.k * gate_value<T>(CL::gg, nv) * nv * nv;
    }
    T wM, wf;
    collision_leaf<CL>(cfg, d, nv, ro + E.rb[g][0], curv, cl, cgnv2, wM, wf);
    if (E.nl[g] > 1) {
      T wm2, wf2;
      collision_leaf<CL>(cfg, d, nv, ro + E.rb[g][1], curv, cl, cgnv2, wm2, wf2);
      wM += wm2;
      wf += wf2;
    }
    if constexpr (CL::generic) {
      wM *= mult;
      wf *= mult;
    }
    T w0 = wM * n[0], w1 = wM * n[1], w2 = wM * n[2];
    acc.A[g][0] += w0 * n[0];
    acc.A[g][1] += w0 * n[1];
    acc.A[g][2] += w0 * n[2];
    acc.A[g][3] += w1 * n[1];
    acc.A[g][4] += w1 * n[2];
    acc.A[g][5] += w2 * n[2];
    acc.b[g][0] += wf * n[0];
    acc.b[g][1] += wf * n[1];
    acc.b[g][2] += wf * n[2];
  }
}

// plane-constraint leaves of all ego points:  x = |a.p + d|/|a| - r_body
template <class SL, typename T, int NP>
__device__ __forceinline__ void accumulate_plane(const DevCfg<T>& cfg, const EgoPts<T, NP>& E, const T* con,
                                                 EgoAcc<T, NP>& acc) {
  T ina = fast_rsqrt(con[0] * con[0] + con[1] * con[1] + con[2] * con[2]);
#pragma unroll
  for (int g = 0; g < NP; ++g) {
    T val = (con[0] * E.p[g][0] + con[1] * E.p[g][1] + con[2] * E.p[g][2] + con[3]) * ina;
    T sg = T(1);
    if (cfg.plane_abs) sg = val < T(0) ? T(-1) : (val > T(0) ? T(1) : T(0));
    T n[3] = {sg * con[0] * ina, sg * con[1] * ina, sg * con[2] * ina};
    T xd = dot3(n, E.v[g]);
    T wM = T(0), wf = T(0);
#pragma unroll
    for (int l = 0; l < 2; ++l) {
      if (l < E.nl[g]) {
        T x = sg * val - E.rb[g][l];
        T m, f;
        scalar_leaf_t<SL>(cfg.pg, cfg.pf, x, xd, m, f);
        wM += m;
        wf += f;
      }
    }
    T w0 = wM * n[0], w1 = wM * n[1], w2 = wM * n[2];
    acc.A[g][0] += w0 * n[0];
    acc.A[g][1] += w0 * n[1];
    acc.A[g][2] += w0 * n[2];
    acc.A[g][3] += w1 * n[1];
    acc.A[g][4] += w1 * n[2];
    acc.A[g][5] += w2 * n[2];
    acc.b[g][0] += wf * n[0];
    acc.b[g][1] += wf * n[1];
    acc.b[g][2] += wf * n[2];
  }
}

// ------------------------------------------------------------------------------------ configuration-space spec
template <int N>
__host__ __device__ constexpr int tri(int i, int j) {  // upper triangle, i <= j
  return i * N - (i * (i - 1)) / 2 + (j - i);
}

template <typename T, int N>
struct QSpec {
  T M[N * (N + 1) / 2];
  T f[N];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int k = 0; k < N * (N + 1) / 2; ++k) M[k] = T(0);
#pragma unroll
    for (int k = 0; k < N; ++k) f[k] = T(0);
  }
};

// M_q += J^T A J ; f_q += J^T t  for a point with NC leading non-zero Jacobian columns J[:,j] = z_j x (p - o_j)
template <typename T, int NC>
__device__ __forceinline__ void pull_point(QSpec<T, 7>& S, const PandaKin<T>& K, const T* p, const T* A6, const T* t) {
  T J[NC][3], AJ[NC][3];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    T r[3] = {p[0] - K.o[j][0], p[1] - K.o[j][1], p[2] - K.o[j][2]};
    cross3(K.z[j], r, J[j]);
    AJ[j][0] = A6[0] * J[j][0] + A6[1] * J[j][1] + A6[2] * J[j][2];
    AJ[j][1] = A6[1] * J[j][0] + A6[3] * J[j][1] + A6[4] * J[j][2];
    AJ[j][2] = A6[2] * J[j][0] + A6[4] * J[j][1] + A6[5] * J[j][2];
    S.f[j] += dot3(J[j], t);
  }
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = i; j < NC; ++j) S.M[tri<7>(i, j)] += dot3(J[i], AJ[j]);
}

// pull_point for an isotropic leaf metric A = a I:  M_q += a J^T J ; f_q += J^T t
template <typename T, int NC>
__device__ __forceinline__ void pull_point_iso(QSpec<T, 7>& S, const PandaKin<T>& K, const T* p, T a, const T* t) {
  T J[NC][3], aJ[NC][3];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    T r[3] = {p[0] - K.o[j][0], p[1] - K.o[j][1], p[2] - K.o[j][2]};
    cross3(K.z[j], r, J[j]);
#pragma unroll
    for (int k = 0; k < 3; ++k) aJ[j][k] = a * J[j][k];
    S.f[j] += dot3(J[j], t);
  }
#pragma unroll
  for (int i = 0; i < NC; ++i)
#pragma unroll
    for (int j = i; j < NC; ++j) S.M[tri<7>(i, j)] += dot3(J[i], aJ[j]);
}

// LDL^T solve of (M + eps I) h = f, M symmetric positive definite, upper-triangle storage (copy is consumed)
template <typename T, int N>
__device__ __forceinline__ void ldl_solve(const QSpec<T, N>& S, T eps, T (&h)[N]) {
  // A = L D L^T with both L_ij and W_ij = L_ij d_j kept, so that every inner term is one fma:
  //   d_j = A_jj - sum_k L_jk W_jk ,   W_ij = A_ij - sum_k L_ik W_jk ,   L_ij = W_ij / d_j
  T L[N * (N + 1) / 2], W[N * (N + 1) / 2];  // entry (i, j), i > j, stored at tri(j, i)
  T dinv[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    T d = S.M[tri<N>(j, j)] + eps;
#pragma unroll
    for (int k = 0; k < j; ++k) d -= L[tri<N>(k, j)] * W[tri<N>(k, j)];
    dinv[j] = fast_rcp(d);
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      T v = S.M[tri<N>(j, i)];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= L[tri<N>(k, i)] * W[tri<N>(k, j)];
      W[tri<N>(j, i)] = v;
      L[tri<N>(j, i)] = v * dinv[j];
    }
  }
  T y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    T v = S.f[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= L[tri<N>(k, i)] * y[k];
    y[i] = v;
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    T v = y[i] * dinv[i];
#pragma unroll
    for (int k = i + 1; k < N; ++k) v -= L[tri<N>(i, k)] * h[k];
    h[i] = v;
  }
}

// attractor leaf quantities on a task value x (dim D): metric scalar 2A and force 2A*grad(psi)
template <typename T, int D>
__device__ __forceinline__ void attractor(const DevCfg<T>& cfg, const T* x, T w, T& twoA, T* f, T& rnorm) {
  T r2 = T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) r2 += x[k] * x[k];
  T r = m_sqrt(r2);
  rnorm = r;
  T ar = cfg.attr_a * r;
  twoA = T(2) * ((cfg.attr_mu - cfg.attr_ml) * fast_exp(-ar * ar) + cfg.attr_ml);
  // grad psi = w k tanh(alpha r) x/r ; 0 at r == 0: for the 1-D attractor that is CasADi's own value (sqrt(sq(x)) -> |x|,
  // derivative sign(x)); for a 3-D task exactly on its goal it is this build's convention (DESIGN.md "deviations" 1)
  T g = r > T(0) ? w * cfg.attr_k * fast_tanh(cfg.attr_alpha * r) * fast_rcp(r) : T(0);
#pragma unroll
  for (int k = 0; k < D; ++k) f[k] = twoA * g * x[k];
}

// energization + damping + action; returns qddot and the action
template <typename T, int N>
__device__ __forceinline__ void finish(const DevCfg<T>& cfg, const T (&qd)[N], bool forced, T alpha_g, const T (&hg)[N],
                                       const T (&hf)[N], T xpsi, T (&qdd)[N], T (&act)[N]) {
  T qq = T(0);
#pragma unroll
  for (int j = 0; j < N; ++j) qq += qd[j] * qd[j];
  if (!forced) {
#pragma unroll
    for (int j = 0; j < N; ++j) qdd[j] = -hg[j] - alpha_g * qd[j];
  } else {
    T qh = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j) qh += qd[j] * hf[j];
    T alpha_f = -qh * fast_rcp(cfg.eps + qq);
    T eta = T(0.5) * (fast_tanh(-cfg.eta_a * qq - cfg.eta_s) + T(1));
    T a_ex = eta * alpha_g + (T(1) - eta) * alpha_f;
    T beta = T(0.5) * (fast_tanh(-cfg.beta_a * (xpsi - cfg.beta_r)) + T(1)) * cfg.beta_b + cfg.beta_s +
             m_max(T(0), alpha_g - a_ex);
#pragma unroll
    for (int j = 0; j < N; ++j) qdd[j] = -hf[j] - (a_ex + beta) * qd[j];
  }
  T nrm = T(0);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    act[j] = cfg.mode == MRF_MODE_VEL ? qd[j] + cfg.dt * qdd[j] : qdd[j];
    nrm += act[j] * act[j];
  }
  if (cfg.zero_small && m_sqrt(nrm) < cfg.eps) {
#pragma unroll
    for (int j = 0; j < N; ++j) act[j] = T(0);
  }
}

// ------------------------------------------------------------------------------------ Panda row solve
// Per-row parameters are read where they are used (coalesced, L2-resident) instead of being held in 29 registers
// across the obstacle loop; the rollout overrides x_goal_0 with its estimate (RF-CV).
template <typename T>
struct PrmView {
  const T* __restrict__ base;
  int64_t rows, r;
  T g0[3];
  bool own_goal;  // x_goal_0 comes from g0 instead of memory
  __device__ __forceinline__ T operator[](int i) const {
    // select chain instead of g0[i]: a runtime index would put g0 into scratch memory
    if (i < 3 && own_goal) return i == 0 ? g0[0] : (i == 1 ? g0[1] : g0[2]);
    return base[i * rows + r];
  }
};

// Row addressing for the kernels whose rows are (first row of the block) + lane: every array element a[c*rows + row]
// is read as  (a + c*rows + first)  [off]  -- a wave-uniform pointer that lives in an SGPR pair and is advanced with scalar
// arithmetic, plus ONE 32-bit lane offset shared by all accesses of the kernel (global_load ... v_off, s[base:base+1]).
// The per-lane 64-bit form (a + row, kept in a VGPR pair per array across the whole step loop) is what put the loop
// invariants of the Cartesian rollout into scratch memory.
template <typename T>
struct RowAddr {
  int64_t rows;   // wave-uniform
  int64_t first;  // wave-uniform: first row of this block
  uint32_t off;   // this lane's row - first (tail lanes clamped onto the last valid row)
  __device__ __forceinline__ const T* at(const T* __restrict__ a, int64_t comp) const { return a + comp * rows + first; }
  __device__ __forceinline__ T* at(T* __restrict__ a, int64_t comp) const { return a + comp * rows + first; }
  __device__ __forceinline__ T load(const T* __restrict__ a, int64_t comp) const { return at(a, comp)[off]; }
  __device__ __forceinline__ void store(T* __restrict__ a, int64_t comp, T v) const { at(a, comp)[off] = v; }
};

// PrmView over RowAddr (same interface; used by k_rollout_cart_panda)
template <typename T>
struct PrmViewU {
  const T* __restrict__ base;
  RowAddr<T> ra;
  T g0[3];
  bool own_goal;
  __device__ __forceinline__ T operator[](int i) const {
    if (i < 3 && own_goal) return i == 0 ? g0[0] : (i == 1 ? g0[1] : g0[2]);
    return ra.load(base, i);
  }
};

// a wave-uniform value computed by vector instructions, moved to an SGPR pair (frees the VGPR pair it would occupy
// across a loop; VOP3 reads it as a scalar operand)
__device__ __forceinline__ double to_uniform(double x) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float to_uniform(float x) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}

template <typename T>
struct PandaState {
  T q[7], qd[7], cq[7], sq[7];
};

// ---------------------------------------------------------------------------- row state
// all fourteen loads first: sincos has branches (argument reduction) the compiler does not move loads across, and a
// load per joint in front of its sincos is seven dependent round trips to HBM at the start of every row
template <typename T>
__device__ __forceinline__ void load_state_values(int64_t rows, int64_t r, const T* __restrict__ q, const T* __restrict__ qd,
                                                  PandaState<T>& R) {
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    R.q[j] = q[j * rows + r];
    R.qd[j] = qd[j * rows + r];
  }
}
template <typename T>
__device__ __forceinline__ void state_sincos(PandaState<T>& R) {
#pragma unroll
  for (int j = 0; j < 7; ++j) m_sincos(R.q[j], &R.sq[j], &R.cq[j]);
}
template <typename T>
__device__ __forceinline__ void load_state(int64_t rows, int64_t r, const T* __restrict__ q, const T* __restrict__ qd,
                                           PandaState<T>& R) {
  load_state_values(rows, r, q, qd, R);
  state_sincos(R);
}

// Body radii of the links that share ego point g (0: link 3, 1: link 4, 2: links 5 and 6, 3: link 7, 4: link 8).
// MASKED = false is the examples' full set (all six links, compile-time counts); MASKED = true reads
// cfg.ego_mask (collision_links subsets, EXJ:91-96 / FPC:20-21) and runs in the runtime-family instantiations.
template <bool MASKED, typename T, class PRM>
__device__ __forceinline__ void ego_point_links(const DevCfg<T>& cfg, const PRM& prm, int g, T& rb0, T& rb1, int& nl) {
  if (g == 2) {
    const T r5 = prm[MRF_P_RADIUS_BODY + 2], r6 = prm[MRF_P_RADIUS_BODY + 3];
    if constexpr (MASKED) {
      const int b5 = (cfg.ego_mask >> 2) & 1, b6 = (cfg.ego_mask >> 3) & 1;
      rb0 = b5 ? r5 : r6;
      rb1 = r6;
      nl = b5 + b6;
    } else {
      rb0 = r5;
      rb1 = r6;
      nl = 2;
    }
    return;
  }
  const int e = g < 2 ? g : g + 1;  // index into radius_body / bit of the mask
  rb0 = prm[MRF_P_RADIUS_BODY + e];
  rb1 = T(0);
  nl = MASKED ? ((cfg.ego_mask >> e) & 1) : 1;
}

template <bool MASKED, typename T, class PRM>
__device__ __forceinline__ void panda_ego_points(const DevCfg<T>& cfg, const PandaKin<T>& K, const PRM& prm,
                                                 EgoPts<T, NG>& E) {
  constexpr int jo[4] = {2, 3, 4, 6};  // joint-origin index of links 3, 4, 5(=6), 7
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      E.p[g][k] = K.o[jo[g]][k];
      E.v[g][k] = K.vo[jo[g]][k];
    }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    E.p[4][k] = K.p8[k];
    E.v[4][k] = K.v8[k];
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) ego_point_links<MASKED>(cfg, prm, g, E.rb[g][0], E.rb[g][1], E.nl[g]);
}

// Everything after the obstacle loop: plane + pullbacks + limits + attractors + solves + damping.
// PLANE_DONE: the caller has already added the plane leaves to acc (the cooperative kernels do it per lane).
template <class LS, bool PLANE_DONE = false, typename T, class PRM>
__device__ __forceinline__ void panda_finish_row(const DevCfg<T>& cfg, const PandaState<T>& R, const PRM& prm,
                                                 const PandaKin<T>& K, const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc,
                                                 T (&qdd)[7], T (&act)[7]) {
  QSpec<T, 7> S;
  S.zero();
#pragma unroll
  for (int j = 0; j < 7; ++j) S.M[tri<7>(j, j)] = cfg.base_mass;
  if (cfg.n_ego > 0) {
    if (!PLANE_DONE && cfg.n_planes > 0) {
      T con[4] = {prm[MRF_P_CONSTRAINT_0], prm[MRF_P_CONSTRAINT_0 + 1], prm[MRF_P_CONSTRAINT_0 + 2], prm[MRF_P_CONSTRAINT_0 + 3]};
      accumulate_plane<typename LS::Plane>(cfg, E, con, acc);
    }
    constexpr int jo[4] = {2, 3, 4, 6};
    // t = b + A c with c = jsign * Jdot qd of the point
#define MRF_PULL(g, NC, pp, aa)                                                                \
  {                                                                                            \
    T c[3] = {cfg.jsign * (aa)[0], cfg.jsign * (aa)[1], cfg.jsign * (aa)[2]};                  \
    T t[3] = {acc.b[g][0] + acc.A[g][0] * c[0] + acc.A[g][1] * c[1] + acc.A[g][2] * c[2],      \
              acc.b[g][1] + acc.A[g][1] * c[0] + acc.A[g][3] * c[1] + acc.A[g][4] * c[2],      \
              acc.b[g][2] + acc.A[g][2] * c[0] + acc.A[g][4] * c[1] + acc.A[g][5] * c[2]};     \
    pull_point<T, NC>(S, K, pp, acc.A[g], t);                                                  \
  }
    MRF_MARK("plane");
    MRF_PULL(0, 2, K.o[jo[0]], K.ao[jo[0]])
    MRF_PULL(1, 3, K.o[jo[1]], K.ao[jo[1]])
    MRF_PULL(2, 4, K.o[jo[2]], K.ao[jo[2]])
    MRF_PULL(3, 6, K.o[jo[3]], K.ao[jo[3]])
    MRF_PULL(4, 6, K.p8, K.a8)
#undef MRF_PULL
  }
  MRF_MARK("pull");
  if (cfg.use_limits) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      T m, f;
      scalar_leaf_t<typename LS::Limit>(cfg.lg, cfg.lf, R.q[j] - cfg.limits[j][0], R.qd[j], m, f);
      S.M[tri<7>(j, j)] += m;
      S.f[j] += f;
      scalar_leaf_t<typename LS::Limit>(cfg.lg, cfg.lf, cfg.limits[j][1] - R.q[j], -R.qd[j], m, f);
      S.M[tri<7>(j, j)] += m;
      S.f[j] -= f;
    }
  }
  MRF_MARK("limits");
  T hg[7], hf[7];
  ldl_solve<T, 7>(S, cfg.eps, hg);
  MRF_MARK("ldl_geometry");
  T qq = T(0), qh = T(0);
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    qq += R.qd[j] * R.qd[j];
    qh += R.qd[j] * hg[j];
  }
  T alpha_g = -qh * fast_rcp(cfg.eps + qq);
  T xpsi = T(0);
  const bool forced = cfg.n_goals > 0;
  if (forced) {
    // attractor 0: panda_hand position -> x_goal_0   (EXJ:32-41)
    {
      T x0[3] = {K.p8[0] - prm[MRF_P_X_GOAL_0], K.p8[1] - prm[MRF_P_X_GOAL_0 + 1], K.p8[2] - prm[MRF_P_X_GOAL_0 + 2]};
      T twoA, f0[3];
      attractor<T, 3>(cfg, x0, prm[MRF_P_WEIGHT_GOAL_0], twoA, f0, xpsi);
      T t[3] = {f0[0] + twoA * cfg.jsign * K.a8[0], f0[1] + twoA * cfg.jsign * K.a8[1], f0[2] + twoA * cfg.jsign * K.a8[2]};
      pull_point_iso<T, 6>(S, K, K.p8, twoA, t);
    }
    MRF_MARK("attractor0");
    if (cfg.n_goals > 1) {
      // attractor 1: R (p_hand - p_link7) -> x_goal_1 ; p_hand - p_link7 = 0.107 z_6   (EXJ:42-52)
      T Rm[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Rm[i] = prm[MRF_P_ANGLE_GOAL_1 + i];
      T d8[3] = {K.p8[0] - K.o[6][0], K.p8[1] - K.o[6][1], K.p8[2] - K.o[6][2]};
      T da[3] = {K.a8[0] - K.ao[6][0], K.a8[1] - K.ao[6][1], K.a8[2] - K.ao[6][2]};
      T x1[3], c1[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        x1[i] = Rm[3 * i] * d8[0] + Rm[3 * i + 1] * d8[1] + Rm[3 * i + 2] * d8[2] - prm[MRF_P_X_GOAL_1 + i];
        c1[i] = cfg.jsign * (Rm[3 * i] * da[0] + Rm[3 * i + 1] * da[1] + Rm[3 * i + 2] * da[2]);
      }
      T twoA, f1[3], rn;
      attractor<T, 3>(cfg, x1, prm[MRF_P_WEIGHT_GOAL_1], twoA, f1, rn);
      T t[3] = {f1[0] + twoA * c1[0], f1[1] + twoA * c1[1], f1[2] + twoA * c1[2]};
      T J[6][3];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        T cz[3];
        cross3(K.z[j], d8, cz);  // d(p8 - o6)/dq_j
#pragma unroll
        for (int i = 0; i < 3; ++i) J[j][i] = Rm[3 * i] * cz[0] + Rm[3 * i + 1] * cz[1] + Rm[3 * i + 2] * cz[2];
        S.f[j] += dot3(J[j], t);
      }
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) S.M[tri<7>(i, j)] += twoA * dot3(J[i], J[j]);
    }
    if (cfg.n_goals > 2) {
      // attractor 2: joint index 6 -> x_goal_2   (EXJ:53-60)
      T x2[1] = {R.q[6] - prm[MRF_P_X_GOAL_2]};
      T twoA, f2[1], rn;
      attractor<T, 1>(cfg, x2, prm[MRF_P_WEIGHT_GOAL_2], twoA, f2, rn);
      S.M[tri<7>(6, 6)] += twoA;
      S.f[6] += f2[0];
    }
    MRF_MARK("attractor12");
    ldl_solve<T, 7>(S, cfg.eps, hf);
    MRF_MARK("ldl_forced");
  } else {
#pragma unroll
    for (int j = 0; j < 7; ++j) hf[j] = hg[j];
  }
  finish<T, 7>(cfg, R.qd, forced, alpha_g, hg, hf, xpsi, qdd, act);
  MRF_MARK("finish");
}

// One fabric solve of a Panda row.  `obstacles(E, acc)` adds the spherical-obstacle leaves.  Two forms, chosen per
// kernel (register pressure is what limits these kernels): SINGLE_WALK = false walks the own chain twice -- once
// for the ego points the obstacle loop needs, once afterwards for the joint axes / origins / curvature terms of the
// pullback -- because recomputing ~400 instructions is cheaper than spilling ~60 values around a heavy loop.
struct NoPublish {
  template <typename K>
  __device__ __forceinline__ void operator()(const K&) const {}
};

// SINGLE_WALK = true keeps the own chain's joint axes / origins / curvature terms alive across the obstacle loop
// (in AGPRs) instead of re-walking the chain afterwards: ~800 fewer instructions per solve where the loop is light
// enough not to spill (measured per kernel with -Rpass-analysis: the link-origin tile loop, the HBM obstacle loop);
// the kernels whose loop walks other robots' chains keep the two-phase form.
template <class LS, bool SINGLE_WALK, typename T, class PRM, class Obst, class Publish = NoPublish, class EmitLink = NoLinkEmit>
__device__ __forceinline__ void panda_solve_row(const DevCfg<T>& cfg, const T* __restrict__ mount, const PandaState<T>& R,
                                                const PRM& prm, Obst obstacles, T (&qdd)[7], T (&act)[7],
                                                Publish publish = Publish(), EmitLink emit_link = EmitLink()) {
  if constexpr (SINGLE_WALK) {
    MRF_MARK("integrate");
    PandaKin<T> K;
    panda_walk_own<T, 7>(mount, R.cq, R.sq, R.qd, K, emit_link);
    EgoPts<T, NG> E;
    panda_ego_points<LS::Collision::generic>(cfg, K, prm, E);
    MRF_MARK("walk");
    publish(K);
    EgoAcc<T, NG> acc;
    acc.zero();
    MRF_MARK("publish");
    if (cfg.n_ego > 0) obstacles(E, acc);
    MRF_MARK("obstacles");
    panda_finish_row<LS>(cfg, R, prm, K, E, acc, qdd, act);
    return;
  }
  EgoPts<T, NG> E;
  {
    PandaKin<T> K1;
    panda_walk_own<T, 7>(mount, R.cq, R.sq, R.qd, K1, emit_link);
    panda_ego_points<LS::Collision::generic>(cfg, K1, prm, E);
    publish(K1);  // coupled kernels: hand this robot's link states to the other lanes of the scenario
  }
  EgoAcc<T, NG> acc;
  acc.zero();
  if (cfg.n_ego > 0) obstacles(E, acc);
  PandaState<T> R2 = R;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    opaque(R2.cq[j]);
    opaque(R2.sq[j]);
  }
  PandaKin<T> K;
  panda_walk_own<T>(mount, R2.cq, R2.sq, R2.qd, K);
  panda_finish_row<LS>(cfg, R, prm, K, E, acc, qdd, act);
}

// Software pipeline of depth one for loops whose operands come from memory while a single wave per SIMD has nothing
// else to overlap the load latency with: the operands of item m+1 are fetched into the other register buffer before
// item m is folded (loop unrolled by two, ping-pong buffers, no copies).  fetch(m, buf) loads, fold(m, buf) consumes.
// Deeper rings (3..6 buffers, loop unrolled by the depth) were measured and lose: the extra buffers push the solve
// into scratch (k_action_panda 0.20 ms at depth one, 0.28 / 0.33 / 0.34 / 0.57 ms as rings of 2 / 3 / 4 / 6).
template <typename T, int NV, class Fetch, class Fold>
__device__ __forceinline__ void pipelined_pairs(int n, Fetch fetch, Fold fold) {
  if (n <= 0) return;
  T A[NV], B[NV];
  fetch(0, A);
  int m = 0;
#pragma unroll 1
  for (; m + 1 < n; m += 2) {
    fetch(m + 1, B);
    fold(m, A);
    fetch(m + 2 < n ? m + 2 : m + 1, A);  // past the end: re-reads the last item, never folded
    fold(m + 1, B);
  }
  if (m < n) fold(m, A);
}

// ------------------------------------------------------------------------------------ link-origin sphere table
// slot of link-origin sphere sp (0..7) in the tile once coincident spheres are merged (DevCfg::lo_merge*)
__host__ __device__ __forceinline__ int lo_slot(int sp, int m01, int m45) { return sp - (sp >= 1 ? m01 : 0) - (sp >= 5 ? m45 : 0); }
// first sphere of a slot and the number of spheres merged into it
__host__ __device__ __forceinline__ int lo_sphere(int slot, int m01, int m45) {
  const int sp = slot + (slot >= 1 ? m01 : 0);
  return sp + (sp >= 5 ? m45 : 0);
}
__host__ __device__ __forceinline__ int lo_count(int slot, int m01, int m45) {
  return ((slot == 0 && m01) || (slot == 4 - m01 && m45)) ? 2 : 1;
}

// The link-origin kernels keep the own chain's kinematics alive across the sphere loop (single walk) only with the
// compile-time leaf policies; the runtime-family leaves need the registers, there the chain is re-walked instead
// (the single-walk form spilled 544 B of scratch per lane in the generic instantiation).
template <class LS>
constexpr bool kSingleWalk = !LS::Collision::generic;


// ------------------------------------------------------------------------------------ planar point robot
// pointRobot1.urdf:91-113: prismatic x (origin z 0.05), prismatic y, revolute theta; collision link base_link.
template <typename T>
struct PlanarRow {
  T q[3], qd[3];
  T prm[MRF_NPARAM];
};

template <typename T>
__device__ __forceinline__ void planar_ego(const PlanarRow<T>& R, EgoPts<T, 1>& E) {
  E.p[0][0] = R.q[0]; E.p[0][1] = R.q[1]; E.p[0][2] = T(0.05);
  E.v[0][0] = R.qd[0]; E.v[0][1] = R.qd[1]; E.v[0][2] = T(0);
  E.rb[0][0] = R.prm[MRF_P_RADIUS_BODY]; E.rb[0][1] = T(0); E.nl[0] = 1;
}

template <typename T>
__device__ __forceinline__ void planar_finish_row(const DevCfg<T>& cfg, const PlanarRow<T>& R, EgoAcc<T, 1>& acc,
                                                  T (&qdd)[3], T (&act)[3]) {
  // J = [e_x e_y 0] (3x3, third column zero), Jdot = 0
  QSpec<T, 3> S;
  S.zero();
#pragma unroll
  for (int j = 0; j < 3; ++j) S.M[tri<3>(j, j)] = cfg.base_mass;
  if (cfg.n_ego > 0) {
    S.M[tri<3>(0, 0)] += acc.A[0][0];
    S.M[tri<3>(0, 1)] += acc.A[0][1];
    S.M[tri<3>(1, 1)] += acc.A[0][3];
    S.f[0] += acc.b[0][0];
    S.f[1] += acc.b[0][1];
  }
  T hg[3], hf[3];
  ldl_solve<T, 3>(S, cfg.eps, hg);
  T qq = T(0), qh = T(0);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    qq += R.qd[j] * R.qd[j];
    qh += R.qd[j] * hg[j];
  }
  T alpha_g = -qh * fast_rcp(cfg.eps + qq);
  T xpsi = T(0);
  const bool forced = cfg.n_goals > 0;
  if (forced) {
    T x0[2] = {R.q[0] - R.prm[MRF_P_X_GOAL_0], R.q[1] - R.prm[MRF_P_X_GOAL_0 + 1]};
    T twoA, f0[2];
    attractor<T, 2>(cfg, x0, R.prm[MRF_P_WEIGHT_GOAL_0], twoA, f0, xpsi);
    S.M[tri<3>(0, 0)] += twoA;
    S.M[tri<3>(1, 1)] += twoA;
    S.f[0] += f0[0];
    S.f[1] += f0[1];
    ldl_solve<T, 3>(S, cfg.eps, hf);
  } else {
#pragma unroll
    for (int j = 0; j < 3; ++j) hf[j] = hg[j];
  }
  finish<T, 3>(cfg, R.qd, forced, alpha_g, hg, hf, xpsi, qdd, act);
}

}  // inline namespace
}  // namespace mrf
