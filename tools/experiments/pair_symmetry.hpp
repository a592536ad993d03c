// experiments/pair_symmetry.hpp -- NOT part of the shipped library.  Included by mrf_kernels.hip only under
// -DMRF_PAIR_SYMMETRY (tools/build_variant.sh pair - -DMRF_PAIR_SYMMETRY [-DMRF_PAIR_TWO_WALKS | -DMRF_PAIR_SCHED_BARRIER]).
// Round-4 experiment asked for by VERDICT r3 next-5; result in profiles/r04_experiments.json: bit-for-bit the same
// rollouts (5e-16), 3.93-4.92 ms against the shipped kernel's 3.10 ms.  Kept so that the number can be reproduced.
#pragma once

// Pair-symmetric form of the link-origin sphere loop (build switch -DMRF_PAIR_SYMMETRY).
// The ego points of a Panda (origins of links 3, 4, 5=6, 7, 8) ARE its sphere slots 1..5, so the leaf of (robot A, ego
// point g) against (robot B, slot t+1) and the leaf of (robot B, ego point t) against (robot A, slot g+1) belong to the
// same pair of points: distance d, unit normal n (up to its sign), n.v_rel and the curvature term kappa are shared;
// only the radii, the obstacle's own acceleration term and the accumulators differ.  For an odd number of robots every
// lane is FIRST towards the robot `off` places after it and SECOND towards the one `off` places before it
// (off = 1 .. (N-1)/2): as first it evaluates the 25 point pairs completely (as the plain loop does) and leaves
// (n, d, n.v_rel, kappa) in registers; as second it pulls those six numbers from the first lane's registers
// (ds_bpermute, no LDS storage) and only finishes its own leaf.  Slot 0 (links 1 and 2, which no robot has as an ego
// point) keeps the plain form.  Requires both coincident pairs merged and the compile-time leaf policy.
template <typename T>
__device__ __forceinline__ T lane_pull(int byte_addr, T v);
template <>
__device__ __forceinline__ double lane_pull<double>(int byte_addr, double v) {
  const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <>
__device__ __forceinline__ float lane_pull<float>(int byte_addr, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

template <class CL, typename T>
__device__ __forceinline__ void obstacles_from_tile_paired(const DevCfg<T>& cfg, const T* __restrict__ tile, int ls, int li,
                                                           int N, const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
  static_assert(!CL::generic, "pair-symmetric loop: compile-time leaf policy only");
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  const int col0 = ls * N;
  // ---- slot 0 of every other robot: the plain fold
#pragma unroll 1
  for (int off = 1; off < N; ++off) {
    int jr = li + off;
    if (jr >= N) jr -= N;
    lds_ptr src = (lds_ptr)(tile + col0 + jr);
    T b[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) b[k] = src[k * 64];
    accumulate_obstacle<CL>(cfg, E, b, b + 3, b + 6, ((lds_ptr)tile)[TILE_RADII], false, acc, ((lds_ptr)tile)[TILE_MULT]);
  }
  // ---- slots 1..5: shared point pairs
#pragma unroll 1
  for (int off = 1; 2 * off < N; ++off) {
    int jf = li + off;  // I am first towards robot jf ...
    if (jf >= N) jf -= N;
    int js = li - off;  // ... and second towards robot js
    if (js < 0) js += N;
    const int pull = (col0 + js) * 4;
#pragma unroll 1
    for (int t = 0; t < NG; ++t) {
      // as first: sphere slot t+1 of robot jf
      lds_ptr s1 = (lds_ptr)(tile + ((t + 1) * 9) * 64 + col0 + jf);
      T xs[3], vs[3], as[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        xs[k] = s1[k * 64];
        vs[k] = s1[(3 + k) * 64];
        as[k] = s1[(6 + k) * 64];
      }
      const T rs = ((lds_ptr)tile)[TILE_RADII + t + 1], ms = ((lds_ptr)tile)[TILE_MULT + t + 1];
      // as second: my ego point t (body radii by a select chain: t is the loop counter, E lives in registers)
      const T rbt = t == 0 ? E.rb[0][0] : (t == 1 ? E.rb[1][0] : (t == 2 ? E.rb[2][0] : (t == 3 ? E.rb[3][0] : E.rb[4][0])));
      T tA[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, tb[3] = {T(0), T(0), T(0)};
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        // ---------------- first: (my ego point g) x (robot jf, slot t+1), complete
        const T dx[3] = {E.p[g][0] - xs[0], E.p[g][1] - xs[1], E.p[g][2] - xs[2]};
        const T vr[3] = {E.v[g][0] - vs[0], E.v[g][1] - vs[1], E.v[g][2] - vs[2]};
        const T d2 = dot3(dx, dx);
        const T id = fast_rsqrt(d2);
        const T dd = d2 * id;
        const T n[3] = {dx[0] * id, dx[1] * id, dx[2] * id};
        const T nv = dot3(n, vr);
        const T kap = (dot3(vr, vr) - nv * nv) * id;
        {
          const T curv = cfg.jsign * kap - dot3(n, as);
          const T cl = (T(2) * ms) * cfg.cf.k * gate_value<T>(CL::gl, nv);
          const T cgnv2 = cfg.cg.k * gate_value<T>(CL::gg, nv) * nv * nv;
          T wM, wf;
          collision_leaf<CL>(cfg, dd, nv, rs + E.rb[g][0], curv, cl, cgnv2, wM, wf);
          if (g == 2) {  // links 5 and 6 share this point
            T wm2, wf2;
            collision_leaf<CL>(cfg, dd, nv, rs + E.rb[g][1], curv, cl, cgnv2, wm2, wf2);
            wM += wm2;
            wf += wf2;
          }
          const T w0 = wM * n[0], w1 = wM * n[1], w2 = wM * n[2];
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
        // ---------------- second: (my ego point t) x (robot js, slot g+1); robot js has just evaluated this pair of
        // points as first (its g, my t): its normal is minus mine
        const T pn[3] = {lane_pull<T>(pull, n[0]), lane_pull<T>(pull, n[1]), lane_pull<T>(pull, n[2])};
        const T pd = lane_pull<T>(pull, dd), pnv = lane_pull<T>(pull, nv), pkap = lane_pull<T>(pull, kap);
        lds_ptr s2 = (lds_ptr)(tile + ((g + 1) * 9 + 6) * 64 + col0 + js);
        const T na2 = -(pn[0] * s2[0] + pn[1] * s2[64] + pn[2] * s2[128]);  // n' . a_o with n' = -pn
        const T curv2 = cfg.jsign * pkap - na2;
        const T rs2 = ((lds_ptr)tile)[TILE_RADII + g + 1];
        const T cl2 = (T(2) * (g == 2 ? T(2) : T(1))) * cfg.cf.k * gate_value<T>(CL::gl, pnv);
        const T cgnv22 = cfg.cg.k * gate_value<T>(CL::gg, pnv) * pnv * pnv;
        T wM, wf;
        collision_leaf<CL>(cfg, pd, pnv, rs2 + rbt, curv2, cl2, cgnv22, wM, wf);
        if (t == 2) {
          T wm2, wf2;
          collision_leaf<CL>(cfg, pd, pnv, rs2 + E.rb[2][1], curv2, cl2, cgnv22, wm2, wf2);
          wM += wm2;
          wf += wf2;
        }
        const T w0 = wM * pn[0], w1 = wM * pn[1], w2 = wM * pn[2];
        tA[0] += w0 * pn[0];
        tA[1] += w0 * pn[1];
        tA[2] += w0 * pn[2];
        tA[3] += w1 * pn[1];
        tA[4] += w1 * pn[2];
        tA[5] += w2 * pn[2];
        tb[0] -= wf * pn[0];
        tb[1] -= wf * pn[1];
        tb[2] -= wf * pn[2];
#ifdef MRF_PAIR_SCHED_BARRIER  // experiment: keep the scheduler from interleaving the five point pairs (register pressure)
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      // the second role's sums belong to my ego point t
#define MRF_ADD_T(G)                                   \
  {                                                    \
    for (int k = 0; k < 6; ++k) acc.A[G][k] += tA[k];  \
    for (int k = 0; k < 3; ++k) acc.b[G][k] += tb[k];  \
  }
      if (t == 0) MRF_ADD_T(0)
      else if (t == 1) MRF_ADD_T(1)
      else if (t == 2) MRF_ADD_T(2)
      else if (t == 3) MRF_ADD_T(3)
      else MRF_ADD_T(4)
#undef MRF_ADD_T
    }
  }
}

