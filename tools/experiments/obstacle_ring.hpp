// EXPERIMENT, not shipped (round 4): the obstacle stream of compute_action prefetched by LDS-DMA.  Included by
// mrf_kernels.hip under -DMRF_OBST_RING only.  Correct (the GPU parity suite passes with it), not faster:
// profiles/r04_experiments.json "k_action_panda obstacle ring" -- the kernel's time is a fixed part + 6.1 us per obstacle
// with the register ping-pong and + 6.5 us per obstacle with this ring, in the one-block-per-64-rows form and in the
// persistent form below alike: the fold of an obstacle (~360 f64 instructions at one wave per SIMD) is what a step of
// the loop costs, not the latency or the bandwidth of its ten loads.
#pragma once

// ---------------------------------------------------------------------------- obstacle ring (LDS-DMA)
// compute_action with its obstacles in HBM arrays is two phases per row: a stream (M obstacles x 10 scalars, at the rate
// HBM delivers them: measured 6.1 us per obstacle and launch = 5.1 TB/s) and arithmetic (chain walk, pullbacks, solve)
// during which the wave loads nothing.  At one wave per SIMD (register-bound) and every wave in step with every other,
// the two phases add up instead of overlapping.  The ring decouples them WITHOUT registers: persistent waves walk the row
// blocks, and the obstacles of the wave's next rows are fetched by global_load_lds_dwordx4 -- straight into a per-wave LDS
// ring, no VGPR destination -- while the current rows are being finished.  One 1 KiB wave-instruction = RING_CPI whole
// components of the wave's 64 rows (the arrays are [.][rows]: a component of 64 consecutive rows is 64 * sizeof(T)
// contiguous bytes = RING_LPC lanes x 16 B).  The wave reads its own ring, so no barrier: a counted s_waitcnt vmcnt
// leaves the younger DMAs in flight (loads retire in order; ordinary loads issued in between only make the wait longer).
// Needs 16-byte aligned arrays, rows % 64 == 0 and 32-bit obstacle strides (host: ring_applies).
typedef __attribute__((address_space(1))) const void* ring_gptr;
typedef __attribute__((address_space(3))) void* ring_lptr;
template <typename T>
constexpr int RING_EPL = 16 / (int)sizeof(T);  // elements per lane and DMA
template <typename T>
constexpr int RING_LPC = 64 / RING_EPL<T>;  // lanes per component
template <typename T>
constexpr int RING_CPI = RING_EPL<T>;  // components per DMA instruction
template <typename T, bool ACC>
constexpr int RING_G = ((ACC ? 10 : 7) + RING_CPI<T> - 1) / RING_CPI<T>;  // DMA instructions per obstacle
template <typename T, bool ACC>
constexpr int RING_SLOT = RING_G<T, ACC> * RING_CPI<T> * 64;  // scalars per ring slot
constexpr int RING_DEPTH = 8;                                 // f64: 8 x 5 KB = 40 KB per wave, four waves per CU

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <bool ACC, typename T, int D_ = RING_DEPTH>
struct ObstRing {
  static constexpr int NV = ACC ? 10 : 7, G = RING_G<T, ACC>, SLOT = RING_SLOT<T, ACC>, D = D_;
  T* ring;  // this wave's [D][G * CPI][64]
  int lane, n_obst;
  const char* src0[G];  // this lane's source of DMA g: obstacle 0 of the wave's first row block
  unsigned stride[G];   // ... its distance to the next obstacle, bytes
  uint64_t block_step;  // ... and to the wave's next row block, bytes (wave-uniform)
  // the wave's items are (row block t, obstacle m) in order; wave-uniform counters
  int64_t total, issued, consumed;
  int it, im;  // (t, m) of the next item to issue
  int slot_issue, slot_read;

  __device__ __forceinline__ ObstRing(T* ring_, int lane_, int64_t rows, int64_t r0, int64_t rows_per_step, int64_t my_blocks,
                                      int n_obst_, const T* __restrict__ ox, const T* __restrict__ ov,
                                      const T* __restrict__ oa, const T* __restrict__ orad, int m_first = 0)
      : ring(ring_), lane(lane_), n_obst(n_obst_) {  // obstacles [m_first, m_first + n_obst) of every row block
    const T* pv = ov ? ov : ox;  // missing arrays: any finite value, zeroed in the fold
    const T* pa = oa ? oa : ox;
    const int sub = lane / RING_LPC<T>;
    const int e = (lane % RING_LPC<T>)*RING_EPL<T>;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      int ci = g * RING_CPI<T> + sub;
      if (ci >= NV) ci = NV - 1;  // padding component: the radius again
      const bool is_rad = ci == NV - 1;
      const T* b = ci < 3 ? ox + ci * rows : (ci < 6 ? pv + (ci - 3) * rows : (is_rad ? orad : pa + (ci - 6) * rows));
      stride[g] = (unsigned)((is_rad ? rows : 3 * rows) * (int64_t)sizeof(T));
      src0[g] = (const char*)(b + r0 + e) + (uint64_t)(unsigned)m_first * stride[g];
    }
    block_step = (uint64_t)rows_per_step * sizeof(T);
    total = my_blocks * n_obst;
    issued = consumed = 0;
    it = im = slot_issue = slot_read = 0;
  }
  __device__ __forceinline__ void issue_next() {
    const uint64_t off = (uint64_t)(unsigned)it * block_step;
#pragma unroll
    for (int g = 0; g < G; ++g)
      __builtin_amdgcn_global_load_lds((ring_gptr)(src0[g] + (uint64_t)(unsigned)im * stride[g] + off),
                                       (ring_lptr)(ring + slot_issue * SLOT + g * RING_CPI<T> * 64), 16, 0, 0);
    ++issued;
    if (++im == n_obst) {
      im = 0;
      ++it;
    }
    slot_issue = slot_issue + 1 == D ? 0 : slot_issue + 1;
  }
  __device__ __forceinline__ void prologue() {
#pragma unroll 1
    while (issued < total && issued < D) issue_next();
  }
  // the oldest unread item has landed when no more than the DMAs of the items issued after it are outstanding
  __device__ __forceinline__ void wait_oldest() const {
    static_assert(D <= 8, "wait_oldest spells out up to 7 counted waits");
    switch ((int)(issued - consumed - 1)) {
      case 0: wait_vm<0>(); break;
      case 1: wait_vm<G>(); break;
      case 2: wait_vm<2 * G>(); break;
      case 3: wait_vm<3 * G>(); break;
      case 4: wait_vm<4 * G>(); break;
      case 5: wait_vm<5 * G>(); break;
      case 6: wait_vm<6 * G>(); break;
      default: wait_vm<7 * G>(); break;
    }
  }
  // the oldest item: wait for it and read it; release() afterwards frees its slot for the next DMA
  __device__ __forceinline__ void take(T (&buf)[NV]) {
    typedef const __attribute__((address_space(3))) T* lds_ptr;
    wait_oldest();
    lds_ptr src = (lds_ptr)(ring + slot_read * SLOT + lane);
#pragma unroll
    for (int c = 0; c < NV; ++c) buf[c] = src[c * 64];
  }
  __device__ __forceinline__ void release() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot has been read: it may be overwritten
    ++consumed;
    slot_read = slot_read + 1 == D ? 0 : slot_read + 1;
    if (issued < total) issue_next();
  }
  // the n_obst obstacles of the current row block: fold(m, buf), m = 0 .. n_obst - 1
  template <class Fold>
  __device__ __forceinline__ void fold_block(Fold fold) {
    typedef const __attribute__((address_space(3))) T* lds_ptr;
#pragma unroll 1
    for (int m = 0; m < n_obst; ++m) {
      wait_oldest();
      T buf[NV];
      lds_ptr src = (lds_ptr)(ring + slot_read * SLOT + lane);
#pragma unroll
      for (int c = 0; c < NV; ++c) buf[c] = src[c * 64];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slot has been read: it may be overwritten
      ++consumed;
      slot_read = slot_read + 1 == D ? 0 : slot_read + 1;
      if (issued < total) issue_next();  // into the slot just read (the ring was full)
      fold(m, buf);
    }
  }
};

template <class CL, bool ACC, typename T>
__device__ __forceinline__ void obstacles_from_ring(const DevCfg<T>& cfg, ObstRing<ACC, T>& ring, int n_static, bool any_v,
                                                    bool any_a, const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
  constexpr int NV = ACC ? 10 : 7;
  ring.fold_block([&](int m, T (&buf)[NV]) {
    const bool is_static = m < n_static;
    const bool has_v = any_v && !is_static, has_a = ACC && any_a && !is_static;
    T xo[3], vo[3], ao[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xo[c] = buf[c];
      vo[c] = has_v ? buf[3 + c] : T(0);
      if constexpr (ACC)
        ao[c] = has_a ? buf[6 + c] : T(0);
      else
        ao[c] = T(0);
    }
    accumulate_obstacle<CL>(cfg, E, xo, vo, ao, buf[NV - 1], false, acc);
  });
}

// Cartesian rollout (measured, not wired in: profiles/r04_experiments.json "k_rollout_cart_panda obstacle ring"; the call
// site was k_rollout_cart_panda's RES branch with a 3-slot ring beside a 7-obstacle resident tile, ring constructed with
// my_blocks = horizon, rows_per_step = 0, m_first = nres): identical rollouts, 4.06 ms against 3.69 ms.
// The first nres obstacles of a row are resident in LDS, the others come round again in every step --
// through the ring, which keeps filling while the step is finished and the next chain is walked.  Resident and streamed
// obstacles are folded alternately so that the ring is drained at an even pace.
template <class CL, bool ACC, typename T, int D>
__device__ __forceinline__ void obstacles_cart_ring(const DevCfg<T>& cfg, const T* __restrict__ res, int lane, int nres,
                                                    ObstRing<ACC, T, D>& ring, int n_static, bool any_v, bool any_a, T tk,
                                                    const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  constexpr int NV = ACC ? 10 : 7;
  auto fold = [&](int m, T (&buf)[NV]) {
    const bool is_static = m < n_static;
    const bool has_v = any_v && !is_static, has_a = ACC && any_a && !is_static;
    T xo[3], vo[3], ao[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      vo[c] = has_v ? buf[3 + c] : T(0);
      xo[c] = buf[c] + tk * vo[c];  // x += dt*v per step (FPC:448-453)
      if constexpr (ACC)
        ao[c] = has_a ? buf[6 + c] : T(0);
      else
        ao[c] = T(0);
    }
    accumulate_obstacle<CL>(cfg, E, xo, vo, ao, buf[NV - 1], false, acc);
  };
  const int ns = ring.n_obst;
  int jr = 0, js = 0;
#pragma unroll 1
  while (jr < nres && js < ns) {  // pairs: one resident, one streamed
    T A[NV], B[NV];
    lds_ptr src = (lds_ptr)(res + jr * (NV * 64) + lane);
#pragma unroll
    for (int c = 0; c < NV; ++c) A[c] = src[c * 64];
    ring.take(B);
    fold(jr, A);
    ring.release();
    fold(nres + js, B);
    ++jr;
    ++js;
  }
#pragma unroll 1
  for (; jr < nres; ++jr) {
    T A[NV];
    lds_ptr src = (lds_ptr)(res + jr * (NV * 64) + lane);
#pragma unroll
    for (int c = 0; c < NV; ++c) A[c] = src[c * 64];
    fold(jr, A);
  }
#pragma unroll 1
  for (; js < ns; ++js) {
    T B[NV];
    ring.take(B);
    ring.release();
    fold(nres + js, B);
  }
}

// compute_action with the obstacle ring: persistent one-wave blocks (four per CU), block b takes the 64-row blocks
// b, b + gridDim.x, ... ; rows % 64 == 0
template <typename T, class LS, bool ACC>
__global__ __launch_bounds__(64) void k_action_panda_ring(const DevCfg<T>* __restrict__ cfgp, int64_t rows,
                                                          const T* __restrict__ q, const T* __restrict__ qd,
                                                          const T* __restrict__ prm, int n_obst, int n_static,
                                                          const T* __restrict__ ox, const T* __restrict__ ov,
                                                          const T* __restrict__ oa, const T* __restrict__ orad,
                                                          T* __restrict__ qdd_out, T* __restrict__ act_out) {
  __shared__ __align__(16) T ring_lds[RING_DEPTH * RING_SLOT<T, ACC>];
  const int lane = threadIdx.x;
  const int64_t nb = rows / 64;
  const int64_t my_blocks = (nb - (int64_t)blockIdx.x + gridDim.x - 1) / gridDim.x;
  const DevCfg<T>& cfg = *cfgp;
  ObstRing<ACC, T> ring(ring_lds, lane, rows, (int64_t)blockIdx.x * 64, (int64_t)gridDim.x * 64, my_blocks, n_obst, ox, ov, oa,
                        orad);
  ring.prologue();  // the first obstacles travel while the first chain is walked
#pragma unroll 1
  for (int64_t t = 0; t < my_blocks; ++t) {
    const int64_t r = ((int64_t)blockIdx.x + t * gridDim.x) * 64 + lane;
    PandaState<T> R;
    load_state(rows, r, q, qd, R);
    PrmView<T> P{prm, rows, r, {T(0), T(0), T(0)}, false};
    T qdd[7], act[7];
    panda_solve_row<LS, kActionSingleWalk>(
        cfg, cfg.mount[(int)(r % cfg.n_robots)], R, P,
        [&](const EgoPts<T, NG>& E, EgoAcc<T, NG>& acc) {
          obstacles_from_ring<typename LS::Collision, ACC>(cfg, ring, n_static, ov != nullptr, oa != nullptr, E, acc);
        },
        qdd, act);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if (qdd_out) qdd_out[j * rows + r] = qdd[j];
      act_out[j * rows + r] = act[j];
    }
  }
}

