// mrf_host.hpp -- host-side state shared by the translation units that implement include/mrf.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/mrf.h"
#include "mrf_device.hpp"

struct mrf_handle {
  mrf_config cfg;
  int device;
  uint64_t serial;        // process-wide creation counter: distinguishes a new handle that reuses a freed address
  int64_t coop_max_scen;  // batches up to this size use the cooperative kernels (auto mode)
  int n_cus = 256;        // compute units of the device (persistent launches: four one-wave blocks per CU)
  void* dcfg;             // DevCfg<double> or DevCfg<float> on the device
  std::string err;
  // mrf_episode_run: cached HIP graph of one control step, the argument tuple it was captured for, and the stream
  // used when the caller passes the (uncapturable) legacy default stream
  void* graph_exec = nullptr;
  std::string graph_key;
  void* own_stream = nullptr;
  // pick-and-place buffers attached by mrf_episode_set_pick_place (mrf_control.hip); all caller-owned
  struct PickPlace {
    bool on = false;
    mrf_state_machine_config sm;
    const void* start_goal = nullptr;
    const void* blocks = nullptr;
    int32_t n_block_arrays = 0;
    void* q_gripper = nullptr;
    int32_t* sm_state = nullptr;
    void* sm_goal = nullptr;
    void* gripper_action = nullptr;
    mrf_handle* h_grasp = nullptr;
    uint64_t grasp_serial = 0;
    void* action_grasp = nullptr;
  } pp;
  // episode recorder attached by mrf_episode_set_recorder (mrf_control.hip); all caller-owned device arrays
  struct Recorder {
    bool on = false;
    void* q_hist = nullptr;
    int32_t* sm_hist = nullptr;
    int64_t* t_begin = nullptr;
    int64_t* t_end = nullptr;
    int32_t* done_at = nullptr;
    int32_t* counter = nullptr;
    int32_t capacity = 0;
    int32_t done_state = 0;
  } rec;
  // mrf_rollout_cartesian_coupled (mrf_control.hip): obstacle arrays assembled on the device, grown on demand (never
  // inside a stream capture: mrf_episode_run sizes them before it captures)
  void* clock_probe = nullptr;  // 8 x int64 written by the first / last workgroup of k_rollout_panda (mrf_rollout_clock) + the call serial
  long long rollout_serial = 0;  // mrf_rollout calls on this handle; the stamping kernels copy it into clock_probe[8]
  void* cart_work = nullptr;
  size_t cart_work_bytes = 0;
  int episode_rollout_kind = 0;  // mrf_episode_set_rollout: which rollout an episode on this ROLLOUT handle runs
  void* staging = nullptr;  // pinned + device staging buffers and the stream of the host-buffer entry points (mrf_hostpath.hip)
  void* comm = nullptr;   // robot-sharded rollout state (mrf_comm.hip): communicator / mapped peer buffers / work buffers
};

namespace mrf_host {

// Every entry point runs on the handle's device, whatever the calling thread's current device is, and leaves the
// caller's current device as it found it (a process that drives several GPUs, or torch after set_device(local_rank)).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int device) {
    if (device < 0 || hipGetDevice(&prev) != hipSuccess) return;
    if (prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

inline int fail(mrf_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

inline int check_hip(mrf_handle* h, hipError_t e, const char* what) {
  if (e == hipSuccess) return MRF_OK;
  return fail(h, MRF_E_LAUNCH, std::string(what) + ": " + hipGetErrorString(e));
}

template <typename K, typename... Args>
int launch(mrf_handle* h, K kernel, dim3 grid, dim3 block, hipStream_t st, Args... args) {
  hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
  return check_hip(h, hipGetLastError(), "kernel launch");
}

// -DMRF_WITH_F32: the float32 instantiations of every kernel are compiled as well.  Not in the default build (r5): they are
// 5-15 % faster than float64 at one wave per SIMD, accurate to 1e-3 on rollouts and unusable below a barrier coordinate of
// ~0.1, and they double the compile time; mrf_create refuses scalar = MRF_F32 without them
// (mrf_build_has_f32()).
// -DMRF_DEV_F64_PANDA_ONLY (development builds, tools/build_variant.sh): only the float64 / reference-leaf-set
// instantiations are compiled (a quarter of the compile time when iterating on one kernel's ISA); any other
// configuration then fails loudly instead of running.
template <typename F>
int dispatch_scalar(mrf_handle* h, F f) {
#if defined(MRF_DEV_F64_PANDA_ONLY) || !defined(MRF_WITH_F32)
  if (h->cfg.scalar != MRF_F64) return fail(h, MRF_E_CONFIG, "this build has no float32 kernels (build with -DMRF_WITH_F32)");
  return f(double{});
#else
  return h->cfg.scalar == MRF_F64 ? f(double{}) : f(float{});
#endif
}


// the reference's Panda leaf strings (EXJ:87-89 + the library's limit / plane-Finsler defaults) get compile-time
// leaf policies; any other configuration runs the generic (runtime-family) instantiation
using LeafSetPanda = mrf::LeafSet<mrf::LeafPow<4, 4, MRF_GATE_NONE, MRF_GATE_NONE>,
                                  mrf::SLeaf<MRF_FAMILY_LOGISTIC, 0, MRF_GATE_NONE, 1, MRF_GATE_NEG>,
                                  mrf::SLeaf<MRF_FAMILY_POW, 1, MRF_GATE_NONE, 1, MRF_GATE_NEG>>;
using LeafSetGeneric = mrf::LeafSet<mrf::LeafGeneric, mrf::SLeafGeneric, mrf::SLeafGeneric>;
inline bool leaf_is(const mrf_leaf_fn& f, int family, int p, int gate) {
  return f.family == family && f.gate == gate && (family == MRF_FAMILY_LOGISTIC || f.p == p);
}
inline bool is_panda_leafset(const mrf_config& c) {  // ... and the examples' full collision-link set
  return (c.n_ego == 0 || c.ego_link_mask == 0x3F) &&
         leaf_is(c.collision_geometry, MRF_FAMILY_POW, 4, MRF_GATE_NONE) &&
         leaf_is(c.collision_finsler, MRF_FAMILY_POW, 4, MRF_GATE_NONE) &&
         leaf_is(c.plane_geometry, MRF_FAMILY_LOGISTIC, 0, MRF_GATE_NONE) &&
         leaf_is(c.plane_finsler, MRF_FAMILY_POW, 1, MRF_GATE_NEG) &&
         leaf_is(c.limit_geometry, MRF_FAMILY_POW, 1, MRF_GATE_NONE) &&
         leaf_is(c.limit_finsler, MRF_FAMILY_POW, 1, MRF_GATE_NEG);
}

// the reference's rollout sphere table: the origins of panda_link1..8 (PM:25-26)
inline bool is_link_origin_table(const mrf_config& c) {
  if (c.n_spheres != 8) return false;
  for (int s = 0; s < 8; ++s)
    if (c.sphere_link[s] != s + 1 || c.sphere_offset[s][0] != 0.0 || c.sphere_offset[s][1] != 0.0 ||
        c.sphere_offset[s][2] != 0.0)
      return false;
  return true;
}

template <typename F>
int dispatch(mrf_handle* h, F f) {  // f(scalar tag, leaf-set tag)
  const bool fast = is_panda_leafset(h->cfg);
#ifdef MRF_DEV_F64_PANDA_ONLY
  if (h->cfg.scalar != MRF_F64 || !fast) return fail(h, MRF_E_CONFIG, "development build: float64, reference leaf set only");
  return f(double{}, LeafSetPanda{});
#else
  if (h->cfg.scalar == MRF_F64) return fast ? f(double{}, LeafSetPanda{}) : f(double{}, LeafSetGeneric{});
#ifdef MRF_WITH_F32
  return fast ? f(float{}, LeafSetPanda{}) : f(float{}, LeafSetGeneric{});
#else
  return fail(h, MRF_E_CONFIG, "this build has no float32 kernels (build with -DMRF_WITH_F32)");
#endif
#endif
}

// mrf_step_action with an explicit robot -> block-position map of sph_all (padded all-gather layouts); NULL = identity
int step_action_slots(mrf_handle* h, int64_t n_scen, int32_t robot_first, int32_t robot_count, const void* q,
                      void* qdot_io, const void* params, const void* sph_all, const int32_t* robot_slot, void* sumsq_io,
                      void* stream);
// mrf_step_action_joints with an explicit robot -> block-position map of jst_all (mrf_shard_step.hip); NULL = identity.
// jst_next_own != NULL: also the position update of the following step, its joint state -> jst_next_own (then q is written)
int step_action_joints_slots(mrf_handle* h, int64_t n_scen, int32_t robot_first, int32_t robot_count, void* q,
                             void* qdot_io, const void* params, const void* jst_all, const int32_t* robot_slot,
                             void* sumsq_io, void* jst_next_own, void* stream);
// the cooperative (one wave per scenario) form of mrf_rollout_cartesian_coupled (mrf_kernels.hip); returns 1 when it does not
// apply (batch above the crossover, kernel_select = 1, a single robot)
bool coop_applies(const mrf_handle* h, int64_t n_scen);  // the batch-size / kernel_select rule of the coupled entry points
int rollout_cartesian_coop(mrf_handle* h, int64_t n_scen, const void* q0, const void* qdot0, const void* params, void* avg_out,
                           void* traj_q, void* traj_qd, void* stream);
bool cartesian_tile_applies(const mrf_handle* h);  // rollout_cartesian_tile will take the call (no obstacle work buffer needed)
int rollout_cartesian_tile(mrf_handle* h, int64_t n_scen, const void* q0, const void* qdot0, const void* params, void* avg_out,
                           void* traj_q, void* traj_qd, void* stream);
// the coupled Cartesian rollout for sphere tables beyond the LDS-tile forms: ONE launch, obstacle assembly in the kernel's
// prologue into the work arrays wx / wv [M][3][rows] (mrf_kernels.hip k_rollout_carts_panda)
int rollout_cartesian_self(mrf_handle* h, int64_t n_scen, const void* q0, const void* qdot0, const void* params, void* wx,
                           void* wv, void* avg_out, void* traj_q, void* traj_qd, void* stream);
// the joint-space rollout as a pair of waves per row (mrf_rollout_wp.hip); the caller has checked that the form applies
int rollout_wave_pair(mrf_handle* h, int64_t n_scen, const void* q0, const void* qdot0, const void* params, void* avg_out,
                      void* traj_q, void* traj_qd, void* stream);
// frees h->comm (mrf_comm.hip); called by mrf_destroy
void comm_release(mrf_handle* h);
// frees h->staging (mrf_hostpath.hip); called by mrf_destroy
void staging_release(mrf_handle* h);
// frees h->cart_work (mrf_control.hip); called by mrf_destroy
void cart_work_release(mrf_handle* h);

}  // namespace mrf_host

#define MRF_GUARD_CAT2(a, b) a##b
#define MRF_GUARD_CAT(a, b) MRF_GUARD_CAT2(a, b)
#define MRF_CHECK_READY(h)                                                                                    \
  if (!(h)) return MRF_E_ARG;                                                                                 \
  if (!(h)->dcfg) return mrf_host::fail((h), MRF_E_DEVICE, "handle has no device state (mrf_create failed)"); \
  mrf_host::DeviceGuard MRF_GUARD_CAT(mrf_device_guard_, __LINE__)((h)->device);
