// mrf_host.hpp -- host-side state shared by the translation units that implement include/mrf.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/mrf.h"

struct mrf_handle {
  mrf_config cfg;
  int device;
  uint64_t serial;        // process-wide creation counter: distinguishes a new handle that reuses a freed address
  int64_t coop_max_scen;  // batches up to this size use the cooperative kernels (auto mode)
  void* dcfg;             // DevCfg<double> or DevCfg<float> on the device
  std::string err;
  // mrf_episode_run: cached HIP graph of one control step, the argument tuple it was captured for, and the stream
  // used when the caller passes the (uncapturable) legacy default stream
  void* graph_exec = nullptr;
  std::string graph_key;
  void* own_stream = nullptr;
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

template <typename F>
int dispatch_scalar(mrf_handle* h, F f) {
  return h->cfg.scalar == MRF_F64 ? f(double{}) : f(float{});
}

}  // namespace mrf_host

#define MRF_GUARD_CAT2(a, b) a##b
#define MRF_GUARD_CAT(a, b) MRF_GUARD_CAT2(a, b)
#define MRF_CHECK_READY(h)                                                                                    \
  if (!(h)) return MRF_E_ARG;                                                                                 \
  if (!(h)->dcfg) return mrf_host::fail((h), MRF_E_DEVICE, "handle has no device state (mrf_create failed)"); \
  mrf_host::DeviceGuard MRF_GUARD_CAT(mrf_device_guard_, __LINE__)((h)->device);
