// mrf_hostpath.hip -- host-buffer variants of the hot-path entry points (include/mrf.h "Host-buffer entry points").
//
// The reference's call sites hand over numpy arrays of ONE scenario (planner.compute_action(**kwargs),
// forwardplanner.get_velocity_rollouts(inputs_action), example_pandas_Jointspace.py:374,441).  For such calls the
// arithmetic is a few microseconds and everything around it decides the latency, so the whole round trip is done here:
// inputs packed into a pinned staging buffer owned by the handle, ONE hipMemcpyAsync to the device, the kernel, ONE
// copy back, one stream synchronisation.  Host arrays are float64 in the layout of the device versions; handles with
// scalar f32 convert while packing.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "mrf_host.hpp"

namespace {
using mrf_host::check_hip;
using mrf_host::fail;

struct Staging {
  unsigned char* pinned = nullptr;
  unsigned char* dev = nullptr;
  size_t bytes = 0;
  hipStream_t stream = nullptr;
};

int ensure(mrf_handle* h, Staging* s, size_t bytes) {
  if (!s->stream && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess)
    return fail(h, MRF_E_LAUNCH, "hipStreamCreate failed");
  if (bytes <= s->bytes) return MRF_OK;
  if (s->pinned) (void)hipHostFree(s->pinned);
  if (s->dev) (void)hipFree(s->dev);
  s->pinned = s->dev = nullptr;
  s->bytes = 0;
  const size_t cap = bytes < 4096 ? 4096 : 2 * bytes;
  hipError_t e = hipHostMalloc((void**)&s->pinned, cap, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc((void**)&s->dev, cap);
  if (e != hipSuccess) return fail(h, MRF_E_DEVICE, std::string("staging buffers: ") + hipGetErrorString(e));
  s->bytes = cap;
  return MRF_OK;
}

// one packed segment: host source (may be NULL = absent), element count, byte offset in the staging buffers
struct Seg {
  const double* src;
  double* dst;
  size_t n;
  size_t off;
};

template <typename T>
void pack(unsigned char* base, const Seg& g) {
  T* d = reinterpret_cast<T*>(base + g.off);
  for (size_t i = 0; i < g.n; ++i) d[i] = (T)g.src[i];
}
template <typename T>
void unpack(const unsigned char* base, const Seg& g) {
  const T* s = reinterpret_cast<const T*>(base + g.off);
  for (size_t i = 0; i < g.n; ++i) g.dst[i] = (double)s[i];
}

struct Plan {
  std::vector<Seg> in, out;
  size_t in_bytes = 0, total = 0;
  size_t sb;
  explicit Plan(size_t scalar_bytes) : sb(scalar_bytes) {}
  size_t add_in(const double* src, size_t n) {
    const size_t off = total;
    if (src) in.push_back({src, nullptr, n, off});
    total += ((n * sb + 63) / 64) * 64;
    in_bytes = total;
    return off;
  }
  size_t add_out(double* dst, size_t n) {
    const size_t off = total;
    if (dst) out.push_back({nullptr, dst, n, off});
    total += ((n * sb + 63) / 64) * 64;
    return off;
  }
};

template <typename Launch>
int round_trip(mrf_handle* h, Plan& P, Launch launch) {
  Staging* s = (Staging*)h->staging;
  if (!s) h->staging = s = new Staging();
  if (int rc = ensure(h, s, P.total)) return rc;
  const bool f64 = h->cfg.scalar == MRF_F64;
  for (const Seg& g : P.in) f64 ? pack<double>(s->pinned, g) : pack<float>(s->pinned, g);
  if (int rc = check_hip(h, hipMemcpyAsync(s->dev, s->pinned, P.in_bytes, hipMemcpyHostToDevice, s->stream), "H2D")) return rc;
  if (int rc = launch(s->dev, (void*)s->stream)) return rc;
  const size_t out_bytes = P.total - P.in_bytes;
  if (out_bytes)
    if (int rc = check_hip(h, hipMemcpyAsync(s->pinned + P.in_bytes, s->dev + P.in_bytes, out_bytes, hipMemcpyDeviceToHost, s->stream), "D2H"))
      return rc;
  if (int rc = check_hip(h, hipStreamSynchronize(s->stream), "hipStreamSynchronize")) return rc;
  for (const Seg& g : P.out) f64 ? unpack<double>(s->pinned, g) : unpack<float>(s->pinned, g);
  return MRF_OK;
}

}  // namespace

void mrf_host::staging_release(mrf_handle* h) {
  if (!h || !h->staging) return;
  Staging* s = (Staging*)h->staging;
  if (s->stream) {
    (void)hipStreamSynchronize(s->stream);
    (void)hipStreamDestroy(s->stream);
  }
  if (s->pinned) (void)hipHostFree(s->pinned);
  if (s->dev) (void)hipFree(s->dev);
  delete s;
  h->staging = nullptr;
}

extern "C" {

int mrf_compute_action_host(mrf_handle* h, int64_t rows, const double* q, const double* qdot, const double* params,
                            int32_t n_obst, int32_t n_obst_static, const double* ox, const double* ov, const double* oa,
                            const double* orad, double* qddot_out, double* action_out) {
  MRF_CHECK_READY(h);
  if (rows == 0) return MRF_OK;
  if (rows < 0 || n_obst < 0 || !q || !qdot || !params || !action_out) return fail(h, MRF_E_ARG, "null/negative argument");
  if (n_obst > 0 && (!ox || !orad)) return fail(h, MRF_E_ARG, "obstacle arrays missing");
  const size_t dof = h->cfg.model == MRF_MODEL_PANDA7 ? 7 : 3, R = (size_t)rows, M = (size_t)n_obst;
  Plan P(h->cfg.scalar == MRF_F64 ? 8 : 4);
  const size_t o_q = P.add_in(q, dof * R), o_qd = P.add_in(qdot, dof * R), o_p = P.add_in(params, MRF_NPARAM * R);
  const size_t o_x = P.add_in(ox, M * 3 * R), o_v = P.add_in(ov, M * 3 * R), o_a = P.add_in(oa, M * 3 * R), o_r = P.add_in(orad, M * R);
  const size_t o_qdd = P.add_out(qddot_out, dof * R), o_act = P.add_out(action_out, dof * R);
  return round_trip(h, P, [&](unsigned char* d, void* st) {
    return mrf_compute_action(h, rows, d + o_q, d + o_qd, d + o_p, n_obst, n_obst_static, M ? d + o_x : nullptr,
                              (M && ov) ? d + o_v : nullptr, (M && oa) ? d + o_a : nullptr, M ? d + o_r : nullptr,
                              qddot_out ? d + o_qdd : nullptr, d + o_act, st);
  });
}

int mrf_rollout_host(mrf_handle* h, int64_t n_scen, const double* q0, const double* qdot0, const double* params,
                     double* avg_out, double* traj_q, double* traj_qd) {
  MRF_CHECK_READY(h);
  if (n_scen == 0) return MRF_OK;
  if (n_scen < 0 || !q0 || !qdot0 || !params || !avg_out) return fail(h, MRF_E_ARG, "null/negative argument");
  if (h->cfg.model != MRF_MODEL_PANDA7)  // the segments below are sized 7 x rows: refuse before anything is read
    return fail(h, MRF_E_CONFIG, "mrf_rollout_host is defined for the panda7 model only");
  const size_t R = (size_t)n_scen * h->cfg.n_robots, H = (size_t)h->cfg.horizon;
  Plan P(h->cfg.scalar == MRF_F64 ? 8 : 4);
  const size_t o_q = P.add_in(q0, 7 * R), o_qd = P.add_in(qdot0, 7 * R), o_p = P.add_in(params, MRF_NPARAM * R);
  const size_t o_avg = P.add_out(avg_out, R), o_tq = P.add_out(traj_q, traj_q ? H * 7 * R : 0),
               o_tqd = P.add_out(traj_qd, traj_qd ? H * 7 * R : 0);
  return round_trip(h, P, [&](unsigned char* d, void* st) {
    return mrf_rollout(h, n_scen, d + o_q, d + o_qd, d + o_p, d + o_avg, traj_q ? d + o_tq : nullptr,
                       traj_qd ? d + o_tqd : nullptr, st);
  });
}

int mrf_rollout_cartesian_host(mrf_handle* h, int64_t rows, const double* q0, const double* qdot0, const double* params,
                               int32_t n_obst, int32_t n_obst_static, const double* ox0, const double* ov, const double* oa,
                               const double* orad, double* avg_out, double* traj_q, double* traj_qd) {
  MRF_CHECK_READY(h);
  if (rows == 0) return MRF_OK;
  if (rows < 0 || n_obst < 0 || !q0 || !qdot0 || !params || !avg_out) return fail(h, MRF_E_ARG, "null/negative argument");
  if (n_obst > 0 && (!ox0 || !orad)) return fail(h, MRF_E_ARG, "obstacle arrays missing");
  if (h->cfg.model != MRF_MODEL_PANDA7)  // the segments below are sized 7 x rows: refuse before anything is read
    return fail(h, MRF_E_CONFIG, "mrf_rollout_cartesian_host is defined for the panda7 model only");
  const size_t R = (size_t)rows, M = (size_t)n_obst, H = (size_t)h->cfg.horizon;
  Plan P(h->cfg.scalar == MRF_F64 ? 8 : 4);
  const size_t o_q = P.add_in(q0, 7 * R), o_qd = P.add_in(qdot0, 7 * R), o_p = P.add_in(params, MRF_NPARAM * R);
  const size_t o_x = P.add_in(ox0, M * 3 * R), o_v = P.add_in(ov, M * 3 * R), o_a = P.add_in(oa, M * 3 * R), o_r = P.add_in(orad, M * R);
  const size_t o_avg = P.add_out(avg_out, R), o_tq = P.add_out(traj_q, traj_q ? H * 7 * R : 0),
               o_tqd = P.add_out(traj_qd, traj_qd ? H * 7 * R : 0);
  return round_trip(h, P, [&](unsigned char* d, void* st) {
    return mrf_rollout_cartesian(h, rows, d + o_q, d + o_qd, d + o_p, n_obst, n_obst_static, M ? d + o_x : nullptr,
                                 (M && ov) ? d + o_v : nullptr, (M && oa) ? d + o_a : nullptr, M ? d + o_r : nullptr,
                                 d + o_avg, traj_q ? d + o_tq : nullptr, traj_qd ? d + o_tqd : nullptr, st);
  });
}

int mrf_fk_spheres_host(mrf_handle* h, int64_t rows, const double* q, const double* qdot, double* x_out, double* v_out,
                        double* a_out) {
  MRF_CHECK_READY(h);
  if (rows == 0) return MRF_OK;
  if (rows < 0 || !q || !x_out) return fail(h, MRF_E_ARG, "null/negative argument");
  if ((v_out || a_out) && !qdot) return fail(h, MRF_E_ARG, "qdot required for v/a");
  if (h->cfg.model != MRF_MODEL_PANDA7)  // the segments below are sized 7 x rows: refuse before anything is read
    return fail(h, MRF_E_CONFIG, "mrf_fk_spheres_host is defined for the panda7 model only");
  const size_t R = (size_t)rows, S = (size_t)h->cfg.n_spheres;
  Plan P(h->cfg.scalar == MRF_F64 ? 8 : 4);
  const size_t o_q = P.add_in(q, 7 * R), o_qd = P.add_in(qdot, 7 * R);
  const size_t o_x = P.add_out(x_out, S * 3 * R), o_v = P.add_out(v_out, v_out ? S * 3 * R : 0), o_a = P.add_out(a_out, a_out ? S * 3 * R : 0);
  return round_trip(h, P, [&](unsigned char* d, void* st) {
    return mrf_fk_spheres(h, rows, d + o_q, qdot ? d + o_qd : nullptr, d + o_x, v_out ? d + o_v : nullptr,
                          a_out ? d + o_a : nullptr, st);
  });
}

}  // extern "C"
