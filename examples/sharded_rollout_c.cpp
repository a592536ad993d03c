// sharded_rollout_c.cpp -- the north-star partitioning (robots of a scenario on different ranks, exchange of predicted
// sphere states at every rollout step) driven from plain C/C++ through include/mrf.h: no Python, no torch, no MPI.
//
// Two processes (fork before any HIP call) share ONE GPU -- the only topology the single-GPU test box offers; on a node
// with several GPUs give each process its own device id instead.  Each process
//   1. creates the same 3-Panda RF-CV handle (mrf_default_config_panda),
//   2. opens the peer transport (mrf_comm_peer_open), swaps the 64-byte IPC handles with the other process over a
//      pipe -- the only thing the host ever carries between the ranks -- and connects (mrf_comm_peer_connect),
//   3. uploads ITS robots' rows of a seeded batch and runs mrf_rollout_sharded (H steps, exchange inside one kernel),
//   4. checks its rows against the fused single-GPU kernel (mrf_rollout on the whole batch) and reports.
// Replaces, for the rollout, the loop and exchange of forward_planner_Jointspace.py:190-249 (exchange step :211-225).
//
//   make -C examples && ./examples/sharded_rollout_c [n_scenarios] [horizon]
#include <hip/hip_runtime.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mrf.h"

static int g_rank = -1;
#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #x, hipGetErrorString(e_));    \
      exit(3);                                                                     \
    }                                                                              \
  } while (0)
#define MRF_OK_(h, x)                                                              \
  do {                                                                             \
    int rc_ = (x);                                                                 \
    if (rc_ != MRF_OK) {                                                           \
      fprintf(stderr, "[rank %d] %s -> %d: %s\n", g_rank, #x, rc_, mrf_last_error(h)); \
      exit(4);                                                                     \
    }                                                                              \
  } while (0)

// the same seeded batch in every process: start pose of parameters_manipulators.py:111-115 +- 0.1 rad, |qdot| < 0.3,
// goals of :126-132 +- 0.05 m, the runtime kwargs of example_pandas_Jointspace.py:421-439
static void make_batch(int n_scen, int N, std::vector<double>& q, std::vector<double>& qd, std::vector<double>& prm) {
  const double pos0[3][7] = {{1.13793529, -0.3227085, -0.02767777, -2.2204281, -0.00917029, 1.88612235, 0.78536134},
                             {1.131, 0.20, 0.12, -1.65, -0.0, 1.86, 0.78539816},
                             {-0.46609715, -0.25025564, -0.40425878, -2.0966941, -0.10682593, 1.84917516, 0.37170524}};
  const double goal[3][3] = {{0.25, 0.6, 1.15}, {0.8, -0.5, 1.15}, {0.4, 0.5, 0.95}};
  const double R1[9] = {0, 0, -1, 0, 1, 0, 1, 0, 0};
  const int64_t rows = (int64_t)n_scen * N;
  q.assign(7 * rows, 0.0);
  qd.assign(7 * rows, 0.0);
  prm.assign((size_t)MRF_NPARAM * rows, 0.0);
  unsigned long long s = 88172645463325252ull;
  auto uni = [&]() {  // xorshift64 in [-1, 1)
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0;
  };
  for (int64_t b = 0; b < n_scen; ++b)
    for (int i = 0; i < N; ++i) {
      const int64_t r = b * N + i;
      for (int j = 0; j < 7; ++j) {
        q[j * rows + r] = pos0[i][j] + 0.1 * uni();
        qd[j * rows + r] = 0.3 * uni();
      }
      for (int c = 0; c < 3; ++c) prm[(MRF_P_X_GOAL_0 + c) * rows + r] = goal[i][c] + 0.05 * uni();
      prm[MRF_P_WEIGHT_GOAL_0 * rows + r] = 2.0;
      for (int c = 0; c < 9; ++c) prm[(MRF_P_ANGLE_GOAL_1 + c) * rows + r] = R1[c];
      prm[(MRF_P_X_GOAL_1 + 0) * rows + r] = 0.107;
      prm[MRF_P_WEIGHT_GOAL_1 * rows + r] = 20.0;
      prm[MRF_P_X_GOAL_2 * rows + r] = 0.78539816339744831;
      prm[MRF_P_WEIGHT_GOAL_2 * rows + r] = 1.0;
      prm[(MRF_P_CONSTRAINT_0 + 2) * rows + r] = 1.0;
      prm[(MRF_P_CONSTRAINT_0 + 3) * rows + r] = -0.65;
      for (int c = 0; c < 6; ++c) prm[(MRF_P_RADIUS_BODY + c) * rows + r] = 0.08;
    }
}

// rows [ncomp][n_scen*N] -> the owned rows [ncomp][n_scen*count], row = scenario*count + (robot - first)
static std::vector<double> owned(const std::vector<double>& a, int ncomp, int n_scen, int N, int first, int count) {
  const int64_t rows = (int64_t)n_scen * N, orows = (int64_t)n_scen * count;
  std::vector<double> o((size_t)ncomp * orows);
  for (int c = 0; c < ncomp; ++c)
    for (int64_t b = 0; b < n_scen; ++b)
      for (int l = 0; l < count; ++l) o[c * orows + b * count + l] = a[c * rows + b * N + first + l];
  return o;
}

static double* to_device(const std::vector<double>& a) {
  double* d = nullptr;
  HIP_OK(hipMalloc((void**)&d, a.size() * sizeof(double)));
  HIP_OK(hipMemcpy(d, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int n_scen = argc > 1 ? atoi(argv[1]) : 200, H = argc > 2 ? atoi(argv[2]) : 12, N = 3, G = 2;
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 2;
  const pid_t pid = fork();  // before any HIP call
  g_rank = pid == 0 ? 1 : 0;
  const int rfd = g_rank == 0 ? c2p[0] : p2c[0], wfd = g_rank == 0 ? p2c[1] : c2p[1];

  mrf_config cfg;
  mrf_default_config_panda(&cfg, N, H);
  cfg.goal_estimate_mask = 0x6;  // RF-CV: the goals of robots 1, 2 are estimated (EXC:355-357)
  mrf_handle* h = nullptr;
  MRF_OK_(h, mrf_create(&cfg, /*device*/ 0, &h));

  unsigned char mine[MRF_IPC_HANDLE_BYTES], all[2 * MRF_IPC_HANDLE_BYTES];
  MRF_OK_(h, mrf_comm_peer_open(h, g_rank, G, n_scen, mine));
  if (write(wfd, mine, sizeof mine) != (ssize_t)sizeof mine) return 2;
  memcpy(all + g_rank * MRF_IPC_HANDLE_BYTES, mine, sizeof mine);
  if (read(rfd, all + (1 - g_rank) * MRF_IPC_HANDLE_BYTES, sizeof mine) != (ssize_t)sizeof mine) return 2;
  MRF_OK_(h, mrf_comm_peer_connect(h, all));
  int32_t first = 0, count = 0;
  MRF_OK_(h, mrf_comm_partition(h, &first, &count));

  std::vector<double> q, qd, prm;
  make_batch(n_scen, N, q, qd, prm);
  const int64_t rows = (int64_t)n_scen * N, orows = (int64_t)n_scen * count;
  double* d_q = to_device(owned(q, 7, n_scen, N, first, count));
  double* d_qd = to_device(owned(qd, 7, n_scen, N, first, count));
  double* d_prm = to_device(owned(prm, MRF_NPARAM, n_scen, N, first, count));
  double* d_avg = nullptr;
  HIP_OK(hipMalloc((void**)&d_avg, orows * sizeof(double)));
  char go = 'g';  // both sides connected: start together
  if (write(wfd, &go, 1) != 1 || read(rfd, &go, 1) != 1) return 2;

  MRF_OK_(h, mrf_rollout_sharded(h, n_scen, d_q, d_qd, d_prm, d_avg, /*stream*/ nullptr));
  MRF_OK_(h, mrf_comm_status(h));  // waits; reports a peer that never published (bounded spin) instead of hanging
  std::vector<double> avg(orows), qend(7 * orows);
  HIP_OK(hipMemcpy(avg.data(), d_avg, orows * sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(qend.data(), d_q, 7 * orows * sizeof(double), hipMemcpyDeviceToHost));

  // the same rollout, unsharded, on this process's own second handle
  mrf_handle* hf = nullptr;
  MRF_OK_(hf, mrf_create(&cfg, 0, &hf));
  double *f_q = to_device(q), *f_qd = to_device(qd), *f_prm = to_device(prm), *f_avg = nullptr, *f_tq = nullptr;
  HIP_OK(hipMalloc((void**)&f_avg, rows * sizeof(double)));
  HIP_OK(hipMalloc((void**)&f_tq, (size_t)H * 7 * rows * sizeof(double)));
  MRF_OK_(hf, mrf_rollout(hf, n_scen, f_q, f_qd, f_prm, f_avg, f_tq, nullptr, nullptr));
  HIP_OK(hipDeviceSynchronize());
  std::vector<double> want(rows), wq((size_t)7 * rows);
  HIP_OK(hipMemcpy(want.data(), f_avg, rows * sizeof(double), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(wq.data(), f_tq + (size_t)(H - 1) * 7 * rows, 7 * rows * sizeof(double), hipMemcpyDeviceToHost));
  double err = 0, scale = 0, errq = 0;
  for (int64_t b = 0; b < n_scen; ++b)
    for (int l = 0; l < count; ++l) {
      const int64_t ro = b * count + l, rf = b * N + first + l;
      err = fmax(err, fabs(avg[ro] - want[rf]));
      scale = fmax(scale, fabs(want[rf]));
      for (int j = 0; j < 7; ++j) errq = fmax(errq, fabs(qend[j * orows + ro] - wq[j * rows + rf]));
    }
  const double rel = err / fmax(scale, 1e-300);
  printf("[rank %d] robots %d..%d of %d, %d scenarios, H = %d: avg-velocity rel err vs fused kernel %.2e, final q abs err %.2e\n",
         g_rank, first, first + count - 1, N, n_scen, H, rel, errq);
  const bool ok = rel < 1e-9 && errq < 1e-9;

  if (write(wfd, &go, 1) != 1 || read(rfd, &go, 1) != 1) return 2;  // keep the mappings alive until both are done
  mrf_destroy(h);
  mrf_destroy(hf);
  int status = 0;
  if (g_rank == 0) waitpid(pid, &status, 0);
  return (ok && status == 0) ? 0 : 1;
}
