// mrf_oracle.cpp -- float64 CPU restatement of the fabric solve and the Rollout-Fabrics
// recurrences.  TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load the library built from this file.  The shipped path never calls it.
//
// PARITY UNPINNED.  The arithmetic of the reference lives in third-party packages that are not
// under /root/reference and cannot be installed here: fabrics==0.9.5, forwardkinematics==1.2.3,
// casadi==3.5.5 (reference pyproject.toml:17-20, poetry.lock:86-87,447-448,551-552); the
// reference's own tests hold no numeric vector (examples/test_examples.py:8-36).  This file
// restates the published algorithm of those packages as the reference's call sites use it, and is
// itself pinned by oracle/autodiff_oracle.py (every derivative by autograd, from the definitions)
// through the fixtures in tests/golden/.
//
// Structure deliberately mirrors the *library's* staged formulation, not the GPU kernel's folded
// closed forms: every leaf is built as a scalar spec (m, f), pulled through its task map(s) with
// dense Jacobians (Spec.pull: M_q = J^T M J, f_q = J^T (f + M c)), and summed.
//
// Reference call sites followed:
//   leaf set, strings, limits, mount, goals   examples/example_pandas_Jointspace.py:25-134
//   coupled rollout recurrence                multi_robot_fabrics/fabrics_planner/forward_planner_Jointspace.py:72-116,190-249
//   Cartesian rollout recurrence              multi_robot_fabrics/fabrics_planner/forward_planner_Cartesian.py:77-92,276-288,421-458
//   fk / J / "jac_dot" = -d(J qd)/dq          multi_robot_fabrics/utils/utils.py:16-54 (sign :28,:37), :87-119
//   goal estimate x_ee + 20*0.01*v_ee         examples/example_pandas_cartesian.py:355-357
//   chain constants                           examples/simulation_environments/urdfs/panda_with_finger.urdf:98-107,150-158,
//                                             201-209,253-261,326-334,378-386,451-459,461-465; pointRobot1.urdf:91-113
#include <omp.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/mrf.h"

namespace {

constexpr int DOF_MAX = 7;

struct Mat7 {
  double a[DOF_MAX][DOF_MAX];
};

struct QSpec {  // spec in configuration space
  double M[DOF_MAX][DOF_MAX];
  double f[DOF_MAX];
  void zero() { std::memset(this, 0, sizeof(*this)); }
};

// ---------------------------------------------------------------- leaf strings
double gate_value(int gate, double xdot) {
  if (gate == MRF_GATE_NONE) return 1.0;
  if (xdot < 0.0) return 1.0;
  if (xdot > 0.0) return 0.0;
  return 0.5;  // casadi: heaviside(0) = 0.5, sign(0) = 0
}

// value of  fn(x, xdot) / xdot^2   (the coefficient in front of xdot**2)
double leaf_coeff(const mrf_leaf_fn& fn, double x, double xdot) {
  double g = gate_value(fn.gate, xdot);
  if (fn.family == MRF_FAMILY_POW) return fn.k / std::pow(x, (double)fn.p) * g;
  return fn.k * (1.0 / (1.0 + fn.c * std::exp(-fn.s * x)) - 1.0) * g;
}
double geometry_h(const mrf_leaf_fn& fn, double x, double xdot) { return leaf_coeff(fn, x, xdot) * xdot * xdot; }
// M = d2/dxdot2 [ coeff(x) * gate(xdot) * xdot^2 ]; gate has zero derivative (casadi sign/heaviside)
double finsler_metric(const mrf_leaf_fn& fn, double x, double xdot) { return 2.0 * leaf_coeff(fn, x, xdot); }

// ---------------------------------------------------------------- small linear algebra
void cross(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// solve (A) h = b by Gaussian elimination with partial pivoting (n <= 7)
void solve(int n, const double A_in[DOF_MAX][DOF_MAX], const double* b_in, double* h) {
  double A[DOF_MAX][DOF_MAX + 1];
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) A[i][j] = A_in[i][j];
    A[i][n] = b_in[i];
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(A[r][c]) > std::fabs(A[piv][c])) piv = r;
    if (piv != c)
      for (int j = 0; j <= n; ++j) std::swap(A[piv][j], A[c][j]);
    for (int r = c + 1; r < n; ++r) {
      double l = A[r][c] / A[c][c];
      for (int j = c; j <= n; ++j) A[r][j] -= l * A[c][j];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = A[i][n];
    for (int j = i + 1; j < n; ++j) s -= A[i][j] * h[j];
    h[i] = s / A[i][i];
  }
}

// ---------------------------------------------------------------- kinematics
struct Frame {
  double R[3][3];
  double p[3];
};

void frame_mul(const Frame& a, const Frame& b, Frame& c) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += a.R[i][k] * b.R[k][j];
      c.R[i][j] = s;
    }
    double s = a.p[i];
    for (int k = 0; k < 3; ++k) s += a.R[i][k] * b.p[k];
    c.p[i] = s;
  }
}

const double kHalfPi = 1.5707963267948966;
// panda_joint1..7: origin xyz and roll (pitch = yaw = 0 for all of them); axis = local z
const double PANDA_XYZ[7][3] = {{0, 0, 0.333}, {0, 0, 0}, {0, -0.316, 0}, {0.0825, 0, 0}, {-0.0825, 0.384, 0}, {0, 0, 0}, {0.088, 0, 0}};
const double PANDA_ROLL[7] = {0, -kHalfPi, kHalfPi, kHalfPi, -kHalfPi, kHalfPi, kHalfPi};
const double PANDA_LINK8_Z = 0.107;

struct Chain {          // world-frame description of one robot at one q
  int dof;              // 7 | 3
  int n_links;          // 8 | 1
  Frame link[8];        // frames of link1..8 (planar3: base_link)
  double z[7][3];       // axis of joint i (revolute) or direction (prismatic)
  double o[7][3];       // a point on the axis of joint i
  int prismatic[7];
};

void panda_chain(const double* q, const double* mount, Chain& C) {
  C.dof = 7;
  C.n_links = 8;
  Frame T;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T.R[i][j] = mount[i * 4 + j];
    T.p[i] = mount[i * 4 + 3];
  }
  for (int i = 0; i < 7; ++i) {
    Frame Jo;  // Trans(xyz) * Rx(roll)
    double cr = std::cos(PANDA_ROLL[i]), sr = std::sin(PANDA_ROLL[i]);
    double Rx[3][3] = {{1, 0, 0}, {0, cr, -sr}, {0, sr, cr}};
    std::memcpy(Jo.R, Rx, sizeof(Rx));
    for (int k = 0; k < 3; ++k) Jo.p[k] = PANDA_XYZ[i][k];
    Frame Rz;
    double c = std::cos(q[i]), s = std::sin(q[i]);
    double Rzz[3][3] = {{c, -s, 0}, {s, c, 0}, {0, 0, 1}};
    std::memcpy(Rz.R, Rzz, sizeof(Rzz));
    Rz.p[0] = Rz.p[1] = Rz.p[2] = 0;
    Frame A, B;
    frame_mul(T, Jo, A);
    frame_mul(A, Rz, B);
    T = B;
    C.link[i] = T;
    for (int k = 0; k < 3; ++k) {
      C.z[i][k] = T.R[k][2];
      C.o[i][k] = T.p[k];
    }
    C.prismatic[i] = 0;
  }
  Frame L8;
  double I3[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  std::memcpy(L8.R, I3, sizeof(I3));
  L8.p[0] = 0;
  L8.p[1] = 0;
  L8.p[2] = PANDA_LINK8_Z;
  frame_mul(T, L8, C.link[7]);
}

void planar_chain(const double* q, const double* mount, Chain& C) {
  // pointRobot1.urdf: world -(prismatic x, origin z 0.05)-> -(prismatic y)-> -(revolute z)-> base_link
  (void)mount;
  C.dof = 3;
  C.n_links = 1;
  double c = std::cos(q[2]), s = std::sin(q[2]);
  double R[3][3] = {{c, -s, 0}, {s, c, 0}, {0, 0, 1}};
  std::memcpy(C.link[0].R, R, sizeof(R));
  C.link[0].p[0] = q[0];
  C.link[0].p[1] = q[1];
  C.link[0].p[2] = 0.05;
  double zz[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) {
      C.z[i][k] = zz[i][k];
      C.o[i][k] = C.link[0].p[k];
    }
  C.prismatic[0] = C.prismatic[1] = 1;
  C.prismatic[2] = 0;
}

struct PointKin {  // a point rigidly attached to a link
  double p[3];
  double J[3][DOF_MAX];
  double v[3];      // J qd
  double jdqd[3];   // d(J qd)/dq qd  (true Jdot qd), *without* the sign convention
};

// number of joints that move link `link` (1-based)
int joints_moving(const Chain& C, int link) { return C.dof == 7 ? (link < 7 ? link : 7) : 3; }

void point_kin(const Chain& C, int link, const double* offset, const double* qd, PointKin& K) {
  const Frame& F = C.link[link - 1];
  for (int i = 0; i < 3; ++i) K.p[i] = F.p[i] + F.R[i][0] * offset[0] + F.R[i][1] * offset[1] + F.R[i][2] * offset[2];
  int nj = joints_moving(C, link);
  std::memset(K.J, 0, sizeof(K.J));
  for (int j = 0; j < nj; ++j) {
    double col[3];
    if (C.prismatic[j]) {
      col[0] = C.z[j][0]; col[1] = C.z[j][1]; col[2] = C.z[j][2];
    } else {
      double r[3] = {K.p[0] - C.o[j][0], K.p[1] - C.o[j][1], K.p[2] - C.o[j][2]};
      cross(C.z[j], r, col);
    }
    for (int i = 0; i < 3; ++i) K.J[i][j] = col[i];
  }
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int j = 0; j < nj; ++j) s += K.J[i][j] * qd[j];
    K.v[i] = s;
  }
  // second derivative: d2p/dqa dqb = z_a x (z_b x (p - o_b)) for a <= b (revolute, serial chain);
  // any prismatic joint in the pair contributes 0 here because the planar robot's prismatic axes
  // are fixed in the world and precede the revolute joint.
  K.jdqd[0] = K.jdqd[1] = K.jdqd[2] = 0;
  for (int a = 0; a < nj; ++a)
    for (int b = 0; b < nj; ++b) {
      int lo = a < b ? a : b, hi = a < b ? b : a;
      if (C.prismatic[lo] || C.prismatic[hi]) continue;
      double r[3] = {K.p[0] - C.o[hi][0], K.p[1] - C.o[hi][1], K.p[2] - C.o[hi][2]};
      double t[3], h[3];
      cross(C.z[hi], r, t);
      cross(C.z[lo], t, h);
      for (int i = 0; i < 3; ++i) K.jdqd[i] += h[i] * qd[a] * qd[b];
    }
}

void build_chain(const mrf_config& cfg, int robot, const double* q, Chain& C) {
  if (cfg.model == MRF_MODEL_PANDA7)
    panda_chain(q, cfg.mount[robot], C);
  else
    planar_chain(q, cfg.mount[robot], C);
}

// ---------------------------------------------------------------- pullbacks
// generic Spec.pull for a task of dimension d (<=3): M_q += J^T M J, f_q += J^T (f + M c)
void pull_add(QSpec& S, int n, int d, const double M[3][3], const double* f, const double J[3][DOF_MAX], const double* c) {
  double t[3];
  for (int i = 0; i < d; ++i) {
    t[i] = f[i];
    for (int k = 0; k < d; ++k) t[i] += M[i][k] * c[k];
  }
  for (int a = 0; a < n; ++a) {
    for (int b = 0; b < n; ++b) {
      double s = 0;
      for (int i = 0; i < d; ++i)
        for (int k = 0; k < d; ++k) s += J[i][a] * M[i][k] * J[k][b];
      S.M[a][b] += s;
    }
    double s = 0;
    for (int i = 0; i < d; ++i) s += J[i][a] * t[i];
    S.f[a] += s;
  }
}

struct Obst {
  double x[3], v[3], a[3], r;
  bool is_static = false;
};

// spherical obstacle leaf, the 3-stage pull of `fabrics` (geometry map, dynamic map, fk):
//   leaf        x = |x_rel|/(r_o + r_b) - 1       (m, f) from the collision strings
//   pull 1      through g(x_rel)                  -> 3x3 spec on x_rel
//   dyn. pull   x_rel = p - x_ref                 f -= M xdd_ref
//   pull 2      through fk p(q)
void obstacle_leaf(const mrf_config& cfg, QSpec& S, int n, const PointKin& K, double r_body, const Obst& o) {
  int d = o.is_static ? 3 : cfg.obst_dim;  // static spheres: 3-D distance; dynamic: dynamic_obstacle_dimension
  double xr[3] = {0, 0, 0}, vr[3] = {0, 0, 0};
  double nn = 0, vv = 0;
  for (int i = 0; i < d; ++i) {
    xr[i] = K.p[i] - o.x[i];
    vr[i] = K.v[i] - o.v[i];
    nn += xr[i] * xr[i];
    vv += vr[i] * vr[i];
  }
  double dist = std::sqrt(nn);
  double R = o.r + r_body;
  double nvec[3] = {xr[0] / dist, xr[1] / dist, xr[2] / dist};
  double x = dist / R - 1.0;
  double nv = 0;
  for (int i = 0; i < d; ++i) nv += nvec[i] * vr[i];
  double xdot = nv / R;
  double m = finsler_metric(cfg.collision_finsler, x, xdot);
  double f = m * geometry_h(cfg.collision_geometry, x, xdot);
  // pull 1: J_g = n^T / R ; Hessian of g = (I - n n^T)/(dist R) ; c_g = sign * vr^T H vr
  double cg = cfg.jdot_sign * (vv - nv * nv) / (dist * R);
  double M3[3][3], f3[3];
  for (int i = 0; i < 3; ++i) {
    for (int k = 0; k < 3; ++k) M3[i][k] = (i < d && k < d) ? nvec[i] * m * nvec[k] / (R * R) : 0.0;
    f3[i] = i < d ? nvec[i] / R * (f + m * cg) : 0.0;
  }
  // dynamic pull
  for (int i = 0; i < d; ++i)
    for (int k = 0; k < d; ++k) f3[i] -= M3[i][k] * o.a[k];
  // pull 2
  double c[3] = {cfg.jdot_sign * K.jdqd[0], cfg.jdot_sign * K.jdqd[1], cfg.jdot_sign * K.jdqd[2]};
  pull_add(S, n, d, M3, f3, K.J, c);
}

void plane_leaf(const mrf_config& cfg, QSpec& S, int n, const PointKin& K, double r_body, const double* con) {
  double na = std::sqrt(con[0] * con[0] + con[1] * con[1] + con[2] * con[2]);
  double val = con[0] * K.p[0] + con[1] * K.p[1] + con[2] * K.p[2] + con[3];
  double sg = 1.0;
  if (cfg.plane_abs && val < 0) sg = -1.0;
  if (cfg.plane_abs && val == 0) sg = 0.0;
  double x = sg * val / na - r_body;
  double Jx[3][DOF_MAX];
  std::memset(Jx, 0, sizeof(Jx));
  double xdot = 0, cx = 0;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < n; ++j) Jx[0][j] += sg * con[i] / na * K.J[i][j];
    xdot += sg * con[i] / na * K.v[i];
    cx += cfg.jdot_sign * sg * con[i] / na * K.jdqd[i];
  }
  double m = finsler_metric(cfg.plane_finsler, x, xdot);
  double f = m * geometry_h(cfg.plane_geometry, x, xdot);
  double M1[3][3] = {{m, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double f1[3] = {f, 0, 0}, c1[3] = {cx, 0, 0};
  pull_add(S, n, 1, M1, f1, Jx, c1);
}

void limit_leaves(const mrf_config& cfg, QSpec& S, int n, const double* q, const double* qd) {
  for (int j = 0; j < n; ++j)
    for (int side = 0; side < 2; ++side) {
      double sgn = side == 0 ? 1.0 : -1.0;
      double x = side == 0 ? q[j] - cfg.limits[j][0] : cfg.limits[j][1] - q[j];
      double xdot = sgn * qd[j];
      double m = finsler_metric(cfg.limit_finsler, x, xdot);
      double f = m * geometry_h(cfg.limit_geometry, x, xdot);
      S.M[j][j] += m;       // J = +-e_j, J^T m J = m
      S.f[j] += sgn * f;    // c = 0
    }
}

// How ca.norm_2(x) of the attractor strings behaves at / near x = 0 is a property of CasADi that this container cannot
// check (DESIGN.md "deviations" 1).  The oracle -- test infrastructure -- can evaluate the candidates so that the real
// reference's vectors decide (tests/reconcile_constants.py); the kernels implement mode 0.  mrfo_set_attractor_norm:
//   0  build convention: r = sqrt(x.x), gradient of the potential defined as 0 at r == 0
//   1  regularised norm: r = sqrt(x.x + eps) everywhere (gradient x / r, finite)
//   2  CasADi as recalled: a 1-D norm_2 is simplified to |x| (sqrt(sq(x)) -> fabs), derivative sign(x) = 0 at 0; a 3-D one
//      keeps x / sqrt(x.x) = 0/0 = NaN at exactly x == 0.  Equal to mode 0 wherever mode 0 is defined by more than convention.
//   3  no simplification: x / sqrt(x.x) in every dimension, NaN at exactly x == 0 (what DESIGN r1-r4 assumed of the reference)
int g_attr_norm_mode = 0;
double g_attr_norm_eps = 0.0;

double attractor_norm(const double* x, int d) {
  double r = 0;
  for (int i = 0; i < d; ++i) r += x[i] * x[i];
  return std::sqrt(g_attr_norm_mode == 1 ? r + g_attr_norm_eps : r);
}

// attractor leaf on a task x (dim d) with Jacobian J and curvature term c
void attractor_leaf(const mrf_config& cfg, QSpec& S, int n, int d, const double* x, const double J[3][DOF_MAX],
                    const double* c, double w) {
  const double r = attractor_norm(x, d);
  double A = (cfg.attr_mu - cfg.attr_ml) * std::exp(-(cfg.attr_a * r) * (cfg.attr_a * r)) + cfg.attr_ml;
  double M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double f[3] = {0, 0, 0};
  for (int i = 0; i < d; ++i) {
    M[i][i] = 2.0 * A;  // L = xdot^T A xdot
    const double gain = w * cfg.attr_k * std::tanh(cfg.attr_alpha * r);
    double grad;
    if (g_attr_norm_mode == 0)
      grad = r > 0 ? gain * x[i] / r : 0.0;  // build convention at r == 0
    else if (g_attr_norm_mode == 2 && d == 1)
      grad = gain * ((x[i] > 0) - (x[i] < 0));
    else
      grad = gain * (x[i] / r);              // 0/0 = NaN at x == 0 unless regularised
    f[i] = M[i][i] * grad;
  }
  pull_add(S, n, d, M, f, J, c);
}

struct Row {  // one (scenario, robot) evaluation
  const double* q;
  const double* qd;
  const double* prm;  // MRF_NPARAM
};

void solve_fabric(const mrf_config& cfg, int robot, const Row& row, const std::vector<Obst>& obst, double* qdd,
                  double* action, QSpec* geo_out = nullptr, QSpec* forced_out = nullptr) {
  Chain C;
  build_chain(cfg, robot, row.q, C);
  const int n = C.dof;
  QSpec geo;
  geo.zero();
  for (int j = 0; j < n; ++j) geo.M[j][j] = cfg.base_mass;  // base geometry h = 0

  const double zero3[3] = {0, 0, 0};
  for (int e = 0; e < cfg.n_ego; ++e) {
    int link = cfg.model == MRF_MODEL_PANDA7 ? 3 + e : 1;
    if (cfg.model == MRF_MODEL_PANDA7 && !((cfg.ego_link_mask >> e) & 1)) continue;  // collision_links subset (EXJ:91-96)
    PointKin K;
    point_kin(C, link, zero3, row.qd, K);
    double r_body = row.prm[MRF_P_RADIUS_BODY + e];
    for (const Obst& o : obst) obstacle_leaf(cfg, geo, n, K, r_body, o);
    for (int pl = 0; pl < cfg.n_planes; ++pl) plane_leaf(cfg, geo, n, K, r_body, row.prm + MRF_P_CONSTRAINT_0);
  }
  if (cfg.use_limits) limit_leaves(cfg, geo, n, row.q, row.qd);

  QSpec forced = geo;
  double x_psi_norm = 0;
  if (cfg.n_goals > 0) {
    int ee_link = cfg.model == MRF_MODEL_PANDA7 ? 8 : 1;
    PointKin K8;
    point_kin(C, ee_link, zero3, row.qd, K8);
    int d0 = cfg.model == MRF_MODEL_PANDA7 ? 3 : 2;
    double x0[3] = {0, 0, 0}, c0[3] = {0, 0, 0};
    for (int i = 0; i < d0; ++i) {
      x0[i] = K8.p[i] - row.prm[MRF_P_X_GOAL_0 + i];
      c0[i] = cfg.jdot_sign * K8.jdqd[i];
      x_psi_norm += x0[i] * x0[i];
    }
    x_psi_norm = std::sqrt(g_attr_norm_mode == 1 ? x_psi_norm + g_attr_norm_eps : x_psi_norm);
    attractor_leaf(cfg, forced, n, d0, x0, K8.J, c0, row.prm[MRF_P_WEIGHT_GOAL_0]);
    if (cfg.n_goals > 1) {  // R (p_hand - p_link7) - x_goal_1
      PointKin K7;
      point_kin(C, 7, zero3, row.qd, K7);
      const double* Rm = row.prm + MRF_P_ANGLE_GOAL_1;
      double x1[3], c1[3], J1[3][DOF_MAX];
      for (int i = 0; i < 3; ++i) {
        x1[i] = -row.prm[MRF_P_X_GOAL_1 + i];
        c1[i] = 0;
        for (int k = 0; k < 3; ++k) {
          x1[i] += Rm[i * 3 + k] * (K8.p[k] - K7.p[k]);
          c1[i] += cfg.jdot_sign * Rm[i * 3 + k] * (K8.jdqd[k] - K7.jdqd[k]);
        }
        for (int j = 0; j < DOF_MAX; ++j) {
          J1[i][j] = 0;
          for (int k = 0; k < 3; ++k) J1[i][j] += Rm[i * 3 + k] * (K8.J[k][j] - K7.J[k][j]);
        }
      }
      attractor_leaf(cfg, forced, n, 3, x1, J1, c1, row.prm[MRF_P_WEIGHT_GOAL_1]);
    }
    if (cfg.n_goals > 2) {  // joint 6 -> x_goal_2
      double x2[3] = {row.q[6] - row.prm[MRF_P_X_GOAL_2], 0, 0};
      double J2[3][DOF_MAX];
      std::memset(J2, 0, sizeof(J2));
      J2[0][6] = 1.0;
      attractor_leaf(cfg, forced, n, 1, x2, J2, zero3, row.prm[MRF_P_WEIGHT_GOAL_2]);
    }
  }
  if (geo_out) *geo_out = geo;
  if (forced_out) *forced_out = forced;

  double Mg[DOF_MAX][DOF_MAX], Mf[DOF_MAX][DOF_MAX], hg[DOF_MAX], hf[DOF_MAX];
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b) {
      Mg[a][b] = geo.M[a][b] + (a == b ? cfg.eps : 0.0);
      Mf[a][b] = forced.M[a][b] + (a == b ? cfg.eps : 0.0);
    }
  solve(n, Mg, geo.f, hg);
  double qq = 0, qhg = 0, qhf = 0;
  for (int j = 0; j < n; ++j) {
    qq += row.qd[j] * row.qd[j];
    qhg += row.qd[j] * hg[j];
  }
  double alpha_g = -qhg / (cfg.eps + qq);
  if (cfg.n_goals == 0) {
    for (int j = 0; j < n; ++j) qdd[j] = -hg[j] - alpha_g * row.qd[j];
  } else {
    solve(n, Mf, forced.f, hf);
    for (int j = 0; j < n; ++j) qhf += row.qd[j] * hf[j];
    double alpha_f = -qhf / (cfg.eps + qq);
    double eta = 0.5 * (std::tanh(-cfg.eta_a * qq - cfg.eta_s) + 1.0);
    double a_ex = eta * alpha_g + (1.0 - eta) * alpha_f;
    // substitute_beta(-a_ex, -alpha_g):  max(0, a_ex_sym - a_le_sym) = max(0, alpha_g - a_ex)
    double beta = 0.5 * (std::tanh(-cfg.beta_a * (x_psi_norm - cfg.beta_r)) + 1.0) * cfg.beta_b + cfg.beta_s +
                  std::fmax(0.0, (-a_ex) - (-alpha_g));
    for (int j = 0; j < n; ++j) qdd[j] = -hf[j] - (a_ex + beta) * row.qd[j];
  }
  double nrm = 0;
  for (int j = 0; j < n; ++j) {
    action[j] = cfg.mode == MRF_MODE_VEL ? row.qd[j] + cfg.dt * qdd[j] : qdd[j];
    nrm += action[j] * action[j];
  }
  if (cfg.zero_small_action && std::sqrt(nrm) < cfg.eps)
    for (int j = 0; j < n; ++j) action[j] = 0;
}

int dof_of(const mrf_config& cfg) { return cfg.model == MRF_MODEL_PANDA7 ? 7 : 3; }

// spheres of one robot: x, v = J qd, a = jdot_sign * Jdot qd   (FPJ:82-100 with qddot = 0)
void robot_spheres(const mrf_config& cfg, int robot, const double* q, const double* qd, double (*x)[3], double (*v)[3],
                   double (*a)[3]) {
  Chain C;
  build_chain(cfg, robot, q, C);
  for (int s = 0; s < cfg.n_spheres; ++s) {
    PointKin K;
    point_kin(C, cfg.sphere_link[s], cfg.sphere_offset[s], qd, K);
    for (int i = 0; i < 3; ++i) {
      x[s][i] = K.p[i];
      v[s][i] = K.v[i];
      a[s][i] = cfg.jdot_sign * K.jdqd[i];
    }
  }
}

}  // namespace

extern "C" {

// candidate behaviours of ca.norm_2 in the attractor strings (see g_attr_norm_mode); process-wide, tests only
void mrfo_set_attractor_norm(int mode, double eps) {
  g_attr_norm_mode = mode;
  g_attr_norm_eps = eps;
}

// threads used by the batch loops below (one scenario / row per thread); returns the previous maximum
int mrfo_set_threads(int n) {
  int prev = omp_get_max_threads();
  if (n > 0) omp_set_num_threads(n);
  return prev;
}

// Same array layouts as include/mrf.h, host pointers, double only.
int mrfo_compute_action(const mrf_config* cfg, int64_t rows, const double* q, const double* qdot, const double* params,
                        int32_t n_obst, int32_t n_static, const double* ox, const double* ov, const double* oa, const double* orad,
                        double* qddot_out, double* action_out) {
  const int n = dof_of(*cfg);
#pragma omp parallel
  {
  std::vector<Obst> obst(n_obst);  // one scratch list per thread
#pragma omp for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    double qv[DOF_MAX], qdv[DOF_MAX], prm[MRF_NPARAM], qdd[DOF_MAX], act[DOF_MAX];
    for (int j = 0; j < n; ++j) {
      qv[j] = q[j * rows + r];
      qdv[j] = qdot[j * rows + r];
    }
    for (int p = 0; p < MRF_NPARAM; ++p) prm[p] = params[p * rows + r];
    for (int m = 0; m < n_obst; ++m) {
      for (int c = 0; c < 3; ++c) {
        obst[m].x[c] = ox[(m * 3 + c) * rows + r];
        obst[m].v[c] = ov ? ov[(m * 3 + c) * rows + r] : 0.0;
        obst[m].a[c] = oa ? oa[(m * 3 + c) * rows + r] : 0.0;
      }
      obst[m].r = orad[m * rows + r];
      obst[m].is_static = m < n_static;
      if (obst[m].is_static)
        for (int c = 0; c < 3; ++c) obst[m].v[c] = obst[m].a[c] = 0.0;
    }
    Row row{qv, qdv, prm};
    solve_fabric(*cfg, (int)(r % cfg->n_robots), row, obst, qdd, act);
    for (int j = 0; j < n; ++j) {
      if (qddot_out) qddot_out[j * rows + r] = qdd[j];
      action_out[j * rows + r] = act[j];
    }
  }
  }
  return 0;
}

// geometry / forced (M, f) of one row, for per-leaf-level debugging against the autodiff oracle
int mrfo_specs(const mrf_config* cfg, int32_t robot, const double* q, const double* qdot, const double* params,
               int32_t n_obst, int32_t n_static, const double* ox, const double* ov, const double* oa, const double* orad, double* Mg,
               double* fg, double* Mf, double* ff) {
  std::vector<Obst> obst(n_obst);
  for (int m = 0; m < n_obst; ++m) {
    for (int c = 0; c < 3; ++c) {
      obst[m].x[c] = ox[m * 3 + c];
      obst[m].v[c] = ov ? ov[m * 3 + c] : 0.0;
      obst[m].a[c] = oa ? oa[m * 3 + c] : 0.0;
    }
    obst[m].r = orad[m];
    obst[m].is_static = m < n_static;
    if (obst[m].is_static)
      for (int c = 0; c < 3; ++c) obst[m].v[c] = obst[m].a[c] = 0.0;
  }
  double qdd[DOF_MAX], act[DOF_MAX];
  QSpec g, f;
  Row row{q, qdot, params};
  solve_fabric(*cfg, robot, row, obst, qdd, act, &g, &f);
  for (int a = 0; a < DOF_MAX; ++a) {
    for (int b = 0; b < DOF_MAX; ++b) {
      Mg[a * DOF_MAX + b] = g.M[a][b];
      Mf[a * DOF_MAX + b] = f.M[a][b];
    }
    fg[a] = g.f[a];
    ff[a] = f.f[a];
  }
  return 0;
}

int mrfo_fk_spheres(const mrf_config* cfg, int64_t rows, const double* q, const double* qdot, double* x_out,
                    double* v_out, double* a_out) {
  const int n = dof_of(*cfg);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    double qv[DOF_MAX], qdv[DOF_MAX];
    for (int j = 0; j < n; ++j) {
      qv[j] = q[j * rows + r];
      qdv[j] = qdot[j * rows + r];
    }
    double x[MRF_MAX_SPHERES][3], v[MRF_MAX_SPHERES][3], a[MRF_MAX_SPHERES][3];
    robot_spheres(*cfg, (int)(r % cfg->n_robots), qv, qdv, x, v, a);
    for (int s = 0; s < cfg->n_spheres; ++s)
      for (int c = 0; c < 3; ++c) {
        x_out[(s * 3 + c) * rows + r] = x[s][c];
        if (v_out) v_out[(s * 3 + c) * rows + r] = v[s][c];
        if (a_out) a_out[(s * 3 + c) * rows + r] = a[s][c];
      }
  }
  return 0;
}

// Coupled joint-space rollout, forward_planner_Jointspace.py:190-249.  Only the 'vel' mode the reference
// runs (parameters_manipulators.py:12) is defined: in 'acc' mode FPJ:233 would assign an acceleration
// to q_dot, so that branch of the reference is not a usable specification.
int mrfo_rollout(const mrf_config* cfg, int64_t n_scen, const double* q0, const double* qdot0, const double* params,
                 double* avg_out, double* traj_q, double* traj_qd) {
  const int n = dof_of(*cfg), N = cfg->n_robots, S = cfg->n_spheres, H = cfg->horizon;
  const int64_t rows = n_scen * N;
  if (cfg->mode != MRF_MODE_VEL) return -2;
#pragma omp parallel
  {
  std::vector<Obst> obst((size_t)S * (N - 1));  // one scratch list per thread
#pragma omp for schedule(static)
  for (int64_t sc = 0; sc < n_scen; ++sc) {
    double q[MRF_MAX_ROBOTS][DOF_MAX], qd[MRF_MAX_ROBOTS][DOF_MAX];
    double prm[MRF_MAX_ROBOTS][MRF_NPARAM], sumsq[MRF_MAX_ROBOTS];
    for (int i = 0; i < N; ++i) {
      int64_t r = sc * N + i;
      for (int j = 0; j < n; ++j) {
        q[i][j] = q0[j * rows + r];
        qd[i][j] = qdot0[j * rows + r];
      }
      for (int p = 0; p < MRF_NPARAM; ++p) prm[i][p] = params[p * rows + r];
      sumsq[i] = 0;
    }
    // RF-CV goal estimate: x_goal_0 <- x_ee + T * J_ee qdot at the initial state (EXC:355-357)
    for (int i = 0; i < N; ++i)
      if (cfg->goal_estimate_mask & (1 << i)) {
        Chain C;
        build_chain(*cfg, i, q[i], C);
        PointKin K;
        const double z3[3] = {0, 0, 0};
        point_kin(C, cfg->model == MRF_MODEL_PANDA7 ? 8 : 1, z3, qd[i], K);
        for (int c = 0; c < 3; ++c) prm[i][MRF_P_X_GOAL_0 + c] = K.p[c] + cfg->goal_estimate_T * K.v[c];
      }
    for (int k = 0; k < H; ++k) {
      double sx[MRF_MAX_ROBOTS][MRF_MAX_SPHERES][3], sv[MRF_MAX_ROBOTS][MRF_MAX_SPHERES][3], sa[MRF_MAX_ROBOTS][MRF_MAX_SPHERES][3];
      for (int i = 0; i < N; ++i) {
        for (int j = 0; j < n; ++j) q[i][j] += cfg->dt * qd[i][j];  // system_step 'vel', FPJ:77-80
        robot_spheres(*cfg, i, q[i], qd[i], sx[i], sv[i], sa[i]);
      }
      double act[MRF_MAX_ROBOTS][DOF_MAX], acc[MRF_MAX_ROBOTS][DOF_MAX];
      for (int i = 0; i < N; ++i) {
        size_t m = 0;
        for (int jr = 0; jr < N; ++jr) {
          if (jr == i) continue;
          for (int s = 0; s < S; ++s, ++m) {
            for (int c = 0; c < 3; ++c) {
              obst[m].x[c] = sx[jr][s][c];
              obst[m].v[c] = cfg->dynamic ? sv[jr][s][c] : 0.0;
              obst[m].a[c] = cfg->dynamic ? sa[jr][s][c] : 0.0;
            }
            obst[m].r = cfg->sphere_radius[s];
          }
        }
        Row row{q[i], qd[i], prm[i]};
        solve_fabric(*cfg, i, row, obst, acc[i], act[i]);
      }
      for (int i = 0; i < N; ++i) {
        int64_t r = sc * N + i;
        for (int j = 0; j < n; ++j) {
          qd[i][j] = act[i][j];  // FPJ:233 (q_ddot stays 0, FPJ:202)
          sumsq[i] += qd[i][j] * qd[i][j];
          if (traj_q) traj_q[((int64_t)k * n + j) * rows + r] = q[i][j];
          if (traj_qd) traj_qd[((int64_t)k * n + j) * rows + r] = qd[i][j];
        }
      }
    }
    for (int i = 0; i < N; ++i) avg_out[sc * N + i] = sumsq[i] / (double)(H * n);
  }
  }
  return 0;
}

// Cartesian constant-velocity rollout, forward_planner_Cartesian.py:421-458: action first, then the
// system step, then every obstacle x += dt * v.
int mrfo_rollout_cartesian(const mrf_config* cfg, int64_t rows, const double* q0, const double* qdot0,
                           const double* params, int32_t n_obst, int32_t n_static, const double* ox0, const double* ov, const double* oa,
                           const double* orad, double* avg_out, double* traj_q, double* traj_qd) {
  const int n = dof_of(*cfg), H = cfg->horizon;
#pragma omp parallel
  {
  std::vector<Obst> obst(n_obst);  // one scratch list per thread
#pragma omp for schedule(static)
  for (int64_t r = 0; r < rows; ++r) {
    double q[DOF_MAX], qd[DOF_MAX], prm[MRF_NPARAM], qdd[DOF_MAX], act[DOF_MAX];
    for (int j = 0; j < n; ++j) {
      q[j] = q0[j * rows + r];
      qd[j] = qdot0[j * rows + r];
    }
    for (int p = 0; p < MRF_NPARAM; ++p) prm[p] = params[p * rows + r];
    for (int m = 0; m < n_obst; ++m) {
      for (int c = 0; c < 3; ++c) {
        obst[m].x[c] = ox0[(m * 3 + c) * rows + r];
        obst[m].v[c] = (ov && m >= n_static) ? ov[(m * 3 + c) * rows + r] : 0.0;
        obst[m].a[c] = (oa && m >= n_static) ? oa[(m * 3 + c) * rows + r] : 0.0;
      }
      obst[m].r = orad[m * rows + r];
      obst[m].is_static = m < n_static;
    }
    double sumsq = 0;
    for (int k = 0; k < H; ++k) {
      Row row{q, qd, prm};
      solve_fabric(*cfg, (int)(r % cfg->n_robots), row, obst, qdd, act);
      for (int j = 0; j < n; ++j) {
        if (cfg->mode == MRF_MODE_VEL) {
          qd[j] = act[j];
          q[j] += cfg->dt * qd[j];
        } else {
          q[j] += cfg->dt * qd[j] + 0.5 * cfg->dt * cfg->dt * act[j];
          qd[j] += cfg->dt * act[j];
        }
        sumsq += qd[j] * qd[j];
        if (traj_q) traj_q[((int64_t)k * n + j) * rows + r] = q[j];
        if (traj_qd) traj_qd[((int64_t)k * n + j) * rows + r] = qd[j];
      }
      for (int m = 0; m < n_obst; ++m)
        for (int c = 0; c < 3; ++c) obst[m].x[c] += cfg->dt * obst[m].v[c];
    }
    avg_out[r] = sumsq / (double)(H * n);
  }
  }
  return 0;
}

}  // extern "C"
