"""Independent autodiff oracle for the fabric solve  --  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic of the reference lives in the un-vendored third-party
packages fabrics==0.9.5, forwardkinematics==1.2.3, casadi==3.5.5 (reference
pyproject.toml:17-20, poetry.lock:86-87,447-448,551-552), none of which is importable
here, and the reference's own tests hold no numeric vectors (examples/test_examples.py:8-36).
This file restates the *published algorithm* of optimization fabrics as the `fabrics`
package implements it (task map -> differential map -> pullback -> sum -> energize ->
damp), from the definitions, with torch float64 autograd doing every derivative.
Nothing here is a closed form: Jacobians, J-dot-qdot terms and Finsler metrics all come
out of autograd, so this is the independent cross-check of the hand-derived closed forms
in oracle/mrf_oracle.cpp (the definition of record) and of the HIP kernels.

Only tests/, __graft_entry__.smoke() and tests/golden/make_golden.py may import it.

Reference call sites this follows:
  leaf set / strings / limits / mount / goals : examples/example_pandas_Jointspace.py:25-134
  point-robot planner                          : examples/example_pointmasses_static.py:102-129,
                                                 examples/example_pointmasses_dynamic.py:102-131
  J-dot sign convention                        : multi_robot_fabrics/utils/utils.py:28,37
  Panda chain constants                        : examples/simulation_environments/urdfs/panda_with_finger.urdf:98-107,
                                                 150-158,201-209,253-261,326-334,378-386,451-459,461-470
"""
import math

import numpy as np
import torch

DT = torch.float64
EPS = 1e-6          # fabrics.helpers.constants.eps  [RECALL]
JDOT_SIGN = -1.0    # DifferentialMap default; mirrored by reference utils.py:28


# --------------------------------------------------------------------------------------
# casadi shim: just enough of `ca.*` for the leaf strings the reference passes
# --------------------------------------------------------------------------------------
class _Ca:
    @staticmethod
    def exp(x):
        return torch.exp(torch.as_tensor(x, dtype=DT))

    @staticmethod
    def log(x):
        return torch.log(x)

    @staticmethod
    def tanh(x):
        return torch.tanh(x)

    @staticmethod
    def sign(x):
        return torch.sign(x)          # zero gradient, sign(0)=0 : same as casadi

    @staticmethod
    def heaviside(x):
        return 0.5 * (torch.sign(x) + 1.0)   # casadi: heaviside(0)=0.5, zero gradient

    @staticmethod
    def norm_2(x):
        # Gradient at x == 0 exactly: defined as 0 (the limit of the attractor force tanh(a r) x/r).  The
        # reference's own start pose sits there for the 1-D attractor (q[6] == x_goal_2 == pi/4, PM:93 / EXJ:428)
        # and evidently runs, so the earlier reading "casadi gives 0/0 = NaN" cannot hold for it: as recalled,
        # casadi simplifies sqrt(sq(x)) of a scalar to fabs(x), whose derivative sign(x) is 0 at 0 -- this
        # convention.  A 3-D task exactly on its goal is the only place where casadi may still give 0/0.
        # DESIGN.md "deviations" 1; tests/reconcile_constants.py enumerates the candidates.
        s = torch.sum(x * x)
        safe = torch.where(s > 0, s, torch.ones_like(s))
        return torch.where(s > 0, torch.sqrt(safe), torch.zeros_like(s))

    @staticmethod
    def dot(a, b):
        return torch.sum(a * b)

    @staticmethod
    def fmax(a, b):
        a = torch.as_tensor(a, dtype=DT)
        b = torch.as_tensor(b, dtype=DT)
        return torch.maximum(a, b)

    @staticmethod
    def fabs(x):
        return torch.abs(x)

    @staticmethod
    def SX(a):
        return torch.as_tensor(np.asarray(a), dtype=DT)


class _SizedTensor(torch.Tensor):
    pass


def _eval_str(expr, **names):
    ns = {"ca": _Ca, "np": np}
    ns.update(names)
    return eval(expr, {"__builtins__": {}}, ns)


# default strings of fabrics' FabricPlannerConfig [RECALL, unverified; overridable]
DEFAULTS = dict(
    base_energy="0.5 * 0.2 * ca.dot(xdot, xdot)",
    collision_geometry="-0.5 / (x ** 5) * (-0.5 * (ca.sign(xdot) - 1)) * xdot ** 2",
    collision_finsler="0.1/(x ** 1) * (-0.5 * (ca.sign(xdot) - 1)) * xdot**2",
    limit_geometry="-0.1 / (x ** 1) * xdot ** 2",
    limit_finsler="0.1/(x**1) * (-0.5 * (ca.sign(xdot) - 1)) * xdot**2",
    geometry_plane_constraint="-0.5 / (x ** 5) * (-0.5 * (ca.sign(xdot) - 1)) * xdot ** 2",
    finsler_plane_constraint="0.1/(x ** 1) * (-0.5 * (ca.sign(xdot) - 1)) * xdot**2",
    attractor_potential="5.0 * (ca.norm_2(x) + 1 / 10 * ca.log(1 + ca.exp(-2 * 10 * ca.norm_2(x))))",
    attractor_metric="((2.0 - 0.3) * ca.exp(-1 * (0.75 * ca.norm_2(x))**2) + 0.3) * ca.SX(np.identity(x.size()[0]))",
    damper_beta="0.5 * (ca.tanh(-0.5 * (ca.norm_2(x) - 0.02)) + 1) * 6.5 + 0.01 + ca.fmax(0, a_ex - a_le)",
    damper_eta="0.5 * (ca.tanh(-0.9 * (1 - 1/2) * ca.dot(xdot, xdot) - 0.5) + 1)",
)

# strings the reference overrides for the Pandas (example_pandas_Jointspace.py:87-89)
PANDA_OVERRIDES = dict(
    geometry_plane_constraint="10*(1/(1+1*ca.exp(-10*x))-1) * (xdot**2)",
    collision_geometry="-0.5 / (x ** 4) * (xdot ** 2)",
    collision_finsler="0.01/(x**4) * xdot**2",
)
# strings the reference overrides for the point robots (example_pointmasses_static.py:106-107)
POINT_OVERRIDES = dict(
    collision_geometry="-2.0 / (x ** 1) * xdot ** 2",
    collision_finsler="1.0/(x**2) * (1 - ca.heaviside(xdot))* xdot**2",
)

PANDA_LIMITS = [[-2.8973, 2.8973], [-1.7628, 1.7628], [-2.8973, 2.8973], [-3.0718, -0.0698],
                [-2.8973, 2.8973], [-0.0175, 3.7525], [-2.8973, 2.8973]]  # EXJ:97-105

# (xyz, rpy) of panda_joint1..7 from the URDF
_HP = math.pi / 2
PANDA_JOINTS = [
    ((0.0, 0.0, 0.333), (0.0, 0.0, 0.0)),
    ((0.0, 0.0, 0.0), (-_HP, 0.0, 0.0)),
    ((0.0, -0.316, 0.0), (_HP, 0.0, 0.0)),
    ((0.0825, 0.0, 0.0), (_HP, 0.0, 0.0)),
    ((-0.0825, 0.384, 0.0), (-_HP, 0.0, 0.0)),
    ((0.0, 0.0, 0.0), (_HP, 0.0, 0.0)),
    ((0.088, 0.0, 0.0), (_HP, 0.0, 0.0)),
]
PANDA_LINK8_XYZ = (0.0, 0.0, 0.107)


def _t(v):
    return torch.as_tensor(v, dtype=DT)


def _rpy_xyz(xyz, rpy):
    r, p, y = rpy
    cr, sr, cp, sp, cy, sy = math.cos(r), math.sin(r), math.cos(p), math.sin(p), math.cos(y), math.sin(y)
    R = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                  [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                  [-sp, cp * sr, cp * cr]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = xyz
    return _t(T)


def _rz(a):
    c, s = torch.cos(a), torch.sin(a)
    z, o = torch.zeros((), dtype=DT), torch.ones((), dtype=DT)
    return torch.stack([torch.stack([c, -s, z, z]), torch.stack([s, c, z, z]),
                        torch.stack([z, z, o, z]), torch.stack([z, z, z, o])])


def panda_link_frames(q, mount):
    """4x4 world frames of panda_link1..8 (index 0..7) for joint vector q (7,)."""
    T = _t(mount)
    frames = []
    for i, (xyz, rpy) in enumerate(PANDA_JOINTS):
        T = T @ _rpy_xyz(xyz, rpy) @ _rz(q[i])
        frames.append(T)
    T8 = T @ _rpy_xyz(PANDA_LINK8_XYZ, (0.0, 0.0, 0.0))
    frames.append(T8)
    return frames


def panda_link_pos(q, mount, link):
    """Origin of panda_link{link} (1..8); panda_hand shares the origin of link8."""
    return panda_link_frames(q, mount)[link - 1][:3, 3]


def panda_sphere_pos(q, mount, link, offset):
    T = panda_link_frames(q, mount)[link - 1]
    return T[:3, :3] @ _t(offset) + T[:3, 3]


def point_pos(q):
    """pointRobot1.urdf: prismatic x (origin z 0.05), prismatic y, revolute theta."""
    return torch.stack([q[0], q[1], torch.as_tensor(0.05, dtype=DT)])


# --------------------------------------------------------------------------------------
# generic differential-geometry helpers, all by autograd
# --------------------------------------------------------------------------------------
def _jac(fun, x):
    x = x.detach().clone().requires_grad_(True)
    return torch.autograd.functional.jacobian(fun, x, create_graph=False).reshape(-1, x.numel())


def _jdotqdot(fun, x, xd):
    """JDOT_SIGN * d(J(x) xd)/dx * xd  -- the `Jdotqdot()` of a DifferentialMap."""
    def jv(xx):
        J = torch.autograd.functional.jacobian(fun, xx, create_graph=True).reshape(-1, xx.numel())
        return J @ xd
    H = torch.autograd.functional.jacobian(jv, x.detach().clone()).reshape(-1, x.numel())
    return JDOT_SIGN * (H @ xd)


def _metric_from_lagrangian(L_str, x, xd):
    """M = d2L/dxd2 at (x, xd) for a leaf Lagrangian string."""
    def L(v):
        return _eval_str(L_str, x=x, xdot=v).reshape(())
    return torch.autograd.functional.hessian(L, xd.detach().clone()).reshape(xd.numel(), xd.numel())


class Spec:
    """(M, f) with M xdd + f = 0, living on some variable of dimension n."""
    def __init__(self, M, f):
        self.M, self.f = M, f

    def pull(self, J, jdqd):
        return Spec(J.T @ self.M @ J, J.T @ (self.f + self.M @ jdqd))

    def dynamic_pull(self, xdd_ref):
        return Spec(self.M, self.f - self.M @ xdd_ref)

    def __add__(self, o):
        return Spec(self.M + o.M, self.f + o.f)


def _leaf_spec(h_str, L_str, x, xd):
    """WeightedGeometry(g=Geometry(h), le=Lagrangian(L)): M from L, f = M h."""
    M = _metric_from_lagrangian(L_str, x, xd)
    h = _eval_str(h_str, x=x, xdot=xd).reshape(-1)
    return Spec(M, M @ h)


def _pull_through(spec_fun, phi, q, qd):
    """Pull a leaf spec, defined on x = phi(q), back to q."""
    x = phi(q).reshape(-1)
    J = _jac(lambda v: phi(v).reshape(-1), q)
    xd = J @ qd
    c = _jdotqdot(lambda v: phi(v).reshape(-1), q, qd)
    return spec_fun(x, xd).pull(J, c)


# --------------------------------------------------------------------------------------
# the planner
# --------------------------------------------------------------------------------------
class Planner:
    """Mirror of ParameterizedFabricPlanner for the two robot models of the reference.

    model='panda'  : dof 7, collision links = link numbers (subset of 3..8 is active; 1,2 are
                     q-independent and skipped exactly as `fabrics` skips sparse FKs, which the
                     reference encodes as `if coll_i > 2`, forward_planner_Jointspace.py:39,164)
    model='point'  : dof 3, collision link base_link, no limits/planes.
    """

    def __init__(self, model="panda", mount=None, collision_links=(3, 4, 5, 6, 7, 8),
                 n_static=0, n_dynamic=0, dyn_dim=3, n_planes=1, limits=PANDA_LIMITS,
                 goals="panda", mode="vel", time_step=0.01, plane_abs=True, **strings):
        self.model = model
        self.dof = 7 if model == "panda" else 3
        self.mount = np.eye(4) if mount is None else np.asarray(mount, dtype=float)
        self.collision_links = [c for c in collision_links if (model != "panda" or c > 2)]
        self.n_static, self.n_dynamic, self.dyn_dim, self.n_planes = n_static, n_dynamic, dyn_dim, n_planes
        self.limits = limits if model == "panda" else None
        self.goals = goals
        self.mode, self.time_step = mode, time_step
        self.plane_abs = plane_abs
        self.s = dict(DEFAULTS)
        self.s.update(PANDA_OVERRIDES if model == "panda" else POINT_OVERRIDES)
        self.s.update(strings)

    # ---- kinematics -------------------------------------------------------------------
    def link_pos(self, q, link):
        if self.model == "panda":
            return panda_link_pos(q, self.mount, link)
        return point_pos(q)

    # ---- the solve --------------------------------------------------------------------
    def solve(self, q, qdot, x_goal_0=None, weight_goal_0=None, angle_goal_1=None, x_goal_1=None,
              weight_goal_1=None, x_goal_2=None, weight_goal_2=None, constraint_0=None,
              radius_body=None, x_obsts=(), radius_obsts=(), x_obsts_dynamic=(),
              xdot_obsts_dynamic=(), xddot_obsts_dynamic=(), radius_obsts_dynamic=(),
              return_parts=False):
        q = _t(q)
        qd = _t(qdot)
        n = self.dof
        s = self.s
        rb = radius_body if radius_body is not None else {}

        # base geometry: h = 0, M from base energy
        Mb = _metric_from_lagrangian(s["base_energy"], q, qd)
        geo = Spec(Mb, torch.zeros(n, dtype=DT))
        leaf_specs = {}

        for link in self.collision_links:
            r_body = float(rb[link]) if isinstance(rb, dict) else float(rb)
            fk = lambda v, link=link: self.link_pos(v, link)
            # static spherical obstacles: one-stage pull through phi(q)=|fk-x_o|/(r+r_b)-1
            for j in range(self.n_static):
                xo, ro = _t(x_obsts[j]), float(radius_obsts[j])
                phi = lambda v, fk=fk, xo=xo, ro=ro: (torch.sqrt(torch.sum((fk(v) - xo) ** 2)) / (ro + r_body) - 1).reshape(1)
                sp = _pull_through(lambda x, xd: _leaf_spec(s["collision_geometry"], s["collision_finsler"], x, xd), phi, q, qd)
                geo = geo + sp
                leaf_specs[("static", link, j)] = sp
            # dynamic spherical obstacles: 3-stage pull (geometry map, dynamic map, fk)
            for j in range(self.n_dynamic):
                d = self.dyn_dim
                xo, vo, ao = _t(x_obsts_dynamic[j])[:d], _t(xdot_obsts_dynamic[j])[:d], _t(xddot_obsts_dynamic[j])[:d]
                ro = float(radius_obsts_dynamic[j])
                fkd = lambda v, fk=fk, d=d: fk(v)[:d]
                p = fkd(q)
                Jp = _jac(fkd, q)
                pd_ = Jp @ qd
                cp = _jdotqdot(fkd, q, qd)
                x_rel, xd_rel = p - xo, pd_ - vo
                gmap = lambda xr, ro=ro: (torch.sqrt(torch.sum(xr ** 2)) / (ro + r_body) - 1).reshape(1)
                pwg1 = _pull_through(lambda x, xd: _leaf_spec(s["collision_geometry"], s["collision_finsler"], x, xd),
                                     gmap, x_rel, xd_rel)
                pwg2 = pwg1.dynamic_pull(ao)
                sp = pwg2.pull(Jp, cp)
                geo = geo + sp
                leaf_specs[("dynamic", link, j)] = sp
            # plane constraints
            for j in range(self.n_planes):
                con = _t(constraint_0)
                def phi(v, fk=fk, con=con):
                    val = torch.sum(con[:3] * fk(v)) + con[3]
                    if self.plane_abs:
                        val = torch.abs(val)
                    return (val / torch.sqrt(torch.sum(con[:3] ** 2)) - r_body).reshape(1)
                sp = _pull_through(lambda x, xd: _leaf_spec(s["geometry_plane_constraint"], s["finsler_plane_constraint"], x, xd), phi, q, qd)
                geo = geo + sp
                leaf_specs[("plane", link, j)] = sp

        if self.limits is not None:
            for j, (lo, hi) in enumerate(self.limits):
                for k, phi in enumerate((lambda v, j=j, lo=lo: (v[j] - lo).reshape(1),
                                         lambda v, j=j, hi=hi: (hi - v[j]).reshape(1))):
                    sp = _pull_through(lambda x, xd: _leaf_spec(s["limit_geometry"], s["limit_finsler"], x, xd), phi, q, qd)
                    geo = geo + sp
                    leaf_specs[("limit", j, k)] = sp

        # attractors
        forced = Spec(geo.M.clone(), geo.f.clone())
        goal_maps = self._goal_maps(x_goal_0, angle_goal_1, x_goal_1, x_goal_2)
        weights = [weight_goal_0, weight_goal_1, weight_goal_2]
        phi_prime = None
        for gi, phi in enumerate(goal_maps):
            w = float(weights[gi])
            def spec_fun(x, xd, w=w):
                xx = x.detach().clone().requires_grad_(True)
                psi = w * _eval_str(s["attractor_potential"], x=xx, xdot=xd)
                h = torch.autograd.grad(psi, xx)[0]
                A = _eval_str(s["attractor_metric"], x=x, xdot=xd)
                L = lambda v: torch.sum(v * (A @ v))
                M = torch.autograd.functional.hessian(L, xd.detach().clone())
                return Spec(M, M @ h)
            sp = _pull_through(spec_fun, phi, q, qd)
            forced = forced + sp
            leaf_specs[("attractor", gi, 0)] = sp
            if gi == 0:
                phi_prime = phi

        eye = torch.eye(n, dtype=DT)
        h_g = torch.linalg.solve(geo.M + EPS * eye, geo.f)
        if not goal_maps:
            alpha_g = -torch.sum(qd * h_g) / (EPS + torch.sum(qd * qd))
            qdd = -h_g - alpha_g * qd
        else:
            h_f = torch.linalg.solve(forced.M + EPS * eye, forced.f)
            # execution energy 0.5 qd.qd : M_ex = I, f_ex = 0
            alpha_g = -torch.sum(qd * h_g) / (EPS + torch.sum(qd * qd))
            alpha_f = -torch.sum(qd * h_f) / (EPS + torch.sum(qd * qd))
            eta = _eval_str(s["damper_eta"], xdot=qd, x=q)
            a_ex = eta * alpha_g + (1 - eta) * alpha_f
            x_psi = phi_prime(q).reshape(-1)
            beta = _eval_str(s["damper_beta"], x=x_psi, a_ex=-a_ex, a_le=-alpha_g)
            qdd = -h_f - (a_ex + beta) * qd
        action = qdd if self.mode == "acc" else qd + self.time_step * qdd
        if torch.linalg.norm(action) < EPS:
            action = torch.zeros_like(action)
        if return_parts:
            return dict(action=action.numpy(), qddot=qdd.numpy(), M_g=geo.M.numpy(), f_g=geo.f.numpy(),
                        M_f=forced.M.numpy(), f_f=forced.f.numpy(),
                        leaves={k: (v.M.numpy(), v.f.numpy()) for k, v in leaf_specs.items()})
        return action.numpy()

    def _goal_maps(self, x_goal_0, angle_goal_1, x_goal_1, x_goal_2):
        if self.goals is None:
            return []
        if self.goals == "panda":
            g0, g1, g2 = _t(x_goal_0), _t(x_goal_1), _t(x_goal_2).reshape(-1)
            R = _t(angle_goal_1).reshape(3, 3)
            return [
                lambda v: self.link_pos(v, 8) - g0,                      # world -> panda_hand (EXJ:32-41)
                lambda v: R @ self.link_pos(v, 8) - R @ self.link_pos(v, 7) - g1,   # link7 -> hand (EXJ:42-52)
                lambda v: v[6:7] - g2,                                   # joint index 6 (EXJ:53-60)
            ]
        if self.goals == "point":
            g0 = _t(x_goal_0)
            return [lambda v: point_pos(v)[:2] - g0]                     # indices [0,1] (pointmass :61-71)
        raise ValueError(self.goals)


# --------------------------------------------------------------------------------------
# kinematics helper of the reference: fk, J, "jac_dot" (= -d(J qd)/dq), utils.py:16-54
# --------------------------------------------------------------------------------------
def panda_link_kinematics(q, qdot, mount, link, offset=(0.0, 0.0, 0.0)):
    """x, v = J qd, a = jac_dot(q,qd) qd = -Jdot qd  (forward_planner_Jointspace.py:82-100, qddot=0)."""
    q, qd = _t(q), _t(qdot)
    fk = lambda v: panda_sphere_pos(v, mount, link, offset)
    J = _jac(fk, q)
    a = _jdotqdot(fk, q, qd)      # JDOT_SIGN already applied -> equals jac_dot_fun @ qd
    return fk(q).numpy(), (J @ qd).numpy(), a.numpy(), J.numpy()


def rollout_jointspace(planners, q0, qdot0, params, H, dt=0.01, dynamic=True, sphere_links=range(1, 9),
                       sphere_radius=0.08):
    """Coupled rollout of forward_planner_Jointspace.py:190-249 ('vel' mode).

    planners: one Planner per robot built with n_dynamic = len(sphere_links)*(N-1).
    params  : list of per-robot kwargs dicts (goals, weights, constraint, radius_body).
    Returns (q_traj[N][H][7], qdot_traj[N][H][7], avg_vel[N]).
    """
    N = len(planners)
    q = [np.array(v, dtype=float) for v in q0]
    qd = [np.array(v, dtype=float) for v in qdot0]
    qs = [[] for _ in range(N)]
    qds = [[] for _ in range(N)]
    for _ in range(H):
        spheres = []
        for i in range(N):
            q[i] = q[i] + dt * qd[i]                    # system_step 'vel' (FPJ:72-80)
            sx, sv, sa = [], [], []
            for link in sphere_links:
                x, v, a, _ = panda_link_kinematics(q[i], qd[i], planners[i].mount, link)
                sx.append(x); sv.append(v); sa.append(a)
            spheres.append((sx, sv, sa))
        new_qd = []
        for i in range(N):
            ox, ov, oa, orad = [], [], [], []
            for j in range(N):
                if j == i:
                    continue
                ox += spheres[j][0]
                if dynamic:
                    ov += spheres[j][1]; oa += spheres[j][2]
                else:
                    ov += [np.zeros(3)] * len(spheres[j][1]); oa += [np.zeros(3)] * len(spheres[j][2])
                orad += [sphere_radius] * len(spheres[j][0])
            new_qd.append(planners[i].solve(q[i], qd[i], x_obsts_dynamic=ox, xdot_obsts_dynamic=ov,
                                            xddot_obsts_dynamic=oa, radius_obsts_dynamic=orad, **params[i]))
        for i in range(N):
            qd[i] = new_qd[i]
            qs[i].append(q[i].copy())
            qds[i].append(qd[i].copy())
    avg = [sum(float(np.sum(v ** 2)) for v in qds[i]) / (H * 7) for i in range(N)]   # FPJ:102-116
    return np.array(qs), np.array(qds), np.array(avg)


def rollout_cartesian(planner, q0, qdot0, params, x_obsts_dyn, v_obsts_dyn, a_obsts_dyn, radius_obsts_dyn, H, dt=0.01):
    """Per-robot rollout of forward_planner_Cartesian.py:421-458 ('vel' mode): action at the current state against the
    obstacles at their current positions, then q += dt * action, then every obstacle advances with its constant
    velocity (x_o += dt * v_o; the accelerations a_o are passed through unchanged, FPC:422-427).
    Returns (q_traj[H][7], qdot_traj[H][7], avg_vel)."""
    q, qd = np.array(q0, dtype=float), np.array(qdot0, dtype=float)
    ox = [np.array(x, dtype=float) for x in x_obsts_dyn]
    ov = [np.array(v, dtype=float) for v in v_obsts_dyn]
    oa = [np.array(a, dtype=float) for a in a_obsts_dyn]
    qs, qds = [], []
    for _ in range(H):
        qd = planner.solve(q, qd, x_obsts_dynamic=ox, xdot_obsts_dynamic=ov, xddot_obsts_dynamic=oa,
                           radius_obsts_dynamic=list(radius_obsts_dyn), **params)          # FPC:430 (action = velocity)
        q = q + dt * qd                                                                   # FPC:440-446
        ox = [x + dt * v for x, v in zip(ox, ov)]                                         # FPC:448-453
        qs.append(q.copy())
        qds.append(qd.copy())
    avg = sum(float(np.sum(v ** 2)) for v in qds) / (H * 7)                               # FPC:276-288
    return np.array(qs), np.array(qds), avg


def hand_estimate(q, qdot, mount, T=0.2):
    """RF-CV goal estimate x_ee + 20 * 0.01 * v_ee with v_ee = J qdot (example_pandas_cartesian.py:355-357)."""
    x, v, _, _ = panda_link_kinematics(q, qdot, mount, 8)
    return x + T * v
