"""Generates tests/golden/*.npz from oracle/autodiff_oracle.py (torch float64 autograd, from the definitions).

The reference's arithmetic lives in un-vendored third-party wheels that cannot be imported here
(SURVEY 8c), so these vectors are NOT outputs of the reference: they pin the C++ oracle and the HIP kernels
to an independent derivation of the same published algorithm.  Run from the repo root:
    python tests/golden/make_golden.py
Inputs are seeded; outputs are data only (inputs + expected values).
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import autodiff_oracle as ao  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
R1 = np.array([[0.0, 0.0, -1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0]])
POS0 = np.array([1.125, 0.19, 0.12, -1.66, -0.0, 1.88, math.pi / 4])
MOUNTS = [ao._t(np.eye(4)).numpy() for _ in range(3)]
MOUNTS[0][:3, 3] = [0.0, 0.0, 0.65]
MOUNTS[1][:3, 3] = [1.0, 0.0, 0.65]
MOUNTS[1][:2, :2] = [[-1.0, 0.0], [0.0, -1.0]]
MOUNTS[2][:3, 3] = [0.7, 0.6, 0.65]
MOUNTS[2][:2, :2] = [[math.cos(2.0), -math.sin(2.0)], [math.sin(2.0), math.cos(2.0)]]   # a non-trivial yaw


def goal_kwargs(g0, w0=2.0, con=(0.0, 0.0, 1.0, -0.65), rb=0.08):
    return dict(x_goal_0=g0, weight_goal_0=w0, angle_goal_1=R1, x_goal_1=[0.107, 0.0, 0.0], weight_goal_1=20.0,
                x_goal_2=[math.pi / 4], weight_goal_2=1.0, constraint_0=list(con), radius_body=rb)


def panda_actions():
    """compute_action cases: dynamic / static / grasp / near-barrier / at-rest / per-link body radii."""
    rng = np.random.default_rng(7)
    cases = []
    kinds = ["dynamic", "dynamic", "dynamic", "static", "static", "grasp", "near", "rest", "radii", "nogoal"]
    for ci, kind in enumerate(kinds):
        robot = ci % 3
        mount = MOUNTS[robot]
        q = POS0 + rng.uniform(-0.3, 0.3, 7)
        qd = rng.uniform(-0.5, 0.5, 7)
        if kind == "rest":
            qd[:] = 0.0
            q[6] = math.pi / 4            # |x| = 0 on attractor 2
        M = 4
        links = rng.choice([3, 4, 5, 7, 8], M)
        gap = rng.uniform(0.22, 0.4, M) if kind != "near" else rng.uniform(0.165, 0.17, M)   # x ~ 0.03..0.06
        dirs = rng.normal(size=(M, 3))
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        dirs[:, 2] = np.abs(dirs[:, 2])
        ox = np.array([ao.panda_link_pos(ao._t(q), mount, int(l)).numpy() for l in links]) + gap[:, None] * dirs
        ov = rng.uniform(-0.3, 0.3, (M, 3))
        oa = rng.uniform(-0.5, 0.5, (M, 3))
        orad = rng.uniform(0.06, 0.1, M) if kind == "radii" else np.full(M, 0.08)
        rb = {l: float(r) for l, r in zip(range(3, 9), rng.uniform(0.05, 0.09, 6))} if kind == "radii" else 0.08
        g0 = np.array([0.2, 0.6, 1.15]) + rng.uniform(-0.1, 0.1, 3)
        kw = goal_kwargs(g0, rb=rb)
        if kind == "static":
            P = ao.Planner(mount=mount, n_static=M)
            out = P.solve(q, qd, x_obsts=ox, radius_obsts=orad, return_parts=True, **kw)
            ov[:] = 0.0
            oa[:] = 0.0
        elif kind == "grasp":
            P = ao.Planner(mount=mount, collision_links=())
            out = P.solve(q, qd, return_parts=True, **kw)
        elif kind == "nogoal":
            P = ao.Planner(mount=mount, n_dynamic=M, goals=None)
            out = P.solve(q, qd, x_obsts_dynamic=ox, xdot_obsts_dynamic=ov, xddot_obsts_dynamic=oa,
                          radius_obsts_dynamic=orad, return_parts=True, **kw)
        else:
            P = ao.Planner(mount=mount, n_dynamic=M)
            out = P.solve(q, qd, x_obsts_dynamic=ox, xdot_obsts_dynamic=ov, xddot_obsts_dynamic=oa,
                          radius_obsts_dynamic=orad, return_parts=True, **kw)
        rbv = np.array([rb[l] for l in range(3, 9)]) if isinstance(rb, dict) else np.full(6, rb)
        cases.append(dict(kind=kind, robot=robot, mount=mount, q=q, qd=qd, g0=g0, rb=rbv, ox=ox, ov=ov, oa=oa, orad=orad,
                          action=out["action"], qddot=out["qddot"], M_g=out["M_g"], f_g=out["f_g"], M_f=out["M_f"],
                          f_f=out["f_f"]))
        print("panda action", ci, kind, out["action"])
    keys = [k for k in cases[0] if k != "kind"]
    np.savez(os.path.join(OUT, "panda_actions.npz"), kinds=np.array([c["kind"] for c in cases]),
             **{k: np.stack([c[k] for c in cases]) for k in keys})


def planar_actions():
    rng = np.random.default_rng(9)
    cases = []
    for ci in range(8):
        dyn = 3 <= ci < 6
        mixed = ci >= 6       # example_pointmasses_dynamic.py: static scene spheres (3-D) + other robots as 2-D dynamic spheres
        q = np.array([rng.uniform(-2.5, 2.5), rng.uniform(-2.5, 3.7), rng.uniform(-1, 1)])
        qd = rng.uniform(-0.5, 0.5, 3)
        g0 = rng.uniform(-2.5, 2.5, 2)
        M = 4
        ang = rng.uniform(0, 2 * math.pi, M)
        dist = rng.uniform(1.4, 3.0, M)
        ox = np.stack([q[0] + dist * np.cos(ang), q[1] + dist * np.sin(ang), np.zeros(M)], axis=1)
        orad = np.array([1.0, 1.0, 0.2, 0.2])
        ov = rng.uniform(-0.3, 0.3, (M, 3)) if dyn else np.zeros((M, 3))
        oa = rng.uniform(-0.3, 0.3, (M, 3)) if dyn else np.zeros((M, 3))
        kw = dict(x_goal_0=g0, weight_goal_0=1.0, radius_body=0.2)
        if mixed:
            ov = rng.uniform(-0.3, 0.3, (M, 3))
            oa = rng.uniform(-0.3, 0.3, (M, 3))
            ov[:2] = 0.0
            oa[:2] = 0.0
            P = ao.Planner(model="point", collision_links=(1,), n_static=2, n_dynamic=2, dyn_dim=2, n_planes=0,
                           goals="point", mode="acc")
            out = P.solve(q, qd, x_obsts=ox[:2], radius_obsts=orad[:2], x_obsts_dynamic=ox[2:],
                          xdot_obsts_dynamic=ov[2:], xddot_obsts_dynamic=oa[2:], radius_obsts_dynamic=orad[2:],
                          return_parts=True, **kw)
        elif dyn:   # dynamic_obstacle_dimension = 2
            P = ao.Planner(model="point", collision_links=(1,), n_dynamic=M, dyn_dim=2, n_planes=0, goals="point", mode="acc")
            out = P.solve(q, qd, x_obsts_dynamic=ox, xdot_obsts_dynamic=ov, xddot_obsts_dynamic=oa,
                          radius_obsts_dynamic=orad, return_parts=True, **kw)
        else:     # example_pointmasses_static.py: static spheres, 3-D distance to (x, y, 0.05)
            P = ao.Planner(model="point", collision_links=(1,), n_static=M, n_planes=0, goals="point", mode="acc")
            out = P.solve(q, qd, x_obsts=ox, radius_obsts=orad, return_parts=True, **kw)
        cases.append(dict(dyn=int(dyn or mixed), n_static=2 if mixed else 0, q=q, qd=qd, g0=g0, ox=ox, ov=ov, oa=oa, orad=orad, action=out["action"]))
        print("planar action", ci, out["action"])
    np.savez(os.path.join(OUT, "planar_actions.npz"), **{k: np.stack([c[k] for c in cases]) for k in cases[0]})


def kinematics():
    """fk, v = J qd, a = jac_dot qd (= -Jdot qd), J for link origins and offset spheres (utils.py:16-54,87-119)."""
    rng = np.random.default_rng(11)
    rec = dict(robot=[], link=[], offset=[], q=[], qd=[], x=[], v=[], a=[], J=[])
    for ci in range(24):
        robot = ci % 3
        link = 1 + ci % 8
        off = np.zeros(3) if ci < 12 else rng.uniform(-0.1, 0.1, 3)
        q = rng.uniform(-2.0, 2.0, 7)
        qd = rng.uniform(-1.0, 1.0, 7)
        x, v, a, J = ao.panda_link_kinematics(q, qd, MOUNTS[robot], link, off)
        for k, val in zip(("robot", "link", "offset", "q", "qd", "x", "v", "a", "J"), (robot, link, off, q, qd, x, v, a, J)):
            rec[k].append(val)
    np.savez(os.path.join(OUT, "panda_kinematics.npz"), mounts=np.stack(MOUNTS), **{k: np.array(v) for k, v in rec.items()})
    print("kinematics: 24 cases")


def rollout():
    """2-Panda coupled rollout, H=3, dynamic and static fabrics (forward_planner_Jointspace.py:190-249)."""
    rng = np.random.default_rng(13)
    out = {}
    for name, dynamic in (("dyn", True), ("stat", False)):
        planners = [ao.Planner(mount=MOUNTS[i], n_dynamic=8) for i in range(2)]
        q0 = [POS0 + rng.uniform(-0.2, 0.2, 7) for _ in range(2)]
        qd0 = [rng.uniform(-0.5, 0.5, 7) for _ in range(2)]
        g0 = [np.array([0.2, 0.6, 1.15]), np.array([0.8, -0.6, 1.15])]
        params = [goal_kwargs(g0[i]) for i in range(2)]
        qs, qds, avg = ao.rollout_jointspace(planners, q0, qd0, params, H=3, dynamic=dynamic)
        out.update({f"{name}_q0": np.array(q0), f"{name}_qd0": np.array(qd0), f"{name}_g0": np.array(g0),
                    f"{name}_q": qs, f"{name}_qd": qds, f"{name}_avg": avg})
        print("rollout", name, avg)
    np.savez(os.path.join(OUT, "panda_rollout.npz"), mounts=np.stack(MOUNTS[:2]), **out)


REF_MOUNTS = [np.eye(4) for _ in range(3)]          # parameters_manipulators.py:101-105,138-150 (N = 3)
for _i, (_p, _yaw) in enumerate((([0.0, 0.0, 0.65], 0.0), ([1.0, 0.0, 0.65], math.pi), ([0.7, 0.6, 0.65], math.pi))):
    REF_MOUNTS[_i][:2, :2] = [[math.cos(_yaw), -math.sin(_yaw)], [math.sin(_yaw), math.cos(_yaw)]]
    REF_MOUNTS[_i][:3, 3] = _p
REF_POS0_3 = np.array([[1.13793529, -0.3227085, -0.02767777, -2.2204281, -0.00917029, 1.88612235, 0.78536134],
                       [1.131, 0.20, 0.12, -1.65, -0.0, 1.86, math.pi / 4],
                       [-0.46609715, -0.25025564, -0.40425878, -2.0966941, -0.10682593, 1.84917516, 0.37170524]])  # PM:111-115
REF_GOALS_3 = np.array([[0.25, 0.6, 1.15], [0.8, -0.5, 1.15], [0.4, 0.5, 0.95]])                                  # PM:130-132


def rollout_c4(H=30):
    """BASELINE config 4: 3-Panda RF-CV H=30 coupled joint-space rollout on the reference's own N = 3 cell (mounts, start
    configurations and start goals of parameters_manipulators); robot 1 rolls out towards the goal ESTIMATED from its
    hand, x_ee + 0.2 v_ee (EXJ:346-348 with the Cartesian driver's proper velocity, EXC:355-357).  Average velocities
    and the last state only."""
    rng = np.random.default_rng(17)
    planners = [ao.Planner(mount=REF_MOUNTS[i], n_dynamic=16) for i in range(3)]
    q0 = [REF_POS0_3[i] + rng.uniform(-0.05, 0.05, 7) for i in range(3)]
    qd0 = [rng.uniform(-0.3, 0.3, 7) for _ in range(3)]
    g0 = [REF_GOALS_3[i].copy() for i in range(3)]
    g_est = ao.hand_estimate(q0[1], qd0[1], REF_MOUNTS[1])
    params = [goal_kwargs(g_est if i == 1 else g0[i]) for i in range(3)]
    qs, qds, avg = ao.rollout_jointspace(planners, q0, qd0, params, H=H, dynamic=True)
    np.savez(os.path.join(OUT, "panda_rollout_c4.npz"), mounts=np.stack(REF_MOUNTS), q0=np.array(q0), qd0=np.array(qd0),
             g0=np.array(g0), estimated_goal_1=g_est, estimate_mask=np.array(0b010), horizon=np.array(H),
             avg=avg, q_last=qs[:, -1], qd_last=qds[:, -1])
    print("rollout C4", avg)


def rollout_cartesian(H=5):
    """FabricsRollouts (forward_planner_Cartesian.py:347-563) for robot 0 and robot 1 of the reference's 2-robot cell:
    the other robot's 8 link-origin spheres as constant-velocity obstacles (n_obst_per_link = 1, EXC:184), zero
    obstacle accelerations (FPC:33), robot 1 with the RF-CV estimated goal."""
    rng = np.random.default_rng(19)
    mounts = REF_MOUNTS[:2]
    q0 = [POS0 + rng.uniform(-0.15, 0.15, 7) for _ in range(2)]
    qd0 = [rng.uniform(-0.4, 0.4, 7) for _ in range(2)]
    g0 = [np.array([0.2, 0.6, 1.15]), np.array([0.8, -0.6, 1.15])]
    out = dict(mounts=np.stack(mounts), q0=np.array(q0), qd0=np.array(qd0), g0=np.array(g0), horizon=np.array(H))
    for i in range(2):
        j = 1 - i
        kin = [ao.panda_link_kinematics(q0[j], qd0[j], mounts[j], link) for link in range(1, 9)]
        ox, ov = np.array([k[0] for k in kin]), np.array([k[1] for k in kin])
        goal = ao.hand_estimate(q0[i], qd0[i], mounts[i]) if i == 1 else g0[i]
        P = ao.Planner(mount=mounts[i], n_dynamic=8)
        qs, qds, avg = ao.rollout_cartesian(P, q0[i], qd0[i], goal_kwargs(goal), ox, ov, np.zeros((8, 3)), [0.08] * 8, H)
        out.update({f"r{i}_ox": ox, f"r{i}_ov": ov, f"r{i}_goal": goal, f"r{i}_q": qs, f"r{i}_qd": qds, f"r{i}_avg": np.array(avg)})
        print("cartesian rollout robot", i, avg)
    np.savez(os.path.join(OUT, "panda_cartesian.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:                  # e.g. `make_golden.py rollout_c4 rollout_cartesian`: only the named fixtures
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    rollout_c4()
    rollout_cartesian()
    kinematics()
    planar_actions()
    panda_actions()
    rollout()
