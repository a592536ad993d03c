"""CPU: the C++ float64 oracle against the committed golden vectors (autodiff-from-definitions) and KATs.

The reference holds no numeric test vector for this path (examples/test_examples.py:8-36 assert only the return
type); these fixtures come from oracle/autodiff_oracle.py via tests/golden/make_golden.py -- parity with the
CasADi path itself stays unpinned.
"""
import math
import os

import numpy as np
import pytest

from multi_robot_fabrics_amd import abi, config

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
R1 = np.array([[0.0, 0.0, -1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0]])


def params_row(g0, rb, w0=2.0):
    p = np.zeros(abi.NPARAM)
    p[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + len(g0)] = g0
    p[abi.P_WEIGHT_GOAL_0] = w0
    p[abi.P_ANGLE_GOAL_1:abi.P_ANGLE_GOAL_1 + 9] = R1.ravel()
    p[abi.P_X_GOAL_1:abi.P_X_GOAL_1 + 3] = [0.107, 0.0, 0.0]
    p[abi.P_WEIGHT_GOAL_1] = 20.0
    p[abi.P_X_GOAL_2] = math.pi / 4
    p[abi.P_WEIGHT_GOAL_2] = 1.0
    p[abi.P_CONSTRAINT_0:abi.P_CONSTRAINT_0 + 4] = [0.0, 0.0, 1.0, -0.65]
    p[abi.P_RADIUS_BODY:abi.P_RADIUS_BODY + len(rb)] = rb
    return p


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def test_panda_actions_match_autodiff(oracle):
    g = np.load(os.path.join(GOLD, "panda_actions.npz"))
    for i, kind in enumerate(g["kinds"]):
        cfg = config.panda_config(n_robots=1, horizon=1, mounts=[g["mount"][i]],
                                  n_ego=0 if kind == "grasp" else 6)
        if kind == "nogoal":
            cfg.n_goals = 0
        prm = params_row(g["g0"][i], g["rb"][i])
        args = (g["ox"][i], g["ov"][i], g["oa"][i], g["orad"][i])
        Mg, fg, Mf, ff = oracle.specs(cfg, 0, g["q"][i], g["qd"][i], prm, *args)
        assert rel(Mg, g["M_g"][i]) < 1e-11, kind
        assert rel(fg, g["f_g"][i]) < 1e-10, kind
        if kind != "nogoal":
            assert rel(Mf, g["M_f"][i]) < 1e-11, kind
            assert rel(ff, g["f_f"][i]) < 1e-10, kind
        qdd, act = oracle.compute_action(cfg, g["q"][i][:, None], g["qd"][i][:, None], prm[:, None],
                                         g["ox"][i][:, :, None], g["ov"][i][:, :, None], g["oa"][i][:, :, None],
                                         g["orad"][i][:, None])
        # the solve amplifies round-off by cond(M): near the barrier the metric reaches 1e6 x the base mass
        tol = 1e-7 if kind in ("near", "nogoal") else 1e-10
        assert rel(qdd[:, 0], g["qddot"][i]) < tol, kind
        assert rel(act[:, 0], g["action"][i]) < tol, kind


def test_planar_actions_match_autodiff(oracle):
    g = np.load(os.path.join(GOLD, "planar_actions.npz"))
    for i in range(len(g["dyn"])):
        cfg = config.planar3_config(n_robots=1, obst_dim=2 if g["dyn"][i] else 3)
        prm = np.zeros(abi.NPARAM)
        prm[0:2] = g["g0"][i]
        prm[abi.P_WEIGHT_GOAL_0] = 1.0
        prm[abi.P_RADIUS_BODY] = 0.2
        _, act = oracle.compute_action(cfg, g["q"][i][:, None], g["qd"][i][:, None], prm[:, None],
                                       g["ox"][i][:, :, None], g["ov"][i][:, :, None], g["oa"][i][:, :, None],
                                       g["orad"][i][:, None], n_static=int(g["n_static"][i]))
        assert rel(act[:, 0], g["action"][i]) < 1e-10


def test_kinematics_match_autodiff(oracle):
    g = np.load(os.path.join(GOLD, "panda_kinematics.npz"))
    for i in range(len(g["link"])):
        cfg = config.panda_config(n_robots=3, horizon=1, mounts=list(g["mounts"]))
        config.set_spheres(cfg, [int(g["link"][i])], [g["offset"][i]])
        rows = 3
        q = np.tile(g["q"][i][:, None], (1, rows))
        qd = np.tile(g["qd"][i][:, None], (1, rows))
        x, v, a = oracle.fk_spheres(cfg, q, qd)
        r = int(g["robot"][i])
        assert np.abs(x[0, :, r] - g["x"][i]).max() < 1e-13
        assert np.abs(v[0, :, r] - g["v"][i]).max() < 1e-12
        assert np.abs(a[0, :, r] - g["a"][i]).max() < 1e-11


def test_fk_known_pose(oracle):
    """link origins at pos0 (PM:93), base frame, 4-decimal KAT."""
    cfg = config.panda_config(n_robots=1, horizon=1, mounts=[np.identity(4)])
    q = np.array([1.125, 0.19, 0.12, -1.66, 0.0, 1.88, math.pi / 4])[:, None]
    x, _, _ = oracle.fk_spheres(cfg, q, np.zeros_like(q))
    want = {1: (0, 0, 0.333), 2: (0, 0, 0.333), 3: (0.0257, 0.0538, 0.6433), 4: (0.0515, 0.1307, 0.6278),
            5: (0.1772, 0.5019, 0.6019), 6: (0.1772, 0.5019, 0.6019), 7: (0.2056, 0.5851, 0.6046),
            8: (0.2044, 0.5891, 0.4977)}
    for link, p in want.items():
        assert np.abs(x[link - 1, :, 0] - np.array(p)).max() < 6e-5


def test_rollout_matches_autodiff(oracle):
    g = np.load(os.path.join(GOLD, "panda_rollout.npz"))
    for name, dynamic in (("dyn", 1), ("stat", 0)):
        cfg = config.panda_config(n_robots=2, horizon=3, dynamic=dynamic, mounts=list(g["mounts"]))
        q0, qd0 = g[f"{name}_q0"].T.copy(), g[f"{name}_qd0"].T.copy()       # [7, rows], rows = 1 scenario x 2 robots
        prm = np.stack([params_row(g[f"{name}_g0"][i], [0.08] * 6) for i in range(2)], axis=1)
        avg, tq, tqd = oracle.rollout(cfg, q0, qd0, prm, traj=True)
        want_q = g[f"{name}_q"].transpose(1, 2, 0)       # [N,H,7] -> [H,7,N]
        want_qd = g[f"{name}_qd"].transpose(1, 2, 0)
        assert rel(tq, want_q) < 1e-11
        assert rel(tqd, want_qd) < 1e-10
        assert rel(avg, g[f"{name}_avg"]) < 1e-10


# ------------------------------------------------------------------------------- analytic properties
def _random_case(seed, n_robots=2):
    from multi_robot_fabrics_amd import scenarios
    cfg = config.panda_config(n_robots=n_robots, horizon=4)
    return cfg, scenarios.panda_batch(cfg, 6, seed=seed)


def test_jacobian_and_jdot_by_finite_differences(oracle):
    """v = J qd and a = -d(J qd)/dq qd against central differences of the oracle's own fk."""
    cfg, b = _random_case(3)
    q, qd = b["q"], b["qdot"]
    x, v, a = oracle.fk_spheres(cfg, q, qd)
    h = 1e-6
    xp, _, _ = oracle.fk_spheres(cfg, q + h * qd, qd)
    xm, _, _ = oracle.fk_spheres(cfg, q - h * qd, qd)
    assert np.abs((xp - xm) / (2 * h) - v).max() < 1e-8
    _, vp, _ = oracle.fk_spheres(cfg, q + h * qd, qd)
    _, vm, _ = oracle.fk_spheres(cfg, q - h * qd, qd)
    assert np.abs(cfg.jdot_sign * (vp - vm) / (2 * h) - a).max() < 1e-7


def test_metric_is_symmetric_positive_definite(oracle):
    cfg, b = _random_case(4)
    sx, sv, sa = oracle.fk_spheres(cfg, b["q"], b["qdot"])
    for r in range(4):
        other = 1 - r % 2
        s = (r // 2) * 2 + other
        Mg, fg, Mf, ff = oracle.specs(cfg, r % 2, b["q"][:, r], b["qdot"][:, r], b["params"][:, r], sx[:, :, s],
                                      sv[:, :, s], sa[:, :, s], np.full(8, 0.08))
        for M in (Mg, Mf):
            assert np.abs(M - M.T).max() < 1e-12 * np.abs(M).max()
            assert np.linalg.eigvalsh(M).min() > 0.19      # base mass 0.2 plus PSD leaf metrics


def test_geometry_conserves_execution_energy(oracle):
    """Without goals the energized geometry  qdd = -h - alpha qd  leaves 0.5 |qd|^2 unchanged: qd . qdd = 0."""
    cfg, b = _random_case(5)
    cfg.n_goals = 0
    cfg.mode = abi.MODE_ACC
    cfg.zero_small_action = 0
    sx, sv, sa = oracle.fk_spheres(cfg, b["q"], b["qdot"])
    from multi_robot_fabrics_amd import scenarios
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, b, sx, sv, sa)
    qdd, _ = oracle.compute_action(cfg, b["q"], b["qdot"], b["params"], ox, ov, oa, orad)
    dots = (qdd * b["qdot"]).sum(0)
    scale = np.linalg.norm(qdd, axis=0) * np.linalg.norm(b["qdot"], axis=0)
    assert np.abs(dots / scale).max() < 1e-4      # qd.qdd = -(qd.h) eps/(eps + |qd|^2): zero up to the eps regulariser


def test_static_equals_dynamic_with_zero_motion_and_permutation_invariance(oracle):
    cfg, b = _random_case(6)
    from multi_robot_fabrics_amd import scenarios
    sx, sv, sa = oracle.fk_spheres(cfg, b["q"], b["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, b, sx, sv, sa)
    _, a_none = oracle.compute_action(cfg, b["q"], b["qdot"], b["params"], ox, None, None, orad)
    _, a_zero = oracle.compute_action(cfg, b["q"], b["qdot"], b["params"], ox, 0 * ov, 0 * oa, orad)
    assert np.array_equal(a_none, a_zero)
    perm = np.random.default_rng(0).permutation(ox.shape[0])
    _, a0 = oracle.compute_action(cfg, b["q"], b["qdot"], b["params"], ox, ov, oa, orad)
    _, a1 = oracle.compute_action(cfg, b["q"], b["qdot"], b["params"], ox[perm], ov[perm], oa[perm], orad[perm])
    assert rel(a1, a0) < 1e-12


def test_h1_rollout_is_euler_step_plus_compute_action(oracle):
    from multi_robot_fabrics_amd import scenarios
    cfg = config.panda_config(n_robots=3, horizon=1)
    b = scenarios.panda_batch(cfg, 5, seed=7)
    avg, tq, tqd = oracle.rollout(cfg, b["q"], b["qdot"], b["params"], traj=True)
    q1 = b["q"] + cfg.dt * b["qdot"]
    sx, sv, sa = oracle.fk_spheres(cfg, q1, b["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, b, sx, sv, sa)
    _, act = oracle.compute_action(cfg, q1, b["qdot"], b["params"], ox, ov, oa, orad)
    assert np.array_equal(tq[0], q1)
    assert rel(tqd[0], act) < 1e-13
    assert rel(avg, (act ** 2).sum(0) / 7) < 1e-13


def test_cartesian_equals_jointspace_when_others_are_frozen(oracle):
    """Cartesian rollout with zero obstacle velocity == coupled rollout seen by a robot whose neighbours do not move:
    checked at H=1, where the only difference left is action-then-step vs step-then-action."""
    from multi_robot_fabrics_amd import scenarios
    cfg = config.panda_config(n_robots=2, horizon=1)
    b = scenarios.panda_batch(cfg, 4, seed=8)
    sx, sv, sa = oracle.fk_spheres(cfg, b["q"], b["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, b, sx, sv, sa)
    avg, tq, tqd = oracle.rollout_cartesian(cfg, b["q"], b["qdot"], b["params"], ox, ov, 0 * oa, orad, traj=True)
    _, act = oracle.compute_action(cfg, b["q"], b["qdot"], b["params"], ox, ov, 0 * oa, orad)
    assert rel(tqd[0], act) < 1e-13
    assert rel(tq[0], b["q"] + cfg.dt * act) < 1e-13


def test_goal_estimate_mask(oracle):
    """RF-CV: masked robots roll out towards x_ee + 0.2 v_ee instead of their communicated goal (EXC:355-357)."""
    from multi_robot_fabrics_amd import scenarios
    cfg = config.panda_config(n_robots=2, horizon=2)
    b = scenarios.panda_batch(cfg, 3, seed=9)
    cfg.goal_estimate_mask = 0b10
    avg_m, _, tqd_m = oracle.rollout(cfg, b["q"], b["qdot"], b["params"], traj=True)
    x, v, _ = oracle.fk_spheres(cfg, b["q"], b["qdot"])
    p2 = b["params"].copy()
    est = x[7] + 0.2 * v[7]                       # sphere 7 = panda_link8 = panda_hand origin
    p2[0:3, 1::2] = est[:, 1::2]
    cfg.goal_estimate_mask = 0
    avg_e, _, tqd_e = oracle.rollout(cfg, b["q"], b["qdot"], p2, traj=True)
    assert rel(tqd_m, tqd_e) < 1e-13


@pytest.mark.parametrize("links", [(7,), (5,), (3, 6, 8)])
def test_collision_link_subsets_match_autodiff(oracle, links):
    """ego_link_mask in the oracle against the autodiff oracle built with the same collision_links (EXJ:91-96)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import autodiff_oracle as ao
    g = np.load(os.path.join(GOLD, "panda_actions.npz"))
    i = 0                                              # a "dynamic" case with 4 obstacle spheres
    P = ao.Planner(mount=g["mount"][i], n_dynamic=4, collision_links=links)
    rb = {l: 0.05 + 0.01 * l for l in range(3, 9)}
    out = P.solve(g["q"][i], g["qd"][i], x_obsts_dynamic=g["ox"][i], xdot_obsts_dynamic=g["ov"][i],
                  xddot_obsts_dynamic=g["oa"][i], radius_obsts_dynamic=g["orad"][i], x_goal_0=g["g0"][i],
                  weight_goal_0=2.0, angle_goal_1=R1, x_goal_1=[0.107, 0.0, 0.0], weight_goal_1=20.0,
                  x_goal_2=[math.pi / 4], weight_goal_2=1.0, constraint_0=[0.0, 0.0, 1.0, -0.65], radius_body=rb,
                  return_parts=True)
    cfg = config.panda_config(n_robots=1, horizon=1, mounts=[g["mount"][i]])
    cfg.ego_link_mask = config.ego_link_mask(links)
    prm = params_row(g["g0"][i], [rb[l] for l in range(3, 9)])
    _, act = oracle.compute_action(cfg, g["q"][i][:, None], g["qd"][i][:, None], prm[:, None], g["ox"][i][:, :, None],
                                   g["ov"][i][:, :, None], g["oa"][i][:, :, None], g["orad"][i][:, None])
    assert rel(act[:, 0], out["action"]) < 1e-10
