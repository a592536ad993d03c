"""Synthetic scenario batches for the Panda configurations, shared by tests and bench.py.

All arrays come out in the C-ABI layout: component-major over rows, row = scenario * n_robots + robot.
The generator is plain numpy on the host (seeded, reproducible); rejection sampling keeps every barrier
coordinate x >= x_min so that the float32 path can be judged on a well-conditioned set, and a separate
`near_barrier=True` set keeps the hard ones.
"""
import math

import numpy as np

from . import abi
from . import config as _config

ROT_GOAL_1 = np.array([[0.0, 0.0, -1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0]])  # PM:121

_HP = math.pi / 2
_XYZ = np.array([[0, 0, 0.333], [0, 0, 0], [0, -0.316, 0], [0.0825, 0, 0], [-0.0825, 0.384, 0], [0, 0, 0], [0.088, 0, 0]])
_ROLL = np.array([0, -_HP, _HP, _HP, -_HP, _HP, _HP])


def pos0(n_robots):
    """Start joint vectors (PM:92-95 for 2 robots, PM:111-115 for 3; the 2-robot pose otherwise)."""
    if n_robots == 3:
        return np.array([[1.13793529, -0.3227085, -0.02767777, -2.2204281, -0.00917029, 1.88612235, 0.78536134],
                         [1.131, 0.20, 0.12, -1.65, -0.0, 1.86, math.pi / 4],
                         [-0.46609715, -0.25025564, -0.40425878, -2.0966941, -0.10682593, 1.84917516, 0.37170524]])
    return np.tile(np.array([1.125, 0.19, 0.12, -1.66, -0.0, 1.88, math.pi / 4]), (n_robots, 1))


def start_goals(n_robots):
    """PM:126-132; for other counts a point 0.5 m above the table in front-left of each mount."""
    z = _config.Z_TABLE
    if n_robots == 2:
        return np.array([[0.2, 0.6, z + 0.5], [0.8, -0.6, z + 0.5]])
    if n_robots == 3:
        return np.array([[0.25, 0.6, z + 0.5], [0.8, -0.5, z + 0.5], [0.4, 0.5, z + 0.3]])
    pos, yaw = _config.mount_positions(n_robots)
    out = []
    for p, y in zip(pos, yaw):
        out.append([p[0] + 0.3 * math.cos(y) - 0.4 * math.sin(y), p[1] + 0.3 * math.sin(y) + 0.4 * math.cos(y), z + 0.5])
    return np.array(out)


def link_origins(q, mount):
    """Batched Panda FK: q [B,7], mount 4x4 -> origins of panda_link1..8, [B,8,3]."""
    B = q.shape[0]
    R = np.tile(mount[:3, :3], (B, 1, 1))
    p = np.tile(mount[:3, 3], (B, 1))
    out = np.zeros((B, 8, 3))
    for j in range(7):
        p = p + np.einsum("bij,j->bi", R, _XYZ[j])
        cr, sr = math.cos(_ROLL[j]), math.sin(_ROLL[j])
        Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
        c, s = np.cos(q[:, j]), np.sin(q[:, j])
        Rz = np.zeros((B, 3, 3))
        Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1], Rz[:, 2, 2] = c, -s, s, c, 1.0
        R = R @ Rx @ Rz
        out[:, j] = p
    out[:, 7] = p + R[:, :, 2] * 0.107
    return out


def cfg_mounts(cfg):
    out = []
    for i in range(cfg.n_robots):
        T = np.identity(4)
        for r in range(3):
            for c in range(4):
                T[r, c] = cfg.mount[i][r * 4 + c]
        out.append(T)
    return out


def min_barrier_coordinate(cfg, q, radius=0.08):
    """Smallest barrier coordinate x of the collision and plane leaves per scenario for joint positions q [b,N,7]:
    x = d/(2 r) - 1 between the ego points (link origins 3..8) of each robot and all 8 link origins of every other
    robot, and height above the table minus r for the plane leaf."""
    N = cfg.n_robots
    mounts = cfg_mounts(cfg)
    org = np.stack([link_origins(q[:, i], mounts[i]) for i in range(N)], axis=1)  # [b,N,8,3]
    xm = np.full(len(q), np.inf)
    for i in range(N):
        for j in range(N):
            if i == j:
                continue
            d = np.linalg.norm(org[:, i, 2:, None, :] - org[:, j, None, :, :], axis=-1)  # ego links 3..8 vs all 8
            xm = np.minimum(xm, (d / (2 * radius) - 1).reshape(len(q), -1).min(axis=1))
        xm = np.minimum(xm, (org[:, i, 2:, 2] - _config.Z_TABLE - radius).min(axis=1))     # plane leaf
    return xm


def panda_batch(cfg, n_scenarios, seed=0, x_min=0.05, q_spread=0.3, qd_spread=0.5, near_barrier=False,
                weight_goal_0=2.0, radius=0.08):
    """Returns dict(q, qdot [7,rows], params [29,rows]) float64 numpy for `cfg` (a panda mrf_config)."""
    N = cfg.n_robots
    rng = np.random.default_rng(seed)
    mounts = cfg_mounts(cfg)
    p0 = pos0(N)
    goals = start_goals(N)
    lim = np.array(_config.PANDA_LIMITS)
    vlim = np.array(_config.PANDA_VEL_LIMITS)
    q = np.zeros((n_scenarios, N, 7))
    need = np.ones(n_scenarios, dtype=bool)
    tries = 0
    while need.any():
        idx = np.nonzero(need)[0]
        cand = p0[None] + rng.uniform(-q_spread, q_spread, (len(idx), N, 7))
        cand = np.clip(cand, lim[:, 0] + 0.1, lim[:, 1] - 0.1)
        xm = min_barrier_coordinate(cfg, cand, radius)
        if near_barrier:
            ok = (xm > 0.01) & (xm < x_min)
        else:
            ok = xm >= x_min
        q[idx[ok]] = cand[ok]
        need[idx[ok]] = False
        tries += 1
        if tries > 2000:
            raise RuntimeError("scenario rejection sampling did not converge")
    qd = np.clip(rng.uniform(-qd_spread, qd_spread, (n_scenarios, N, 7)), -vlim, vlim)
    rows = n_scenarios * N
    prm = np.zeros((abi.NPARAM, rows))
    g0 = goals[None] + rng.uniform(-0.1, 0.1, (n_scenarios, N, 3))
    prm[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + 3] = g0.reshape(rows, 3).T
    prm[abi.P_WEIGHT_GOAL_0] = weight_goal_0
    prm[abi.P_ANGLE_GOAL_1:abi.P_ANGLE_GOAL_1 + 9] = ROT_GOAL_1.reshape(9, 1)
    prm[abi.P_X_GOAL_1:abi.P_X_GOAL_1 + 3] = np.array([[0.107], [0.0], [0.0]])
    prm[abi.P_WEIGHT_GOAL_1] = 20.0
    prm[abi.P_X_GOAL_2] = math.pi / 4
    prm[abi.P_WEIGHT_GOAL_2] = 1.0
    prm[abi.P_CONSTRAINT_0:abi.P_CONSTRAINT_0 + 4] = np.array([[0.0], [0.0], [1.0], [-_config.Z_TABLE]])
    prm[abi.P_RADIUS_BODY:abi.P_RADIUS_BODY + 6] = radius
    return dict(q=np.ascontiguousarray(q.reshape(rows, 7).T), qdot=np.ascontiguousarray(qd.reshape(rows, 7).T),
                params=prm)


def other_robot_obstacles(cfg, batch, spheres_x, spheres_v=None, spheres_a=None):
    """Host-side obstacle assembly of the control loop (EXJ:394-412): for every row the spheres of all
    *other* robots of its scenario.  spheres_* are [S,3,rows] (as mrf_fk_spheres returns them).
    Returns obst_x, obst_v, obst_a [M,3,rows], obst_r [M,rows] with M = S*(N-1)."""
    N, S = cfg.n_robots, cfg.n_spheres
    rows = spheres_x.shape[2]
    B = rows // N
    M = S * (N - 1)
    xp = np if isinstance(spheres_x, np.ndarray) else None
    if xp is None:
        import torch
        zeros = lambda shape: torch.zeros(shape, dtype=spheres_x.dtype, device=spheres_x.device)
    else:
        zeros = lambda shape: np.zeros(shape)
    ox, ov, oa, orad = zeros((M, 3, rows)), zeros((M, 3, rows)), zeros((M, 3, rows)), zeros((M, rows))
    sx = spheres_x.reshape(S, 3, B, N)
    sv = None if spheres_v is None else spheres_v.reshape(S, 3, B, N)
    sa = None if spheres_a is None else spheres_a.reshape(S, 3, B, N)
    oxv, ovv, oav, orv = ox.reshape(M, 3, B, N), ov.reshape(M, 3, B, N), oa.reshape(M, 3, B, N), orad.reshape(M, B, N)
    for i in range(N):
        m = 0
        for j in range(N):
            if j == i:
                continue
            oxv[m:m + S, :, :, i] = sx[:, :, :, j]
            if sv is not None:
                ovv[m:m + S, :, :, i] = sv[:, :, :, j]
            if sa is not None:
                oav[m:m + S, :, :, i] = sa[:, :, :, j]
            for s in range(S):
                orv[m + s, :, i] = cfg.sphere_radius[s]
            m += S
    return ox, ov, oa, orad


# ------------------------------------------------------------------------------------------------ BASELINE.json configs
BASELINE_CONFIGS = ("C2", "C3", "C5", "CART", "CARTC", "CART32", "CARTC32")


def baseline_config(name, scalar=abi.F64):
    """The BASELINE.json configurations beside the bench headline (C4 = 3-Panda RF-CV H=30), built in ONE place for
    bench.py's `configs` block, tools/prof_configs.py (the counter passes behind profiles/traffic.json) and the tests:
        C2    2-Panda MRDF, no rollout, 10 collision spheres per robot      -> mrf_compute_action_coupled
        C3    2-Panda Rollout Fabrics H=20, link-origin spheres             -> mrf_rollout
        C5    8-Panda RF-CV H=50, 20 spheres per robot (config.c5_sphere_table, 140 obstacle spheres per robot) -> mrf_rollout
        CART  3-Panda Cartesian rollout H=30, M=16 constant-velocity obstacle spheres per robot (the other
              robots' link origins at the start state), no obstacle accelerations (FPC:33) -> mrf_rollout_cartesian
        CARTC the same rollout through mrf_rollout_cartesian_coupled, the call the Cartesian example driver makes (EXC:330-399):
              the other robots' start states never leave the chip (LDS tile, k_rollout_cartc_panda)
        CART32 / CARTC32  the reference's DEFAULT Cartesian example (panda_config.yaml: 2 robots, n_obst_per_link: 4 -> 32 spheres
              per robot, EXC:184) at H=30: through obstacle arrays (mrf_rollout_cartesian) and through the coupled entry point,
              whose kernel assembles the obstacle arrays of its own rows in its prologue (k_rollout_carts_panda, round 6)
    -> dict(cfg, kind, batch = keyword arguments of panda_batch, scenarios_per_cu_round, label)."""
    if name == "C2":
        cfg = _config.panda_config(n_robots=2, horizon=1, scalar=scalar)
        links, offs = _config.sphere_offsets_per_link(2)          # 16 candidates; 10 of them, spread over links 2..8
        keep = [2, 3, 4, 6, 8, 9, 10, 12, 14, 15]
        _config.set_spheres(cfg, [links[i] for i in keep], [offs[i] for i in keep])
        return dict(cfg=cfg, kind="action_coupled", batch=dict(x_min=0.15), label="2-Panda MRDF jointspace, 10 spheres/robot")
    if name == "C3":
        cfg = _config.panda_config(n_robots=2, horizon=20, scalar=scalar)
        return dict(cfg=cfg, kind="rollout", batch={}, label="2-Panda RF H=20")
    if name == "C5":
        cfg = _config.panda_config(n_robots=8, horizon=50, scalar=scalar)
        links, offs = _config.c5_sphere_table()
        _config.set_spheres(cfg, links, offs)
        cfg.goal_estimate_mask = 0xFE
        return dict(cfg=cfg, kind="rollout", batch=dict(x_min=0.3, q_spread=0.15), label="8-Panda RF-CV H=50, 20 spheres/robot")
    if name == "CART":
        cfg = _config.panda_config(n_robots=3, horizon=30, scalar=scalar)
        return dict(cfg=cfg, kind="rollout_cartesian", batch=dict(x_min=0.1), label="3-Panda Cartesian rollout H=30, M=16")
    if name == "CARTC":
        cfg = _config.panda_config(n_robots=3, horizon=30, scalar=scalar)
        return dict(cfg=cfg, kind="rollout_cartesian_coupled", batch=dict(x_min=0.1),
                    label="3-Panda Cartesian rollout H=30 against the other robots' spheres (coupled entry point)")
    if name in ("CART32", "CARTC32"):
        cfg = _config.panda_config(n_robots=2, horizon=30, scalar=scalar)
        links, offs = _config.sphere_offsets_per_link(4)
        _config.set_spheres(cfg, links, offs)
        if name == "CART32":
            return dict(cfg=cfg, kind="rollout_cartesian", batch=dict(x_min=0.1),
                        label="2-Panda Cartesian rollout H=30, M=32 (n_obst_per_link=4) through obstacle arrays")
        return dict(cfg=cfg, kind="rollout_cartesian_coupled", batch=dict(x_min=0.1),
                    label="2-Panda Cartesian rollout H=30 against the other robot's 32 spheres (coupled entry point, one launch)")
    raise KeyError(f"unknown baseline configuration {name!r}: {BASELINE_CONFIGS}")


def tiled_batch(cfg, n_scenarios, seed, unique=4096, **kw):
    """A batch of n_scenarios scenarios whose first `unique` ones are drawn by panda_batch and then repeated: large timing
    batches without minutes of host-side rejection sampling (the oracle spot checks look at the first scenarios)."""
    N = cfg.n_robots
    u = min(unique, n_scenarios)
    b = panda_batch(cfg, u, seed=seed, **kw)
    reps = -(-n_scenarios // u)
    return {k: np.ascontiguousarray(np.tile(v, (1, reps))[:, :n_scenarios * N]) for k, v in b.items()}
