"""Builders of the immutable `mrf_config` (include/mrf.h) from the reference's configuration surface.

Every constant that the survey could only *recall* from the un-vendored `fabrics` package is a named,
overridable field here (DESIGN.md "spec"), so a later session with `fabrics` installed can reconcile by
configuration rather than by code.
"""
import math
import os

import numpy as np

from . import abi
from .leafspec import parse_leaf

# strings the reference passes for the Pandas (example_pandas_Jointspace.py:87-89)
PANDA_STRINGS = dict(
    geometry_plane_constraint="10*(1/(1+1*ca.exp(-10*x))-1) * (xdot**2)",
    collision_geometry="-0.5 / (x ** 4) * (xdot ** 2)",
    collision_finsler="0.01/(x**4) * xdot**2",
)
# strings the reference passes for the point robots (example_pointmasses_static.py:106-107)
POINT_STRINGS = dict(
    collision_geometry="-2.0 / (x ** 1) * xdot ** 2",
    collision_finsler="1.0/(x**2) * (1 - ca.heaviside(xdot))* xdot**2",
)
# library defaults of fabrics.FabricPlannerConfig -- RECALLED, unverified
LIBRARY_STRINGS = dict(
    collision_geometry="-0.5 / (x ** 5) * (-0.5 * (ca.sign(xdot) - 1)) * xdot ** 2",
    collision_finsler="0.1/(x ** 1) * (-0.5 * (ca.sign(xdot) - 1)) * xdot**2",
    limit_geometry="-0.1 / (x ** 1) * xdot ** 2",
    limit_finsler="0.1/(x**1) * (-0.5 * (ca.sign(xdot) - 1)) * xdot**2",
    geometry_plane_constraint="-0.5 / (x ** 5) * (-0.5 * (ca.sign(xdot) - 1)) * xdot ** 2",
    finsler_plane_constraint="0.1/(x ** 1) * (-0.5 * (ca.sign(xdot) - 1)) * xdot**2",
)

PANDA_LIMITS = [[-2.8973, 2.8973], [-1.7628, 1.7628], [-2.8973, 2.8973], [-3.0718, -0.0698],
                [-2.8973, 2.8973], [-0.0175, 3.7525], [-2.8973, 2.8973]]  # EXJ:97-105
PANDA_VEL_LIMITS = [2.175, 2.175, 2.175, 2.175, 2.61, 2.61, 2.61]          # EXJ:221
Z_TABLE = 0.65                                                            # PM:81


def mount_positions(n_robots):
    """PM:83-105 for 2 and 3 robots; for other counts a build-defined ring around (0.5, 0, z_table) of
    radius max(0.75, 0.15 N) (neighbouring mounts ~0.9-1.0 m apart, like the reference's 1.0 m), the
    robots facing the centre -- the reference defines no layout there."""
    if n_robots == 1:
        return [np.array([0.0, 0.0, Z_TABLE])], [0.0]
    if n_robots == 2:
        return [np.array([0.0, 0.0, Z_TABLE]), np.array([1.0, 0.0, Z_TABLE])], [0.0, math.pi]
    if n_robots == 3:
        return ([np.array([0.0, 0.0, Z_TABLE]), np.array([1.0, 0.0, Z_TABLE]), np.array([0.7, 0.6, Z_TABLE])],
                [0.0, math.pi, math.pi])
    pos, yaw = [], []
    rad = max(0.75, 0.15 * n_robots)
    for i in range(n_robots):
        ang = 2.0 * math.pi * i / n_robots
        pos.append(np.array([0.5 + rad * math.cos(ang), rad * math.sin(ang), Z_TABLE]))
        yaw.append(ang + math.pi)
    return pos, yaw


def mount_transform(position, yaw):
    """T_0 = [Rz(yaw) | position]  (EXJ:107-118, PM:138-150)."""
    T = np.identity(4)
    T[0:2, 0:2] = np.array([[math.cos(yaw), -math.sin(yaw)], [math.sin(yaw), math.cos(yaw)]])
    T[0:3, 3] = position
    return T


def _set_leaf(dst, src):
    dst.family, dst.gate, dst.p, dst.reserved = src.family, src.gate, src.p, 0
    dst.k, dst.c, dst.s = src.k, src.c, src.s


def _common(cfg):
    cfg.abi_version = abi.MRF_ABI_VERSION
    cfg.scalar = abi.F64
    cfg.eps = 1e-6
    cfg.jdot_sign = -1.0            # utils.py:28
    cfg.goal_estimate_T = 20 * 0.01  # EXJ:347
    cfg.goal_estimate_mask = 0
    cfg.base_mass = 0.2             # base_energy "0.5 * 0.2 * ca.dot(xdot, xdot)"
    cfg.attr_k, cfg.attr_alpha = 5.0, 10.0
    cfg.attr_mu, cfg.attr_ml, cfg.attr_a = 2.0, 0.3, 0.75
    cfg.beta_a, cfg.beta_r, cfg.beta_b, cfg.beta_s = 0.5, 0.02, 6.5, 0.01
    cfg.eta_a, cfg.eta_s = 0.9 * (1 - 1 / 2), 0.5
    cfg.plane_abs = 1
    cfg.zero_small_action = 1
    cfg.ego_link_mask = 0x3F        # collision + plane leaves on all of links 3..8
    _apply_reconciled_fields(cfg)


# Reconciled constants (tests/reconcile_constants.py --write): every value the survey could only RECALL from the un-vendored
# `fabrics` package is a named field or a library string; once the reference's own vectors exist (tests/golden/
# make_reference_golden.py) the fit is written to a JSON file and picked up here -- no code change.  Looked for at
# $MRF_CONSTANTS, else multi-robot-fabrics_amd/constants.json; absent = the recalled defaults.  Format:
#   {"fields": {"attr_k": 5.0, "jdot_sign": -1.0, ...}, "strings": {"limit_finsler": "...", "finsler_plane_constraint": "..."}}
# strings are LIBRARY defaults: a string the caller passes explicitly (EXJ:87-89) wins.
CONSTANTS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "constants.json")
LIBRARY_STRING_KEYS = ("limit_geometry", "limit_finsler", "finsler_plane_constraint")


_constants_cache = {}      # (path, mtime) -> parsed file: configurations are built often, the file changes never or once


# Only the RECALLED values may be reconciled: a constants file cannot rewrite the ABI version, the scalar type,
# the model, robot counts or any other structural field of mrf_config.
RECONCILABLE_FIELDS = ("eps", "jdot_sign", "goal_estimate_T", "base_mass", "attr_k", "attr_alpha", "attr_mu", "attr_ml", "attr_a",
                       "beta_a", "beta_r", "beta_b", "beta_s", "eta_a", "eta_s", "plane_abs", "zero_small_action")


def reconciled_constants():
    explicit = os.environ.get("MRF_CONSTANTS")
    path = explicit or CONSTANTS_FILE
    try:
        key = (path, os.stat(path).st_mtime_ns)
    except OSError as e:
        if explicit:        # a typo must not silently run the recalled defaults
            raise FileNotFoundError(f"MRF_CONSTANTS={explicit!r} cannot be read: {e}") from e
        return {}
    if key not in _constants_cache:
        import json
        with open(path) as f:
            d = json.load(f)
        unknown = [k for k in d if k not in ("fields", "strings", "_meta")]
        if unknown:
            raise ValueError(f"{path}: unknown sections {unknown}")
        _constants_cache.clear()
        _constants_cache[key] = d
    return _constants_cache[key]


def _apply_reconciled_fields(cfg):
    for k, v in reconciled_constants().get("fields", {}).items():
        if k not in RECONCILABLE_FIELDS:
            raise KeyError(f"reconciled constants: {k!r} is not a reconcilable mrf_config field {RECONCILABLE_FIELDS}")
        setattr(cfg, k, type(getattr(cfg, k))(v))


def reconciled_constants_source():
    """None when the recalled defaults are in force, else {"path", "fields", "strings"} of the constants file every
    planner of this process is built with (surfaced by bench.py's line and FabricHandle.constants_source)."""
    d = reconciled_constants()
    if not d:
        return None
    return {"path": os.environ.get("MRF_CONSTANTS") or CONSTANTS_FILE, "fields": dict(d.get("fields", {})),
            "strings": dict(d.get("strings", {}))}


def _library_strings():
    s = dict(LIBRARY_STRINGS)
    for k, v in reconciled_constants().get("strings", {}).items():
        if k not in LIBRARY_STRING_KEYS:
            raise KeyError(f"reconciled constants: {k!r} is not a library-default string {LIBRARY_STRING_KEYS}")
        s[k] = v
    return s


def set_strings(cfg, **strings):
    """Apply leaf strings (same keyword names as ParameterizedFabricPlanner's config)."""
    mapping = dict(collision_geometry="collision_geometry", collision_finsler="collision_finsler",
                   geometry_plane_constraint="plane_geometry", finsler_plane_constraint="plane_finsler",
                   limit_geometry="limit_geometry", limit_finsler="limit_finsler")
    for key, val in strings.items():
        if key not in mapping:
            raise KeyError(f"unsupported planner string {key!r}; supported: {sorted(mapping)}")
        _set_leaf(getattr(cfg, mapping[key]), parse_leaf(val))


def set_spheres(cfg, links, offsets=None, radii=None):
    """Exchanged collision spheres of one robot: parent link number, link-local offset, radius."""
    n = len(links)
    if n > abi.MRF_MAX_SPHERES:
        raise ValueError(f"at most {abi.MRF_MAX_SPHERES} spheres per robot")
    cfg.n_spheres = n
    for s in range(n):
        cfg.sphere_link[s] = int(links[s])
        off = (0.0, 0.0, 0.0) if offsets is None else offsets[s]
        for c in range(3):
            cfg.sphere_offset[s][c] = float(off[c])
        cfg.sphere_radius[s] = 0.08 if radii is None else float(radii[s])


def ego_link_mask(link_numbers):
    """Bit mask of mrf_config.ego_link_mask for a list of Panda link numbers; links 1 and 2 do not move with q, so
    fabrics creates no leaf for them (FPJ:39,164: `coll_i > 2`) and they are dropped here too."""
    mask = 0
    for l in link_numbers:
        l = int(l)
        if l > 8:
            l = 8                   # 'panda_hand' (EXJ:95-96) has the origin of panda_link8
        if l > 2:
            mask |= 1 << (l - 3)
    return mask


def sphere_offsets_per_link(n_obst_per_link):
    """Link-local z offsets of the simulator's collision spheres (create_simulation_manipulators.py:188-245)
    for links 1..8: returns (links, offsets) with n_obst_per_link spheres per link."""
    length = [0.333, 0.2, 0.3164, 0.2, 0.3840, 0.2, 0.088, 0.2]
    kind = ["linear", "rotational"] * 4
    links, offsets = [], []
    for li in range(8):
        z_start = length[li] if kind[li] == "linear" else length[li] / 2
        for i in range(n_obst_per_link):
            off = [0.0, 0.0, -z_start + i * length[li] / n_obst_per_link]
            if li == 7:      # hand link (urdf index 16): two spheres shifted in x,y (SIM:232-239)
                if i == 1:
                    off = [0.03, 0.03, -z_start + (i + 1) * length[li] / n_obst_per_link]
                elif i == 2:
                    off[0], off[1] = -0.03, -0.03
            if li == 4 and i in (2, 3):   # bend of link 5 (urdf index 11) (SIM:240-246)
                off[0], off[1] = 0.0, (0.02 if i == 2 else 0.06)
            links.append(li + 1)
            offsets.append(off)
    return links, offsets


def c5_sphere_table():
    """BASELINE.json configs[4] "20 collision spheres/robot": the simulator's 3-per-link table (24 spheres) with every
    sixth entry dropped, so that every link keeps at least two -- in particular link 8 / the hand, the link that
    actually meets the other robots.  Returns (links, offsets)."""
    links, offs = sphere_offsets_per_link(3)
    keep = [i for i in range(len(links)) if i % 6 != 5]
    return [links[i] for i in keep], [offs[i] for i in keep]


def panda_config(n_robots=2, horizon=10, dynamic=1, n_ego=6, scalar=abi.F64, mounts=None, **strings):
    """Reference Panda planner (set_planner_panda, EXJ:64-134) for n_robots robots per scenario."""
    if not 1 <= n_robots <= abi.MRF_MAX_ROBOTS:
        raise ValueError(f"n_robots must be in 1..{abi.MRF_MAX_ROBOTS}")
    cfg = abi.Config()
    _common(cfg)
    cfg.model, cfg.scalar, cfg.mode = abi.MODEL_PANDA7, scalar, abi.MODE_VEL
    cfg.n_robots, cfg.horizon, cfg.dynamic = n_robots, horizon, dynamic
    cfg.n_ego, cfg.n_planes, cfg.use_limits, cfg.n_goals, cfg.obst_dim = n_ego, 1, 1, 3, 3
    cfg.dt = 0.01
    if mounts is None:
        pos, yaw = mount_positions(n_robots)
        mounts = [mount_transform(p, y) for p, y in zip(pos, yaw)]
    for i in range(n_robots):
        T = np.asarray(mounts[i], dtype=float)
        for r in range(3):
            for c in range(4):
                cfg.mount[i][r * 4 + c] = T[r, c]
    for j in range(7):
        cfg.limits[j][0], cfg.limits[j][1] = PANDA_LIMITS[j]
    set_spheres(cfg, list(range(1, 9)))       # 8 link origins, r = 0.08 (PM:23-26)
    s = _library_strings()
    s.update(PANDA_STRINGS)
    s.update(strings)
    set_strings(cfg, **s)
    return cfg


def planar3_config(n_robots=4, n_goals=1, obst_dim=3, scalar=abi.F64, **strings):
    """Point-robot planner (set_planner_point, example_pointmasses_static.py:102-129): dof 3,
    collision link base_link, no limits, no planes, mode 'acc'."""
    cfg = abi.Config()
    _common(cfg)
    cfg.model, cfg.scalar, cfg.mode = abi.MODEL_PLANAR3, scalar, abi.MODE_ACC
    cfg.n_robots, cfg.horizon, cfg.dynamic = n_robots, 1, 1
    cfg.n_ego, cfg.n_planes, cfg.use_limits, cfg.n_goals, cfg.obst_dim = 1, 0, 0, n_goals, obst_dim
    cfg.dt = 0.01
    eye = np.identity(4)
    for i in range(n_robots):
        for r in range(3):
            for c in range(4):
                cfg.mount[i][r * 4 + c] = eye[r, c]
    set_spheres(cfg, [1], radii=[0.2])
    s = _library_strings()
    s.update(POINT_STRINGS)
    s.update(strings)
    set_strings(cfg, **s)
    return cfg
