"""Host-side mirror of the reference's kinematics helpers, evaluated by the HIP sphere-kinematics kernel.

Mirrors (same names, argument meaning and return shapes):
  GenericURDFFk                              forwardkinematics' URDF front-end as the reference uses it
                                             (example_pandas_Jointspace.py:83,118; utils.py:99-108)
  UtilsKinematics                            multi_robot_fabrics/utils/utils.py:7-136
  compute_x_obsts_dyn_0, compute_endeffector multi_robot_fabrics/utils/utils_apply_fk.py:3-44

The reference builds CasADi functions; here every returned "function" is a small callable that launches
mrf_fk_spheres for its link and returns a `DM`-like array (`.full()` as in CasADi), so call sites such as
`fk_dict["jac_fun"][i][l](q) @ qdot` and `.full().transpose()[0]` read unchanged.
"""
import math
import re

import numpy as np

from . import abi
from . import config as _config

PANDA_LINKS = ["panda_link%d" % i for i in range(1, 9)] + ["panda_hand"]
_PANDA_JOINTS = {  # name: (xyz, rpy) -- the constants compiled into csrc/mrf_device.hpp
    "panda_joint1": ((0, 0, 0.333), (0, 0, 0)),
    "panda_joint2": ((0, 0, 0), (-math.pi / 2, 0, 0)),
    "panda_joint3": ((0, -0.316, 0), (math.pi / 2, 0, 0)),
    "panda_joint4": ((0.0825, 0, 0), (math.pi / 2, 0, 0)),
    "panda_joint5": ((-0.0825, 0.384, 0), (-math.pi / 2, 0, 0)),
    "panda_joint6": ((0, 0, 0), (math.pi / 2, 0, 0)),
    "panda_joint7": ((0.088, 0, 0), (math.pi / 2, 0, 0)),
    "panda_joint8": ((0, 0, 0.107), (0, 0, 0)),
}


class DM(np.ndarray):
    """numpy array with CasADi's `.full()`: vectors come back as columns, like a casadi.DM."""

    def __new__(cls, a):
        return np.asarray(a, dtype=np.float64).view(cls)

    def full(self):
        a = np.asarray(self)
        return a.reshape(-1, 1) if a.ndim < 2 else a


def link_number(name):
    """'panda_link5' -> 5, 'panda_hand' -> 8 (same origin as panda_link8, URDF :466-470), 'base_link' -> 1."""
    if name == "panda_hand":
        return 8
    if name == "base_link":
        return 1
    m = re.fullmatch(r"panda_link([1-8])", name)
    if not m:
        raise KeyError(f"link {name!r} is not on the compiled kinematic chain")
    return int(m.group(1))


def _check_urdf(urdf_text, model):
    """The chain constants are compiled into the kernels; refuse a URDF that disagrees with them."""
    import xml.etree.ElementTree as ET
    try:
        root = ET.fromstring(urdf_text)
    except ET.ParseError as e:
        raise ValueError(f"URDF does not parse: {e}") from None
    if model != "panda":
        return
    found = 0
    for j in root.findall("joint"):
        name = j.get("name")
        if name in _PANDA_JOINTS:
            org = j.find("origin")
            xyz = tuple(float(v) for v in (org.get("xyz") or "0 0 0").split())
            rpy = tuple(float(v) for v in (org.get("rpy") or "0 0 0").split())
            wx, wr = _PANDA_JOINTS[name]
            if max(abs(a - b) for a, b in zip(xyz + rpy, tuple(wx) + tuple(wr))) > 1e-9:
                raise ValueError(f"URDF joint {name} = {xyz},{rpy} differs from the chain compiled into the kernels")
            found += 1
    if found != len(_PANDA_JOINTS):
        raise ValueError("URDF does not contain the Panda chain panda_joint1..8")


class GenericURDFFk:
    """Forward-kinematics descriptor.  Supported chains: the Panda of panda_with_finger.urdf
    (root 'panda_link0', end 'panda_leftfinger') and the point robot of pointRobot1.urdf ('world' -> 'base_link')."""

    def __init__(self, urdf, root_link=None, end_links=None, rootLink=None, end_link=None):
        root = root_link or rootLink
        end = end_links or end_link
        end = end[0] if isinstance(end, (list, tuple)) else end
        if root == "panda_link0":
            self.model = "panda"
            self._n = 8          # 7 revolute + finger (utils.py:100-102 pads q with a zero)
        elif root == "world" and end == "base_link":
            self.model = "point"
            self._n = 3
        else:
            raise NotImplementedError(f"kinematic chain {root!r} -> {end!r} is not compiled into the kernels")
        if urdf:
            _check_urdf(urdf, self.model)
        self._mount = np.identity(4)

    def n(self):
        return self._n

    def set_mount_transformation(self, mount_transformation):
        T = np.asarray(mount_transformation, dtype=float)
        if T.shape != (4, 4):
            raise ValueError("mount transformation must be 4x4")
        self._mount = T.copy()

    @property
    def mount(self):
        return self._mount


class _SphereEvaluator:
    """One FabricHandle per (mount, sphere table); evaluates x, v, a for a handful of rows."""

    _cache = {}

    def __init__(self, mount, links, offsets):
        from .runtime import FabricHandle
        cfg = _config.panda_config(n_robots=1, horizon=1, mounts=[mount])
        _config.set_spheres(cfg, links, offsets)
        self.S = len(links)
        self.h = FabricHandle(cfg, getattr(self, "device", None))

    @classmethod
    def get(cls, mount, links, offsets=None):
        offsets = [(0.0, 0.0, 0.0)] * len(links) if offsets is None else [tuple(float(v) for v in o) for o in offsets]
        import torch
        dev = torch.cuda.current_device() if torch.cuda.is_available() else -1   # handles are per device
        key = (dev, np.asarray(mount, dtype=float).tobytes(), tuple(links), tuple(offsets))
        if key not in cls._cache:
            cls._cache[key] = cls(mount, list(links), offsets)
        return cls._cache[key]

    def eval(self, q_rows, qd_rows=None):
        """q_rows [rows,7] -> x, v, a  each [S,3,rows] numpy (v, a None without qd)."""
        q = np.ascontiguousarray(np.asarray(q_rows, dtype=float)[:, :7].T)
        qd = None if qd_rows is None else np.ascontiguousarray(np.asarray(qd_rows, dtype=float)[:, :7].T)
        return self.h.fk_spheres_host(q, qd)          # numpy in, numpy out (mrf_fk_spheres_host)


def _vec(q):
    return np.asarray(q, dtype=float).reshape(-1)


class _LinkFunctions:
    """fk, jac, jac_dot of one link origin (utils.py:31-49), sign convention jac_dot = -d(J qd)/dq (:28,:37)."""

    def __init__(self, mount, link):
        self.ev = _SphereEvaluator.get(mount, [link])

    def fk(self, q):
        x, _, _ = self.ev.eval(_vec(q)[None, :7])
        return DM(x[0, :, 0])

    def jac(self, q):
        q = _vec(q)[:7]
        x, v, _ = self.ev.eval(np.tile(q, (7, 1)), np.identity(7))       # v(q, e_j) = J[:, j]
        return DM(v[0].copy())                                            # [3, 7]

    def jac_dot(self, q, qdot):
        # a(q, qd) = jac_dot(q, qd) qd is a quadratic form in qd; its matrix follows by polarisation:
        # D[:, k] = (a(qd + e_k) - a(qd) - a(e_k)) / 2
        q, qd = _vec(q)[:7], _vec(qdot)[:7]
        rows = np.concatenate([qd[None], np.identity(7), qd[None] + np.identity(7)])
        _, _, a = self.ev.eval(np.tile(q, (15, 1)), rows)
        a = a[0]
        return DM((a[:, 8:15] - a[:, 0:1] - a[:, 1:8]) / 2.0)


class UtilsKinematics:
    """Mirror of multi_robot_fabrics/utils/utils.py:7-136."""

    def __init__(self):
        self.nr_robots = 0

    def necessary_kinematics(self, planner, center=False, i_robot=0):
        if center:
            raise NotImplementedError("link-centre kinematics are not used by the reference's examples (utils.py:74)")
        mount = planner._forward_kinematics.mount
        fk_fun, jac_fun, jac_dot_fun = {}, {}, {}
        for link in PANDA_LINKS:          # links[0:index('panda_leftfinger')] of the URDF chain (utils.py:29-31)
            f = _LinkFunctions(mount, link_number(link))
            fk_fun[link], jac_fun[link], jac_dot_fun[link] = f.fk, f.jac, f.jac_dot
        return fk_fun, jac_fun, jac_dot_fun

    def define_forward_kinematics(self, planners, collision_links_nrs, collision_links):
        nr_robots = len(collision_links_nrs)
        self.nr_robots = nr_robots
        keys = ["fk_fun_center", "jac_fun_center", "jac_dot_fun_center", "fk_fun", "jac_fun", "jac_dot_fun"]
        fk_dict = {k: [[] for _ in range(nr_robots)] for k in keys}
        for i_robot in range(nr_robots):
            fk_fun, jac_fun, jac_dot_fun = self.necessary_kinematics(planners[i_robot])
            for link in collision_links[i_robot]:
                fk_dict["fk_fun"][i_robot].append(fk_fun[link])
                fk_dict["jac_fun"][i_robot].append(jac_fun[link])
                fk_dict["jac_dot_fun"][i_robot].append(jac_dot_fun[link])
        return fk_dict

    def define_symbolic_collision_link_poses(self, urdf_files, collision_links, sphere_transformations,
                                             n_obst_per_link=1, mount_transform=()):
        """fk_spheres[i]["fk_fun"](q) -> 3 x S positions, ["vel_fun"](q, qdot) -> 3 x S velocities (utils.py:87-119);
        q, qdot carry the finger joint as an eighth entry (utils_apply_fk.py:12-13), which no sphere depends on."""
        nr_robots = len(sphere_transformations)
        self.nr_robots = nr_robots
        out = []
        for i_robot in range(nr_robots):
            links, offsets = [], []
            for i_link, link in enumerate(collision_links[i_robot]):
                for i_sphere in range(n_obst_per_link):
                    T = np.asarray(sphere_transformations[i_robot][i_link][i_sphere], dtype=float)
                    links.append(link_number(link))
                    offsets.append(T[0:3, 3])
            ev = _SphereEvaluator.get(mount_transform[i_robot], links, offsets)
            out.append({"fk_fun": (lambda q, ev=ev: DM(ev.eval(_vec(q)[None, :7])[0][:, :, 0].T)),
                        "vel_fun": (lambda q, qd, ev=ev: DM(ev.eval(_vec(q)[None, :7], _vec(qd)[None, :7])[1][:, :, 0].T))})
        return out

    def define_symbolic_endeffector(self, planners):
        if not self.nr_robots:
            self.nr_robots = len(planners)
        out = []
        for i_robot in range(self.nr_robots):
            ev = _SphereEvaluator.get(planners[i_robot]._forward_kinematics.mount, [8])   # panda_hand
            out.append({"fk_fun_ee": (lambda q, ev=ev: DM(ev.eval(_vec(q)[None, :7])[0][0, :, 0])),
                        "vel_fun_ee": (lambda q, qd, ev=ev: DM(ev.eval(_vec(q)[None, :7], _vec(qd)[None, :7])[1][0, :, 0]))})
        return out


def compute_x_obsts_dyn_0(q_robots, qdot_robots, x_collision_sphere_poses=None, nr_robots=2, fk_dict_spheres=(),
                          nr_dyn_obsts=(0, 0)):
    """utils_apply_fk.py:3-33: positions from the simulator's sphere dictionary, velocities from the sphere FK."""
    q = [np.append(_vec(q_robots[i]), 0) for i in range(nr_robots)]
    qdot = [np.append(_vec(qdot_robots[i]), 0) for i in range(nr_robots)]
    x_dyns_obsts = [[] for _ in range(nr_robots)]
    v_dyns_obsts = [[] for _ in range(nr_robots)]
    x_per_robot = [[] for _ in range(nr_robots)]
    for i_robot in range(nr_robots):
        x_per_robot[i_robot] = [x for key, x in x_collision_sphere_poses.items() if str(i_robot) in key[0]]
        for i_other in (i for i in range(nr_robots) if i != i_robot):
            x_dyns_obsts[i_other] = x_dyns_obsts[i_other] + x_per_robot[i_robot]
            vel = fk_dict_spheres[i_other]["vel_fun"](q[i_other], qdot[i_other]).full().transpose()
            v_dyns_obsts[i_robot].extend(np.vsplit(vel, vel.shape[0]))    # one (1,3) block per sphere
    return x_dyns_obsts, v_dyns_obsts, x_per_robot


def compute_endeffector(q_robots, qdot_robots, fk_endeff, nr_robots=2):
    """utils_apply_fk.py:35-44."""
    x_ee, v_ee = [], []
    for i in range(nr_robots):
        x_ee.append(fk_endeff[i]["fk_fun_ee"](q_robots[i]).full().transpose()[0])
        v_ee.append(fk_endeff[i]["vel_fun_ee"](q_robots[i], qdot_robots[i]).full().transpose()[0])
    return x_ee, v_ee
