"""`ParameterizedFabricPlanner` with the call surface the reference uses, evaluated by the HIP kernels.

Reference call sites mirrored (the class itself lives in the un-vendored `fabrics` package):
    ParameterizedFabricPlanner(dof, fk, **leaf_strings)        example_pandas_Jointspace.py:84-90
    planner._forward_kinematics.set_mount_transformation(T)    :118
    planner.set_components(collision_links=..., goal=..., number_obstacles=..., number_dynamic_obstacles=...,
                           dynamic_obstacle_dimension=..., number_plane_constraints=..., limits=...)   :123-132
    planner.concretize(mode='vel', time_step=0.01)             :133      (pointmass: concretize() -> 'acc')
    planner.compute_action(**kwargs) -> np.ndarray[dof]        :441,444; example_pointmasses_static.py:191-199
    planner._funs._input_keys / planner._funs._function        forward_planner_Jointspace.py:184-185

Error behaviour follows the Python reference: bad configuration raises at set_components/concretize,
a missing runtime argument raises KeyError at compute_action.  There is no CPU evaluation path.
"""
import numpy as np

from . import abi
from . import config as _config
from .kinematics import GenericURDFFk, link_number  # noqa: F401  (re-exported for drop-in imports)
from .leafspec import LeafSpecError  # noqa: F401

_STRING_KEYS = ("collision_geometry", "collision_finsler", "geometry_plane_constraint", "finsler_plane_constraint",
                "limit_geometry", "limit_finsler")
_DEFAULT_ONLY_KEYS = ("base_energy", "attractor_potential", "attractor_metric", "damper_beta", "damper_eta",
                      "self_collision_geometry", "self_collision_finsler")


class _Variables:
    """Stand-in for fabrics' Variables: only names, there is no symbolic graph here."""

    def __init__(self, dof):
        self._dof = dof

    def position_variable(self):
        return ["q_%d" % i for i in range(self._dof)]

    def velocity_variable(self):
        return ["qdot_%d" % i for i in range(self._dof)]


class _Funs:
    """`planner._funs`: the sorted input names and a positional evaluator (FPJ:184-185,227-233)."""

    def __init__(self, planner):
        self._planner = planner
        self._input_keys = planner.input_keys()

    def _function(self, *args):
        if len(args) != len(self._input_keys):
            raise TypeError(f"expected {len(self._input_keys)} positional inputs {self._input_keys}, got {len(args)}")
        return self._planner.compute_action(**dict(zip(self._input_keys, args)))


class ParameterizedFabricPlanner:
    def __init__(self, dof, forward_kinematics, **kwargs):
        if not isinstance(forward_kinematics, GenericURDFFk):
            raise TypeError("forward_kinematics must be a multi_robot_fabrics_amd GenericURDFFk")
        self._dof = dof
        self._forward_kinematics = forward_kinematics
        self._model = forward_kinematics.model
        want_dof = 7 if self._model == "panda" else 3
        if dof != want_dof:
            raise ValueError(f"the compiled {self._model} chain has {want_dof} degrees of freedom, got dof={dof}")
        self._strings = {}
        for key, val in kwargs.items():
            if key in _STRING_KEYS:
                self._strings[key] = val
            elif key in _DEFAULT_ONLY_KEYS:
                raise NotImplementedError(f"{key} is fixed to the library default family; override its constants "
                                          "through planner.constants instead of a string")
            else:
                raise TypeError(f"unknown planner configuration {key!r}")
        self.constants = {}          # e.g. {'attr_k': 5.0, 'eps': 1e-6, 'jdot_sign': -1.0}: any scalar field of mrf_config
        self.variables = _Variables(dof)
        self.leaves = {}
        self._components = None
        self._handle = None
        self._funs = None

    # ------------------------------------------------------------------ configuration
    def set_components(self, collision_links=None, self_collision_pairs=None, goal=None, number_obstacles=1,
                       number_dynamic_obstacles=0, dynamic_obstacle_dimension=3, number_plane_constraints=0,
                       limits=None, **unsupported):
        if unsupported:
            raise NotImplementedError(f"set_components arguments not on the hot path: {sorted(unsupported)}")
        if self_collision_pairs:
            raise NotImplementedError("self-collision leaves are commented out in the reference (EXJ:120-124)")
        collision_links = list(collision_links or [])
        if self._model == "panda":
            # q-independent links are skipped exactly as fabrics skips sparse FKs (FPJ:39,164: `coll_i > 2`)
            active = sorted({min(link_number(name), 8) for name in collision_links if link_number(name) > 2})
            n_ego = 6 if active else 0               # any subset of links 3..8: mrf_config.ego_link_mask
            if limits is not None:
                lim = np.asarray(limits, dtype=float)
                if lim.shape != (7, 2):
                    raise ValueError("limits must be 7 x [lower, upper]")
        else:
            if any(name != "base_link" for name in collision_links):
                raise NotImplementedError("the point robot has the single collision link 'base_link'")
            n_ego = 1 if collision_links else 0
            if limits is not None or number_plane_constraints:
                raise NotImplementedError("the point-robot planner has no limit / plane leaves (pointmass :122-127)")
        n_goals = 0
        if goal is not None:
            subs = goal.sub_goals()
            n_goals = len(subs)
            self._check_goal(subs)
        if n_ego == 0:
            number_obstacles = number_dynamic_obstacles = 0      # no collision link -> no obstacle / plane leaves
        self._ego_links = active if self._model == "panda" else []
        self._components = dict(n_ego=n_ego, n_static=int(number_obstacles), n_dynamic=int(number_dynamic_obstacles),
                                obst_dim=int(dynamic_obstacle_dimension), n_planes=int(number_plane_constraints) if n_ego else 0,
                                limits=None if limits is None else np.asarray(limits, dtype=float), n_goals=n_goals)
        links = self._ego_links if (self._model == "panda" and n_ego) else ([1] if n_ego else [])
        self.leaves = {}
        for l in links:
            name = "panda_link%d" % l if self._model == "panda" else "base_link"
            for j in range(number_obstacles):
                self.leaves[f"{name}_obst_{j}_leaf"] = ("static", l, j)
            for j in range(number_dynamic_obstacles):
                self.leaves[f"{name}_obst_dynamic_{j}_leaf"] = ("dynamic", l, j)
            for j in range(self._components["n_planes"]):
                self.leaves[f"{name}_constraint_{j}_leaf"] = ("plane", l, j)
        if limits is not None:
            for j in range(7):
                self.leaves[f"limit_joint_{j}_0_leaf"] = ("limit", j, 0)
                self.leaves[f"limit_joint_{j}_1_leaf"] = ("limit", j, 1)
        for g in range(n_goals):
            self.leaves[f"goal_{g}_leaf"] = ("attractor", g, 0)

    def _check_goal(self, subs):
        if self._model == "panda":
            if len(subs) > 3:
                raise NotImplementedError("at most the three sub-goals of create_dummy_goal_panda (EXJ:25-62)")
            want = [("staticSubGoal", [0, 1, 2], "panda_hand"), ("staticSubGoal", [0, 1, 2], "panda_hand"),
                    ("staticJointSpaceSubGoal", [6], None)]
            for g, (typ, idx, child) in zip(subs, want):
                if g.type() != typ or g.indices() != idx or (child and g.child_link() != child):
                    raise NotImplementedError(f"sub-goal {g.name()} is not one of the reference's Panda sub-goals")
            if len(subs) > 1 and subs[1].parent_link() != "panda_link7":
                raise NotImplementedError("sub-goal 1 must be panda_link7 -> panda_hand (EXJ:42-52)")
        else:
            if len(subs) != 1 or subs[0].indices() != [0, 1] or subs[0].child_link() != "base_link":
                raise NotImplementedError("the point robot has one 2-D goal on base_link (pointmass :61-71)")

    def concretize(self, mode="acc", time_step=None):
        if self._components is None:
            raise RuntimeError("call set_components() before concretize()")
        if mode not in ("acc", "vel"):
            raise ValueError("mode must be 'acc' or 'vel'")
        if mode == "vel" and not time_step:
            raise ValueError("mode 'vel' needs a time_step")
        c = self._components
        if self._model == "panda":
            cfg = _config.panda_config(n_robots=1, horizon=1, n_ego=c["n_ego"], mounts=[self._forward_kinematics.mount],
                                       **self._strings)
            cfg.n_planes = c["n_planes"]
            if c["n_ego"]:
                cfg.ego_link_mask = _config.ego_link_mask(self._ego_links)
            cfg.use_limits = 0 if c["limits"] is None else 1
            if c["limits"] is not None:
                for j in range(7):
                    cfg.limits[j][0], cfg.limits[j][1] = c["limits"][j]
        else:
            cfg = _config.planar3_config(n_robots=1, n_goals=c["n_goals"], **self._strings)
            cfg.n_ego = c["n_ego"]
        cfg.n_goals = c["n_goals"]
        cfg.obst_dim = c["obst_dim"]
        cfg.mode = abi.MODE_VEL if mode == "vel" else abi.MODE_ACC
        if time_step:
            cfg.dt = float(time_step)
        for key, val in self.constants.items():
            if not hasattr(cfg, key):
                raise KeyError(f"unknown mrf_config field {key!r}")
            setattr(cfg, key, val)
        self._mode = mode
        self.config = cfg
        from .runtime import FabricHandle
        self._handle = FabricHandle(cfg, getattr(self, "device", None))      # raises without the HIP library or a GPU: there is no CPU path
        self._funs = _Funs(self)

    # ------------------------------------------------------------------ evaluation
    def input_keys(self):
        """Runtime parameter names in the order of the reference's positional call (FPJ:227-232)."""
        c = self._components
        body = "radius_body_panda_link%d" if self._model == "panda" else "radius_body_base_link"
        keys = []
        if self._model == "panda" and c["n_goals"] > 1:
            keys.append("angle_goal_1")
        keys += ["constraint_%d" % j for j in range(c["n_planes"])]
        keys += ["q", "qdot"]
        if c["n_ego"]:
            keys += [body % l for l in self._ego_links] if self._model == "panda" else [body]
        keys += ["radius_obst_%d" % j for j in range(c["n_static"])]
        keys += ["radius_obst_dynamic_%d" % j for j in range(c["n_dynamic"])]
        keys += ["weight_goal_%d" % g for g in range(c["n_goals"])]
        keys += ["x_goal_%d" % g for g in range(c["n_goals"])]
        keys += ["x_obst_%d" % j for j in range(c["n_static"])]
        keys += ["x_obst_dynamic_%d" % j for j in range(c["n_dynamic"])]
        keys += ["xddot_obst_dynamic_%d" % j for j in range(c["n_dynamic"])]
        keys += ["xdot_obst_dynamic_%d" % j for j in range(c["n_dynamic"])]
        return keys

    @staticmethod
    def _listed(kw, plural, singular, count, what):
        """list-valued kwargs are expanded to name_0, name_1, ... as fabrics does (EXJ:430-438)."""
        if count == 0:
            return []
        if plural in kw:
            vals = list(kw[plural])
            if len(vals) < count:
                raise KeyError(f"{plural}: the planner was built with {count} {what}, got {len(vals)}")
            return vals[:count]
        try:
            return [kw[singular % j] for j in range(count)]
        except KeyError as e:
            raise KeyError(f"compute_action: missing argument {e.args[0]!r} (or the list form {plural!r})") from None

    def params_row(self, kw):
        c = self._components
        p = np.zeros(abi.NPARAM)
        for g in range(c["n_goals"]):
            if f"x_goal_{g}" not in kw or f"weight_goal_{g}" not in kw:
                raise KeyError(f"compute_action: missing argument 'x_goal_{g}' / 'weight_goal_{g}'")
        if c["n_goals"] > 0:
            g0 = np.asarray(kw["x_goal_0"], dtype=float).reshape(-1)
            p[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + len(g0)] = g0
            p[abi.P_WEIGHT_GOAL_0] = float(np.asarray(kw["weight_goal_0"]).reshape(-1)[0])
        if self._model == "panda":
            if c["n_goals"] > 1:
                if "angle_goal_1" not in kw:
                    raise KeyError("compute_action: missing argument 'angle_goal_1'")
                p[abi.P_ANGLE_GOAL_1:abi.P_ANGLE_GOAL_1 + 9] = np.asarray(kw["angle_goal_1"], dtype=float).reshape(9)
                p[abi.P_X_GOAL_1:abi.P_X_GOAL_1 + 3] = np.asarray(kw["x_goal_1"], dtype=float).reshape(3)
                p[abi.P_WEIGHT_GOAL_1] = float(np.asarray(kw["weight_goal_1"]).reshape(-1)[0])
            if c["n_goals"] > 2:
                p[abi.P_X_GOAL_2] = float(np.asarray(kw["x_goal_2"]).reshape(-1)[0])
                p[abi.P_WEIGHT_GOAL_2] = float(np.asarray(kw["weight_goal_2"]).reshape(-1)[0])
            if c["n_planes"]:
                if "constraint_0" not in kw:
                    raise KeyError("compute_action: missing argument 'constraint_0'")
                p[abi.P_CONSTRAINT_0:abi.P_CONSTRAINT_0 + 4] = np.asarray(kw["constraint_0"], dtype=float).reshape(4)
            if c["n_ego"]:
                # only the links that carry leaves need a body radius (the planner has no parameter for the others)
                if "radius_body_panda_links" in kw:
                    rb = kw["radius_body_panda_links"]
                    vals = {l: float(np.asarray(rb[str(l)]).reshape(-1)[0]) for l in self._ego_links}
                else:
                    try:
                        vals = {l: float(np.asarray(kw["radius_body_panda_link%d" % l]).reshape(-1)[0]) for l in self._ego_links}
                    except KeyError as e:
                        raise KeyError(f"compute_action: missing argument {e.args[0]!r}") from None
                for l, r in vals.items():
                    p[abi.P_RADIUS_BODY + l - 3] = r
        elif c["n_ego"]:
            if "radius_body_base_link" not in kw:
                raise KeyError("compute_action: missing argument 'radius_body_base_link'")
            p[abi.P_RADIUS_BODY] = float(np.asarray(kw["radius_body_base_link"]).reshape(-1)[0])
        return p

    def obstacle_arrays(self, kw):
        """-> ox, ov, oa [M,3], orad [M], n_static  with the static spheres first."""
        c = self._components
        ns, nd = c["n_static"], c["n_dynamic"]
        xs = self._listed(kw, "x_obsts", "x_obst_%d", ns, "static obstacles")
        rs = self._listed(kw, "radius_obsts", "radius_obst_%d", ns, "static obstacles")
        xd = self._listed(kw, "x_obsts_dynamic", "x_obst_dynamic_%d", nd, "dynamic obstacles")
        vd = self._listed(kw, "xdot_obsts_dynamic", "xdot_obst_dynamic_%d", nd, "dynamic obstacles")
        ad = self._listed(kw, "xddot_obsts_dynamic", "xddot_obst_dynamic_%d", nd, "dynamic obstacles")
        rd = self._listed(kw, "radius_obsts_dynamic", "radius_obst_dynamic_%d", nd, "dynamic obstacles")
        M = ns + nd
        ox, ov, oa, orad = np.zeros((M, 3)), np.zeros((M, 3)), np.zeros((M, 3)), np.zeros(M)

        def put(dst, row, val):
            v = np.asarray(val, dtype=float).reshape(-1)
            dst[row, :len(v)] = v[:3]

        def block(dst, lo, vals):
            """all entries at once when they are uniform vectors (the usual case), entry by entry otherwise"""
            try:
                a = np.asarray(vals, dtype=float).reshape(len(vals), -1)
                dst[lo:lo + len(vals), :min(3, a.shape[1])] = a[:, :3]
            except ValueError:
                for j, v in enumerate(vals):
                    put(dst, lo + j, v)

        def radii(lo, vals):
            try:
                orad[lo:lo + len(vals)] = np.asarray(vals, dtype=float).reshape(len(vals), -1)[:, 0]
            except ValueError:
                for j, v in enumerate(vals):
                    orad[lo + j] = float(np.asarray(v).reshape(-1)[0])

        if ns:
            block(ox, 0, xs)
            radii(0, rs)
        if nd:
            block(ox, ns, xd); block(ov, ns, vd); block(oa, ns, ad)
            radii(ns, rd)
        return ox, ov, oa, orad, ns

    def compute_action(self, **kwargs):
        if self._handle is None:
            raise RuntimeError("call concretize() before compute_action()")
        if "q" not in kwargs or "qdot" not in kwargs:
            raise KeyError("compute_action: missing argument 'q' / 'qdot'")
        h = self._handle
        q = np.asarray(kwargs["q"], dtype=float).reshape(-1)[:self._dof]
        qd = np.asarray(kwargs["qdot"], dtype=float).reshape(-1)[:self._dof]
        prm = self.params_row(kwargs)
        ox, ov, oa, orad, ns = self.obstacle_arrays(kwargs)
        M = ox.shape[0]
        # host arrays in, host array out: one packed copy each way inside the library (mrf_compute_action_host)
        if M:
            # zero accelerations (what the reference's drivers pass, EXJ:411) select the kernel without the acceleration
            # loads and the n.a_o term: obst_a = NULL means zeros (include/mrf.h)
            act = h.compute_action_host(q[:, None], qd[:, None], prm[:, None], ox[:, :, None], ov[:, :, None],
                                        oa[:, :, None] if oa.any() else None, orad[:, None], n_static=ns)
        else:
            act = h.compute_action_host(q[:, None], qd[:, None], prm[:, None])
        return act[:, 0].copy()

    # ------------------------------------------------------------------ kinematics access (EXJ:235, utils.py:35)
    def get_forward_kinematics(self, link_name, position_only=True):
        """Returns a callable q -> position of the link origin (the reference gets a CasADi expression)."""
        from .kinematics import _LinkFunctions
        if self._model != "panda":
            return lambda q: np.array([q[0], q[1], 0.05])
        return _LinkFunctions(self._forward_kinematics.mount, link_number(link_name)).fk

    def get_leaves(self, leaf_names):
        return [self.leaves[name] for name in leaf_names]


def panda_planner(urdf_path, mount, goal, link_numbers, n_static, n_dynamic, dt=0.01, device=None, **strings):
    """One concretized Panda planner as the example drivers configure it: chain read from `urdf_path` (and checked against
    the compiled one), mounted at the 4x4 `mount`, collision + table-plane leaves on `link_numbers` (numbers above 8 mean
    the hand), joint-limit leaves, `goal`, velocity output with time step dt.  Leaf strings default to the drivers'
    (config.PANDA_STRINGS = EXJ:87-89)."""
    with open(urdf_path, "r") as f:
        fk = GenericURDFFk(f.read(), "panda_link0", "panda_leftfinger")
    fk.set_mount_transformation(np.asarray(mount, dtype=float))
    planner = ParameterizedFabricPlanner(7, fk, **dict(_config.PANDA_STRINGS, **strings))
    if device is not None:
        planner.device = device
    names = ["panda_hand" if int(l) > 8 else "panda_link%d" % int(l) for l in link_numbers]
    planner.set_components(collision_links=names, goal=goal, number_obstacles=int(n_static), number_dynamic_obstacles=int(n_dynamic),
                           dynamic_obstacle_dimension=3, number_plane_constraints=1, limits=_config.PANDA_LIMITS)
    planner.concretize(mode="vel", time_step=dt)
    return planner


def point_planner(urdf_path, goal, n_static, n_dynamic=0, dynamic_dimension=3, device=None, **strings):
    """One concretized point-robot planner as the point-mass examples configure it: x-y-heading chain read from `urdf_path`,
    collision leaves on `base_link`, acceleration output.  Leaf strings default to those examples' (config.POINT_STRINGS)."""
    with open(urdf_path, "r") as f:
        fk = GenericURDFFk(f.read(), "world", "base_link")
    planner = ParameterizedFabricPlanner(3, fk, **dict(_config.POINT_STRINGS, **strings))
    if device is not None:
        planner.device = device
    planner.set_components(collision_links=["base_link"], goal=goal, number_obstacles=int(n_static),
                           number_dynamic_obstacles=int(n_dynamic), dynamic_obstacle_dimension=int(dynamic_dimension))
    planner.concretize()
    return planner
