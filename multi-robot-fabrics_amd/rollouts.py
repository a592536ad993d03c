"""Rollout Fabrics with the reference's two class surfaces, evaluated by the HIP rollout kernels.

    ForwardFabricsPlanner   multi_robot_fabrics/fabrics_planner/forward_planner_Jointspace.py:12-555
                            (joint-space coupling: every robot's fabric sees the other robots' predicted spheres)
    FabricsRollouts         multi_robot_fabrics/fabrics_planner/forward_planner_Cartesian.py:8-563
                            (per-robot rollout, obstacles extrapolated at constant Cartesian velocity)

The reference unrolls the horizon into one CasADi graph at construction time
(`forward_multi_fabrics_symbolic`, `symbolic_forward_fabrics`); here those calls create the device handle, and
the per-step entry points launch one persistent kernel.  The *_batch methods are the same entry points for
many scenarios at once (device tensors in, device tensors out).
"""
import numpy as np
import torch

from . import abi
from . import config as _config
from .kinematics import DM


def _scalar(v):
    return float(np.asarray(v, dtype=float).reshape(-1)[0])


def _params_row(angle, constraint, weights, goals, radius_bodies):
    p = np.zeros(abi.NPARAM)
    g0 = np.asarray(goals[0], dtype=float).reshape(-1)
    p[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + 3] = g0[:3]
    p[abi.P_WEIGHT_GOAL_0] = _scalar(weights[0])
    if len(goals) > 1:
        p[abi.P_ANGLE_GOAL_1:abi.P_ANGLE_GOAL_1 + 9] = np.asarray(angle, dtype=float).reshape(9)
        p[abi.P_X_GOAL_1:abi.P_X_GOAL_1 + 3] = np.asarray(goals[1], dtype=float).reshape(-1)[:3]
        p[abi.P_WEIGHT_GOAL_1] = _scalar(weights[1])
    if len(goals) > 2:
        p[abi.P_X_GOAL_2] = _scalar(goals[2])
        p[abi.P_WEIGHT_GOAL_2] = _scalar(weights[2])
    if constraint is not None:
        p[abi.P_CONSTRAINT_0:abi.P_CONSTRAINT_0 + 4] = np.asarray(constraint, dtype=float).reshape(4)
    rb = [_scalar(r) for r in radius_bodies]
    p[abi.P_RADIUS_BODY:abi.P_RADIUS_BODY + len(rb)] = rb
    return p


class ForwardFabricsPlanner:
    """Joint-space Rollout Fabrics (FPJ:12-336).  `planners` are this package's ParameterizedFabricPlanner objects
    built by set_planner_panda-style code; their mounts, leaf strings and limits define the rollout handle."""

    def __init__(self, params, planners, N_steps, fk_dict, goal_struct_robots, ROLLOUTS_PLOTTING=0,
                 goal_estimate_mask=None):
        self.ROLLOUTS_PLOTTING = ROLLOUTS_PLOTTING
        self.N_horizon = params.N_HORIZON
        self.dt = params.dt
        self.dof = params.dof
        self.nr_robots = len(self.dof)
        self.planners = planners
        self.N_steps = N_steps
        self.fk_dict = fk_dict
        self.collision_links_nrs = params.collision_links_nrs
        self.goal_struct_robots = goal_struct_robots
        self.fabrics_mode = params.fabrics_mode
        self.other_robot_static_dynamic = params.STATIC_OR_DYN_FABRICS
        self.r_robots = params.r_robots
        self.nr_constraints = params.nr_constraints
        self.rotation_matrices_pandas = params.rotation_matrix_pandas
        self.nr_subgoals = [len(g._config) for g in goal_struct_robots]
        # radius bodies of the active ego links only (FPJ:37-40)
        self.r_robots_args = [[r for z, r in zip(self.collision_links_nrs[i], self.r_robots[i]) if z > 2]
                              for i in range(self.nr_robots)]
        # RF-CV: which robots roll out towards an estimated goal (the reference rewrites robot 1's goal on the
        # host, EXJ:346-348; with a mask the estimate x_ee + 0.2 v_ee is taken on the device instead)
        self.goal_estimate_mask = 0 if goal_estimate_mask is None else int(goal_estimate_mask)
        self._handle = None

    # -- construction ------------------------------------------------------------------------------------------
    @classmethod
    def for_cell(cls, params, planner_of, goals, n_steps=100, fk_dict=None, **kw):
        """Rollout object of a whole cell, ready to use: planner_of(i) -> the rollout planner of robot i."""
        rollout = cls(params, [planner_of(i) for i in range(params.nr_robots)], n_steps, fk_dict, goals, **kw)
        rollout.forward_multi_fabrics_symbolic()
        return rollout

    def forward_multi_fabrics_symbolic(self):
        if self.fabrics_mode != "vel":
            raise NotImplementedError("the joint-space rollout is defined for fabrics_mode 'vel' only (FPJ:233)")
        N = self.nr_robots
        p0 = self.planners[0]
        if any(p._strings != p0._strings for p in self.planners):
            raise NotImplementedError("all robots of one rollout share their leaf strings")
        cfg = _config.panda_config(n_robots=N, horizon=self.N_horizon, dynamic=1 if self.other_robot_static_dynamic else 0,
                                   mounts=[p._forward_kinematics.mount for p in self.planners], **p0._strings)
        comp = p0._components
        cfg.n_ego, cfg.n_planes, cfg.n_goals = comp["n_ego"], comp["n_planes"], comp["n_goals"]
        if any(p._ego_links != p0._ego_links for p in self.planners):
            raise NotImplementedError("all robots of one rollout share their collision-link set")
        if comp["n_ego"]:
            cfg.ego_link_mask = _config.ego_link_mask(p0._ego_links)
        cfg.use_limits = 0 if comp["limits"] is None else 1
        if comp["limits"] is not None:
            for j in range(7):
                cfg.limits[j][0], cfg.limits[j][1] = comp["limits"][j]
        cfg.dt = self.dt
        links = list(self.collision_links_nrs[0])
        if any(list(c) != links for c in self.collision_links_nrs):
            raise NotImplementedError("all robots of one rollout share their sphere table")
        _config.set_spheres(cfg, links, None, self.r_robots[0])
        cfg.goal_estimate_mask = self.goal_estimate_mask
        for key, val in p0.constants.items():
            setattr(cfg, key, val)
        from .runtime import FabricHandle
        self.config = cfg
        self._handle = FabricHandle(cfg, getattr(self, "device", None))
        return {}

    # -- argument marshalling (FPJ:303-329) ------------------------------------------------------------------------
    def _rows(self, inputs_action):
        N = self.nr_robots
        q = np.zeros((7, N))
        qd = np.zeros((7, N))
        prm = np.zeros((abi.NPARAM, N))
        for i in range(N):
            q[:, i] = np.asarray(inputs_action["q_robots"][i], dtype=float).reshape(-1)[:7]
            qd[:, i] = np.asarray(inputs_action["q_dot_robots"][i], dtype=float).reshape(-1)[:7]
            ns = self.nr_subgoals[i]
            weights = [inputs_action["weight_goals" + str(g)][i] for g in range(ns)]
            goals = [inputs_action["x_goals" + str(g)][i] for g in range(ns)]
            con = inputs_action["constraints"][i] if self.nr_constraints[i] else None
            prm[:, i] = _params_row(self.rotation_matrices_pandas[i], con, weights, goals, self.r_robots_args[i])
        return q, qd, prm

    def _run(self, inputs_action, traj):
        if self._handle is None:
            raise RuntimeError("call forward_multi_fabrics_symbolic() first")
        q, qd, prm = self._rows(inputs_action)
        return self._handle.rollout_host(q, qd, prm, want_traj=traj)      # numpy in, numpy out (mrf_rollout_host)

    # -- the reference's entry points --------------------------------------------------------------------------------
    def get_velocity_rollouts(self, inputs_action):
        """-> list over robots of np.ndarray shape (1,): mean squared joint velocity over the horizon (FPJ:298-336)."""
        avg = self._run(inputs_action, traj=False)
        return [avg[i:i + 1].copy() for i in range(self.nr_robots)]

    def rollouts_numerical(self, inputs_action=None, **_ignored):
        """-> (q_N, q_dot_N, q_ddot_N): dicts 'robot_i' -> [ndarray(7, H)] (FPJ:338-423; q_ddot is zero, FPJ:202)."""
        avg, tq, tqd = self._run(inputs_action, traj=True)     # [H, 7, N]
        qN, qdN, qddN = {}, {}, {}
        for i in range(self.nr_robots):
            qN["robot_%d" % i] = [tq[:, :, i].T.copy()]
            qdN["robot_%d" % i] = [tqd[:, :, i].T.copy()]
            qddN["robot_%d" % i] = [np.zeros_like(tq[:, :, i].T)]
        return qN, qdN, qddN

    def rollouts_numerical_obstacles(self, inputs_action):
        """-> (x_N, v_N, a_N): dicts 'robot_i' -> list over the H steps of ndarray(3, 8*(N-1)): the spheres of the
        other robots as robot i sees them at step k (FPJ:425-512 evaluating the graph outputs of FPJ:211-225,250-253):
        positions after the step's position update, v = J qdot and a = jdot_sign * Jdot qdot with the velocities the
        robots had BEFORE this step's actions; zeros for static fabrics (FPJ:215-217)."""
        h = self._handle
        if h is None:
            raise RuntimeError("call forward_multi_fabrics_symbolic() first")
        q, qd, prm = h.upload(*self._rows(inputs_action))
        _, tq, tqd = h.rollout(q, qd, prm, want_traj=True)                      # [H, 7, N]
        H, N, S = self.N_horizon, self.nr_robots, h.cfg.n_spheres
        # [H, S, 3, N] -> [S, 3, H, N]: what every robot publishes at step k (mrf_rollout_sphere_traj)
        x, v, a = (t.permute(1, 2, 0, 3).cpu().numpy() for t in h.rollout_sphere_traj(qd, tq, tqd))
        dyn = bool(h.cfg.dynamic)
        xN, vN, aN = ({"robot_%d" % i: [] for i in range(N)} for _ in range(3))
        for i in range(N):
            others = [j for j in range(N) if j != i]
            for k in range(H):
                xs = np.concatenate([x[:, :, k, j].T for j in others], axis=1)   # (3, S*(N-1))
                xN["robot_%d" % i].append(xs)
                vN["robot_%d" % i].append(np.concatenate([v[:, :, k, j].T for j in others], axis=1) if dyn else np.zeros_like(xs))
                aN["robot_%d" % i].append(np.concatenate([a[:, :, k, j].T for j in others], axis=1) if dyn else np.zeros_like(xs))
        return xN, vN, aN

    def compute_velocity_average(self, q_dot_robots_N):
        """FPJ:102-116 on numeric trajectories: sum of squares / (H * dof) per robot."""
        out = []
        for i in range(self.nr_robots):
            traj = q_dot_robots_N["robot_%d" % i]
            arr = np.asarray(traj[0] if isinstance(traj, list) and np.ndim(traj[0]) == 2 else traj, dtype=float)
            out.append(float((arr ** 2).sum() / (self.N_horizon * self.dof[i])))
        return out

    def system_step(self, pos, vel, action, i_robot, dt):
        """FPJ:72-80."""
        if self.fabrics_mode == "acc":
            return pos + dt * vel + 0.5 * dt ** 2 * action, vel + dt * action
        return pos + dt * action, action

    # -- batched form --------------------------------------------------------------------------------------------------
    def get_velocity_rollouts_batch(self, q, qdot, params, want_traj=False, stream=None):
        """Device tensors [7, B*N], [7, B*N], [29, B*N] (row = scenario*N + robot) -> avg_vel [B*N] (, traj_q, traj_qdot)."""
        return self._handle.rollout(q, qdot, params, want_traj=want_traj, stream=stream)


class FabricsRollouts:
    """Cartesian constant-velocity Rollout Fabrics (FPC:8-563), one instance per robot as in EXC:172-192."""

    def __init__(self, N, dt, nx, nu, dof, nr_obsts, bool_ring, nr_obsts_dyn=0, v_obsts_dyn=(), fabrics_mode="acc",
                 collision_links_nrs=(7,), nr_constraints=0, radius_sphere=0.08, constraints=None, nr_goals=3):
        self.N = N
        self.dt = dt
        self.Ts = dt
        self.nx, self.nu, self.dof = nx, nu, dof
        self.ring = bool_ring
        self.nr_obsts = nr_obsts
        self.nr_goals = nr_goals
        self.nr_obsts_dyn = nr_obsts_dyn
        self.v_obsts_dyn = list(v_obsts_dyn)
        self.a_obsts_dyn = [np.zeros((3,))] * len(self.v_obsts_dyn)
        self.fabrics_mode = fabrics_mode
        self.collision_links_nrs = list(collision_links_nrs)
        self.nr_constraints = nr_constraints
        self.radius_sphere = radius_sphere
        self.rotation_matrix_panda = np.array([[0, 0, -1], [0, 1, 0], [1, 0, 0]])
        self.radius_obsts_dyn = []
        self.radius_obsts = []
        self.constraints = constraints
        self.radius_body_panda_links = {str(l): np.array(radius_sphere) for l in self.collision_links_nrs if l > 2}
        self._handle = None

    @classmethod
    def for_robot(cls, params, i_robot, planner, goal):
        """The rollout object of robot i as the Cartesian driver sets it up (EXC:172-192): horizon params.N_HORIZON, all
        spheres of the other robots as constant-velocity obstacles, bound to `planner`."""
        dof = params.dof[i_robot]
        n_dyn = params.nr_obsts_dyn_all[i_robot]
        r = cls(params.N_HORIZON, params.dt, 2 * dof, dof, dof, params.nr_obsts[i_robot], False, nr_obsts_dyn=n_dyn,
                v_obsts_dyn=[np.zeros(3)] * n_dyn, fabrics_mode=params.fabrics_mode,
                collision_links_nrs=params.collision_links_nrs[i_robot], nr_constraints=params.nr_constraints[i_robot],
                radius_sphere=params.radius_sphere, constraints=params.constraints[i_robot], nr_goals=len(goal.sub_goals()))
        r.symbolic_forward_fabrics(planner, goal)
        r.preset_radii_obsts_dyn(params.r_dyns_obsts[i_robot])
        return r

    def preset_radii_obsts_dyn(self, radii_obst_dyn):
        self.radius_obsts_dyn = radii_obst_dyn

    def reset_v_obsts_dyn(self, v_obsts_dyn):
        self.v_obsts_dyn = v_obsts_dyn

    def system_step(self, pos, vel, input, dt, fabrics_mode="vel"):
        """FPC:77-92."""
        if fabrics_mode == "acc":
            return pos + dt * vel + 0.5 * dt ** 2 * input, vel + dt * input
        if fabrics_mode == "vel":
            return pos + dt * input, input
        raise ValueError("nonexisting fabrics mode inserted, should be vel or acc")

    def symbolic_forward_fabrics(self, planner, goal_struct):
        """Binds the rollout to `planner`'s configuration with horizon N (FPC:347-489)."""
        self.nr_subgoals = len(goal_struct._config)
        cfg = planner.config.copy()
        cfg.horizon = self.N
        cfg.dt = self.dt
        cfg.mode = abi.MODE_VEL if self.fabrics_mode == "vel" else abi.MODE_ACC
        from .runtime import FabricHandle
        self.config = cfg
        self._planner = planner
        self._handle = FabricHandle(cfg, getattr(self, "device", None))
        return {}

    def define_arguments_numerical(self, q_robot, q_dot_robot, weight_goals, x_goals, x_obsts, x_obsts_dyn, v_obsts_dyn,
                                   constraints=()):
        """Same positional list as FPC:507-536 (angle, constraints, q, q_dot, weights, goals, static x/r, body radii,
        dynamic radii, dynamic x, v, a)."""
        arguments = []
        if self.nr_subgoals > 1:
            arguments.append(self.rotation_matrix_panda)
        for _ in range(self.nr_constraints):
            arguments.append(constraints)
        arguments.append(q_robot)
        arguments.append(q_dot_robot)
        for g in range(self.nr_subgoals):
            arguments.append(weight_goals["subgoal" + str(g)])
        for g in range(self.nr_subgoals):
            arguments.append(x_goals["subgoal" + str(g)])
        for j in range(self.nr_obsts):
            arguments.append(x_obsts[j])
        for j in range(self.nr_obsts):
            arguments.append(self.radius_obsts[j])
        if self.nr_obsts + self.nr_obsts_dyn > 0:
            arguments.extend(self.radius_body_panda_links.values())
        arguments.extend(self.radius_obsts_dyn)
        for j in range(self.nr_obsts_dyn):
            arguments.append(x_obsts_dyn[j])
        for j in range(self.nr_obsts_dyn):
            arguments.append(v_obsts_dyn[j])
        for j in range(self.nr_obsts_dyn):
            arguments.append(self.a_obsts_dyn[j] if j < len(self.a_obsts_dyn) else np.zeros(3))
        self.arguments = arguments
        return arguments

    def _unpack(self, arguments):
        a = list(arguments)
        k = 0
        angle = None
        if self.nr_subgoals > 1:
            angle = a[k]; k += 1
        con = None
        for _ in range(self.nr_constraints):
            con = a[k]; k += 1
        q, qd = a[k], a[k + 1]; k += 2
        weights = a[k:k + self.nr_subgoals]; k += self.nr_subgoals
        goals = a[k:k + self.nr_subgoals]; k += self.nr_subgoals
        xs = a[k:k + self.nr_obsts]; k += self.nr_obsts
        rs = a[k:k + self.nr_obsts]; k += self.nr_obsts
        nb = len(self.radius_body_panda_links) if self.nr_obsts + self.nr_obsts_dyn > 0 else 0
        rb = a[k:k + nb]; k += nb
        nd = self.nr_obsts_dyn
        rd = a[k:k + nd]; k += nd
        xd = a[k:k + nd]; k += nd
        vd = a[k:k + nd]; k += nd
        ad = a[k:k + nd]; k += nd
        if k != len(a):
            raise TypeError(f"expected {k} rollout arguments, got {len(a)}")
        if len(rb) not in (0, 6):
            raise NotImplementedError("the kernels carry ego leaves on links 3..8 (all) or none")
        prm = _params_row(angle, con, weights, goals, rb if rb else [self.radius_sphere] * 6)
        M = self.nr_obsts + nd
        ox, ov, oa, orad = np.zeros((M, 3)), np.zeros((M, 3)), np.zeros((M, 3)), np.zeros(M)
        for j in range(self.nr_obsts):
            ox[j] = np.asarray(xs[j], dtype=float).reshape(-1)[:3]
            orad[j] = _scalar(rs[j])
        for j in range(nd):
            r = self.nr_obsts + j
            ox[r] = np.asarray(xd[j], dtype=float).reshape(-1)[:3]
            ov[r] = np.asarray(vd[j], dtype=float).reshape(-1)[:3]
            oa[r] = np.asarray(ad[j], dtype=float).reshape(-1)[:3]
            orad[r] = _scalar(rd[j])
        return (np.asarray(q, dtype=float).reshape(-1)[:7], np.asarray(qd, dtype=float).reshape(-1)[:7], prm, ox, ov, oa, orad)

    def _run(self, arguments, traj):
        if self._handle is None:
            raise RuntimeError("call symbolic_forward_fabrics() first")
        h = self._handle
        q, qd, prm, ox, ov, oa, orad = self._unpack(arguments)
        if ox.shape[0]:
            # FPC:33 presets the obstacle accelerations to zero and the drivers never change them: obst_a = NULL selects the
            # rollout kernel that neither loads nor folds them and keeps 10 instead of 7 obstacles resident on chip
            obst = (ox[:, :, None], ov[:, :, None], oa[:, :, None] if oa.any() else None, orad[:, None])
        else:
            obst = (None, None, None, None)
        return h.rollout_cartesian_host(q[:, None], qd[:, None], prm[:, None], *obst, want_traj=traj, n_static=self.nr_obsts)

    def get_velocity_rollouts(self, arguments):
        """-> DM-like array of shape (1,) whose .full() is (1,1), as `avg_vel_fun(*arguments)` (FPC:561-563)."""
        return DM(self._run(arguments, traj=False))

    def rollouts_numerical(self, arguments):
        """-> q_N, q_dot_N, q_ddot_N each ndarray (7, H) (FPC:538-559; q_ddot is zero in mode 'vel', FPC:431)."""
        avg, tq, tqd = self._run(arguments, traj=True)
        tq, tqd = tq[:, :, 0].T.copy(), tqd[:, :, 0].T.copy()
        if self.fabrics_mode == "vel":
            return tq, tqd, np.zeros_like(tq)
        qdd = np.diff(np.concatenate([np.asarray(self._unpack(arguments)[1])[:, None], tqd], axis=1), axis=1) / self.dt
        return tq, tqd, qdd

    # -- the numeric twin of the rollout (FPC:132-273): one planner.compute_action per horizon step -------------------
    def get_action(self, planner, pos, vel, x_obsts, x_obsts_dyn, x_goals, weight_goals):
        """planner.compute_action with this rollout's constants filled in (FPC:132-191; the `ring` and the plain
        branch pass the same arguments)."""
        x_goals, weight_goals = list(x_goals), list(weight_goals)
        for _ in range(3 - self.nr_goals):        # FPC:143-147: pad to three sub-goals
            x_goals.append(0)
            weight_goals.append(0)
        return planner.compute_action(
            q=pos, qdot=vel, x_obsts=x_obsts, radius_obsts=self.radius_obsts, angle_goal_1=self.rotation_matrix_panda,
            x_goal_0=x_goals[0], x_goal_1=x_goals[1], x_goal_2=x_goals[2], weight_goal_0=weight_goals[0],
            weight_goal_1=weight_goals[1], weight_goal_2=weight_goals[2], x_obsts_dynamic=x_obsts_dyn,
            xdot_obsts_dynamic=self.v_obsts_dyn, xddot_obsts_dynamic=[np.array([0.0, 0.0, 0.0])] * self.nr_obsts_dyn,
            radius_obsts_dynamic=self.radius_obsts_dyn, radius_body_panda_links=self.radius_body_panda_links,
            radius_body_panda_hand=np.array([0.08]), constraint_0=self.constraints)

    def get_x_obsts_dyn_N(self, x_obsts_dyn):
        """FPC:195-216: obstacle positions at k = 0..N as (3, n) arrays, and per step k = 0..N-1 as lists of 3-vectors."""
        x0 = np.stack([np.asarray(p, dtype=float).reshape(-1)[:3] for p in x_obsts_dyn]).transpose()
        v = np.stack([np.asarray(p, dtype=float).reshape(-1)[:3] for p in self.v_obsts_dyn]).transpose()
        arrays = [x0 + k * self.Ts * v for k in range(self.N + 1)]
        lists = [[arrays[k][:, j].copy() for j in range(self.nr_obsts_dyn)] for k in range(self.N)]
        return arrays, lists

    def forward_fabrics(self, planner, pos_k, vel_k, ob_robot=None, goal=None, x_obsts_dyn_0=None, x_goals_struct=None,
                        weight_goals_struct=None, x_obsts=()):
        """FPC:218-273: numeric rollout, action-then-step, obstacles at constant velocity.  Returns lists over the
        horizon of q, q_dot, q_ddot (the last one empty in mode 'vel').  Goals and obstacle positions must be passed
        explicitly (the reference can also pull them out of the simulator's observation `ob_robot`)."""
        if x_goals_struct is None or weight_goals_struct is None or (self.nr_obsts_dyn > 0 and x_obsts_dyn_0 is None):
            raise NotImplementedError("pass x_obsts_dyn_0, x_goals_struct and weight_goals_struct (no simulator observation here)")
        steps = self.get_x_obsts_dyn_N(x_obsts_dyn_0)[1] if self.nr_obsts_dyn > 0 else [[] for _ in range(self.N)]
        pos_k, vel_k = np.asarray(pos_k, dtype=float).copy(), np.asarray(vel_k, dtype=float).copy()
        q_stacked, qdot_stacked, qddot_stacked = [], [], []
        for k in range(self.N):
            u_k = self.get_action(planner, pos_k, vel_k, x_obsts=list(x_obsts), x_obsts_dyn=steps[k],
                                  x_goals=list(x_goals_struct.values()), weight_goals=list(weight_goals_struct.values()))
            pos_k, vel_k = self.system_step(pos_k, vel_k, u_k, dt=self.dt, fabrics_mode=self.fabrics_mode)
            if self.fabrics_mode == "acc":
                qddot_stacked.append(np.array(u_k, copy=True))
                qdot_stacked.append(vel_k.copy())
            else:
                qdot_stacked.append(np.array(u_k, copy=True))
            q_stacked.append(pos_k.copy())
        return q_stacked, qdot_stacked, qddot_stacked

    def x_obsts_dyn_numerical(self, pos_obsts_dyn):
        """Obstacle positions after each step, x += dt*v (FPC:448-453,491-505): list over k of (3, n_obst) arrays."""
        x = np.stack([np.asarray(p, dtype=float).reshape(-1)[:3] for p in pos_obsts_dyn]) if len(pos_obsts_dyn) else np.zeros((0, 3))
        v = np.stack([np.asarray(p, dtype=float).reshape(-1)[:3] for p in self.v_obsts_dyn]) if len(pos_obsts_dyn) else np.zeros((0, 3))
        out = []
        for _ in range(self.N):
            x = x + self.dt * v
            out.append(x.T.copy())
        return out

    def compute_velocity_average(self, q_dot_N):
        """FPC:276-288 on a numeric (7, H) trajectory."""
        arr = np.asarray(q_dot_N, dtype=float)
        return [float((arr ** 2).sum() / (self.N * self.dof))]

    def get_velocity_rollouts_batch(self, q, qdot, params, obst_x0, obst_v, obst_a, obst_r, want_traj=False, stream=None):
        return self._handle.rollout_cartesian(q, qdot, params, obst_x0, obst_v, obst_a, obst_r, want_traj=want_traj,
                                              n_static=self.nr_obsts, stream=stream)
