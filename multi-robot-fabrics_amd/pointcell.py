"""The point-robot arena of the reference's two point-mass examples, stepped as a batch on the device.

Four (or any number of) holonomic point robots cross an arena with fixed spheres in it; every robot's fabric sees the
spheres and the other robots -- as more static spheres (example_pointmasses_static.py) or as moving ones described by
position and velocity in the plane (example_pointmasses_dynamic.py).  The reference asks one robot at a time
(`planner.compute_action(**kwargs)` per robot per step) and lets a gym environment integrate the accelerations.  Here a
control step of ALL robots of ALL `scenes` copies of the arena is one `mrf_compute_action` launch on rows = scenes x
robots (the other robots' states are gathered into each row's obstacle list on the device), followed by the integration
of the commanded accelerations (urdfenvs mode 'acc': qdot += dt a, q += dt qdot) and the clearance bookkeeping as device
tensor arithmetic.  There is no contact physics: a fabric's barrier is a geometry, not a hard constraint, and a robot
that creeps against a sphere (xdot -> 0 makes the repulsion -2/x xdot^2 vanish) can end up inside it; the run reports
the first step at which that happens instead of stopping the robot there as a simulator would.
"""
import numpy as np
import torch

from . import abi
from .runtime import MrfError


class PointRobotArena:
    def __init__(self, planner, starts, goals, spheres, sphere_radii, robot_radius=0.2, goal_weight=1.0, dt=0.01, scenes=1,
                 start_jitter=0.0, rng=None):
        """planner: a concretized point-robot ParameterizedFabricPlanner (its obstacle counts decide how the other robots
        are seen: all static = spheres + others static; dynamic = the others as moving obstacles of dimension 2);
        starts [R,3] (x, y, heading), goals [R,2], spheres [K,3], sphere_radii [K]."""
        h = planner._handle
        if h is None or planner._model == "panda":
            raise MrfError("a concretized point-robot planner is required")
        comp = planner._components
        starts, goals = np.asarray(starts, dtype=float), np.asarray(goals, dtype=float)
        spheres, sphere_radii = np.asarray(spheres, dtype=float), np.asarray(sphere_radii, dtype=float)
        R, K = len(starts), len(spheres)
        self.h, self.R, self.K, self.scenes, self.dt = h, R, K, int(scenes), float(dt)
        self.n_static, self.n_dynamic = comp["n_static"], comp["n_dynamic"]
        if (self.n_static, self.n_dynamic) not in ((K + R - 1, 0), (K, R - 1)):
            raise MrfError(f"the planner sees {self.n_static} static + {self.n_dynamic} dynamic obstacles; this arena offers "
                           f"{K} spheres + {R - 1} other robots (all static, or the others dynamic)")
        rows = self.scenes * R
        t = h.tensor
        q0 = np.tile(starts.T, (1, self.scenes))                                   # [3, rows], row = scene*R + robot
        if start_jitter:
            q0[:2] += (np.random.default_rng() if rng is None else rng).uniform(-start_jitter, start_jitter, (2, rows))
        self.q, self.qd = t(q0), t(np.zeros_like(q0))
        prm = np.zeros((abi.NPARAM, rows))
        prm[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + 2] = np.tile(goals.T, (1, self.scenes))
        prm[abi.P_WEIGHT_GOAL_0] = goal_weight
        prm[abi.P_RADIUS_BODY] = robot_radius
        self.prm, self.goal = t(prm), t(np.tile(goals.T, (1, self.scenes)))
        self.robot_radius = float(robot_radius)
        M = K + R - 1
        ox = np.zeros((M, 3, rows))
        ox[:K] = spheres[:, :, None]
        orad = np.zeros((M, rows))
        orad[:K] = sphere_radii[:, None]
        orad[K:] = robot_radius
        self.ox, self.ov, self.orad = t(ox), t(np.zeros_like(ox)), t(orad)
        self.sphere_xy, self.sphere_r = t(spheres[:, :2]), t(sphere_radii)
        # row (scene, i) takes its k-th "other robot" from row (scene, others[i][k])
        others = np.array([[j for j in range(R) if j != i] for i in range(R)])        # [R, R-1]
        src = (np.arange(self.scenes)[:, None, None] * R + others[None]).reshape(rows, R - 1)
        self.src = torch.as_tensor(src.T.copy(), device=h.device)                    # [R-1, rows]

    def step(self):
        h, K = self.h, self.K
        self.ox[K:, :2] = self.q[:2][:, self.src].permute(1, 0, 2)                  # the others' positions, heading dropped
        self.ox[K:, 2] = 0.0
        if self.n_dynamic:
            self.ov[K:, :2] = self.qd[:2][:, self.src].permute(1, 0, 2)
        acc = h.compute_action(self.q, self.qd, self.prm, self.ox, self.ov if self.n_dynamic else None, None, self.orad,
                               n_static=self.n_static)
        self.qd += self.dt * acc
        self.q += self.dt * self.qd

    def clearance(self):
        """[scenes]: smallest surface distance robot-robot and robot-sphere in every scene, now."""
        xy = self.q[:2].view(2, self.scenes, self.R)
        d_rr = (xy[:, :, :, None] - xy[:, :, None, :]).norm(dim=0) - 2 * self.robot_radius
        d_rr = d_rr + torch.eye(self.R, dtype=d_rr.dtype, device=d_rr.device)[None] * 1e9
        d_rs = (xy[:, :, :, None] - self.sphere_xy.t()[:, None, None, :]).norm(dim=0) - self.sphere_r - self.robot_radius
        return torch.minimum(d_rr.flatten(1).min(dim=1).values, d_rs.flatten(1).min(dim=1).values)

    def run(self, n_steps):
        """-> dict for scene 0 (+ 'all_scenes' when there are several): distances to the goals, smallest clearance over
        the run, first step with a negative clearance (None: never), final speeds."""
        low = torch.full((self.scenes,), float("inf"), dtype=self.q.dtype, device=self.q.device)
        first = torch.full((self.scenes,), -1, dtype=torch.int64, device=self.q.device)
        for k in range(int(n_steps)):
            self.step()
            c = self.clearance()
            low = torch.minimum(low, c)
            first = torch.where((c < 0) & (first < 0), torch.full_like(first, k), first)
        dist = (self.q[:2] - self.goal).norm(dim=0).view(self.scenes, self.R).cpu().numpy()
        speed = self.qd.norm(dim=0).view(self.scenes, self.R).cpu().numpy()
        low, first = low.cpu().numpy(), first.cpu().numpy()
        out = {"steps": int(n_steps), "distance_to_goal_m": [float(v) for v in dist[0]], "min_clearance_m": float(low[0]),
               "first_contact_step": int(first[0]) if first[0] >= 0 else None, "final_speed": [float(v) for v in speed[0]]}
        if self.scenes > 1:
            out["all_scenes"] = {"min_clearance_m": low.tolist(), "mean_distance_to_goal_m": dist.mean(axis=1).tolist()}
        return out
