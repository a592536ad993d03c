"""The Panda pick-and-place cell of the reference's two manipulator examples as ONE device-resident loop.

The reference steps its cell from Python: per control step it reads the simulator, runs a state machine per robot,
evaluates the Rollout Fabrics, the deadlock logic and one `compute_action` per robot, and steps the simulator
(examples/example_pandas_Jointspace.py:280-458, example_pandas_cartesian.py:290-470).  This build keeps all of that on
the GPU: `PandaCell` owns the state of `scenes` independent copies of the cell (N robots each) and advances them with
runtime.ControlLoop -- one replayed HIP graph per control step (mrf_episode_run): hand FK, pick-and-place state machine,
RF-CV goal estimate, Rollout Fabrics (coupled joint-space or per-robot Cartesian), deadlock logic, compute_action of the
main and of the grasp planner, gripper command, velocity integration -- and the step's own record (joint positions,
state-machine states, who is done since when, wall-clock stamps at its first and after its last kernel;
mrf_episode_set_recorder).  Steps are queued a chunk at a time with no host work in between; the host reads the record
after a chunk.

What stands in for the simulator: joints integrate their velocity command exactly (urdfenvs 'vel'
mode), finger joints likewise between their stops, a cube travels with a closed gripper (mrf_state_machine_config.model
= 1).  There is no renderer.
"""
import math
import time

import numpy as np
import torch

from . import abi
from . import config as _config
from .runtime import ControlLoop, FabricHandle, MrfError

DONE = 10                    # state-machine state "all blocks delivered" (SM:150-199)
CUBE_HALF = 0.025            # cubes are 0.05 m boxes resting on the table (SIM:78-112)
HAND_ABOVE_CUBE = 0.1        # the hand is sent 0.1 m above the cube's centre (EXJ:297, EXC:309)


# ------------------------------------------------------------------------------------------------ scene data
def cube_layout(params, random_scene=False, n_cubes=None, rng=None, scenes=1):
    """Cube centres on the table, [scenes, n_cubes, 3]; cubes [i*k, (i+1)*k) are robot i's (k = n_cubes / N, EXJ:294-297).

    Fixed scene: two columns x = 0.4 / 0.6 between the mounts, y in {0, -0.15, +0.15} (+0.2 for three robots), the
    reference's assignment of places to robots (SIM:37-60).  Random scene: uniform in that 0.2 x 0.3 m strip, no two cubes
    closer than edge + 0.06 m (SIM:62-76), drawn per scene with `rng` (numpy Generator)."""
    N = params.nr_robots
    n_cubes = params.n_cubes if n_cubes is None else int(n_cubes)
    z = params.z_table + CUBE_HALF
    shift = 0.2 if N == 3 else 0.0
    out = np.zeros((scenes, n_cubes, 3))
    out[:, :, 2] = z
    if not random_scene:
        near = [(0.4, 0.0), (0.4, -0.15), (0.4, 0.15)]
        far = [(0.6, -0.15), (0.6, 0.0), (0.6, 0.15)]
        if N == 2:
            places = [near, far]
        elif N == 3:
            places = [near[:2], far, [near[2], far[2]]]
        else:       # no reference layout beyond three robots: three cubes 0.45 m in front of every mount
            places = []
            for T in params.mount_transform:
                T = np.asarray(T, dtype=float)
                places.append([tuple((T[:3, 3] + T[:3, :3] @ np.array([0.45, dy, 0.0]))[:2]) for dy in (0.0, -0.15, 0.15)])
        per = n_cubes // N
        flat = [places[i][k] for i in range(N) for k in range(per)]
        for c, (x, y) in enumerate(flat):
            out[:, c, 0], out[:, c, 1] = x, y + (shift if N <= 3 else 0.0)
        return out
    rng = np.random.default_rng() if rng is None else rng
    min_gap = 2 * CUBE_HALF + 0.06
    for s in range(scenes):
        placed = []
        misses = 0
        while len(placed) < n_cubes:
            cand = np.array([rng.uniform(0.4, 0.6), rng.uniform(-0.15, 0.15) + shift])
            if all(np.hypot(*(cand - p)) > min_gap for p in placed):
                placed.append(cand)
            else:
                misses += 1
                if misses > 200:       # a jammed draw (six cubes barely fit the strip): start the scene over
                    placed, misses = [], 0
        out[s, :, :2] = np.array(placed)
    return out


def nominal_parameters(params, scenes=1):
    """[29, scenes*N] runtime parameters of the example planners (EXJ:421-429): start goal, weights 2 / 20 / 1, hand
    orientation target, joint-7 target pi/4, table plane, body radii."""
    N = params.nr_robots
    prm = np.zeros((abi.NPARAM, scenes * N))
    for i in range(N):
        prm[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + 3, i::N] = np.asarray(params.start_goals[i], dtype=float)[:, None]
        prm[abi.P_ANGLE_GOAL_1:abi.P_ANGLE_GOAL_1 + 9, i::N] = np.asarray(params.rotation_matrix_pandas[i], dtype=float).reshape(9, 1)
    prm[abi.P_WEIGHT_GOAL_0], prm[abi.P_WEIGHT_GOAL_1], prm[abi.P_WEIGHT_GOAL_2] = 2.0, 20.0, 1.0
    prm[abi.P_X_GOAL_1] = 0.107
    prm[abi.P_X_GOAL_2] = math.pi / 4
    prm[abi.P_CONSTRAINT_0 + 2], prm[abi.P_CONSTRAINT_0 + 3] = 1.0, -params.z_table
    prm[abi.P_RADIUS_BODY:abi.P_RADIUS_BODY + 6] = params.radius_sphere
    return prm


def _coupled_config(params, template, horizon, dynamic, n_ego=None, sphere_table=True):
    """mrf_config of all N robots of the cell from one robot's planner (`template`: a concretized
    ParameterizedFabricPlanner, or None for the example defaults)."""
    kw = {}
    if template is not None:
        kw.update(template._strings)
    cfg = _config.panda_config(n_robots=params.nr_robots, horizon=int(horizon), dynamic=int(bool(dynamic)),
                               mounts=params.mount_transform, **kw)
    if template is not None:
        comp = template._components
        cfg.n_ego, cfg.n_planes, cfg.n_goals = comp["n_ego"], comp["n_planes"], comp["n_goals"]
        if comp["n_ego"]:
            cfg.ego_link_mask = _config.ego_link_mask(template._ego_links)
        cfg.use_limits = 0 if comp["limits"] is None else 1
        if comp["limits"] is not None:
            for j in range(7):
                cfg.limits[j][0], cfg.limits[j][1] = comp["limits"][j]
        for key, val in template.constants.items():
            setattr(cfg, key, val)
    if n_ego is not None:
        cfg.n_ego = n_ego
    cfg.dt = params.dt
    if sphere_table:     # the simulator's collision spheres: n_obst_per_link per link, link-local offsets (SIM:188-245)
        links, offs = _config.sphere_offsets_per_link(params.n_obst_per_link)
        _config.set_spheres(cfg, links, offs, [params.radius_sphere] * len(links))
    return cfg


# ------------------------------------------------------------------------------------------------ episode record
class EpisodeLog:
    """What one PandaCell.run() observed.  Arrays are host numpy; `scene` 0 is the one the reference-shaped result
    dictionary describes."""

    def __init__(self, cell, steps, solver_s, wall_s, done_at, states, q_hist, picked, failed, deadlock_steps, nonfinite):
        self.cell, self.steps = cell, steps
        self.solver_s, self.wall_s = solver_s, wall_s            # [steps]
        self.done_at = done_at                                   # [scenes, N] first step reporting DONE, -1 = never
        self.states = states                                     # [steps, scenes, N]
        self.q_hist = q_hist                                     # device [steps, 7, rows] (joint state after each step)
        self.picked, self.failed = picked, failed                # [scenes, N]
        self.deadlock_steps, self.nonfinite = deadlock_steps, nonfinite

    def min_clearance(self, batch=None):
        """[scenes] smallest sphere-surface distance between two robots of a scene over the recorded steps (the
        reference tracks robots 0 and 1 with one radius index for both, EXJ:460-470; here every pair, each sphere with
        its own radius), evaluated in one batched sphere-FK call per `batch` snapshots."""
        cell = self.cell
        h, N, B = cell.ha, cell.N, cell.scenes
        S = h.cfg.n_spheres
        rad = h.tensor(np.array(h.cfg.sphere_radius[:S]))
        best = torch.full((B,), float("inf"), dtype=h.dtype, device=h.device)
        if batch is None:                       # snapshots per call: the [S, S, n, B] distance block stays below 256 MB
            batch = max(1, (1 << 25) // (S * S * B))
        for lo in range(0, self.steps, batch):
            q = self.q_hist[lo:lo + batch]                                       # [n, 7, rows]
            n = q.shape[0]
            x, _, _ = h.fk_spheres(q.permute(1, 0, 2).reshape(7, n * B * N).contiguous())
            x = x.view(S, 3, n, B, N)
            for i in range(N):
                for j in range(i + 1, N):
                    d = (x[:, None, :, :, :, i] - x[None, :, :, :, :, j]).norm(dim=2)      # [S, S, n, B]
                    gap = d - rad[:, None, None, None] - rad[None, :, None, None]
                    best = torch.minimum(best, gap.reshape(S * S * n, B).min(dim=0).values)
        return best.cpu().numpy()

    def reference_result(self, scene=0):
        """The dictionary the reference's run_panda_example returns (EXJ:509-515 = EXC:518-523), for one scene."""
        cell = self.cell
        done = self.done_at[scene]
        first = [float(d) if d >= 0 else float("nan") for d in done]
        solver, wall = self.solver_s, self.wall_s
        nan = float("nan")
        nb = cell.blocks_per_robot
        res = {
            "success_rate": float((self.picked[scene, -1] - self.failed[scene, -1]) / nb),      # the last robot's (EXJ:506)
            "n_steps_panda": first[0], "n_steps_robot2": first[1] if cell.N > 1 else nan,
            "step_time_mean": float(wall.mean()) if len(wall) else nan, "step_time_std": float(wall.std()) if len(wall) else nan,
            "total_time": max(first[:2]) * cell.dt if not any(math.isnan(f) for f in first[:2]) else nan,
            "dt": cell.dt,
            "solver_time_mean": float(solver.mean()) if len(solver) else nan,
            "solver_time_std": float(solver.std()) if len(solver) else nan,
            "min clearance": float(self.min_clearance()[scene]) if self.steps else 100,
            "solver_times": solver,
        }
        # beside the reference's keys
        res.update({
            "control_steps": int(self.steps),
            "blocks_picked": [int(v) for v in self.picked[scene]],
            "states": [int(v) for v in self.states[-1, scene]] if self.steps else [],
            "states_visited": [sorted(set(int(v) for v in self.states[:, scene, i])) for i in range(cell.N)],
            "time_in_deadlock_steps": int(self.deadlock_steps[scene]) if self.deadlock_steps is not None else 0,
            "q_final": self.q_hist[-1].view(7, cell.scenes, cell.N)[:, scene].T.cpu().numpy() if self.steps else None,
            "solver_time_is": "device time of one control step: wall-clock stamps of its first and after its last kernel",
        })
        return res


# ------------------------------------------------------------------------------------------------ the cell
class PandaCell:
    """`scenes` copies of N Pandas picking their cubes.  Build one with from_planners (the example drivers: settings taken
    from the planner objects they define) or from_parameters (the planner defaults of the examples, any batch)."""

    def __init__(self, params, cfg_action, cfg_grasp, cfg_rollout, *, cartesian=False, deadlock=True, estimate="off",
                 cubes=None, scenes=1, stop_margin=-1.0, q_jitter=0.0, rng=None, device=None):
        self.params, self.N, self.scenes, self.dt = params, params.nr_robots, int(scenes), float(params.dt)
        N, B = self.N, self.scenes
        rows = B * N
        self.ha = FabricHandle(cfg_action, device)
        self.hg = FabricHandle(cfg_grasp, device) if cfg_grasp is not None else None
        self.hr = FabricHandle(cfg_rollout, device) if cfg_rollout is not None else None
        if estimate not in ("off", "rollouts", "reference"):
            raise MrfError("estimate: 'off', 'rollouts' (RF-CV inside the rollouts only) or 'reference' (EXJ:346-348: "
                           "the estimate replaces robot 1's goal for the deadlock logic and its own action too)")
        self.estimate = estimate
        cubes = cube_layout(params, scenes=B) if cubes is None else np.asarray(cubes, dtype=float).reshape(B, -1, 3)
        if cubes.shape[1] % N:
            raise MrfError(f"{cubes.shape[1]} cubes do not divide among {N} robots")
        self.cubes = cubes
        self.blocks_per_robot = per = cubes.shape[1] // N
        blocks = np.zeros((per, 3, rows))                       # hand targets: block b of robot i, scene-major rows
        for i in range(N):
            for b in range(per):
                blocks[b, :, i::N] = (cubes[:, i * per + b] + np.array([0.0, 0.0, HAND_ABOVE_CUBE])).T
        start = np.zeros((3, rows))
        q0 = np.zeros((7, rows))
        grip = np.zeros((2, rows))
        for i in range(N):
            start[:, i::N] = np.asarray(params.start_goals[i], dtype=float)[:, None]
            p = np.asarray(params.pos0[i], dtype=float).reshape(-1)
            q0[:, i::N] = p[:7, None]
            grip[:, i::N] = p[7] if len(p) > 7 else 0.02
        if q_jitter:
            q0 += (np.random.default_rng() if rng is None else rng).uniform(-q_jitter, q_jitter, q0.shape)
        t = self.ha.tensor
        self.loop = ControlLoop(
            self.ha, self.hr, t(q0), t(np.zeros_like(q0)), t(nominal_parameters(params, B)), _config.PANDA_VEL_LIMITS,
            deadlock=bool(deadlock), apply_estimate=(estimate == "reference"), stop_margin=stop_margin,
            cartesian_rollouts=bool(cartesian),
            pick_place=dict(start_goal=t(start), blocks=t(blocks), nr_blocks=per, q_gripper=t(grip), model=1, h_grasp=self.hg))

    # -- construction -------------------------------------------------------------------------------------------------
    @classmethod
    def from_planners(cls, params, planners, planners_grasp=None, rollout=None, *, cartesian=False, cubes=None, scenes=1,
                      deadlock=None, device=None, **kw):
        """planners / planners_grasp: concretized ParameterizedFabricPlanner per robot; rollout: the object the drivers
        build for their Rollout Fabrics -- a rollouts.ForwardFabricsPlanner, or the list of per-robot
        rollouts.FabricsRollouts of the Cartesian driver -- or None (plain MRDF)."""
        main = planners[0]
        if any(p._strings != main._strings or p._components["n_dynamic"] != main._components["n_dynamic"] for p in planners):
            raise MrfError("the robots of one cell share their planner definition")
        n_seen = main._components["n_static"] + main._components["n_dynamic"]
        n_table = 8 * params.n_obst_per_link * (params.nr_robots - 1)
        if main._components["n_ego"] and n_seen != n_table:
            raise MrfError(f"the main planners were built for {n_seen} obstacle spheres, the cell's sphere table has {n_table} "
                           f"(8 links x n_obst_per_link = {params.n_obst_per_link} x {params.nr_robots - 1} other robots)")
        cfg_action = _coupled_config(params, main, 1, dynamic=main._components["n_dynamic"] > 0)
        cfg_grasp = _coupled_config(params, planners_grasp[0], 1, dynamic=0) if planners_grasp else None
        cfg_rollout = None
        if rollout is not None and not cartesian:
            cfg_rollout = rollout.config.copy()                  # ForwardFabricsPlanner: N robots, link-origin spheres, H
        elif rollout is not None:
            per_robot = list(rollout)
            r0 = per_robot[0]
            cfg_rollout = _coupled_config(params, r0._planner, r0.N, dynamic=params.STATIC_OR_DYN_FABRICS)
            cfg_rollout.mode = r0.config.mode
        estimate = "reference" if (params.ESTIMATE_GOAL and cfg_rollout is not None) else "off"
        if estimate != "off":
            cfg_rollout.goal_estimate_mask = 0b10                # robot 1's goal is the estimated one (EXJ:346-348)
        if deadlock is None:
            deadlock = cfg_rollout is not None
        return cls(params, cfg_action, cfg_grasp, cfg_rollout, cartesian=cartesian, deadlock=deadlock, estimate=estimate,
                   cubes=cubes, scenes=scenes, device=device, **kw)

    @classmethod
    def from_parameters(cls, params, *, rollouts="jointspace", dynamic=True, estimate="off", deadlock=None, **kw):
        """The example planners without the planner objects: rollouts in (None, 'jointspace', 'cartesian')."""
        cfg_action = _coupled_config(params, None, 1, dynamic)
        cfg_grasp = _coupled_config(params, None, 1, 0, n_ego=0)
        cfg_rollout = None
        if rollouts == "jointspace":
            cfg_rollout = _coupled_config(params, None, params.N_HORIZON, dynamic, sphere_table=False)
        elif rollouts == "cartesian":
            cfg_rollout = _coupled_config(params, None, params.N_HORIZON, dynamic)
        elif rollouts is not None:
            raise MrfError("rollouts: None, 'jointspace' or 'cartesian'")
        if rollouts == "cartesian" and estimate == "rollouts":
            # no Cartesian rollout kernel reads goal_estimate_mask: the reference applies the estimate in the driver
            # (EXC:355-357), which is estimate='reference' here
            raise MrfError("estimate='rollouts' is defined for joint-space rollouts only; Cartesian rollouts take "
                           "estimate='reference' (the driver-side estimate of EXC:355-357) or 'off'")
        if cfg_rollout is not None and estimate != "off":
            cfg_rollout.goal_estimate_mask = ((1 << params.nr_robots) - 1) & ~1 if estimate == "rollouts" else 0b10
        if deadlock is None:
            deadlock = cfg_rollout is not None
        return cls(params, cfg_action, cfg_grasp, cfg_rollout, cartesian=(rollouts == "cartesian"), deadlock=deadlock,
                   estimate=estimate if cfg_rollout is not None else "off", **kw)

    # -- stepping -----------------------------------------------------------------------------------------------------
    def run(self, n_steps, chunk=64, stop_when_done=True, history_limit_bytes=1 << 30):
        """Up to n_steps control steps of every scene, queued `chunk` at a time with no host work in between: the steps
        record themselves (mrf_episode_set_recorder: joint positions, state-machine states, first-DONE step per row,
        wall-clock stamps at the first and after the last kernel of every step).  Stops after the chunk in which every
        robot of every scene has reported DONE and truncates the record to that step (the reference leaves its loop
        there, EXJ:311-312)."""
        loop, N, B = self.loop, self.N, self.scenes
        rows = B * N
        dev = self.ha.device
        n_steps = int(n_steps)
        if n_steps * rows * 7 * loop.q.element_size() > history_limit_bytes:
            raise MrfError("joint history would exceed history_limit_bytes: run fewer steps per call or fewer scenes")
        rec = loop.attach_recorder(n_steps, done_state=DONE)
        tick_s = 1e-9 / self.ha.rollout_clock()["wall_clock_ghz"]          # seconds per tick of the device's wall clock
        wall = []
        w = 0
        while w < n_steps:
            n = min(chunk, n_steps - w)
            t0 = time.perf_counter()
            loop.run(n)
            torch.cuda.synchronize(dev)
            wall += [(time.perf_counter() - t0) / n] * n
            w += n
            if stop_when_done and bool((rec["done_at"] >= 0).all()):
                w = int(rec["done_at"].max())   # the step at which the last robot reported DONE: nothing is kept past it
                break
        loop.recorder = None
        solver = (rec["t_end"][:w] - rec["t_begin"][:w]).double().cpu().numpy() * tick_s
        picked = loop.sm_state[abi.SM_PICKED].view(B, N).cpu().numpy()
        failed = loop.sm_state[abi.SM_FAILED].view(B, N).cpu().numpy()
        dl = loop.dl_state
        return EpisodeLog(self, w, solver, np.asarray(wall[:w], dtype=float), rec["done_at"].view(B, N).cpu().numpy(),
                          rec["sm_hist"][:w].view(w, B, N).cpu().numpy(), rec["q_hist"][:w], picked, failed,
                          dl[abi.DL_TIME_IN_DEADLOCK].cpu().numpy() if dl is not None else None,
                          dl[abi.DL_NONFINITE].cpu().numpy() if dl is not None else None)
