#!/usr/bin/env python3
"""Random pick-and-place scenes for two Pandas under the three methods the reference compares

    "dynamic"                      multi-robot dynamic fabrics: every arm avoids the other arm's moving spheres, no rollouts
    "rollouts dynamic"             + joint-space Rollout Fabrics every control step, deadlock logic on their velocity signal
    "rollouts dynamic estimated"   + robot 1's goal is not communicated but estimated from its hand motion (RF-CV)

`define_run_evaluations(n_steps=100, render=False, n_runs=1)` is the reference's entry point
(examples/evaluation/evaluate_random_dynamic_scenarios.py) and reports its statistics -- time to success, collision
episodes, minimum clearance, solver / step time, success rate; mean +- std per case, as a dictionary and a text table --
but the n_runs random scenes of a case are not run one after the other: they are the `scenes` of ONE
multi_robot_fabrics_amd.cell.PandaCell and advance together, one HIP graph launch per control step for all of them.
`run_case` is the same thing for thousands of scenes with cubes spread over each arm's reach.

Settings as the reference's script leaves them: one sphere per link, dynamic obstacle leaves, horizon from the parameter
class; in the third case the estimate x_ee + 0.2 v_ee replaces robot 1's goal everywhere (its own planner included), and
the deadlock logic stays on (the joint-space driver ignores RESOLVE_DEADLOCKS).  `run_case` keeps the estimate inside
the rollouts instead (`estimate="rollouts"`), which is what RF-CV means in the paper.

What stands in for pybullet: exact velocity integration, a cube that travels with the closed gripper.
A behavioural evaluation of the planner specification, not a parity test.

usage: python examples/evaluation/evaluate_random_dynamic_scenarios.py [--runs 8 --steps 7000]
       python examples/evaluation/evaluate_random_dynamic_scenarios.py --device [--scenarios 512] [--steps 4000] [--blocks 2]
"""
import argparse
import json
import math
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np

from multi_robot_fabrics_amd.cell import CUBE_HALF, PandaCell, cube_layout
from multi_robot_fabrics_amd.parameters import manipulator_parameters

CASES = {  # case -> PandaCell.from_parameters settings
    "dynamic": dict(rollouts=None),
    "rollouts dynamic": dict(rollouts="jointspace"),
    "rollouts dynamic estimated": dict(rollouts="jointspace", estimate="reference"),
}


def get_std(list_of_std: list) -> float:
    """Standard deviation of pooled, equally long runs from the runs' own standard deviations."""
    return math.sqrt(float(np.mean(np.square(list_of_std))))


def _mean_std(values):
    v = np.asarray(values, dtype=float)
    v = v[np.isfinite(v)]
    return {"mean": float(v.mean()) if v.size else float("nan"), "std": float(v.std()) if v.size else float("nan")}


def define_run_evaluations(n_steps=100, render=False, n_runs=1, *, out_path="results_dynamic_scenarios", seed=0):
    params = manipulator_parameters(nr_robots=2)
    params.define_settings(STATIC_OR_DYN_FABRICS=1, N_HORIZON=params.N_HORIZON)
    cubes = cube_layout(params, random_scene=True, rng=np.random.default_rng(seed), scenes=n_runs)
    out = {"n_runs": n_runs, "n_steps": n_steps, "cases": {}}
    for case, settings in CASES.items():
        log = PandaCell.from_parameters(params, dynamic=True, cubes=cubes, scenes=n_runs, **settings).run(n_steps)
        done = log.done_at.astype(float)
        done[done < 0] = np.nan
        finished = ~np.isnan(done).any(axis=1)                                   # every robot of the scene delivered
        rate = (log.picked[:, -1] - log.failed[:, -1]) / log.cell.blocks_per_robot    # the last robot's, as the reference counts
        clearance = log.min_clearance()
        ok = rate == 1
        out["cases"][case] = {
            "time_to_success_s": _mean_std(np.where(finished, np.nanmax(np.nan_to_num(done, nan=-1.0), axis=1), np.nan) * params.dt),
            "collision_episode_rate": float((clearance[ok] < 0).mean()) if ok.any() else 0.0,
            "min_clearance_m": _mean_std(clearance[ok]),
            "solver_time_s": {"mean": float(log.solver_s.mean()), "std": get_std([log.solver_s.std()])},
            "step_time_s": {"mean": float(log.wall_s.mean()), "std": get_std([log.wall_s.std()])},
            "success_rate": _mean_std(rate),
            "scenes_advanced_per_launch": n_runs}
        if out_path:       # the reference leaves the solver times of the last case's runs behind
            with open(out_path, "wb") as fp:
                pickle.dump([log.solver_s[None, :].copy() for _ in range(n_runs)], fp)
    head = ["", "Time-to-Success", "# Collision Episodes", "Min Clearance", "Solver-Time", "Step-Time", "Success-Rate"]
    pm = lambda d: "%.4f+-%.4f" % (d["mean"], d["std"])
    rows = [head] + [[case, pm(c["time_to_success_s"]), "%.6f" % c["collision_episode_rate"], pm(c["min_clearance_m"]),
                      pm(c["solver_time_s"]), pm(c["step_time_s"]), pm(c["success_rate"])] for case, c in out["cases"].items()]
    widths = [max(len(r[k]) for r in rows) for k in range(len(head))]
    out["table"] = "\n".join(" | ".join(cell.ljust(w) for cell, w in zip(r, widths)) for r in rows)
    return out


def reachable_cubes(params, scenes, n_blocks, rng):
    """[scenes, N * n_blocks, 3]: robot i's cubes 0.35-0.6 m from its base, within +-1 rad of the direction it faces."""
    N = params.nr_robots
    out = np.zeros((scenes, N * n_blocks, 3))
    out[:, :, 2] = params.z_table + CUBE_HALF
    for i, T in enumerate(np.asarray(params.mount_transform, dtype=float)):
        facing = math.atan2(T[1, 0], T[0, 0])
        for b in range(n_blocks):
            reach, angle = rng.uniform(0.35, 0.6, scenes), facing + rng.uniform(-1.0, 1.0, scenes)
            out[:, i * n_blocks + b, 0] = T[0, 3] + reach * np.cos(angle)
            out[:, i * n_blocks + b, 1] = T[1, 3] + reach * np.sin(angle)
    return out


def run_case(case, params, B, steps, n_blocks, seed, monitor=64):
    """One method on B random scenes at once; RF-CV stays inside the rollouts.  -> statistics of the batch."""
    rng = np.random.default_rng(seed)
    settings = dict(CASES[case])
    if settings.get("estimate"):
        settings["estimate"] = "rollouts"
    cell = PandaCell.from_parameters(params, dynamic=True, scenes=B, cubes=reachable_cubes(params, B, n_blocks, rng),
                                     stop_margin=1e-3, q_jitter=0.05, rng=rng, **settings)
    t0 = time.perf_counter()
    log = cell.run(steps, chunk=monitor)
    wall = time.perf_counter() - t0
    ok = (log.done_at >= 0).all(axis=1)
    clearance = log.min_clearance()
    res = {"case": case, "scenarios": B, "control_steps_run": log.steps, "blocks_per_robot": n_blocks,
           "success_rate": float(ok.mean()),
           "mean_time_to_success_s": float(log.done_at[ok].max(axis=1).mean() * params.dt) if ok.any() else None,
           "mean_blocks_picked_per_robot": float(log.picked.mean()),
           "min_clearance_m": float(clearance.min()), "collision_episodes": int((clearance < 0).sum()),
           "all_finite": bool(np.isfinite(log.q_hist[-1].cpu().numpy()).all()),
           "wall_s": wall, "scenario_control_steps_per_s": B * log.steps / wall}
    if log.deadlock_steps is not None:
        res["episodes_with_deadlock_resolution"] = int((log.deadlock_steps > 0).sum())
        res["mean_steps_in_deadlock"] = float(log.deadlock_steps.mean())
        res["steps_with_nonfinite_rollout_signal"] = int(log.nonfinite.sum())
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=8)
    ap.add_argument("--device", action="store_true", help="thousands of scenes with cubes over each arm's reach (run_case)")
    ap.add_argument("--scenarios", type=int, default=512)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--blocks", type=int, default=2)
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--robots", type=int, default=2)
    args = ap.parse_args()
    if args.device:
        params = manipulator_parameters(nr_robots=args.robots, n_obst_per_link=1)
        params.set_horizon(args.horizon)
        print(json.dumps([run_case(c, params, args.scenarios, args.steps or 4000, args.blocks, seed=7) for c in CASES], indent=1))
    else:
        out = define_run_evaluations(n_steps=args.steps or 7000, render=False, n_runs=args.runs)
        print(out.pop("table"))
        print(json.dumps(out, indent=1))
