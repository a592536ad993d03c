#!/usr/bin/env python3
"""Solver time per control step against the rollout horizon K -- the measurement behind the reference's only recorded
numbers (its pickle evaluation/results_horizon), under the reference's entry point
`define_run_evaluations(n_steps=100, render=False, n_runs=1)`: two Pandas, joint-space Rollout Fabrics with dynamic
obstacle leaves, one sphere per link, one random cube scene, n_steps control steps for each K in (1, 10, 20).

Here a control step is one replayed HIP graph (multi_robot_fabrics_amd.cell.PandaCell) and its "solver time" is the
graph's device time between two HIP events -- rollouts, deadlock logic, state machine, both planners and the integration
step together; the reference times its Python calls of the rollouts, the deadlock logic and half of the planner calls.
The pickle written to `out_path` has the reference's layout: a list with one float array [1, n_steps] per horizon.

usage: python examples/evaluation/evaluate_horizon.py [--steps 100] [--out results_horizon]
"""
import argparse
import json
import math
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from multi_robot_fabrics_amd.cell import PandaCell, cube_layout
from multi_robot_fabrics_amd.parameters import manipulator_parameters

# mean / median solver time [ms] in the reference's committed pickle (hardware not recorded)
RECORDED_MS = {1: (8.457, 8.241), 10: (40.416, 38.874), 20: (76.270, 75.320)}
WARMUP_STEPS = 5          # graph capture and first launches land in the first steps of a run


def get_std(list_of_std: list) -> float:
    """Standard deviation of pooled, equally long runs from the runs' own standard deviations."""
    return math.sqrt(float(np.mean(np.square(list_of_std))))


def define_run_evaluations(n_steps=100, render=False, n_runs=1, *, out_path="results_horizon", horizons=(1, 10, 20), seed=0):
    params = manipulator_parameters(nr_robots=2, n_obst_per_link=1)
    params.define_settings(ROLLOUT_FABRICS=True, STATIC_OR_DYN_FABRICS=1, RESOLVE_DEADLOCKS=True, ESTIMATE_GOAL=False,
                           N_HORIZON=params.N_HORIZON, n_obst_per_link=1)
    cubes = cube_layout(params, random_scene=True, rng=np.random.default_rng(seed))
    runs = {}
    for K in horizons:
        params.set_horizon(K)
        cell = PandaCell.from_parameters(params, rollouts="jointspace", dynamic=True, cubes=cubes)
        runs[K] = cell.run(n_steps).solver_s
    data = [runs[K][None, :] for K in horizons]
    if out_path:
        with open(out_path, "wb") as fp:
            pickle.dump(data, fp)
    table = {}
    for K in horizons:
        ms = 1e3 * runs[K][min(WARMUP_STEPS, len(runs[K]) - 1):]
        row = {"mean_ms": float(ms.mean()), "median_ms": float(np.median(ms)), "max_ms": float(ms.max()),
               "control_steps_per_s": 1e3 / float(ms.mean())}
        if K in RECORDED_MS:
            row.update(reference_recorded_mean_ms=RECORDED_MS[K][0], reference_recorded_median_ms=RECORDED_MS[K][1],
                       ratio_of_means=RECORDED_MS[K][0] / float(ms.mean()))
        table[f"K = {K}"] = row
    return {"protocol": "2 Pandas, joint-space RF, dynamic fabrics, n_obst_per_link = 1, one random scene", "steps": n_steps,
            "horizons": list(horizons), "data": data, "solver_time": table,
            "note": "reference figures: its committed pickle, hardware unknown; different things are inside the two timers"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--out", default="results_horizon")
    args = ap.parse_args()
    out = define_run_evaluations(n_steps=args.steps, render=False, n_runs=1, out_path=args.out)
    print(json.dumps({k: v for k, v in out.items() if k != "data"}, indent=1))
