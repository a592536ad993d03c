#!/usr/bin/env python3
"""The reference's examples/evaluation/evaluate_horizon.py -- the script behind its only recorded numbers
(evaluation/results_horizon: solver time per control step for K = 1, 10, 20) -- with the same
entry point, `define_run_evaluations(n_steps=100, render=False, n_runs=1)` (:25), on the mirrored classes:
2 Pandas, case "rollouts dynamic", n_obst_per_link = 1, a random scene, one run of n_steps control steps per horizon
through examples/example_pandas_Jointspace.run_panda_example (:78-92).
solver time = (all compute_action calls) / 2 + get_velocity_rollouts + deadlock_checking (EXJ:353-386,414-457).
The pickle `results_horizon` is written in the reference's format (a list of arrays [1, n_steps], one per horizon);
instead of the pandas / seaborn box plot the numbers are returned (and printed by the command line).

usage: python examples/evaluation/evaluate_horizon.py [--steps 100] [--out results_horizon]
"""
import argparse
import copy
import json
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import examples.parameters_manipulators as parameters_manipulators
from examples.example_pandas_Jointspace import define_planners, define_rollout_planners, run_panda_example
from examples.simulation_environments.create_simulation_manipulators import create_manipulators_simulation
from multi_robot_fabrics_amd.kinematics import UtilsKinematics

REFERENCE_MS = {1: (8.457, 8.241), 10: (40.416, 38.874), 20: (76.270, 75.320)}   # mean, median of the reference's pickle


def get_std(list_of_std: list) -> float:
    """:15-23: the standard deviation of pooled runs from the runs' standard deviations."""
    return float(np.sqrt(sum(std ** 2 for std in list_of_std) / len(list_of_std)))


def define_run_evaluations(n_steps=100, render=False, n_runs=1, *, out_path="results_horizon"):
    random_scene = True                                                                      # :29
    cases = ["rollouts dynamic"]
    param = parameters_manipulators.manipulator_parameters(nr_robots=2, n_obst_per_link=1)   # :45
    simulation_class = create_manipulators_simulation(params=param)
    kinematics_class = UtilsKinematics()
    random_obstacles = [simulation_class.create_scene(random_scene, n_cubes=param.n_cubes) for _ in range(n_runs)]   # :50-53
    horizons = [1, 10, 20]                                                                   # :78
    results = []
    for case in cases:
        [ROLLOUT_FABRICS, ROLLOUTS_PLOTTING, STATIC_OR_DYN_FABRICS, RESOLVE_DEADLOCKS, ESTIMATE_GOAL, N_HORIZON, MPC_LAYER] = param.get_settings()
        if case in ("dynamic", "rollouts dynamic", "rollouts dynamic estimated"):
            STATIC_OR_DYN_FABRICS = 1
        if case in ("rollouts static", "rollouts dynamic", "rollouts dynamic estimated"):
            ROLLOUT_FABRICS = True
        if case == "rollouts dynamic estimated":
            ESTIMATE_GOAL, RESOLVE_DEADLOCKS = True, True
        param.define_settings(ROLLOUT_FABRICS=ROLLOUT_FABRICS, ROLLOUTS_PLOTTING=ROLLOUTS_PLOTTING,
                              STATIC_OR_DYN_FABRICS=STATIC_OR_DYN_FABRICS, RESOLVE_DEADLOCKS=RESOLVE_DEADLOCKS,
                              ESTIMATE_GOAL=ESTIMATE_GOAL, N_HORIZON=N_HORIZON, MPC_LAYER=False)    # :66-72
        planners, planners_grasp, goal_structs = define_planners(params=param)
        fk_dict = kinematics_class.define_forward_kinematics(planners=planners, collision_links_nrs=param.collision_links_nrs,
                                                             collision_links=param.collision_links)
        results = []
        for h in horizons:
            param.set_horizon(h)
            forwardplanner = (define_rollout_planners(params=param, fk_dict=fk_dict, goal_structs=goal_structs)
                              if case.startswith("rollouts") else None)
            env = simulation_class.initialize_environment(render=render, random_scene=random_scene, obstacles=random_obstacles[0])
            res = run_panda_example(param, n_steps=n_steps, planners=planners, planners_grasp=planners_grasp,
                                    goal_structs=goal_structs, env=env, fk_dict=fk_dict, forwardplanner=forwardplanner)
            env.close()
            results.append(res)
    data = [np.expand_dims(np.array(res["solver_times"]), 0) for res in results]            # :96-98
    if out_path:
        with open(out_path, "wb") as fp:                                                     # :100-101
            pickle.dump(data, fp)
    table = {}
    for h, d in zip(horizons, data):
        ms = 1e3 * d[0][min(5, d.shape[1] - 1):]          # the first steps carry one-time costs (handles, first launches)
        table[f"K = {h}"] = {"mean_ms": float(ms.mean()), "median_ms": float(np.median(ms)), "max_ms": float(ms.max()),
                            "control_steps_per_s": float(1e3 / ms.mean()),
                            "reference_recorded_mean_ms": REFERENCE_MS[h][0], "reference_recorded_median_ms": REFERENCE_MS[h][1],
                            "ratio_of_means": REFERENCE_MS[h][0] / float(ms.mean())}
    return {"protocol": "evaluate_horizon.py: 2 Pandas, jointspace RF, dynamic fabrics, n_obst_per_link=1", "steps": n_steps,
            "horizons": horizons, "data": data, "solver_time": table,
            "note": "reference numbers: its committed pickle, hardware unknown (BASELINE.md)"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--out", default="results_horizon")
    args = ap.parse_args()
    out = define_run_evaluations(n_steps=args.steps, render=False, n_runs=1, out_path=args.out)
    print(json.dumps({k: v for k, v in out.items() if k != "data"}, indent=1))
