#!/usr/bin/env python3
"""The reference's examples/evaluation/evaluate_horizon.py -- the script behind its only recorded numbers
(evaluation/results_horizon: solver time per control step for K = 1, 10, 20) -- on the mirrored
classes: 2 Pandas, joint-space Rollout Fabrics, dynamic fabrics, n_obst_per_link = 1, 100 control steps per horizon
(evaluate_horizon.py:24-103).  solver time = (all compute_action calls) / 2 + get_velocity_rollouts + deadlock_checking
(example_pandas_Jointspace.py:353-386,414-457).  The simulator is replaced by exact velocity integration
(examples/example_pandas_jointspace.py); no pandas / seaborn plot, the numbers are printed and the pickle
`results_horizon` is written in the reference's format (a list of arrays [1, n_steps], one per horizon).

usage: python examples/evaluation/evaluate_horizon.py [--steps 100] [--out results_horizon]
"""
import argparse
import importlib.util
import json
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from multi_robot_fabrics_amd import config
from multi_robot_fabrics_amd.kinematics import UtilsKinematics
from multi_robot_fabrics_amd.parameters import manipulator_parameters

REFERENCE_MS = {1: (8.457, 8.241), 10: (40.416, 38.874), 20: (76.270, 75.320)}   # mean, median of the reference's pickle


def _example():
    spec = importlib.util.spec_from_file_location("example_pandas_jointspace",
                                                  os.path.join(ROOT, "examples", "example_pandas_jointspace.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def define_run_evaluations(n_steps=100):
    ex = _example()
    param = manipulator_parameters(nr_robots=2, n_obst_per_link=1)                          # :45
    [_, ROLLOUTS_PLOTTING, _, RESOLVE_DEADLOCKS, ESTIMATE_GOAL, N_HORIZON, _] = param.get_settings()
    param.define_settings(ROLLOUT_FABRICS=True, ROLLOUTS_PLOTTING=ROLLOUTS_PLOTTING, STATIC_OR_DYN_FABRICS=1,
                          RESOLVE_DEADLOCKS=RESOLVE_DEADLOCKS, ESTIMATE_GOAL=ESTIMATE_GOAL, N_HORIZON=N_HORIZON,
                          MPC_LAYER=False)                                                   # case "rollouts dynamic", :57-72
    planners, planners_grasp, goal_structs = ex.define_planners(params=param)
    kin = UtilsKinematics()
    fk_dict = kin.define_forward_kinematics(planners=planners, collision_links_nrs=param.collision_links_nrs,
                                            collision_links=param.collision_links)
    links, offs = config.sphere_offsets_per_link(param.n_obst_per_link)
    sphere_T = []
    for _ in range(param.nr_robots):
        per_link = [[np.identity(4) for _ in range(param.n_obst_per_link)] for _ in range(8)]
        for s, off in enumerate(offs):
            per_link[links[s] - 1][s % param.n_obst_per_link][0:3, 3] = off
        sphere_T.append(per_link)
    fk_spheres = kin.define_symbolic_collision_link_poses(None, param.collision_links, sphere_T,
                                                          n_obst_per_link=param.n_obst_per_link,
                                                          mount_transform=param.mount_transform)
    results, horizons = [], [1, 10, 20]                                                     # :78
    for h in horizons:
        param.set_horizon(h)
        forwardplanner = ex.define_rollout_planners(params=param, fk_dict=fk_dict, goal_structs=goal_structs)
        results.append(ex.run_panda_example(param, n_steps, planners, goal_structs, fk_dict, fk_spheres, forwardplanner))
    data = [np.expand_dims(np.array(res["solver_times"]), 0) for res in results]            # :96-98
    return horizons, data


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--out", default="results_horizon")
    args = ap.parse_args()
    horizons, data = define_run_evaluations(args.steps)
    with open(args.out, "wb") as fp:                                                         # :100-101
        pickle.dump(data, fp)
    table = {}
    for h, d in zip(horizons, data):
        ms = 1e3 * d[0][min(5, d.shape[1] - 1):]          # the first steps carry one-time costs (handles, first launches)
        table[f"K = {h}"] = {"mean_ms": float(ms.mean()), "median_ms": float(np.median(ms)), "max_ms": float(ms.max()),
                            "control_steps_per_s": float(1e3 / ms.mean()),
                            "reference_recorded_mean_ms": REFERENCE_MS[h][0], "reference_recorded_median_ms": REFERENCE_MS[h][1],
                            "ratio_of_means": REFERENCE_MS[h][0] / float(ms.mean())}
    print(json.dumps({"protocol": "evaluate_horizon.py: 2 Pandas, jointspace RF, dynamic fabrics, n_obst_per_link=1",
                      "steps": args.steps, "solver_time": table,
                      "note": "reference numbers: its committed pickle, hardware unknown (BASELINE.md)"}, indent=1))
