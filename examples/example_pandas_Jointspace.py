#!/usr/bin/env python3
"""Two (or three) Pandas pick their cubes from a shared table and carry them home; each arm's fabric treats the other
arms' collision spheres as obstacles, and Rollout Fabrics in JOINT space -- the coupled H-step forward simulation of all
arms -- feed the deadlock detection.

Same entry points as the reference's examples/example_pandas_Jointspace.py (names, positional parameters, defaults and
the keys of the returned dictionary; tests/test_examples_contract.py), a different program underneath: the functions
below only DESCRIBE the cell -- planner objects, rollout object, cube layout -- and hand it to
multi_robot_fabrics_amd.cell.PandaCell, which runs every control step on the GPU as one replayed HIP graph (state
machine, RF-CV goal estimate, rollouts, deadlock logic, main / grasp planner, gripper, velocity integration;
include/mrf.h mrf_episode_run).  No simulator object, no per-step host arithmetic: `env` is the cube layout,
`fk_dict` is unused, `render=True` is ignored with a warning (there is no renderer).

Behaviour kept from the reference, stated where it is decided: the main planners are built BEFORE the YAML flags are
applied and therefore take the other arm's spheres as STATIC obstacles in this driver (the evaluation scripts apply the
flags first and get dynamic ones); the joint-space driver runs the deadlock logic whenever rollouts are on; with
ESTIMATE_GOAL the estimate x_ee + 0.2 v_ee replaces robot 1's goal everywhere, its own planner included.
Deviation: that estimate uses the hand VELOCITY (the Cartesian driver's form); the reference's joint-space driver feeds
a flattened Jacobian there.

usage: python examples/example_pandas_Jointspace.py [--steps 7000] [--config other.yaml] [--scenes 1]
"""
import argparse
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from multi_robot_fabrics_amd import config
from multi_robot_fabrics_amd.cell import PandaCell, cube_layout
from multi_robot_fabrics_amd.goals import panda_pick_place_goal
from multi_robot_fabrics_amd.parameters import load_yaml_settings
from multi_robot_fabrics_amd.planner import panda_planner
from multi_robot_fabrics_amd.rollouts import ForwardFabricsPlanner



def create_dummy_goal_panda():
    return panda_pick_place_goal(orientation_weight=10.0)


def set_planner_panda(degrees_of_freedom: int = 7, nr_obst=0, nr_obst_dyn=1, collision_links_nr=[5], urdf_links={},
                      mount_param={}, i_robot=0):
    """-> (planner, goal) of robot i_robot: base at mount_param['mount_positions'][i_robot], robot 0 facing +x and the
    others turned by pi to face it."""
    if degrees_of_freedom != 7:
        raise ValueError("the Panda chain has seven joints")
    goal = create_dummy_goal_panda()
    base = config.mount_transform(mount_param["mount_positions"][i_robot], 0.0 if i_robot == 0 else np.pi)
    return panda_planner(urdf_links["URDF_file_panda"], base, goal, collision_links_nr, nr_obst, nr_obst_dyn), goal


def define_planners(params):
    """-> (main planners, grasp planners, goals), one of each per robot.  A grasp planner has no collision links: it
    takes over while the hand descends onto its cube.  The main planners see n_obst_per_link spheres on each of the other
    robots' eight links, as static obstacles (STATIC_OR_DYN_FABRICS == 0) or as moving ones."""
    def robot(i, links, counts):
        return set_planner_panda(params.dof[i], *counts, links, params.urdf_links, params.mount_param, i)

    rows = [robot(i, links, params.obstacle_counts(i)) + (robot(i, [], (0, 0))[0],)
            for i, links in enumerate(params.collision_links_nrs)]
    main, goals, grasp = (list(column) for column in zip(*rows))
    return main, grasp, goals


def define_rollout_planners(params, fk_dict=None, goal_structs=None, n_steps=100):
    """-> the joint-space rollout object (horizon params.N_HORIZON): its planners see ONE sphere per link of the other
    robots, the link origins (params.nr_obsts_dyn)."""
    def link_origin_planner(i):
        return set_planner_panda(*params.rollout_planner_spec(i), params.urdf_links, params.mount_param, i)[0]

    return ForwardFabricsPlanner.for_cell(params, link_origin_planner, goal_structs, n_steps, fk_dict)


def run_panda_example(params, n_steps=5000, planners=[], planners_grasp=[], goal_structs=[], env=None, fk_dict=None,
                      forwardplanner=None, *, scenes=1) -> dict:
    """Runs the cell for at most n_steps control steps and reports what the reference's driver reports.  `env`: cube
    centres [n_cubes, 3] (or [scenes, n_cubes, 3]); None = the fixed layout.  `forwardplanner`: the object from
    define_rollout_planners, or None for plain multi-robot dynamic fabrics."""
    cell = PandaCell.from_planners(params, planners, planners_grasp, forwardplanner if params.ROLLOUT_FABRICS else None,
                                   cubes=env, scenes=scenes)
    log = cell.run(n_steps)
    result = log.reference_result()
    if params.ROLLOUTS_PLOTTING and forwardplanner is not None and log.steps:
        # the predicted joint trajectories of the last control step (the reference draws them with matplotlib)
        q, qd = cell.loop.q, cell.loop.qdot
        _, traj_q, _ = cell.hr.rollout(q, qd, cell.loop.params_work, want_traj=True)
        result["rollout_plot_data"] = traj_q.cpu().numpy()
    if scenes > 1:
        result["all_scenes"] = {"success": (log.done_at >= 0).all(axis=1).tolist(), "min_clearance_m": log.min_clearance().tolist()}
    return result


def define_run_panda_example(n_steps=100, render=True, *, config_path=None, scenes=1):
    if render:
        warnings.warn("multi-robot-fabrics_amd has no renderer: render=True is ignored", RuntimeWarning, stacklevel=2)
    params, setup = load_yaml_settings(config_path, apply_flags=False)
    built = define_planners(params)                  # before the flags: static-obstacle main planners (see the header)
    params.apply_yaml(setup)
    rollout = define_rollout_planners(params, goal_structs=built[2]) if params.ROLLOUT_FABRICS else None
    result = run_panda_example(params, n_steps, *built, env=cube_layout(params, scenes=scenes), forwardplanner=rollout, scenes=scenes)
    result["config"] = setup
    return result


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=7000)
    ap.add_argument("--config", default=None)
    ap.add_argument("--scenes", type=int, default=1)
    args = ap.parse_args()
    res = define_run_panda_example(n_steps=args.steps, render=False, config_path=args.config, scenes=args.scenes)
    print(json.dumps({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in res.items() if k != "solver_times"}, indent=1))
