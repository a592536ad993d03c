#!/usr/bin/env python3
"""The pick-and-place cell of example_pandas_Jointspace.py with the CARTESIAN variant of Rollout Fabrics: every arm
rolls out its OWN fabric over the horizon against the other arms' collision spheres extrapolated at constant velocity --
no coupling between the rollouts, one rollout object per robot.

Same entry points as the reference's examples/example_pandas_cartesian.py (names, positional parameters, defaults,
result keys; tests/test_examples_contract.py).  As in the joint-space example the functions here describe the cell and
multi_robot_fabrics_amd.cell.PandaCell steps it on the GPU; the per-robot rollouts of one control step are ONE launch for
all robots of all scenes, with the obstacle lists (the other arms' n_obst_per_link spheres per link, their positions and
velocities J qdot at the start of the horizon) assembled on the device (include/mrf.h mrf_rollout_cartesian_coupled).
`env` is the cube layout; `fk_dict`, `fk_dict_spheres` and `utils_class` are unused; `render=True` is ignored with a
warning.

Behaviour kept from the reference: the YAML flags are applied before the planners are built (STATIC_OR_DYN_FABRICS
decides how the main planners see the other arm); the deadlock logic runs only with RESOLVE_DEADLOCKS; with ESTIMATE_GOAL
robot 1's goal is x_ee + 0.2 v_ee everywhere.

usage: python examples/example_pandas_cartesian.py [--steps 7000] [--config other.yaml] [--cubes 2]
"""
import argparse
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from multi_robot_fabrics_amd.cell import PandaCell, cube_layout
from multi_robot_fabrics_amd.goals import panda_pick_place_goal
from multi_robot_fabrics_amd.parameters import load_yaml_settings
from multi_robot_fabrics_amd.planner import panda_planner
from multi_robot_fabrics_amd.rollouts import FabricsRollouts



def create_dummy_goal_panda():
    return panda_pick_place_goal(orientation_weight=20.0)


def set_planner_panda(degrees_of_freedom: int = 7, nr_obst=0, nr_obst_dyn=1, collision_links_nr=[5], urdf_links={},
                      mount_transform=[], i_robot=0):
    """-> (planner, goal) of robot i_robot, mounted at mount_transform[i_robot] (4x4)."""
    if degrees_of_freedom != 7:
        raise ValueError("the Panda chain has seven joints")
    target = create_dummy_goal_panda()
    return panda_planner(urdf_links["URDF_file_panda"], mount_transform[i_robot], target, collision_links_nr, nr_obst,
                         nr_obst_dyn), target


def define_planners(params):
    """-> (main planners, grasp planners, goals) per robot; see the joint-space example."""
    def robot(i, links, counts):
        return set_planner_panda(params.dof[i], *counts, links, params.urdf_links, params.mount_transform, i)

    rows = [robot(i, links, params.obstacle_counts(i)) + (robot(i, [], (0, 0))[0],)
            for i, links in enumerate(params.collision_links_nrs)]
    main, goals, grasp = (list(column) for column in zip(*rows))
    return main, grasp, goals


def define_rollout_planners(params, fk_dict=None, goal_structs=None, n_steps=100, planners=[], nr_robots=2):
    """-> one rollout object per robot, horizon params.N_HORIZON, bound to that robot's main planner: it rolls the
    planner out against all nr_obsts_dyn_all spheres of the other robots, each moving with the velocity it has now."""
    return [FabricsRollouts.for_robot(params, i, planners[i], goal_structs[i]) for i in range(nr_robots)]


def run_panda_example(params, n_steps=5000, planners=[], planners_grasp=[], goal_structs=[], env=None, fk_dict=None,
                      forwardplanners=None, fk_dict_spheres=None, utils_class=None, *, scenes=1) -> dict:
    """Runs the cell for at most n_steps control steps; `env`: cube centres ([n_cubes, 3] or [scenes, n_cubes, 3]) or None
    for the fixed layout; `forwardplanners`: the list from define_rollout_planners or None."""
    rollouts = forwardplanners if params.ROLLOUT_FABRICS else None
    cell = PandaCell.from_planners(params, planners, planners_grasp, rollouts, cartesian=True, cubes=env, scenes=scenes,
                                   deadlock=bool(rollouts) and bool(params.RESOLVE_DEADLOCKS))
    log = cell.run(n_steps)
    result = log.reference_result()
    if params.ROLLOUTS_PLOTTING and rollouts and log.steps:
        _, traj_q, _ = cell.hr.rollout_cartesian_coupled(cell.loop.q, cell.loop.qdot, cell.loop.params_work, want_traj=True)
        result["rollout_plot_data"] = traj_q.cpu().numpy()
    return result


def define_run_panda_example(n_steps=100, render=True, *, config_path=None, n_cubes=None, scenes=1):
    if render:
        warnings.warn("multi-robot-fabrics_amd has no renderer: render=True is ignored", RuntimeWarning, stacklevel=2)
    params, setup = load_yaml_settings(config_path)          # flags first: they decide how the planners see the other arm
    if n_cubes is not None:
        params.n_cubes = int(n_cubes)
    built = define_planners(params)
    rollouts = None
    if params.ROLLOUT_FABRICS:
        rollouts = define_rollout_planners(params, goal_structs=built[2], n_steps=n_steps, planners=built[0], nr_robots=params.nr_robots)
    result = run_panda_example(params, n_steps, *built, env=cube_layout(params, scenes=scenes), forwardplanners=rollouts, scenes=scenes)
    result["config"] = setup
    return result


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=7000)
    ap.add_argument("--config", default=None)
    ap.add_argument("--cubes", type=int, default=None)
    args = ap.parse_args()
    res = define_run_panda_example(n_steps=args.steps, render=False, config_path=args.config, n_cubes=args.cubes)
    print(json.dumps({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in res.items()
                      if k not in ("solver_times", "rollout_plot_data")}, indent=1))
