#!/usr/bin/env python3
"""The arena of example_pointmasses_static.py with the other three robots as MOVING obstacles: every robot's fabric gets
their positions and velocities in the plane (dynamic obstacle leaves of dimension 2, zero accelerations), the six spheres
stay static leaves.

Same entry points as the reference's examples/example_pointmasses_dynamic.py -- `set_planner_point(goal, n_obstacles=2,
n_dyn_obstacles=0)`, `run_point_example(n_steps=1000, render=True)` -- over this build's batched arena
(multi_robot_fabrics_amd.pointcell.PointRobotArena): one launch per control step for all robots of all scenes.  The remarks
of the static example on rendering and on creeping contact apply.

usage: python examples/example_pointmasses_dynamic.py [--steps 1000] [--scenes 1]
"""
import argparse
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from examples.example_pointmasses_static import GOALS, ROBOT_RADIUS, SPHERES, STARTS, URDF
from multi_robot_fabrics_amd.goals import point_robot_goal
from multi_robot_fabrics_amd.planner import point_planner
from multi_robot_fabrics_amd.pointcell import PointRobotArena


def set_planner_point(goal, n_obstacles: int = 2, n_dyn_obstacles=0):
    """-> the point-robot planner with n_obstacles static and n_dyn_obstacles planar moving sphere leaves."""
    return point_planner(URDF, goal, n_static=n_obstacles, n_dynamic=n_dyn_obstacles, dynamic_dimension=2)


def run_point_example(n_steps=1000, render=True, *, scenes=1, start_jitter=0.0):
    if render:
        warnings.warn("multi-robot-fabrics_amd has no renderer: render=True is ignored", RuntimeWarning, stacklevel=2)
    goal = point_robot_goal()
    planner = set_planner_point(goal, n_obstacles=len(SPHERES), n_dyn_obstacles=len(STARTS) - 1)
    arena = PointRobotArena(planner, STARTS, GOALS, SPHERES, [1.0] * len(SPHERES), robot_radius=ROBOT_RADIUS,
                            goal_weight=goal.sub_goals()[0].weight(), scenes=scenes, start_jitter=start_jitter)
    return arena.run(n_steps)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--scenes", type=int, default=1)
    args = ap.parse_args()
    print(json.dumps(run_point_example(n_steps=args.steps, render=False, scenes=args.scenes,
                                       start_jitter=0.05 if args.scenes > 1 else 0.0), indent=1))
