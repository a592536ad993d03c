#!/usr/bin/env python3
"""Four point robots swap sides of an arena with six fixed spheres in it; every robot's fabric treats the spheres AND
the other three robots as static obstacles (BASELINE.json configs[0]).

Same entry points as the reference's examples/example_pointmasses_static.py -- `set_planner_point(goal, n_obstacles=2,
degrees_of_freedom=7, obstacle_resolution=1)`, `run_point_example(n_steps=1000, render=True)` -- over this build's batched
arena (multi_robot_fabrics_amd.pointcell.PointRobotArena): one `mrf_compute_action` launch per control step for all
robots (and all `scenes` copies of the arena), the other robots' states gathered into every robot's obstacle list on the
device, accelerations integrated as urdfenvs' 'acc' mode does.  No simulator, no renderer (`render=True` is ignored with a
warning), no contact physics: the result says when a robot first touched something instead of stopping it there.

usage: python examples/example_pointmasses_static.py [--steps 1000] [--scenes 1]
"""
import argparse
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multi_robot_fabrics_amd.goals import point_robot_goal
from multi_robot_fabrics_amd.planner import point_planner
from multi_robot_fabrics_amd.pointcell import PointRobotArena

URDF = os.path.join(ROOT, "examples", "simulation_environments", "urdfs", "pointRobot1.urdf")
# the arena of the reference's example: sphere centres (radius 1), robot starts (x, y, heading) and goals
SPHERES = [(1, 1.25, 0), (1, 3.75, 0), (1, -1.25, 0), (-1.1, 0, 0), (-1.1, 2.5, 0), (-1.1, -2.5, 0)]
STARTS = [(-2.5, 0.01, 0.0), (-2.5, -2.49, 0.0), (2.5, 1.26, 0.0), (2.5, 3.74, 0.0)]
GOALS = [(1.5, 3.76), (1.5, 1.26), (-2.5, 0.01), (-2.5, -2.49)]
ROBOT_RADIUS = 0.2


def set_planner_point(goal, n_obstacles: int = 2, degrees_of_freedom: int = 7, obstacle_resolution=1):
    """-> the point-robot planner with n_obstacles static sphere leaves (the chain has three joints whatever
    degrees_of_freedom says -- the reference overrides that argument too; obstacle_resolution is unused there as well)."""
    return point_planner(URDF, goal, n_static=n_obstacles)


def run_point_example(n_steps=1000, render=True, *, scenes=1, start_jitter=0.0):
    if render:
        warnings.warn("multi-robot-fabrics_amd has no renderer: render=True is ignored", RuntimeWarning, stacklevel=2)
    goal = point_robot_goal()
    planner = set_planner_point(goal, n_obstacles=len(SPHERES) + len(STARTS) - 1)
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
