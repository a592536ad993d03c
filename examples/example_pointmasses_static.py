#!/usr/bin/env python3
"""The reference's examples/example_pointmasses_static.py (BASELINE.json configs[0]: 4 point-mass robots, static
fabrics) through the mirrored classes: one planner (set_planner_point, :102-129) evaluated for each of the four robots
with the six scene spheres and the other three robots as static obstacles (run_point_example, :130-201).  The gym
simulator is replaced by arithmetic: the robots are acceleration-controlled point masses (urdfenvs mode 'acc'),
integrated with dt = 0.01.  There is no contact physics here: a fabric's barrier is a geometry, not a hard constraint --
a robot that is pushed against an obstacle while it creeps (xdot -> 0 makes the repulsion -2/x xdot^2 vanish) gets
arbitrarily close, and where pybullet would stop it at the surface this loop reports `first_contact_step` and goes on
(robot 3 is squeezed between two scene spheres on its diagonal after ~6 s; the CPU oracle does the same with dt = 0.002).

usage: python examples/example_pointmasses_static.py [--steps 1000]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from multi_robot_fabrics_amd.goals import GoalComposition
from multi_robot_fabrics_amd.kinematics import GenericURDFFk
from multi_robot_fabrics_amd.planner import ParameterizedFabricPlanner


def set_planner_point(goal, n_obstacles: int = 2, degrees_of_freedom: int = 7, obstacle_resolution=1):
    """:102-129."""
    degrees_of_freedom = 3                      # the reference overrides its own argument the same way
    with open(os.path.join(ROOT, "examples", "simulation_environments", "urdfs", "pointRobot1.urdf"), "r") as file:
        urdf = file.read()
    fk = GenericURDFFk(urdf, "world", "base_link")
    planner = ParameterizedFabricPlanner(
        degrees_of_freedom, fk,
        collision_geometry="-2.0 / (x ** 1) * xdot ** 2",
        collision_finsler="1.0/(x**2) * (1 - ca.heaviside(xdot))* xdot**2")
    planner.set_components(["base_link"], {}, goal=goal, number_obstacles=n_obstacles)
    planner.concretize()
    return planner


def run_point_example(n_steps=1000, render=True):
    """The reference's signature (:131); there is no renderer in this build, `render` is ignored with a warning."""
    if render:
        import warnings
        warnings.warn("multi-robot-fabrics_amd has no renderer: render=True is ignored", RuntimeWarning, stacklevel=2)
    obstacles_pos = [[1, 1.25, 0], [1, 3.75, 0], [1, -1.25, 0], [-1.1, 0, 0], [-1.1, 2.5, 0], [-1.1, -2.5, 0]]    # :142
    obstacles_radius = [1, 1, 1, 1, 1, 1]
    robots_pos = np.array([[-2.5, 0.01, 0.0], [-2.5, -2.49, 0.0], [2.5, 1.26, 0.0], [2.5, 3.74, 0.0]])             # :146
    goal_robots = [np.array([1.5, 3.76]), np.array([1.5, 1.26]), np.array([-2.5, 0.01]), np.array([-2.5, -2.49])]
    r_robots = [np.array(0.2)] * 4
    n_robots = 4
    goal = GoalComposition(name="goal", content_dict={
        "subgoal0": {"weight": 1, "is_primary_goal": True, "indices": [0, 1], "parent_link": "world",
                     "child_link": "base_link", "desired_position": [1.5, 0.99], "epsilon": 0.1, "type": "staticSubGoal"}})
    planner_point = set_planner_point(goal, n_obstacles=n_robots - 1 + len(obstacles_pos))
    q = robots_pos.copy()
    qdot = np.zeros_like(q)
    dt = 0.01
    min_clear, first_contact = np.inf, None
    for step in range(n_steps):
        action = np.zeros((n_robots, 3))
        for i in range(n_robots):                                                                                  # :180-199
            pos_obs = [np.array(p, dtype=float) for p in obstacles_pos] + [q[j] for j in range(n_robots) if j != i]
            radius_obs = list(obstacles_radius) + [r_robots[j] for j in range(n_robots) if j != i]
            action[i] = planner_point.compute_action(q=q[i], qdot=qdot[i], x_goal_0=goal_robots[i],
                                                     weight_goal_0=goal.sub_goals()[0].weight(), x_obsts=pos_obs,
                                                     radius_obsts=radius_obs, radius_body_base_link=r_robots[i])
        qdot = qdot + dt * action                       # acceleration-controlled point masses
        q = q + dt * qdot
        for i in range(n_robots):
            for j in range(i + 1, n_robots):
                min_clear = min(min_clear, float(np.linalg.norm(q[i, :2] - q[j, :2]) - 0.4))
            for p, r in zip(obstacles_pos, obstacles_radius):
                min_clear = min(min_clear, float(np.linalg.norm(q[i, :2] - np.array(p[:2])) - r - 0.2))
        if first_contact is None and min_clear < 0.0:
            first_contact = step
    dist = [float(np.linalg.norm(q[i, :2] - goal_robots[i])) for i in range(n_robots)]
    return {"steps": n_steps, "distance_to_goal_m": dist, "min_clearance_m": min_clear, "first_contact_step": first_contact,
            "final_speed": [float(np.linalg.norm(v)) for v in qdot]}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    print(json.dumps(run_point_example(n_steps=ap.parse_args().steps, render=False), indent=1))
