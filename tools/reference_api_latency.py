#!/usr/bin/env python3
"""The reference's own "solver time" measurement (examples/evaluation/evaluate_horizon.py:45-103,
example_pandas_Jointspace.py:353-386,414-450,457) through the reference-shaped Python classes:
2 Pandas, joint-space Rollout Fabrics, dynamic fabrics, n_obst_per_link = 1, K in {1, 10, 20}:
    solver_time = (time of all robots' compute_action) / 2 + time of get_velocity_rollouts
Recorded in the reference's pickle (hardware unknown): 8.457 / 40.416 / 76.270 ms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from multi_robot_fabrics_amd import config, scenarios
from multi_robot_fabrics_amd.parameters import manipulator_parameters
from multi_robot_fabrics_amd.rollouts import ForwardFabricsPlanner
from test_gpu_host_api import set_planner_panda

REF_MS = {1: 8.457, 10: 40.416, 20: 76.270}
params = manipulator_parameters(nr_robots=2, n_obst_per_link=1)
cfg = config.panda_config(n_robots=2, horizon=1, mounts=params.mount_transform)
b = scenarios.panda_batch(cfg, 1, seed=3, x_min=0.08)
planners = [set_planner_panda(params, i, nr_obst=0, nr_obst_dyn=params.nr_obsts_dyn_all[i])[0] for i in range(2)]
from multi_robot_fabrics_amd.runtime import FabricHandle
hk = FabricHandle(cfg, 0)
sx, sv, _ = hk.fk_spheres(hk.tensor(b["q"]), hk.tensor(b["qdot"]))
sx, sv = sx.cpu().numpy(), sv.cpu().numpy()
for K in (1, 10, 20):
    params.define_settings(ROLLOUT_FABRICS=True, STATIC_OR_DYN_FABRICS=1, N_HORIZON=K)
    rp, gs = zip(*[set_planner_panda(params, i, nr_obst=0, nr_obst_dyn=params.nr_obsts_dyn[i]) for i in range(2)])
    fp = ForwardFabricsPlanner(params=params, planners=list(rp), N_steps=100, fk_dict=None, goal_struct_robots=list(gs))
    fp.forward_multi_fabrics_symbolic()
    inputs_action = {"q_robots": [b["q"][:, i] for i in range(2)], "q_dot_robots": [b["qdot"][:, i] for i in range(2)],
                     "x_obsts": [[]], "x_goals0": [b["params"][0:3, i] for i in range(2)],
                     "x_goals1": [g._config.subgoal1.desired_position for g in gs],
                     "x_goals2": [g._config.subgoal2.desired_position for g in gs],
                     "weight_goals0": [2.0] * 2, "weight_goals1": [20.0] * 2, "weight_goals2": [1.0] * 2,
                     "constraints": [np.array([0, 0, 1, -0.65])] * 2}
    kw = []
    for i in range(2):
        j = 1 - i
        kw.append(dict(q=b["q"][:, i], qdot=b["qdot"][:, i], x_goal_0=b["params"][0:3, i], weight_goal_0=2.0,
                       angle_goal_1=params.rotation_matrix_pandas[i], x_goal_1=np.array([0.107, 0, 0]), weight_goal_1=20.0,
                       x_goal_2=np.array([np.pi / 4]), weight_goal_2=1.0, constraint_0=params.constraints[i],
                       radius_body_panda_links=params.radius_body_panda_links,
                       x_obsts_dynamic=[sx[s, :, j] for s in range(8)], xdot_obsts_dynamic=[sv[s, :, j] for s in range(8)],
                       xddot_obsts_dynamic=[np.zeros(3)] * 8, radius_obsts_dynamic=[0.08] * 8))
    ts = []
    for it in range(120):
        t0 = time.perf_counter()
        vel_avg = fp.get_velocity_rollouts(inputs_action=inputs_action)
        t_roll = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i in range(2):
            planners[i].compute_action(**kw[i])
        t_act = (time.perf_counter() - t0) / 2
        if it >= 20:
            ts.append(t_roll + t_act)
    ts = np.array(ts) * 1e3
    print(f"K={K:2d}: solver time mean {ts.mean():.3f} ms (std {ts.std():.3f}, median {np.median(ts):.3f})  "
          f"reference recorded {REF_MS[K]:.3f} ms  -> {REF_MS[K] / ts.mean():.0f}x")
