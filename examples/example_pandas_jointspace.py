#!/usr/bin/env python3
"""The planner side of the reference's examples/example_pandas_Jointspace.py, function by function, on the HIP kernels:

    configs/panda_config.yaml -> manipulator_parameters -> define_planners / define_rollout_planners
        -> run_panda_example: per control step  get_velocity_rollouts -> deadlock_checking -> compute_action per robot

with the reference's own call surface (EXJ:136-193 planners, :280-458 loop, :517-538 driver).  What the reference gets
from its pybullet simulator is replaced by arithmetic: the observation is the state itself, env.step integrates the
clipped velocity command exactly (urdfenvs 'vel' mode), collision-sphere centres come from the sphere forward
kinematics (create_simulation_manipulators.py:188-245 offsets).  There are no blocks, so the pick-and-place state
machine stays in its "approach" state and the goals are the start goals of parameters_manipulators.

After the host-API loop the same configuration runs as a device-resident episode (runtime.ControlLoop =
mrf_episode_run: no host round trip per step) and the two trajectories are compared.

usage: python examples/example_pandas_jointspace.py [--config examples/configs/panda_config.yaml] [--steps 200]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np

from multi_robot_fabrics_amd import config
from multi_robot_fabrics_amd.deadlock import deadlockprevention
from multi_robot_fabrics_amd.goals import GoalComposition
from multi_robot_fabrics_amd.kinematics import GenericURDFFk, UtilsKinematics
from multi_robot_fabrics_amd.parameters import load_yaml_settings
from multi_robot_fabrics_amd.planner import ParameterizedFabricPlanner
from multi_robot_fabrics_amd.rollouts import ForwardFabricsPlanner


def create_dummy_goal_panda():
    """EXJ:25-62."""
    goal_dict = {
        "subgoal0": {"weight": 2.0, "is_primary_goal": True, "indices": [0, 1, 2], "parent_link": "world",
                     "child_link": "panda_hand", "desired_position": [0.1, 0.6, 0.8], "epsilon": 0.05,
                     "type": "staticSubGoal"},
        "subgoal1": {"weight": 10.0, "is_primary_goal": False, "indices": [0, 1, 2], "parent_link": "panda_link7",
                     "child_link": "panda_hand", "desired_position": [0.107, 0.0, 0.0],
                     "angle": [-0.366, 0.0, 0.0, 0.3305], "epsilon": 0.05, "type": "staticSubGoal"},
        "subgoal2": {"weight": 1.0, "is_primary_goal": False, "indices": [6], "desired_position": [np.pi / 4],
                     "epsilon": 0.05, "type": "staticJointSpaceSubGoal"},
    }
    return GoalComposition(name="goal", content_dict=goal_dict)


def set_planner_panda(degrees_of_freedom=7, nr_obst=0, nr_obst_dyn=1, collision_links_nr=(5,), mount_transform=None):
    """EXJ:64-134 (the URDF is compiled into the kernels, so GenericURDFFk takes no text)."""
    goal = create_dummy_goal_panda()
    fk = GenericURDFFk(None, "panda_link0", "panda_leftfinger")
    planner = ParameterizedFabricPlanner(
        degrees_of_freedom, fk,
        geometry_plane_constraint="10*(1/(1+1*ca.exp(-10*x))-1) * (xdot**2)",
        collision_geometry="-0.5 / (x ** 4) * (xdot ** 2)",
        collision_finsler="0.01/(x**4) * xdot**2",
    )
    collision_links = ["panda_link%d" % l if l < 9 else "panda_hand" for l in collision_links_nr]
    planner._forward_kinematics.set_mount_transformation(mount_transform)
    planner.set_components(collision_links=collision_links, goal=goal, number_obstacles=nr_obst,
                           number_dynamic_obstacles=nr_obst_dyn, dynamic_obstacle_dimension=3,
                           number_plane_constraints=1, limits=config.PANDA_LIMITS)
    planner.concretize(mode="vel", time_step=0.01)
    return planner, goal


def define_planners(params):
    """EXJ:136-170."""
    if params.STATIC_OR_DYN_FABRICS == 0:
        nr_obst, nr_obst_dyn = params.nr_obsts_dyn_all, [0] * params.nr_robots
    else:
        nr_obst, nr_obst_dyn = [0] * params.nr_robots, params.nr_obsts_dyn_all
    planners, planners_grasp, goal_structs = [], [], []
    for i in range(params.nr_robots):
        p, g = set_planner_panda(params.dof[i], nr_obst[i], nr_obst_dyn[i], params.collision_links_nrs[i],
                                 params.mount_transform[i])
        pg, _ = set_planner_panda(params.dof[i], i, i, [], params.mount_transform[i])
        planners.append(p)
        goal_structs.append(g)
        planners_grasp.append(pg)
    return planners, planners_grasp, goal_structs


def define_rollout_planners(params, fk_dict=None, goal_structs=None, n_steps=100):
    """EXJ:172-193."""
    planners_rollout = [set_planner_panda(params.dof[i], params.nr_obsts[i], params.nr_obsts_dyn[i],
                                          params.collision_links_nrs[i], params.mount_transform[i])[0]
                        for i in range(params.nr_robots)]
    fwd = ForwardFabricsPlanner(params=params, planners=planners_rollout, N_steps=n_steps, fk_dict=fk_dict,
                                goal_struct_robots=goal_structs)
    fwd.forward_multi_fabrics_symbolic()
    return fwd


def run_panda_example(params, n_steps, planners, goal_structs, fk_dict, fk_spheres, forwardplanner):
    """EXJ:195-515 without the simulator: returns the joint trajectory and the timing statistics."""
    N = params.nr_robots
    limit_vel = np.array(config.PANDA_VEL_LIMITS)
    constraints = [np.array([0, 0, 1, 0.0 - params.mount_param["z_table"]])] * N
    q = [np.array(params.pos0[i][:7], dtype=float) for i in range(N)]
    qdot = [np.zeros(7) for _ in range(N)]
    ee = UtilsKinematics().define_symbolic_endeffector(planners)
    deadlock_prevention = deadlockprevention(params.dof, N, params.N_HORIZON) if params.ROLLOUT_FABRICS else None
    time_deadlock_out = 1000
    solver_times, traj = [], []
    for w in range(n_steps):
        goal_pandas = [np.array(g, dtype=float) for g in params.start_goals]
        goal_weights = [2.0] * N                                        # state machine in "approach" (SM weight 2)
        state_machine_pandas = [0] * N
        x_robots_ee = [np.asarray(ee[i]["fk_fun_ee"](q[i])).reshape(-1) for i in range(N)]
        v_robots_ee = [np.asarray(ee[i]["vel_fun_ee"](q[i], qdot[i])).reshape(-1) for i in range(N)]
        # link-origin velocities, repeated n_obst_per_link times below (EXJ:330-344, 409-410)
        v_robots = [[np.asarray(fk_dict["jac_fun"][i][l](q[i]) @ qdot[i]).reshape(-1) if params.STATIC_OR_DYN_FABRICS
                     else np.zeros(3) for l in range(len(params.collision_links_nrs[i]))] for i in range(N)]
        if params.ESTIMATE_GOAL:
            goal_pandas[1] = x_robots_ee[1] + 20 * 0.01 * v_robots_ee[1]               # EXC:355-357 form
        t_rollouts = 0.0
        if params.ROLLOUT_FABRICS:
            t0 = time.perf_counter()
            inputs_action = {"q_robots": q, "q_dot_robots": qdot, "x_obsts": [[] * N], "x_goals0": goal_pandas,
                             "x_goals1": [g._config.subgoal1.desired_position for g in goal_structs],
                             "x_goals2": [g._config.subgoal2.desired_position for g in goal_structs],
                             "weight_goals0": goal_weights,
                             "weight_goals1": [g._config.subgoal1.weight for g in goal_structs],
                             "weight_goals2": [g._config.subgoal2.weight for g in goal_structs],
                             "constraints": constraints}
            vel_avg = forwardplanner.get_velocity_rollouts(inputs_action=inputs_action)
            vel_avg_tot = sum(vel_avg) / N
            if params.RESOLVE_DEADLOCKS:
                goal_pandas, goal_weights, time_deadlock_out = deadlock_prevention.deadlock_checking(
                    x_robots=x_robots_ee, goal_robots=goal_pandas, goal_weights=goal_weights, time_step=w,
                    time_deadlock_out=time_deadlock_out, avg_sum=vel_avg_tot, state_machine_robots=state_machine_pandas)
            t_rollouts = time.perf_counter() - t0
        # collision spheres of every robot (the simulator's env.collision_links_poses, EXJ:394-396)
        x_spheres = [np.asarray(fk_spheres[i]["fk_fun"](np.append(q[i], 0))).T for i in range(N)]      # [S,3]
        t0 = time.perf_counter()
        action = []
        for i in range(N):
            others = [j for j in range(N) if j != i]
            x_dyn = [x for j in others for x in x_spheres[j]]
            v_dyn = [v for j in others for v in v_robots[j] for _ in range(params.n_obst_per_link)]
            a_dyn = [np.zeros(3)] * len(x_dyn)
            r_dyn = [params.r_robots[j][s] for j in others for s in range(8) for _ in range(params.n_obst_per_link)]
            arguments_robot = dict(q=q[i], qdot=qdot[i], x_goal_0=np.array(goal_pandas[i]), weight_goal_0=goal_weights[i],
                                   angle_goal_1=params.rotation_matrix_pandas[i], x_goal_1=np.array([0.107, 0.0, 0.0]),
                                   weight_goal_1=20.0, x_goal_2=np.array([np.pi / 4]), weight_goal_2=1.0,
                                   x_obsts=x_dyn, radius_obsts=r_dyn, constraint_0=constraints[i],
                                   radius_body_panda_links=params.radius_body_panda_links,
                                   radius_body_panda_hand=np.array([params.radius_sphere]),
                                   x_obsts_dynamic=x_dyn, xdot_obsts_dynamic=v_dyn, xddot_obsts_dynamic=a_dyn,
                                   radius_obsts_dynamic=r_dyn)
            action.append(planners[i].compute_action(**arguments_robot))
        t_actions = (time.perf_counter() - t0) / 2                     # EXJ:450 (sic)
        for i in range(N):                                             # env.step: clip + exact integration (EXJ:452-453)
            a = np.clip(action[i], -limit_vel, limit_vel)
            q[i] = q[i] + params.dt * a
            qdot[i] = a
        solver_times.append(t_actions + t_rollouts)
        traj.append(np.array(q))
    return {"q": np.array(traj), "solver_times": np.array(solver_times),
            "time_in_deadlock": deadlock_prevention.time_in_deadlock if deadlock_prevention else 0}


def device_resident_episode(params, n_steps):
    """The same control loop without the host in it: mrf_episode_run over one scenario (or thousands)."""
    import torch
    from multi_robot_fabrics_amd import abi
    from multi_robot_fabrics_amd.runtime import ControlLoop, FabricHandle
    N = params.nr_robots
    cfg_act = config.panda_config(n_robots=N, horizon=1, dynamic=params.STATIC_OR_DYN_FABRICS, mounts=params.mount_transform)
    links, offs = config.sphere_offsets_per_link(params.n_obst_per_link)
    config.set_spheres(cfg_act, links, offs, [params.radius_sphere] * len(links))
    ha = FabricHandle(cfg_act)
    hr = None
    if params.ROLLOUT_FABRICS:
        cfg_roll = config.panda_config(n_robots=N, horizon=params.N_HORIZON, dynamic=params.STATIC_OR_DYN_FABRICS,
                                       mounts=params.mount_transform)
        cfg_roll.goal_estimate_mask = 0b10 if params.ESTIMATE_GOAL else 0
        hr = FabricHandle(cfg_roll)
    prm = np.zeros((abi.NPARAM, N))
    for i in range(N):
        prm[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + 3, i] = params.start_goals[i]
        prm[abi.P_ANGLE_GOAL_1:abi.P_ANGLE_GOAL_1 + 9, i] = np.asarray(params.rotation_matrix_pandas[i]).ravel()
    prm[abi.P_WEIGHT_GOAL_0], prm[abi.P_WEIGHT_GOAL_1], prm[abi.P_WEIGHT_GOAL_2] = 2.0, 20.0, 1.0
    prm[abi.P_X_GOAL_1:abi.P_X_GOAL_1 + 3] = np.array([[0.107], [0.0], [0.0]])
    prm[abi.P_X_GOAL_2] = np.pi / 4
    prm[abi.P_CONSTRAINT_0:abi.P_CONSTRAINT_0 + 4] = np.array([[0.0], [0.0], [1.0], [-params.z_table]])
    prm[abi.P_RADIUS_BODY:abi.P_RADIUS_BODY + 6] = params.radius_sphere
    q = ha.tensor(np.array([p[:7] for p in params.pos0], dtype=float).T)
    loop = ControlLoop(ha, hr, q, torch.zeros_like(q), ha.tensor(prm), config.PANDA_VEL_LIMITS,
                       deadlock=bool(params.RESOLVE_DEADLOCKS), apply_estimate=bool(params.ESTIMATE_GOAL), stop_margin=-1.0)
    loop.run(n_steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.run(n_steps)
    torch.cuda.synchronize()
    per_step = (time.perf_counter() - t0) / n_steps
    loop2 = ControlLoop(ha, hr, q, torch.zeros_like(q), ha.tensor(prm), config.PANDA_VEL_LIMITS,
                        deadlock=bool(params.RESOLVE_DEADLOCKS), apply_estimate=bool(params.ESTIMATE_GOAL), stop_margin=-1.0)
    loop2.run(n_steps)
    torch.cuda.synchronize()
    return loop2.q.cpu().numpy().T, per_step


def define_run_panda_example(config_path, n_steps=100):
    """EXJ:517-538."""
    params, setup = load_yaml_settings(config_path)
    planners, planners_grasp, goal_structs = define_planners(params)
    utils_class = UtilsKinematics()
    fk_dict = utils_class.define_forward_kinematics(planners=planners, collision_links=params.collision_links,
                                                    collision_links_nrs=params.collision_links_nrs)
    links, offs = config.sphere_offsets_per_link(params.n_obst_per_link)
    sphere_T = []
    for i in range(params.nr_robots):
        per_link = [[np.identity(4) for _ in range(params.n_obst_per_link)] for _ in range(8)]
        for s, off in enumerate(offs):
            per_link[links[s] - 1][s % params.n_obst_per_link][0:3, 3] = off
        sphere_T.append(per_link)
    fk_spheres = utils_class.define_symbolic_collision_link_poses(None, params.collision_links, sphere_T,
                                                                  n_obst_per_link=params.n_obst_per_link,
                                                                  mount_transform=params.mount_transform)
    forwardplanner = define_rollout_planners(params, fk_dict=fk_dict, goal_structs=goal_structs) if params.ROLLOUT_FABRICS else None
    res = run_panda_example(params, n_steps, planners, goal_structs, fk_dict, fk_spheres, forwardplanner)
    q_dev, dev_step = device_resident_episode(params, n_steps)
    st = res["solver_times"][min(10, n_steps - 1):] * 1e3
    ee = UtilsKinematics().define_symbolic_endeffector(planners)
    dist = [float(np.linalg.norm(np.asarray(ee[i]["fk_fun_ee"](res["q"][-1][i])).reshape(-1) - np.array(params.start_goals[i])))
            for i in range(params.nr_robots)]
    return {"config": setup, "steps": n_steps,
            "solver_time_ms_mean": float(st.mean()), "solver_time_ms_median": float(np.median(st)),
            "reference_recorded_solver_time_ms": {"K=1": 8.457, "K=10": 40.416, "K=20": 76.270,
                                                  "note": "2 Pandas, n_obst_per_link=1, hardware unknown (BASELINE.md)"},
            "device_resident_ms_per_control_step": 1e3 * dev_step,
            "host_api_vs_device_episode_max_abs_dq": float(np.abs(res["q"][-1] - q_dev).max()),
            "ee_distance_to_goal_m": dist, "time_in_deadlock_steps": int(res["time_in_deadlock"])}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(ROOT, "examples", "configs", "panda_config.yaml"))
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    print(json.dumps(define_run_panda_example(args.config, args.steps), indent=1))
