#!/usr/bin/env python3
"""The reference's examples/example_pandas_Jointspace.py on the HIP kernels, function by function and with the same
signatures (EXJ = the reference file):

    define_run_panda_example(n_steps=100, render=True)                                         EXJ:517-538
        panda_config.yaml -> manipulator_parameters -> create_manipulators_simulation -> define_planners
        -> UtilsKinematics.define_forward_kinematics -> define_rollout_planners -> run_panda_example
    run_panda_example(params, n_steps, planners, planners_grasp, goal_structs, env, fk_dict, forwardplanner) -> dict
        per control step: observation -> pick-and-place state machine -> (RF-CV goal estimate) ->
        get_velocity_rollouts -> deadlock_checking -> obstacle assembly -> compute_action(**kwargs) of the main or the
        grasp planner per robot -> gripper action -> env.step                                   EXJ:195-515

Two pandas (or three) pick their cubes and carry them home; the other robot's collision spheres are dynamic obstacles
of each planner, and Rollout Fabrics in JOINT space (the coupled H-step forward simulation) feed the deadlock logic.

The planner classes are this package's mirrors of `fabrics` / `forwardkinematics` / `mpscenes` (every numeric call is a
launch of csrc/libmrf_hip.so); the simulator is the kinematic stand-in of multi-robot-fabrics_amd/scene.py -- there is
no pybullet and no renderer here, `render=True` is ignored with a warning.  The result dictionary carries the
reference's keys (EXJ:509-515).  Deviations from the reference file, all documented where they occur: the RF-CV goal
estimate uses the hand VELOCITY J qdot (the Cartesian driver's form, EXC:355-357; EXJ:329 feeds the flattened Jacobian),
and RESOLVE_DEADLOCKS=0 switches the deadlock logic off (EXJ:379 calls it regardless of the flag).

usage: python examples/example_pandas_Jointspace.py [--steps 7000] [--device-episode]
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import yaml

import examples.parameters_manipulators as parameters_manipulators
from examples.simulation_environments import create_simulation_manipulators
from multi_robot_fabrics_amd import config
from multi_robot_fabrics_amd.deadlock import deadlockprevention
from multi_robot_fabrics_amd.goals import GoalComposition
from multi_robot_fabrics_amd.kinematics import GenericURDFFk, UtilsKinematics
from multi_robot_fabrics_amd.pick_place import StateMachine
from multi_robot_fabrics_amd.planner import ParameterizedFabricPlanner
from multi_robot_fabrics_amd.rollouts import ForwardFabricsPlanner

CONFIG_PATH = os.path.join("examples", "configs", "panda_config.yaml")       # EXJ:518, relative to the repository root


def create_dummy_goal_panda() -> GoalComposition:
    """EXJ:25-62: a goal whose numbers are replaced through the planner's parameters at run time."""
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


def set_planner_panda(degrees_of_freedom: int = 7, nr_obst=0, nr_obst_dyn=1, collision_links_nr=[5], urdf_links={},
                      mount_param={}, i_robot=0):
    """EXJ:64-134.  The URDF named by urdf_links["URDF_file_panda"] is read and CHECKED against the chain compiled into
    the kernels (GenericURDFFk refuses one that disagrees); the mount is Rz(yaw_i) at mount_param["mount_positions"][i]."""
    with open(urdf_links["URDF_file_panda"], "r") as file:
        urdf = file.read()
    goal = create_dummy_goal_panda()
    fk = GenericURDFFk(urdf, "panda_link0", "panda_leftfinger")
    planner = ParameterizedFabricPlanner(
        degrees_of_freedom, fk,
        geometry_plane_constraint="10*(1/(1+1*ca.exp(-10*x))-1) * (xdot**2)",
        collision_geometry="-0.5 / (x ** 4) * (xdot ** 2)",
        collision_finsler="0.01/(x**4) * xdot**2",
    )
    collision_links = ["panda_link" + str(l) if l < 9 else "panda_hand" for l in collision_links_nr]
    panda_limits = [[-2.8973, 2.8973], [-1.7628, 1.7628], [-2.8973, 2.8973], [-3.0718, -0.0698], [-2.8973, 2.8973],
                    [-0.0175, 3.7525], [-2.8973, 2.8973]]
    angle_rot = np.pi if i_robot in (1, 2) else 0.0                           # EXJ:108-113
    T_0 = np.identity(4)
    T_0[0:2, 0:2] = np.array([[np.cos(angle_rot), -np.sin(angle_rot)], [np.sin(angle_rot), np.cos(angle_rot)]])
    T_0[0:3, 3] = mount_param["mount_positions"][i_robot]
    planner._forward_kinematics.set_mount_transformation(T_0)
    planner.set_components(collision_links=collision_links, goal=goal, number_obstacles=nr_obst,
                           number_dynamic_obstacles=nr_obst_dyn, dynamic_obstacle_dimension=3,
                           number_plane_constraints=1, limits=panda_limits)
    planner.concretize(mode="vel", time_step=0.01)       # the planner's output is a joint VELOCITY
    return planner, goal


def define_planners(params):
    """EXJ:136-170: a main planner per robot and a "grasp" planner without collision links (pregrasp -> grasp)."""
    if params.STATIC_OR_DYN_FABRICS == 0:
        nr_obst_planners, nr_obst_dyn_planners = params.nr_obsts_dyn_all, [0] * params.nr_robots
    else:
        nr_obst_planners, nr_obst_dyn_planners = [0] * params.nr_robots, params.nr_obsts_dyn_all
    planners, goal_structs, planners_grasp = [], [], []
    for i_robot in range(params.nr_robots):
        planner_i, goal_struct_i = set_planner_panda(degrees_of_freedom=params.dof[i_robot], nr_obst=nr_obst_planners[i_robot],
                                                     nr_obst_dyn=nr_obst_dyn_planners[i_robot],
                                                     collision_links_nr=params.collision_links_nrs[i_robot],
                                                     urdf_links=params.urdf_links, mount_param=params.mount_param,
                                                     i_robot=i_robot)
        planner_grasp_i, _ = set_planner_panda(degrees_of_freedom=params.dof[i_robot], nr_obst=i_robot, nr_obst_dyn=i_robot,
                                               collision_links_nr=[], urdf_links=params.urdf_links,
                                               mount_param=params.mount_param, i_robot=i_robot)
        planners.append(planner_i)
        goal_structs.append(goal_struct_i)
        planners_grasp.append(planner_grasp_i)
    return planners, planners_grasp, goal_structs


def define_rollout_planners(params, fk_dict=None, goal_structs=None, n_steps=100):
    """EXJ:172-193: the rollout planners see one sphere per link of the other robots."""
    planners_rollout = []
    for i_robot in range(params.nr_robots):
        planner_i, _ = set_planner_panda(degrees_of_freedom=params.dof[i_robot], nr_obst=params.nr_obsts[i_robot],
                                         nr_obst_dyn=params.nr_obsts_dyn[i_robot],
                                         collision_links_nr=params.collision_links_nrs[i_robot],
                                         urdf_links=params.urdf_links, mount_param=params.mount_param, i_robot=i_robot)
        planners_rollout.append(planner_i)
    forwardplanner = ForwardFabricsPlanner(params=params, planners=planners_rollout, N_steps=n_steps, fk_dict=fk_dict,
                                           goal_struct_robots=goal_structs)
    forwardplanner.forward_multi_fabrics_symbolic()
    return forwardplanner


def run_panda_example(params, n_steps=5000, planners=[], planners_grasp=[], goal_structs=[], env=None, fk_dict=None,
                      forwardplanner=None) -> dict:
    """EXJ:195-515: the control loop; returns the reference's dictionary of evaluation metrics."""
    dof = params.dof
    n_steps_panda, n_steps_panda2 = np.nan, np.nan
    success = [False, False]
    step_times, solver_times = [], []
    min_clearance = 100
    constraints = [np.array([0, 0, 1, 0.0 - params.mount_param["z_table"]])] * params.nr_robots
    nr_robots = len(params.collision_links_nrs)
    dof_index = [0]
    for i_robot in range(nr_robots):
        dof_index.append(dof_index[i_robot] + dof[0] + 2)
    limit_vel_panda = np.array([2.1750, 2.1750, 2.1750, 2.1750, 2.61, 2.61, 2.61])                   # EXJ:221
    limits_action = np.concatenate([np.concatenate((limit_vel_panda, np.array([2, 2]))) for _ in range(nr_robots)])

    action = np.zeros(env.n())
    ob, *_ = env.step(action)

    # hand position / velocity functions (EXJ:229-238; `define_symbolic_endeffector` is what the Cartesian driver uses)
    fk_endeff = UtilsKinematics().define_symbolic_endeffector(planners)
    fk_fun_endeff = [fk_endeff[i]["fk_fun_ee"] for i in range(nr_robots)]
    env.reconfigure_camera(2.5, -5., -42., (0.3, 1., -0.5))

    deadlock_prevention = deadlockprevention(dof, params.nr_robots, params.N_HORIZON) if params.ROLLOUT_FABRICS else None
    state_machines = [StateMachine(start_goal=params.start_goals[i], nr_robots=nr_robots, nr_blocks=params.n_cubes / nr_robots,
                                   fk_fun_ee=fk_fun_endeff[i], robot_types=params.robot_types) for i in range(nr_robots)]

    q_pandas = [[] for _ in range(nr_robots)]
    qdot_pandas = [[] for _ in range(nr_robots)]
    q_pandas_gripper = [[] for _ in range(nr_robots)]
    state_machine_pandas = [[] for _ in range(nr_robots)]
    goal_pandas = [[] for _ in range(nr_robots)]
    goal_weights = [[] for _ in range(nr_robots)]
    goal_pandas_block = [[] for _ in range(nr_robots)]
    time_deadlock_out = 1000
    q_robots_N = None

    for w in range(n_steps):
        t_start_loop = time.perf_counter()
        # --- states and the cube each robot is after (EXJ:284-297) ---
        first_index = list(ob["robot_0"]["FullSensor"]["obstacles"].keys())[0]
        per_robot = params.n_cubes / nr_robots
        for i_robot in range(nr_robots):
            ob_i = ob["robot_" + str(i_robot)]
            q_pandas[i_robot] = ob_i["joint_state"]["position"][0:dof[0]]
            q_pandas_gripper[i_robot] = ob_i["joint_state"]["position"][dof[0]:dof[0] + 2]
            qdot_pandas[i_robot] = np.clip(ob_i["joint_state"]["velocity"][0:dof[0]], -limit_vel_panda, limit_vel_panda)
            picked = state_machines[i_robot].get_nr_blocks_picked()
            if picked < per_robot:
                goal_pandas_block[i_robot] = copy.deepcopy(
                    ob["robot_0"]["FullSensor"]["obstacles"][first_index + picked + int(i_robot * per_robot)]["position"])
                goal_pandas_block[i_robot][2] += 0.1
        # --- state machine (EXJ:299-316) ---
        for i_robot in range(nr_robots):
            state_machine_pandas[i_robot] = state_machines[i_robot].get_state_machine_panda(
                q_robot=q_pandas[i_robot], q_robot_gripper=q_pandas_gripper[i_robot], goal_block=goal_pandas_block[i_robot],
                robot_type="panda")
        if state_machine_pandas[0] == 10 and not success[0]:
            n_steps_panda, success[0] = w, True
        if state_machine_pandas[1] == 10 and not success[1]:
            n_steps_panda2, success[1] = w, True
        if all(state_machine_pandas[i] == 10 for i in range(nr_robots)):
            break
        for i_robot in range(nr_robots):
            goal_pandas[i_robot] = state_machines[i_robot].get_goal_robot()
            goal_weights[i_robot] = state_machines[i_robot].get_weight_goal0()

        # --- link-origin positions / velocities of every robot (EXJ:318-343) ---
        x_robots = [[] for _ in range(nr_robots)]
        v_robots = [[] for _ in range(nr_robots)]
        x_robots_ee, v_robots_ee = [], []
        for i_robot in range(nr_robots):
            q_num = ob["robot_" + str(i_robot)]["joint_state"]["position"][0:dof[i_robot]]
            q_dot_num = ob["robot_" + str(i_robot)]["joint_state"]["velocity"][0:dof[i_robot]]
            x_robots_ee.append(fk_endeff[i_robot]["fk_fun_ee"](q_num).full().transpose()[0])
            v_robots_ee.append(fk_endeff[i_robot]["vel_fun_ee"](q_num, q_dot_num).full().transpose()[0])
            for i_link in range(len(params.collision_links_nrs[i_robot])):
                x_robots[i_robot].append(fk_dict["fk_fun"][i_robot][i_link](q_num).full().transpose()[0])
                if params.STATIC_OR_DYN_FABRICS == 0:
                    v_robots[i_robot].append(np.zeros((3,)))
                else:
                    v_robots[i_robot].append(np.asarray(fk_dict["jac_fun"][i_robot][i_link](q_num) @ q_dot_num).reshape(-1))
        # --- RF-CV: robot 1's goal is not communicated but extrapolated from its hand (EXJ:345-348) ---
        if params.ESTIMATE_GOAL:
            goal_pandas[1] = x_robots_ee[1] + 20 * 0.01 * v_robots_ee[1]

        t_rollouts = 0
        if params.ROLLOUT_FABRICS:
            t_start_rollouts = time.perf_counter()
            inputs_action = {"q_robots": q_pandas, "q_dot_robots": qdot_pandas, "x_obsts": [[] * nr_robots],
                             "x_goals0": goal_pandas,
                             "x_goals1": [goal_structs[i]._config.subgoal1.desired_position for i in range(nr_robots)],
                             "x_goals2": [goal_structs[i]._config.subgoal2.desired_position for i in range(nr_robots)],
                             "weight_goals0": goal_weights,
                             "weight_goals1": [goal_structs[i]._config.subgoal1.weight for i in range(nr_robots)],
                             "weight_goals2": [goal_structs[i]._config.subgoal2.weight for i in range(nr_robots)],
                             "constraints": constraints}
            if params.ROLLOUTS_PLOTTING:       # EXJ:371-372 (called there with keywords the method does not take)
                q_robots_N, q_dot_robots_N, q_ddot_robots_N = forwardplanner.rollouts_numerical(inputs_action=inputs_action)
            vel_avg = forwardplanner.get_velocity_rollouts(inputs_action=inputs_action)
            vel_avg_tot = sum(vel_avg) / nr_robots
            if params.RESOLVE_DEADLOCKS:
                goal_pandas, goal_weights, time_deadlock_out = deadlock_prevention.deadlock_checking(
                    x_robots=x_robots_ee, goal_robots=goal_pandas, goal_weights=goal_weights, time_step=w,
                    time_deadlock_out=time_deadlock_out, avg_sum=vel_avg_tot, state_machine_robots=state_machine_pandas)
            t_rollouts = time.perf_counter() - t_start_rollouts

        # --- obstacles of every planner: the collision spheres of the other robots (EXJ:394-412) ---
        x_dyns_obsts = [[] for _ in range(nr_robots)]
        v_dyns_obsts = [[] for _ in range(nr_robots)]
        a_dyns_obsts = [[] for _ in range(nr_robots)]
        r_dyns_obsts = [[] for _ in range(nr_robots)]
        x_dyns_obsts_per_robot = [[] for _ in range(nr_robots)]
        env.update_collision_links()
        x_collision_sphere_poses = env.collision_links_poses(position_only=True)
        for i_robot in range(nr_robots):
            x_dyns_obsts_per_robot[i_robot] = [x for key, x in x_collision_sphere_poses.items() if str(i_robot) in key[0]]
            for i_other_robot in (i for i in range(nr_robots) if i != i_robot):
                x_dyns_obsts[i_other_robot] = x_dyns_obsts[i_other_robot] + x_dyns_obsts_per_robot[i_robot]
                for i_sphere in range(len(v_robots[i_other_robot])):
                    v_dyns_obsts[i_robot] = v_dyns_obsts[i_robot] + [v_robots[i_other_robot][i_sphere]] * params.n_obst_per_link
                    a_dyns_obsts[i_robot] = a_dyns_obsts[i_robot] + [np.zeros((3,))] * params.n_obst_per_link
                    r_dyns_obsts[i_robot] = r_dyns_obsts[i_robot] + [params.r_robots[i_other_robot][i_sphere]] * params.n_obst_per_link

        t_start_actions = time.perf_counter()
        # --- actions (EXJ:416-448) ---
        for i_robot in range(nr_robots):
            lo = dof_index[i_robot]
            if state_machine_pandas[i_robot] == 3 or state_machine_pandas[i_robot] == 5:
                action[lo:lo + dof[i_robot]] = np.zeros(dof[0])
            else:
                arguments_robot = dict(q=q_pandas[i_robot], qdot=qdot_pandas[i_robot],
                                       x_goal_0=np.array(goal_pandas[i_robot]), weight_goal_0=goal_weights[i_robot],
                                       angle_goal_1=params.rotation_matrix_pandas[i_robot],
                                       x_goal_1=np.array([0.107, 0.0, 0.0]), weight_goal_1=20.0,
                                       x_goal_2=np.array([np.pi / 4]), weight_goal_2=1.0,
                                       x_obsts=x_dyns_obsts[i_robot], radius_obsts=r_dyns_obsts[i_robot],
                                       constraint_0=constraints[i_robot],
                                       radius_body_panda_links=params.radius_body_panda_links,
                                       radius_body_panda_hand=np.array([params.radius_sphere]),
                                       x_obsts_dynamic=x_dyns_obsts[i_robot], xdot_obsts_dynamic=v_dyns_obsts[i_robot],
                                       xddot_obsts_dynamic=a_dyns_obsts[i_robot], radius_obsts_dynamic=r_dyns_obsts[i_robot])
                if state_machine_pandas[i_robot] == 2:       # descending onto the cube: goal reaching only
                    action[lo:lo + dof[i_robot]] = planners_grasp[i_robot].compute_action(**arguments_robot)
                else:
                    action[lo:lo + dof[i_robot]] = planners[i_robot].compute_action(**arguments_robot)
            action[lo + dof[i_robot]:dof_index[i_robot + 1]] = state_machines[i_robot].get_gripper_action_panda(q_pandas_gripper[i_robot])
        t_actions = (time.perf_counter() - t_start_actions) / 2                                     # EXJ:450 (sic)

        action = np.clip(action, -limits_action, limits_action)
        ob, *_ = env.step(action)
        t_end_loop = time.perf_counter()
        solver_times = np.append(solver_times, t_actions + t_rollouts)
        step_times = np.append(step_times, t_end_loop - t_start_loop)

        # --- sphere clearance between robots 0 and 1 (EXJ:460-470; the reference indexes both radii with k) ---
        for k, x_panda_1 in enumerate(x_dyns_obsts_per_robot[0]):
            for j, x_panda_2 in enumerate(x_dyns_obsts_per_robot[1]):
                dist_x_r = np.linalg.norm(x_panda_1 - x_panda_2, 2) - r_dyns_obsts[0][k] - r_dyns_obsts[1][k]
                if dist_x_r < min_clearance:
                    min_clearance = dist_x_r

    solver_times, step_times = np.asarray(solver_times, dtype=float), np.asarray(step_times, dtype=float)
    nan = float("nan")
    return {"success_rate": state_machines[-1].get_success_rate(),                                  # EXJ:506 (last robot's)
            "n_steps_panda": n_steps_panda, "n_steps_robot2": n_steps_panda2,
            "step_time_mean": float(np.mean(step_times)) if len(step_times) else nan,
            "step_time_std": float(np.std(step_times)) if len(step_times) else nan,
            "total_time": max([n_steps_panda, n_steps_panda2]) * 0.01, "dt": params.dt,
            "solver_time_mean": float(np.mean(solver_times)) if len(solver_times) else nan,
            "solver_time_std": float(np.std(solver_times)) if len(solver_times) else nan,
            "min clearance": min_clearance, "solver_times": solver_times,
            # extras of this build, beside the reference's keys
            "control_steps": int(len(solver_times)),
            "blocks_picked": [m.get_nr_blocks_picked() for m in state_machines],
            "states": [int(s) for s in state_machine_pandas],
            "time_in_deadlock_steps": int(deadlock_prevention.time_in_deadlock) if deadlock_prevention else 0,
            "q_final": np.array([ob["robot_%d" % i]["joint_state"]["position"][0:7] for i in range(nr_robots)]),
            "rollout_plot_data": q_robots_N}


def device_resident_episode(params, n_steps, cubes, dynamic_action=None):
    """The same control loop without the host in it: mrf_episode_run (runtime.ControlLoop) with the pick-and-place state
    machine, the grasp planner and the minimal cube / gripper model on the device.  Returns q [N,7] after n_steps and the
    time per control step of a second, warmed-up run."""
    import torch
    from multi_robot_fabrics_amd import abi
    from multi_robot_fabrics_amd.runtime import ControlLoop, FabricHandle
    N = params.nr_robots
    dyn = 1 if params.STATIC_OR_DYN_FABRICS else 0
    dyn_act = dyn if dynamic_action is None else int(dynamic_action)      # what the MAIN planners were built with
    cfg_act = config.panda_config(n_robots=N, horizon=1, dynamic=dyn_act, mounts=params.mount_transform)
    links, offs = config.sphere_offsets_per_link(params.n_obst_per_link)
    config.set_spheres(cfg_act, links, offs, [params.radius_sphere] * len(links))
    cfg_grasp = config.panda_config(n_robots=N, horizon=1, dynamic=dyn_act, n_ego=0, mounts=params.mount_transform)
    ha, hg, hr = FabricHandle(cfg_act), FabricHandle(cfg_grasp), None
    if params.ROLLOUT_FABRICS:
        cfg_roll = config.panda_config(n_robots=N, horizon=params.N_HORIZON, dynamic=dyn, mounts=params.mount_transform)
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
    per_robot = int(params.n_cubes / N)
    blocks = np.zeros((per_robot, 3, N))
    for i in range(N):
        for b in range(per_robot):
            blocks[b, :, i] = np.asarray(cubes[b + i * per_robot], dtype=float) + np.array([0.0, 0.0, 0.1])   # EXJ:297
    start = np.array(params.start_goals, dtype=float).T
    q0 = np.array([np.asarray(p, dtype=float)[:7] for p in params.pos0]).T
    grip0 = np.array([[np.asarray(p, dtype=float)[7] if len(p) > 7 else 0.02 for p in params.pos0]] * 2)

    def loop():
        q = ha.tensor(q0)
        return ControlLoop(ha, hr, q, torch.zeros_like(q), ha.tensor(prm), config.PANDA_VEL_LIMITS,
                           deadlock=bool(params.RESOLVE_DEADLOCKS), apply_estimate=bool(params.ESTIMATE_GOAL), stop_margin=-1.0,
                           pick_place=dict(start_goal=ha.tensor(start), blocks=ha.tensor(blocks), nr_blocks=per_robot,
                                           q_gripper=ha.tensor(grip0), model=1, h_grasp=hg))
    first = loop()
    first.run(n_steps)
    torch.cuda.synchronize()
    timed = loop()
    t0 = time.perf_counter()
    timed.run(n_steps)
    torch.cuda.synchronize()
    return first.q.cpu().numpy().T, (time.perf_counter() - t0) / max(1, n_steps)


def define_run_panda_example(n_steps=100, render=True, *, config_path=None, device_episode=False):
    """EXJ:517-538.  Keyword-only extras of this build: `config_path` (another YAML with the same eight keys) and
    `device_episode` (also run the configuration as a device-resident episode and report how the two compare)."""
    path = config_path if config_path is not None else (CONFIG_PATH if os.path.exists(CONFIG_PATH) else os.path.join(ROOT, CONFIG_PATH))
    with open(path, "r") as setup_stream:
        setup = yaml.safe_load(setup_stream)
    random_scene = False
    nr_robots = setup["n_robots"]
    param = parameters_manipulators.manipulator_parameters(nr_robots=nr_robots, n_obst_per_link=setup["n_obst_per_link"])
    simulation_class = create_simulation_manipulators.create_manipulators_simulation(param)
    utils_class = UtilsKinematics()
    random_obstacles = simulation_class.create_scene(random_scene=random_scene, n_cubes=param.n_cubes)
    env = simulation_class.initialize_environment(render=render, random_scene=random_scene, obstacles=random_obstacles)
    planners, planners_grasp, goal_structs = define_planners(params=param)
    fk_dict = utils_class.define_forward_kinematics(planners=planners, collision_links=param.collision_links,
                                                    collision_links_nrs=param.collision_links_nrs)
    param.define_settings(ROLLOUT_FABRICS=setup["ROLLOUT_FABRICS"], ROLLOUTS_PLOTTING=setup["ROLLOUTS_PLOTTING"],
                          STATIC_OR_DYN_FABRICS=setup["STATIC_OR_DYN_FABRICS"], RESOLVE_DEADLOCKS=setup["RESOLVE_DEADLOCKS"],
                          ESTIMATE_GOAL=setup["ESTIMATE_GOAL"], N_HORIZON=setup["N_HORIZON"],
                          n_obst_per_link=setup["n_obst_per_link"])
    # NOTE the reference builds the planners BEFORE define_settings (EXJ:527-531), i.e. with the constructor's
    # STATIC_OR_DYN_FABRICS = 0: its main planners are always the static-obstacle ones.  Kept as is.
    forwardplanner = define_rollout_planners(params=param, fk_dict=fk_dict, goal_structs=goal_structs) if param.ROLLOUT_FABRICS else None
    res = run_panda_example(params=param, n_steps=n_steps, planners=planners, planners_grasp=planners_grasp,
                            goal_structs=goal_structs, env=env, fk_dict=fk_dict, forwardplanner=forwardplanner)
    env.close()
    res["config"] = setup
    if device_episode:
        cubes = [np.asarray(o._config.geometry.position, dtype=float) for o in random_obstacles]
        cubes = [np.array([c[0], c[1], param.z_table + 0.025]) for c in cubes]              # settled on the table top
        q_dev, dev_step = device_resident_episode(param, res["control_steps"], cubes,
                                                  dynamic_action=planners[0]._components["n_dynamic"] > 0)
        res["device_resident_ms_per_control_step"] = 1e3 * dev_step
        res["host_api_vs_device_episode_max_abs_dq"] = float(np.abs(res["q_final"] - q_dev).max())
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=7000)
    ap.add_argument("--config", default=None)
    ap.add_argument("--device-episode", action="store_true")
    args = ap.parse_args()
    res = define_run_panda_example(n_steps=args.steps, render=True, config_path=args.config, device_episode=args.device_episode)
    print(json.dumps({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in res.items() if k != "solver_times"}, indent=1))
