#!/usr/bin/env python3
"""The planner side of the reference's examples/example_pandas_cartesian.py on the HIP kernels: the Cartesian variant of
Rollout Fabrics, in which every robot rolls out ITS OWN fabric against the other robots' collision spheres moving with
constant velocity (FPC:421-458) -- no coupling between the rollouts, one FabricsRollouts object per robot --

    manipulator_parameters -> define_planners / define_rollout_planners (EXC:124-192)
        -> run_panda_example (EXC:194-520): per control step
             state machine -> compute_x_obsts_dyn_0 / compute_endeffector -> RF-CV goal estimate
             -> define_arguments_numerical + get_velocity_rollouts per robot -> deadlock_checking
             -> compute_action(**kwargs) of the main or the grasp planner per robot -> gripper action -> env.step

with the reference's own call surface.  Here the pick-and-place loop is complete: the state machine of
others_planner/state_machine.py sequences pregrasp / grasp / lift / carry / release for `n_cubes` blocks.  What the
reference gets from pybullet is replaced by the same minimal model the device-resident episodes use:
env.step integrates the clipped velocity command exactly, finger joints follow their velocity command, a block travels
with the closed gripper and stays where it is released, collision-sphere centres come from the sphere forward
kinematics.

usage: python examples/example_pandas_cartesian.py [--robots 2] [--steps 6000] [--horizon 10] [--no-rollouts]
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import numpy as np

from example_pandas_jointspace import define_planners                                     # EXC:124-158 = EXJ:136-170
from multi_robot_fabrics_amd import config
from multi_robot_fabrics_amd.deadlock import deadlockprevention
from multi_robot_fabrics_amd.kinematics import UtilsKinematics, compute_endeffector, compute_x_obsts_dyn_0
from multi_robot_fabrics_amd.parameters import manipulator_parameters
from multi_robot_fabrics_amd.pick_place import StateMachine
from multi_robot_fabrics_amd.rollouts import FabricsRollouts


def define_rollout_planners(params, goal_structs, planners):
    """EXC:160-192: one independent rollout object per robot, built on that robot's main planner."""
    v_obsts_dyn = [np.zeros((3,))] * params.nr_obsts_dyn_all[0]
    forwardplanners = []
    for i in range(params.nr_robots):
        fp = FabricsRollouts(N=params.N_HORIZON, dt=params.dt, nx=params.dof[i] * 2, nu=params.dof[i], dof=params.dof[i],
                             nr_obsts=params.nr_obsts[i], bool_ring=False, nr_obsts_dyn=params.nr_obsts_dyn_all[i],
                             v_obsts_dyn=v_obsts_dyn, fabrics_mode=params.fabrics_mode,
                             collision_links_nrs=params.collision_links_nrs[i], nr_constraints=params.nr_constraints[i],
                             radius_sphere=params.radius_sphere, constraints=params.constraints[i],
                             nr_goals=len(goal_structs[i]._config))
        fp.symbolic_forward_fabrics(planner=planners[i], goal_struct=goal_structs[i])
        forwardplanners.append(fp)
    return forwardplanners


def block_positions(params, rng):
    """Cubes on the table inside each robot's reach (the reference places them in its simulation scene)."""
    per_robot = params.n_cubes // params.nr_robots
    out = []
    for i in range(params.nr_robots):
        T = np.asarray(params.mount_transform[i])
        yaw = np.arctan2(T[1, 0], T[0, 0])
        for _ in range(per_robot):
            r, a = rng.uniform(0.4, 0.55), yaw + rng.uniform(-0.8, 0.8)
            out.append(np.array([T[0, 3] + r * np.cos(a), T[1, 3] + r * np.sin(a), params.z_table + 0.025]))
    return out


def run_panda_example(params, n_steps, planners, planners_grasp, goal_structs, forwardplanners, fk_dict_spheres, seed=0):
    """EXC:194-520 without the simulator."""
    N, dof = params.nr_robots, params.dof
    limit_vel = np.array(config.PANDA_VEL_LIMITS)
    fk_endeff = UtilsKinematics().define_symbolic_endeffector(planners)
    q = [np.array(params.pos0[i][:7], dtype=float) for i in range(N)]
    qdot = [np.zeros(7) for _ in range(N)]
    q_gripper = [np.array([0.04, 0.04]) for _ in range(N)]
    blocks = block_positions(params, np.random.default_rng(seed))
    per_robot = params.n_cubes // N
    held = [None] * N                                               # index of the block the closed gripper carries
    state_machines = [StateMachine(start_goal=params.start_goals[i], nr_robots=N, nr_blocks=per_robot,
                                   fk_fun_ee=fk_endeff[i]["fk_fun_ee"], robot_types=["panda"] * N) for i in range(N)]
    deadlock_prevention = deadlockprevention(dof, N, params.N_HORIZON)
    if params.ROLLOUT_FABRICS:
        for i in range(N):
            forwardplanners[i].preset_radii_obsts_dyn(radii_obst_dyn=params.r_dyns_obsts[i])
    weight_goals = {"robot_%d" % i: {} for i in range(N)}
    x_goals = {"robot_%d" % i: {} for i in range(N)}
    time_deadlock_out = 1000
    success_step = [None] * N
    solver_times, min_clearance, states_seen = [], 100.0, [set() for _ in range(N)]
    for w in range(n_steps):
        # --- state machine (EXC:297-321) ---
        state = [0] * N
        for i in range(N):
            picked = state_machines[i].get_nr_blocks_picked()
            goal_block = np.zeros(3)
            if picked < per_robot:
                goal_block = copy.deepcopy(blocks[picked + i * per_robot])
                goal_block[2] += 0.1                                # hand target above the cube (EXC:309)
            state[i] = state_machines[i].get_state_machine_panda(q_robot=q[i], q_robot_gripper=q_gripper[i],
                                                                 goal_block=goal_block, robot_type="panda")
            states_seen[i].add(int(state[i]))
            if state[i] == 10 and success_step[i] is None:
                success_step[i] = w
        if all(s == 10 for s in state):
            break
        for i in range(N):                                          # EXC:323-336
            key = "robot_%d" % i
            cfg_goal = goal_structs[i]._config
            x_goals[key] = {"subgoal0": state_machines[i].get_goal_robot(),
                            "subgoal1": cfg_goal["subgoal1"]["desired_position"],
                            "subgoal2": cfg_goal["subgoal2"]["desired_position"]}
            weight_goals[key] = {"subgoal0": state_machines[i].get_weight_goal0(),
                                 "subgoal1": cfg_goal["subgoal1"]["weight"], "subgoal2": cfg_goal["subgoal2"]["weight"]}
        # --- obstacle spheres of the other robots and end effectors (EXC:338-352) ---
        poses = {}
        for i in range(N):
            xs = np.asarray(fk_dict_spheres[i]["fk_fun"](np.append(q[i], 0))).T
            for s, x in enumerate(xs):
                poses[("robot_%d" % i, s)] = x
        x_dyn, v_dyn, x_per_robot = compute_x_obsts_dyn_0(q_robots=q, qdot_robots=qdot, x_collision_sphere_poses=poses,
                                                          nr_robots=N, fk_dict_spheres=fk_dict_spheres,
                                                          nr_dyn_obsts=params.nr_obsts_dyn_all)
        x_ee, v_ee = compute_endeffector(q, qdot, fk_endeff, nr_robots=N)
        if params.ESTIMATE_GOAL:                                    # EXC:355-357
            x_goals["robot_1"]["subgoal0"] = x_ee[1] + 20 * 0.01 * v_ee[1]
        # --- rollouts and the deadlock logic on their velocity signal (EXC:361-423) ---
        t_rollouts = 0.0
        if params.ROLLOUT_FABRICS:
            t0 = time.perf_counter()
            arguments = [forwardplanners[i].define_arguments_numerical(
                q_robot=q[i], q_dot_robot=qdot[i], constraints=params.constraints[i],
                weight_goals=weight_goals["robot_%d" % i], x_goals=x_goals["robot_%d" % i], x_obsts=[],
                x_obsts_dyn=x_dyn[i], v_obsts_dyn=v_dyn[i]) for i in range(N)]
            vel_avg = [forwardplanners[i].get_velocity_rollouts(arguments[i]).full()[0] for i in range(N)]
            t_rollouts = time.perf_counter() - t0
            if params.RESOLVE_DEADLOCKS:
                goal_d, weight_d, time_deadlock_out = deadlock_prevention.deadlock_checking(
                    x_robots=x_ee, goal_robots=[x_goals["robot_%d" % i]["subgoal0"] for i in range(N)],
                    goal_weights=[weight_goals["robot_%d" % i]["subgoal0"] for i in range(N)], time_step=w,
                    time_deadlock_out=time_deadlock_out, avg_sum=sum(vel_avg) / N, state_machine_robots=state)
                for i in range(N):
                    x_goals["robot_%d" % i]["subgoal0"] = goal_d[i]
                    weight_goals["robot_%d" % i]["subgoal0"] = weight_d[i]
        # --- actions (EXC:425-462) ---
        t0 = time.perf_counter()
        action, grip_action = [], []
        for i in range(N):
            key = "robot_%d" % i
            if state[i] in (3, 5):
                action.append(np.zeros(7))
            else:
                arguments_robot = dict(
                    q=q[i], qdot=qdot[i], x_goal_0=np.array(x_goals[key]["subgoal0"]),
                    x_goal_1=np.array(x_goals[key]["subgoal1"]), x_goal_2=np.array(x_goals[key]["subgoal2"]),
                    weight_goal_0=weight_goals[key]["subgoal0"], weight_goal_1=weight_goals[key]["subgoal1"],
                    weight_goal_2=weight_goals[key]["subgoal2"], angle_goal_1=params.rotation_matrix_pandas[i],
                    x_obsts=x_dyn[i], radius_obsts=params.r_dyns_obsts[i], constraint_0=params.constraints[i],
                    radius_body_panda_links=params.radius_body_panda_links,
                    radius_body_panda_hand=np.array([params.radius_sphere]), x_obsts_dynamic=x_dyn[i],
                    xdot_obsts_dynamic=v_dyn[i], xddot_obsts_dynamic=params.a_dyns_obsts[i],
                    radius_obsts_dynamic=params.r_dyns_obsts[i])
                planner = planners_grasp[i] if state[i] == 2 else planners[i]      # descending: goal reaching only
                action.append(planner.compute_action(**arguments_robot))
            grip_action.append(state_machines[i].get_gripper_action_panda(q_gripper[i]))
        solver_times.append((time.perf_counter() - t0) / 2 + t_rollouts)           # EXC:464 (sic)
        # --- env.step: clip, integrate, fingers, blocks (EXC:466-468 + the minimal scene model) ---
        for i in range(N):
            a = np.clip(action[i], -limit_vel, limit_vel)
            q[i] = q[i] + params.dt * a
            qdot[i] = a
            q_gripper[i] = np.clip(q_gripper[i] + params.dt * np.asarray(grip_action[i], dtype=float), 0.0, 0.04)
            hand = np.asarray(fk_endeff[i]["fk_fun_ee"](q[i])).reshape(-1)
            closed = state_machines[i].get_gripper_status() == "closed"
            picked = state_machines[i].get_nr_blocks_picked()
            if closed and held[i] is None and picked < per_robot and state[i] in (3, 12, 4):
                held[i] = picked + i * per_robot
            if not closed:
                held[i] = None
            if held[i] is not None:
                blocks[held[i]] = hand - np.array([0.0, 0.0, 0.1])
        for xa in x_per_robot[0]:                                                   # EXC:474-481
            for k, xb in enumerate(x_per_robot[1]):
                min_clearance = min(min_clearance, float(np.linalg.norm(xa - xb)) - params.r_dyns_obsts[0][k] - params.r_dyns_obsts[1][k])
    st = np.array(solver_times[min(10, len(solver_times) - 1):]) * 1e3
    return {"n_robots": N, "control_steps": w + 1, "success": [s is not None for s in success_step],
            "steps_to_success": success_step, "blocks_picked": [m.get_nr_blocks_picked() for m in state_machines],
            "states_visited": [sorted(s) for s in states_seen], "min_clearance_m": min_clearance,
            "time_in_deadlock_steps": int(deadlock_prevention.time_in_deadlock),
            "solver_time_ms_mean": float(st.mean()), "solver_time_ms_median": float(np.median(st))}


def define_run_panda_example(n_robots=2, n_steps=6000, horizon=10, rollouts=True, estimate_goal=False, n_cubes=None,
                             n_obst_per_link=1):
    """EXC:522-556."""
    params = manipulator_parameters(nr_robots=n_robots, n_obst_per_link=n_obst_per_link)
    params.define_settings(ROLLOUT_FABRICS=rollouts, ROLLOUTS_PLOTTING=False, STATIC_OR_DYN_FABRICS=1,
                           RESOLVE_DEADLOCKS=int(rollouts), ESTIMATE_GOAL=estimate_goal, N_HORIZON=horizon)
    if n_cubes is not None:
        params.n_cubes = n_cubes
    planners, planners_grasp, goal_structs = define_planners(params)
    for g in goal_structs:
        g._config["subgoal1"]["weight"] = 20.0             # EXC:42 (the joint-space driver's dummy goal carries 10)
    utils_class = UtilsKinematics()
    links, offs = config.sphere_offsets_per_link(params.n_obst_per_link)
    sphere_T = []
    for i in range(params.nr_robots):
        per_link = [[np.identity(4) for _ in range(params.n_obst_per_link)] for _ in range(8)]
        for s, off in enumerate(offs):
            per_link[links[s] - 1][s % params.n_obst_per_link][0:3, 3] = off
        sphere_T.append(per_link)
    fk_dict_spheres = utils_class.define_symbolic_collision_link_poses(None, params.collision_links, sphere_T,
                                                                       n_obst_per_link=params.n_obst_per_link,
                                                                       mount_transform=params.mount_transform)
    forwardplanners = define_rollout_planners(params, goal_structs, planners) if rollouts else None
    return run_panda_example(params, n_steps, planners, planners_grasp, goal_structs, forwardplanners, fk_dict_spheres)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--robots", type=int, default=2)
    ap.add_argument("--steps", type=int, default=6000)
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--no-rollouts", action="store_true")
    ap.add_argument("--estimate-goal", action="store_true")
    args = ap.parse_args()
    print(json.dumps(define_run_panda_example(args.robots, args.steps, args.horizon, not args.no_rollouts,
                                              args.estimate_goal), indent=1))
