"""Host-side deadlock detection / resolution with the reference's call surface
(multi_robot_fabrics/others_planner/deadlock_prevention.py:4-118).  It consumes one float per robot from the hot
path (the rollout's mean squared joint velocity) and rewrites the follower's goal and the two goal weights; it is
O(N^2) scalar logic per control step and stays on the host (SURVEY 8f-1 lists the device version as the next row).
"""
import itertools

import numpy as np


class deadlockprevention:
    def __init__(self, dof, n_robots, N_horizon):
        self.dof = dof
        self.n_robots = n_robots
        self.N_horizon = N_horizon
        self.i_leader, self.i_follower = 0, 1
        point = self.dof[0] == 2                      # point-mass thresholds, DP:12-19; manipulators DP:20-27
        self.avg_vel_constant = 0.03 if point else 0.16
        self.dist_constant = 1 if point else 0
        self.goal_weight_follower = 10 if point else 2
        self.goal_weight_leader = 1 if point else 3
        self.time_wait = 50 if point else 300
        self.nr_goal_scale = 100 if point else 2
        self.goal_robot0 = np.zeros(2 if point else 3)
        self.robot_combinations = list(itertools.combinations(range(n_robots), 2))
        self.deadlock_robots = [0] * n_robots
        self.deadlock_combinations = [0] * len(self.robot_combinations)
        self.i_robots_dead = [0, 1]
        self.time_in_deadlock = 0

    def compute_velocity_average(self, q_dot_robots_N):
        """DP:36-43: mean absolute joint velocity over the horizon, summed over robots."""
        total = 0.0
        for i in range(self.n_robots):
            for df in range(self.dof[i]):
                traj = np.asarray(q_dot_robots_N["robot_" + str(i)][df], dtype=float)
                total += np.abs(traj).sum() / (self.N_horizon * self.dof[i])
        return total

    @staticmethod
    def compute_distance_to_goal(x_robot, goal_robot):
        return float(np.linalg.norm(np.asarray(x_robot) - np.asarray(goal_robot)))

    def deadlock_checking(self, x_robots, goal_robots, goal_weights, time_step, time_deadlock_out, avg_sum,
                          state_machine_robots=()):
        """DP:50-118.  Mutates and returns (goal_robots, goal_weights, time_deadlock_out) like the reference."""
        deadlock = False
        pair_dist = [100.0] * len(self.robot_combinations)
        to_goal = [self.compute_distance_to_goal(x_robots[i], goal_robots[i]) for i in range(self.n_robots)]
        for z, (a, b) in enumerate(self.robot_combinations):
            approaching = state_machine_robots[a] in (0, 1) and state_machine_robots[b] in (0, 1)
            d_ee = float(np.linalg.norm(np.asarray(x_robots[a]) - np.asarray(x_robots[b])))
            if (avg_sum < self.avg_vel_constant and to_goal[a] + to_goal[b] > self.dist_constant and time_step > 10
                    and approaching and d_ee < 0.35):
                for i in (a, b):
                    self.deadlock_robots[i] += 1
                    self.deadlock_combinations[z] += 1
                    pair_dist[z] = d_ee
                deadlock = True
                best = 100.0
                for zz in range(len(self.deadlock_combinations)):
                    if pair_dist[zz] < best:
                        best = pair_dist[zz]
                        self.i_robots_dead = list(self.robot_combinations[zz])
        dead = self.i_robots_dead
        if deadlock and time_step > 10:
            # the robot closer to its goal leads (DP:84-90)
            if to_goal[dead[0]] > to_goal[dead[1]]:
                self.i_leader, self.i_follower = dead[1], dead[0]
            else:
                self.i_leader, self.i_follower = dead[0], dead[1]
            diff = np.asarray(x_robots[self.i_leader], dtype=float) - np.asarray(x_robots[self.i_follower], dtype=float)
            diff_goal = diff * self.nr_goal_scale
            if np.linalg.norm(diff_goal) > 0.05:
                self.goal_robot0 = np.asarray(x_robots[self.i_follower], dtype=float) - 0.3 / np.linalg.norm(diff_goal) * diff_goal
            else:
                self.goal_robot0 = np.asarray(x_robots[self.i_follower], dtype=float) - diff_goal
            if self.goal_robot0[2] < 0:
                self.goal_robot0[2] = 0.1
            goal_weights[self.i_leader] = self.goal_weight_leader
            goal_weights[self.i_follower] = self.goal_weight_follower
            goal_robots[self.i_follower] = self.goal_robot0
            self.time_in_deadlock += 1
            time_deadlock_out = 0
        elif state_machine_robots[dead[0]] == 2 or state_machine_robots[dead[1]] == 2:
            time_deadlock_out = 400
        elif time_deadlock_out < self.time_wait:
            goal_weights[self.i_leader] = self.goal_weight_leader
            goal_weights[self.i_follower] = self.goal_weight_follower
            goal_robots[self.i_follower] = self.goal_robot0
            time_deadlock_out += 1
        return goal_robots, goal_weights, time_deadlock_out
