"""Host-side deadlock detection / resolution with the reference's call surface
(multi_robot_fabrics/others_planner/deadlock_prevention.py:4-118; driver use EXJ:244,377-383).

The logic consumes one scalar from the hot path (the rollout's mean squared joint velocity, averaged over the robots)
plus the hand positions, and rewrites the follower's goal and the goal weights of one robot pair.  This module is the
one-scenario host form for drivers written against the reference; `mrf_deadlock_step` (csrc/mrf_control.hip) is the
batched device form.  Both are pinned by sequences recorded from the reference's module (tests/test_deadlock.py,
tests/test_gpu_control.py).
"""
from dataclasses import dataclass
from itertools import combinations

import numpy as np


@dataclass(frozen=True)
class Thresholds:
    """DP:12-27 (two presets) and the literals of DP:50-118."""
    avg_vel: float
    goal_dist_sum: float
    weight_follower: float
    weight_leader: float
    hold_steps: int
    goal_scale: float
    hand_distance: float = 0.35       # DP:63
    push_back: float = 0.3            # DP:95
    tiny_offset: float = 0.05         # DP:94
    floor_z: float = 0.1              # DP:98-99
    warmup_steps: int = 10            # DP:66,81
    grasp_state: int = 2              # DP:108
    grasp_hold: int = 400             # DP:109

    @staticmethod
    def preset(point_mass):
        if point_mass:
            return Thresholds(0.03, 1, 10, 1, 50, 100)
        return Thresholds(0.16, 0, 2, 3, 300, 2)


def _closest_stuck_pair(pairs, x, to_goal, states, avg_sum, step, th):
    """The pair of approaching robots (states 0/1) whose hands are closest among those that look stuck: slow on
    average, not yet at their goals, hands within reach of each other (DP:59-79).  None if there is no such pair."""
    if not (avg_sum < th.avg_vel and step > th.warmup_steps):
        return None
    best, best_d = None, 100.0
    for a, b in pairs:
        if states[a] not in (0, 1) or states[b] not in (0, 1):
            continue
        if not to_goal[a] + to_goal[b] > th.goal_dist_sum:
            continue
        d = float(np.linalg.norm(x[a] - x[b]))
        if d < th.hand_distance and d < best_d:
            best, best_d = (a, b), d
    return best


class deadlockprevention:
    """Same constructor, attributes and `deadlock_checking` signature as the reference class."""

    def __init__(self, dof, n_robots, N_horizon):
        self.dof, self.n_robots, self.N_horizon = dof, n_robots, N_horizon
        th = Thresholds.preset(point_mass=(dof[0] == 2))
        self._th = th
        # the reference's attribute names, for drivers that read or tune them
        self.avg_vel_constant, self.dist_constant = th.avg_vel, th.goal_dist_sum
        self.goal_weight_follower, self.goal_weight_leader = th.weight_follower, th.weight_leader
        self.time_wait, self.nr_goal_scale = th.hold_steps, th.goal_scale
        self.robot_combinations = list(combinations(range(n_robots), 2))
        self.i_leader, self.i_follower = 0, 1
        self.i_robots_dead = [0, 1]
        self.goal_robot0 = np.zeros(2 if dof[0] == 2 else 3)
        self.time_in_deadlock = 0
        self.deadlock_robots = [0] * n_robots                        # diagnostics only, as in the reference
        self.deadlock_combinations = [0] * len(self.robot_combinations)

    def _thresholds(self):
        t = self._th
        return Thresholds(self.avg_vel_constant, self.dist_constant, self.goal_weight_follower, self.goal_weight_leader,
                          self.time_wait, self.nr_goal_scale, t.hand_distance, t.push_back, t.tiny_offset, t.floor_z,
                          t.warmup_steps, t.grasp_state, t.grasp_hold)

    def compute_velocity_average(self, q_dot_robots_N):
        """DP:36-43: mean absolute joint velocity over the horizon, summed over the robots."""
        return sum(float(np.abs(np.asarray(q_dot_robots_N["robot_%d" % i][j], dtype=float)).sum()) / (self.N_horizon * self.dof[i])
                   for i in range(self.n_robots) for j in range(self.dof[i]))

    @staticmethod
    def compute_distance_to_goal(x_robot, goal_robot):
        return float(np.linalg.norm(np.asarray(x_robot) - np.asarray(goal_robot)))

    def _apply(self, goal_robots, goal_weights):
        goal_weights[self.i_leader] = self.goal_weight_leader
        goal_weights[self.i_follower] = self.goal_weight_follower
        goal_robots[self.i_follower] = self.goal_robot0

    def deadlock_checking(self, x_robots, goal_robots, goal_weights, time_step, time_deadlock_out, avg_sum,
                          state_machine_robots=()):
        """DP:50-118.  Mutates and returns (goal_robots, goal_weights, time_deadlock_out) like the reference."""
        th = self._thresholds()
        x = [np.asarray(p, dtype=float) for p in x_robots]
        to_goal = [self.compute_distance_to_goal(x[i], goal_robots[i]) for i in range(self.n_robots)]
        pair = _closest_stuck_pair(self.robot_combinations, x, to_goal, state_machine_robots, avg_sum, time_step, th)
        if pair is not None:
            self.i_robots_dead = list(pair)
            for z, (a, b) in enumerate(self.robot_combinations):      # bookkeeping counters (never read back)
                if (a, b) == pair:
                    self.deadlock_combinations[z] += 2
            for i in pair:
                self.deadlock_robots[i] += 1
            a, b = pair
            # the robot closer to its goal keeps going, the other one backs off (DP:83-99)
            self.i_leader, self.i_follower = (b, a) if to_goal[a] > to_goal[b] else (a, b)
            away = (x[self.i_leader] - x[self.i_follower]) * th.goal_scale
            length = np.linalg.norm(away)
            retreat = x[self.i_follower] - (th.push_back / length * away if length > th.tiny_offset else away)
            if retreat[2] < 0:
                retreat[2] = th.floor_z
            self.goal_robot0 = retreat
            self._apply(goal_robots, goal_weights)
            self.time_in_deadlock += 1
            return goal_robots, goal_weights, 0
        d0, d1 = self.i_robots_dead
        if state_machine_robots[d0] == th.grasp_state or state_machine_robots[d1] == th.grasp_state:
            return goal_robots, goal_weights, th.grasp_hold
        if time_deadlock_out < th.hold_steps:                          # keep the resolution for a while (DP:111-115)
            self._apply(goal_robots, goal_weights)
            time_deadlock_out += 1
        return goal_robots, goal_weights, time_deadlock_out
