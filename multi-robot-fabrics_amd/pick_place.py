"""Pick-and-place sequencing of one arm with the reference's call surface
(multi_robot_fabrics/others_planner/state_machine.py, class StateMachine; driver use EXJ:249-316,448).

States (SM:133-214): 0 go home, gripper open -> 1 above the block -> 2 down to the block -> 3 close the gripper for
0.3 s -> 12 lift -> 4 carry home -> 5 release -> 0 ...; 10 = all blocks done.  The logic consumes the hot path's
end-effector position (a callable q -> x_ee, e.g. `UtilsKinematics`' hand FK) and produces the goal / goal weight the
planner is fed with; it is integer bookkeeping per robot and stays on the host.  Pinned by sequences recorded from the
reference module (tests/golden/make_state_machine_golden.py).
"""
import numpy as np

HOME, PREGRASP, DESCEND, GRIP, CARRY, RELEASE, DONE, LIFT = 0, 1, 2, 3, 4, 5, 10, 12


class StateMachine:
    # distances [m] / durations [control steps] of the transitions (SM:150-199)
    REACH_HOME, REACH_PREGRASP, REACH_BLOCK, REACH_LIFT, REACH_DROP = 0.05, 0.013, 0.013, 0.04, 0.15
    GRIP_STEPS = 0.3 / 0.01
    OPEN_TOL_PANDA, OPEN_TOL_KINOVA = 0.005, 0.1
    DROPPED_BELOW_Z = 0.6

    def __init__(self, start_goal, nr_robots, nr_blocks, fk_fun_ee, robot_types):
        self.nr_robots, self.robot_types, self.fk_fun_ee = nr_robots, robot_types, fk_fun_ee
        self.start_goal = start_goal
        self.goal = start_goal
        self.weight_goal = 2
        self.weight_panda_high, self.weight_panda_low = 2, 0
        self.state_machine_panda = PREGRASP
        self.nr_blocks_panda = nr_blocks
        self.nr_blocks_panda_success = self.nr_blocks_panda_failed = 0
        self.time_gripping_panda = self.time_block_move_panda = self.time_start_move_panda = self.stop_time_panda = 0
        self.gripper_panda = self.gripper_robot2 = "open"
        # second-robot constants (SM:18-27); only the goal weight differs between a Kinova and a Panda partner
        self.weight_robot2_high = 4 if robot_types[1] == "kinova" else 10
        self.dist_start_constant_robot2, self.dist_block_constant_robot2, self.constant_time_gripping2 = 0.03, 0.01, 1.3
        self.q_panda_gripper_opened = np.array([0.04, 0.04])
        self.q_kinova_gripper_opened = np.array([0.96, 0.21, -0.96, -0.21])

    # ------------------------------------------------------------------ accessors (SM:40-64,125-131)
    def get_goal_robot(self):
        return self.goal

    def get_nr_blocks_picked(self):
        return self.nr_blocks_panda_success

    def get_weight_goal0(self):
        return self.weight_goal

    def get_gripper_status(self):
        return self.gripper_panda, self.gripper_robot2

    def get_success_rate(self):
        return (self.nr_blocks_panda_success - self.nr_blocks_panda_failed) / self.nr_blocks_panda

    def get_x_ee(self, q_robot):
        return self.fk_fun_ee(q_robot)

    def get_distance_ee_goal(self, q_robot, goal):          # horizontal distance only
        return np.linalg.norm(self.get_x_ee(q_robot)[:2] - goal[:2])

    def get_distance_ee_goal3(self, q_robot, goal):
        return np.linalg.norm(self.get_x_ee(q_robot) - goal)

    def get_distance_ee_start(self, q_robot):
        return np.linalg.norm(self.get_x_ee(q_robot) - self.start_goal)

    # ------------------------------------------------------------------ gripper velocity commands (SM:66-123)
    def get_gripper_action_panda(self, q_panda_gripper):
        q = np.asarray(q_panda_gripper)
        if self.gripper_panda == "close":
            return np.full(2, -0.05)
        if self.gripper_panda == "open" and np.linalg.norm(q - self.q_panda_gripper_opened) > self.OPEN_TOL_PANDA:
            return np.where(q > self.q_panda_gripper_opened, -0.4, 0.4)
        return np.zeros(2)

    def get_gripper_action_kinova(self, q_gripper):
        q = np.asarray(q_gripper)
        if self.gripper_robot2 == "close":
            lo = np.array([-0.10, -1.04, -0.97, -0.51])     # finger joint range of the Kinova manual (SM:90-91)
            hi = np.array([0.97, 0.22, 0.10, 0.22])
            act = np.array([-4.0, -4.0, 4.0, 4.0])
            act[q < lo] = 0.4
            act[q > hi] = -0.4                               # applied after the lower bound, as in the reference
            return act
        if self.gripper_robot2 == "open" and np.linalg.norm(q - self.q_kinova_gripper_opened) > self.OPEN_TOL_KINOVA:
            return np.where(q > self.q_kinova_gripper_opened, -0.4, 0.4)
        return np.zeros(4)

    # ------------------------------------------------------------------ one update per control step (SM:133-214)
    def get_state_machine_panda(self, q_robot, q_robot_gripper, goal_block, robot_type):
        block = np.asarray(goal_block)
        above = np.array(block, dtype=float, copy=True)
        above[2] += 0.1
        d_home = self.get_distance_ee_start(q_robot)
        d_above = self.get_distance_ee_goal(q_robot, goal=above)
        d_block = self.get_distance_ee_goal3(q_robot, goal=block)
        d_open = np.linalg.norm(np.asarray(q_robot_gripper) - self.q_panda_gripper_opened)

        if self.nr_blocks_panda_success > self.nr_blocks_panda - 1:
            self.state_machine_panda = DONE
        elif block[2] < self.DROPPED_BELOW_Z:                # the block fell off the table: count it and start over
            self.nr_blocks_panda_success += 1
            self.nr_blocks_panda_failed += 1
            self.state_machine_panda = HOME

        s = self.state_machine_panda
        if s == HOME:
            self.goal, self.gripper_panda = self.start_goal, "open"
            if d_home < self.REACH_HOME:
                self.state_machine_panda = PREGRASP
        elif s == PREGRASP:
            self.goal = above
            if d_above < self.REACH_PREGRASP:
                self.state_machine_panda = DESCEND
        elif s == DESCEND:
            self.goal = goal_block
            if d_block < self.REACH_BLOCK:
                self.gripper_panda, self.weight_goal = "close", self.weight_panda_low
                self.state_machine_panda = GRIP
        elif s == GRIP:
            self.goal = goal_block
            self.goal_above_block = np.array(block, dtype=float, copy=True)
            self.goal_above_block[2] += 0.15
            self.time_gripping_panda += 1
            if self.time_gripping_panda > self.GRIP_STEPS:
                self.time_gripping_panda = 0
                self.goal, self.weight_goal = self.start_goal, self.weight_panda_high
                self.state_machine_panda = LIFT
        elif s == LIFT:
            self.goal = self.goal_above_block
            if self.get_distance_ee_goal(q_robot, goal=self.goal) < self.REACH_LIFT:
                self.state_machine_panda = CARRY
        elif s == CARRY:
            self.goal = self.start_goal
            if d_home < self.REACH_DROP:
                self.state_machine_panda, self.gripper_panda, self.time_start_move_panda = RELEASE, "open", 0
        elif s == RELEASE:
            if d_open < self.OPEN_TOL_PANDA:
                self.state_machine_panda = HOME
                self.nr_blocks_panda_success += 1
                self.goal = self.start_goal
        elif s == DONE:
            self.stop_time_panda = 1
        return self.state_machine_panda
