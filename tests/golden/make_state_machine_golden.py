#!/usr/bin/env python3
"""Golden sequences for the pick-and-place state machine, recorded by importing the reference's module
(numpy + copy only; runs in the build container):  /root/reference/multi_robot_fabrics/others_planner/state_machine.py

A scripted end effector is moved through the phases of several pick-and-place cycles (approach, descend, grip, lift,
carry, release, a dropped block, all blocks done); inputs and the reference object's outputs per control step are
stored -- data only.     usage: python3 tests/golden/make_state_machine_golden.py
"""
import contextlib
import io
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from multi_robot_fabrics.others_planner.state_machine import StateMachine  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def scripted_run(seed, robot2, nr_blocks, T):
    rng = np.random.default_rng(seed)
    start = np.array([0.4, 0.0, 0.9]) + rng.uniform(-0.05, 0.05, 3)
    x = {"ee": start + rng.uniform(-0.2, 0.2, 3)}
    sm = StateMachine(start_goal=start, nr_robots=2, nr_blocks=nr_blocks, fk_fun_ee=lambda q: x["ee"].copy(),
                      robot_types=["panda", robot2])
    block = np.array([0.6, 0.2, 0.7]) + rng.uniform(-0.05, 0.05, 3)
    grip = np.array([0.04, 0.04])
    kin = np.array([0.9, 0.2, -0.9, -0.2])
    rec = {k: [] for k in ("x_ee", "block", "grip", "kin", "state", "goal", "weight", "picked", "grip_act", "kin_act",
                           "success", "status")}
    for t in range(T):
        goal = np.asarray(sm.get_goal_robot(), dtype=float)
        x["ee"] = x["ee"] + np.clip(goal - x["ee"], -0.02, 0.02) + rng.normal(0, 0.002, 3)   # the arm tracks the goal
        status = sm.get_gripper_status()[0]
        grip = np.clip(grip + (-0.004 if status == "close" else 0.004) + rng.normal(0, 0.0005, 2), 0.0, 0.0415)
        kin = kin + rng.normal(0, 0.05, 4)
        if t % 97 == 96:
            sm.gripper_robot2 = "close" if sm.gripper_robot2 == "open" else "open"
        if rng.random() < 0.004:
            block = block.copy(); block[2] = 0.3                  # dropped from the table
        elif block[2] < 0.6:
            block = np.array([0.6, 0.2, 0.7]) + rng.uniform(-0.05, 0.05, 3)   # next block
        rec["x_ee"].append(x["ee"].copy()); rec["block"].append(block.copy()); rec["grip"].append(grip.copy())
        rec["kin"].append(kin.copy())
        with contextlib.redirect_stdout(io.StringIO()):
            s = sm.get_state_machine_panda(q_robot=None, q_robot_gripper=grip.copy(), goal_block=block.copy(), robot_type="panda")
        rec["state"].append(s); rec["goal"].append(np.asarray(sm.get_goal_robot(), dtype=float).copy())
        rec["weight"].append(sm.get_weight_goal0()); rec["picked"].append(sm.get_nr_blocks_picked())
        rec["grip_act"].append(sm.get_gripper_action_panda(grip.copy())); rec["kin_act"].append(sm.get_gripper_action_kinova(kin.copy()))
        rec["success"].append(sm.get_success_rate()); rec["status"].append(0 if sm.get_gripper_status()[0] == "open" else 1)
        if s in (0,) and sm.get_nr_blocks_picked() > rec["picked"][max(0, t - 1)]:
            block = np.array([0.6, 0.2, 0.7]) + rng.uniform(-0.05, 0.05, 3)
    out = {k: np.array(v) for k, v in rec.items()}
    out["start"] = start
    return out


def main():
    out = {}
    for name, seed, robot2, nb, T in (("panda_a", 0, "panda", 2, 1500), ("kinova_b", 1, "kinova", 3, 2500),
                                      ("panda_c", 2, "panda", 1, 800)):
        r = scripted_run(seed, robot2, nb, T)
        for k, v in r.items():
            out[f"{name}/{k}"] = v
        print(name, "states visited:", sorted(set(r["state"].tolist())), "blocks picked:", int(r["picked"][-1]))
        out[f"{name}/meta"] = np.array([nb, 1 if robot2 == "kinova" else 0])
    np.savez_compressed(os.path.join(HERE, "state_machine_sequences.npz"), **out)


if __name__ == "__main__":
    main()
