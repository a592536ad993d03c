"""Minimal mirror of mpscenes' GoalComposition as the reference uses it (example_pandas_Jointspace.py:25-62,
361-365; example_pointmasses_static.py:61-83): a named dict of sub-goals with attribute access."""
import math

import numpy as np


class _SubGoalConfig(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


class SubGoal:
    def __init__(self, name, content):
        self._name = name
        self._config = _SubGoalConfig(content)

    def name(self):
        return self._name

    def weight(self):
        return self._config["weight"]

    def is_primary_goal(self):
        return bool(self._config.get("is_primary_goal", False))

    def indices(self):
        return list(self._config["indices"])

    def dimension(self):
        return len(self._config["indices"])

    def parent_link(self):
        return self._config.get("parent_link")

    def child_link(self):
        return self._config.get("child_link")

    def position(self):
        return np.asarray(self._config["desired_position"], dtype=float)

    def angle(self):
        return self._config.get("angle")

    def type(self):
        return self._config["type"]


class _CompositionConfig(dict):
    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key) from None


class GoalComposition:
    def __init__(self, name, content_dict):
        self._name = name
        self._config = _CompositionConfig((k, _SubGoalConfig(v)) for k, v in content_dict.items())
        self._sub_goals = [SubGoal(k, v) for k, v in content_dict.items()]

    def sub_goals(self):
        return self._sub_goals

    def primary_goal(self):
        for g in self._sub_goals:
            if g.is_primary_goal():
                return g
        return self._sub_goals[0]


def panda_pick_place_goal(orientation_weight=10.0):
    """The three-part goal of the Panda drivers (EXJ:25-62 / EXC:24-61; the Cartesian driver weighs the orientation
    part 20): hand position in the world, hand axis relative to link 7 (the 0.107 m offset of the flange), joint 7
    angle.  Every number here is a placeholder: positions, weights and the rotation are runtime parameters."""
    hand = dict(indices=[0, 1, 2], child_link="panda_hand", epsilon=0.05, type="staticSubGoal")
    parts = [
        dict(hand, weight=2.0, is_primary_goal=True, parent_link="world", desired_position=[0.1, 0.6, 0.8]),
        dict(hand, weight=float(orientation_weight), is_primary_goal=False, parent_link="panda_link7",
             desired_position=[0.107, 0.0, 0.0], angle=[-0.366, 0.0, 0.0, 0.3305]),
        dict(weight=1.0, is_primary_goal=False, indices=[6], desired_position=[math.pi / 4], epsilon=0.05,
             type="staticJointSpaceSubGoal"),
    ]
    return GoalComposition(name="goal", content_dict={"subgoal%d" % k: part for k, part in enumerate(parts)})


def point_robot_goal(position=(1.5, 0.99), weight=1.0):
    """The single planar goal of the point-mass examples (a placeholder: every robot's own goal is a runtime parameter)."""
    return GoalComposition(name="goal", content_dict={"subgoal0": dict(
        weight=weight, is_primary_goal=True, indices=[0, 1], parent_link="world", child_link="base_link",
        desired_position=list(position), epsilon=0.1, type="staticSubGoal")})
