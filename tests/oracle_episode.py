"""Oracle-side composition of one control step / an episode (test infrastructure): what mrf_episode_run does on the
device, built from the CPU oracle's pieces -- examples/example_pandas_Jointspace.py:280-458 minus the simulator."""
import os
import sys

import numpy as np

import oracle_lib
from multi_robot_fabrics_amd import abi, scenarios

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import deadlock_oracle  # noqa: E402


def hand_state(cfg_roll, q, qd):
    """x_ee, v_ee [3,rows]: the rollout planner's sphere table is the 8 link origins, the last one is the hand."""
    assert cfg_roll.n_spheres == 8 and cfg_roll.sphere_link[7] == 8
    sx, sv, _ = oracle_lib.fk_spheres(cfg_roll, q, qd)
    return sx[7], sv[7]


def control_step(cfg_roll, cfg_act, q, qd, prm, states, K, w, vel_limit, stop_margin, apply_estimate=True, sm=None,
                 deadlock=True):
    N = cfg_act.n_robots
    rows = q.shape[1]
    x_ee, v_ee = hand_state(cfg_roll, q, qd)
    work = prm.copy()
    if apply_estimate:                                                         # EXJ:346-348
        for i in range(N):
            if (cfg_roll.goal_estimate_mask >> i) & 1:
                work[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + 3, i::N] = x_ee[:, i::N] + cfg_roll.goal_estimate_T * v_ee[:, i::N]
    avg, _, _ = oracle_lib.rollout(cfg_roll, q, qd, work)                      # EXJ:354-375
    if deadlock:
        states, work = deadlock_oracle.step_batch(states, K, x_ee, work, avg, w, sm, N)   # EXJ:377-383
    sx, sv, _ = oracle_lib.fk_spheres(cfg_act, q, qd)                          # EXJ:394-412 (a = 0, EXJ:411)
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg_act, dict(q=q), sx, sv if cfg_act.dynamic else None, None)
    _, act = oracle_lib.compute_action(cfg_act, q, qd, work, ox, ov, oa, orad)  # EXJ:417-448
    vl = np.asarray(vel_limit, dtype=float)[:, None]
    act = np.clip(act, -vl, vl)                                                # EXJ:452
    qn = q + cfg_act.dt * act
    if stop_margin >= 0:
        lo = np.array([cfg_act.limits[j][0] for j in range(7)])[:, None] + stop_margin
        hi = np.array([cfg_act.limits[j][1] for j in range(7)])[:, None] - stop_margin
        qn = np.minimum(np.maximum(qn, lo), hi)
    return qn, act, act, states, work, avg, x_ee


def episode(cfg_roll, cfg_act, q, qd, prm, n_steps, K, vel_limit, stop_margin, apply_estimate=True, sm=None,
            deadlock=True):
    B = q.shape[1] // cfg_act.n_robots
    states = [deadlock_oracle.initial_state() for _ in range(B)]
    hist = []
    act = None
    for w in range(n_steps):
        q, qd, act, states, work, avg, x_ee = control_step(cfg_roll, cfg_act, q, qd, prm, states, K, w, vel_limit,
                                                           stop_margin, apply_estimate, sm, deadlock)
        hist.append(dict(q=q.copy(), work=work, avg=avg, x_ee=x_ee))
    return q, qd, act, states, hist
