"""CPU restatement of the reference's deadlock detection / resolution -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product never does.

Restates  multi_robot_fabrics/others_planner/deadlock_prevention.py:50-118  (`deadlockprevention.deadlock_checking`)
as a pure function of (state, inputs) -> (state', outputs), looping over scenarios in plain Python.  PINNED: it is
checked against sequences recorded from the reference's own module (tests/golden/deadlock_sequences.npz, made by
tests/golden/make_deadlock_golden.py which imports /root/reference in the build container) in tests/test_deadlock.py.
"""
import itertools
import math

import numpy as np


def constants(point_mass=False):
    """deadlock_prevention.py:12-27 plus the literals used in :50-118."""
    return dict(avg_vel_constant=0.03 if point_mass else 0.16, dist_constant=1.0 if point_mass else 0.0,
                goal_weight_follower=10.0 if point_mass else 2.0, goal_weight_leader=1.0 if point_mass else 3.0,
                time_wait=50 if point_mass else 300, nr_goal_scale=100.0 if point_mass else 2.0,
                ee_distance=0.35, follower_offset=0.3, min_goal_norm=0.05, z_floor=0.1, min_time_step=10,
                grasp_state=2, grasp_timeout=400)


def initial_state():
    """DP:9-10,28,33-34 and the driver's time_deadlock_out = 1000 (EXJ:273)."""
    return dict(leader=0, follower=1, dead=(0, 1), goal=np.zeros(3), time_in_deadlock=0, time_deadlock_out=1000)


def _norm(v):
    return math.sqrt(sum(float(c) * float(c) for c in v))


def step(state, K, x, goals, weights, avg_sum, time_step, sm):
    """One call of deadlock_checking for one scenario.
    x, goals [N,3]; weights [N]; sm [N] ints.  Returns (new_state, goals', weights')."""
    n = len(x)
    s = dict(state)
    goals = np.array(goals, dtype=float, copy=True)
    weights = np.array(weights, dtype=float, copy=True)
    to_goal = [_norm(x[i] - goals[i]) for i in range(n)]                      # DP:57
    found, best = False, 100.0
    for a, b in itertools.combinations(range(n), 2):                           # DP:30,59
        d_ee = _norm(x[a] - x[b])                                              # DP:62
        if (avg_sum < K["avg_vel_constant"] and to_goal[a] + to_goal[b] > K["dist_constant"]
                and time_step > K["min_time_step"] and sm[a] in (0, 1) and sm[b] in (0, 1) and d_ee < K["ee_distance"]):
            found = True                                                       # DP:66-73
            if d_ee < best:                                                    # DP:74-79: closest pair, first on ties
                best, s["dead"] = d_ee, (a, b)
    d0, d1 = s["dead"]
    if found and time_step > K["min_time_step"]:                               # DP:81
        s["leader"], s["follower"] = (d1, d0) if to_goal[d0] > to_goal[d1] else (d0, d1)   # DP:83-88
        diff = x[s["leader"]] - x[s["follower"]]
        dg = diff * K["nr_goal_scale"]
        nrm = _norm(dg)
        if nrm > K["min_goal_norm"]:
            g = x[s["follower"]] - K["follower_offset"] / nrm * dg             # DP:95
        else:
            g = x[s["follower"]] - diff * K["nr_goal_scale"]                   # DP:97
        g = np.array(g, dtype=float)
        if g[2] < 0:
            g[2] = K["z_floor"]                                                # DP:98-99
        s["goal"] = g
        apply = True
        s["time_in_deadlock"] += 1
        s["time_deadlock_out"] = 0
    elif sm[d0] == K["grasp_state"] or sm[d1] == K["grasp_state"]:             # DP:108-109
        apply = False
        s["time_deadlock_out"] = K["grasp_timeout"]
    elif s["time_deadlock_out"] < K["time_wait"]:                              # DP:111-115
        apply = True
        s["time_deadlock_out"] += 1
    else:
        apply = False
    if apply:
        weights[s["leader"]] = K["goal_weight_leader"]
        weights[s["follower"]] = K["goal_weight_follower"]
        goals[s["follower"]] = s["goal"]
    return s, goals, weights


def step_batch(states, K, x_ee, params, avg_vel, time_step, sm, n_robots, P_X_GOAL_0=0, P_WEIGHT_GOAL_0=3):
    """Batched form on the build's layouts: x_ee [3,rows], params [29,rows] (modified copy returned), avg_vel [rows],
    sm [rows] or None; avg_sum = sum_i avg_vel_i / N (EXJ:375)."""
    rows = x_ee.shape[1]
    B = rows // n_robots
    prm = np.array(params, dtype=float, copy=True)
    out_states = []
    for b in range(B):
        r = slice(b * n_robots, (b + 1) * n_robots)
        x = x_ee[:, r].T
        goals = prm[P_X_GOAL_0:P_X_GOAL_0 + 3, r].T
        w = prm[P_WEIGHT_GOAL_0, r]
        smb = [0] * n_robots if sm is None else [int(v) for v in sm[r]]
        st, g2, w2 = step(states[b], K, x, goals, w, float(np.sum(avg_vel[r]) / n_robots), time_step, smb)
        prm[P_X_GOAL_0:P_X_GOAL_0 + 3, r] = g2.T
        prm[P_WEIGHT_GOAL_0, r] = w2
        out_states.append(st)
    return out_states, prm
