#!/usr/bin/env python3
"""Simulator-free closed loop of the N-Panda example (the planner side of examples/example_pandas_Jointspace.py:280-458):
every control step  [Rollout Fabrics -> avg velocity -> deadlock logic]  +  compute_action for all robots against each
other's collision spheres, then the joint velocity command is applied exactly (q += dt * clip(action)), as urdfenvs
does in 'vel' mode.  The whole loop runs on the device (runtime.ControlLoop = mrf_episode_run); the host only looks at
the state every `--monitor` steps.

Runs B independent scenarios at once on the GPU and reports behavioural statistics: how many end-effectors reach
their goal, the minimum sphere clearance between robots, joint-limit margins.  This is a plausibility check of the
fabric specification, not a parity test.

usage: python examples/closed_loop_pandas.py [--robots 2] [--scenarios 256] [--steps 1000] [--rollouts] [--deadlock]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import ControlLoop, FabricHandle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--robots", type=int, default=2)
    ap.add_argument("--scenarios", type=int, default=256)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--horizon", type=int, default=10)
    ap.add_argument("--rollouts", action="store_true", help="also run the Rollout Fabrics every step (avg-velocity monitor)")
    ap.add_argument("--deadlock", action="store_true", help="with --rollouts: run the deadlock detection/resolution too")
    ap.add_argument("--monitor", type=int, default=10, help="host-side statistics every this many control steps")
    ap.add_argument("--n-obst-per-link", type=int, default=1)
    args = ap.parse_args()
    N, B = args.robots, args.scenarios
    cfg = config.panda_config(n_robots=N, horizon=args.horizon)
    if args.n_obst_per_link > 1:
        links, offs = config.sphere_offsets_per_link(args.n_obst_per_link)
        config.set_spheres(cfg, links, offs)
    batch = scenarios.panda_batch(cfg, B, seed=0, qd_spread=0.0)
    h = FabricHandle(cfg, 0)
    roll = None
    if args.rollouts:
        cr = config.panda_config(n_robots=N, horizon=args.horizon)
        roll = FabricHandle(cr, 0)
    q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
    vlim = h.tensor(np.array(config.PANDA_VEL_LIMITS)[:, None])
    lim = h.tensor(np.array(config.PANDA_LIMITS))
    goal = prm[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + 3]                       # [3, rows]
    S = cfg.n_spheres
    rad = h.tensor(np.array(cfg.sphere_radius[:S]))
    min_clear = torch.full((B,), 1e9, dtype=h.dtype, device="cuda")
    min_limit = torch.full((B * N,), 1e9, dtype=h.dtype, device="cuda")
    reached_at = torch.full((B * N,), -1, dtype=torch.int64, device="cuda")
    avg_hist = []
    bad_rollouts = 0
    loop = ControlLoop(h, roll, q, qd, prm, config.PANDA_VEL_LIMITS, deadlock=args.deadlock, apply_estimate=False,
                       stop_margin=1e-3)
    t = -1
    while t < args.steps - 1:
        n = min(args.monitor, args.steps - 1 - t)
        loop.run(n)
        t += n
        q, qd = loop.q, loop.qdot
        if roll is not None:
            # a state parked on a hard joint stop sits 1e-3 rad from a limit barrier: the 30-step explicit-Euler
            # prediction from there can step across the barrier and blow up (the reference's rollouts have no stops
            # either); such rows are counted, not averaged
            avg_hist.append(float(torch.nanmean(loop.avg)))
            bad_rollouts = max(bad_rollouts, int((~torch.isfinite(loop.avg)).sum()))
        if True:  # statistics on the state after these steps
            x, _, _ = h.fk_spheres(q)                                  # [S,3,rows]
            xs = x.view(S, 3, B, N)
            for i in range(N):
                for j in range(i + 1, N):
                    d = (xs[:, None, :, :, i] - xs[None, :, :, :, j]).norm(dim=2)          # [S,S,B]
                    clear = d - rad[:, None, None] - rad[None, :, None]
                    min_clear = torch.minimum(min_clear, clear.reshape(S * S, B).min(0).values)
            ee = x[max(s for s in range(S) if cfg.sphere_link[s] == 8)]                     # a sphere on link 8
            if all(cfg.sphere_offset[s][k] == 0.0 for s in range(S) for k in range(3)):
                dist = (ee - goal).norm(dim=0)
                newly = (dist < 0.05) & (reached_at < 0)
                reached_at[newly] = t
            margin = torch.minimum(q - lim[:, 0:1], lim[:, 1:2] - q).min(0).values
            min_limit = torch.minimum(min_limit, margin)
    torch.cuda.synchronize()
    out = {
        "robots": N, "scenarios": B, "steps": args.steps, "dt": cfg.dt, "spheres_per_robot": S,
        "all_finite": bool(torch.isfinite(q).all()),
        "goal_reached_fraction": float((reached_at >= 0).double().mean()),
        "median_steps_to_goal": float(reached_at[reached_at >= 0].double().median()) if (reached_at >= 0).any() else None,
        "final_ee_goal_distance_mean": float((h.fk_spheres(q)[0][S - 1] - goal).norm(dim=0).mean()),
        "min_sphere_clearance_m": float(min_clear.min()), "scenarios_with_contact": int((min_clear < 0).sum()),
        "min_joint_limit_margin_rad": float(min_limit.min()),
        "rows_resting_on_a_hard_stop_at_the_end": int(((q - lim[:, 0:1] < 1.5e-3) | (lim[:, 1:2] - q < 1.5e-3)).any(0).sum()),
        "final_speed_mean": float(qd.norm(dim=0).mean()),
    }
    if avg_hist:
        out["rollout_avg_velocity_first_last"] = [avg_hist[0], avg_hist[-1]]
        out["max_rows_with_nonfinite_rollout_prediction"] = bad_rollouts
    if loop.dl_state is not None:
        from multi_robot_fabrics_amd import abi as _abi
        tid = loop.dl_state[_abi.DL_TIME_IN_DEADLOCK]
        out["scenarios_that_entered_deadlock_resolution"] = int((tid > 0).sum())
        out["mean_steps_in_deadlock"] = float(tid.double().mean())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
