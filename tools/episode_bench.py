#!/usr/bin/env python3
"""Device-resident closed loop (mrf_episode_run): control steps per second with and without the replayed HIP graph.
usage: python3 tools/episode_bench.py [N] [H] [steps]   (runs B in {1, 64, 1024, 6144, 129024})"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import config, scenarios
from multi_robot_fabrics_amd.runtime import ControlLoop, FabricHandle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
H = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cfg_roll = config.panda_config(n_robots=N, horizon=H)
cfg_roll.goal_estimate_mask = ((1 << N) - 1) & ~1
cfg_act = config.panda_config(n_robots=N, horizon=1)
hr, ha = FabricHandle(cfg_roll, 0), FabricHandle(cfg_act, 0)
out = []
for B in (1, 64, 1024, 6144, 129024):
    batch = scenarios.panda_batch(cfg_roll, B, seed=7, qd_spread=0.2)
    q, qd, prm = (ha.tensor(batch[k]) for k in ("q", "qdot", "params"))
    n = steps if B <= 6144 else max(10, steps // 10)
    row = {"scenarios": B, "steps": n}
    for graph in (False, True):
        loop = ControlLoop(ha, hr, q, qd, prm, config.PANDA_VEL_LIMITS, deadlock=True, apply_estimate=False, use_graph=graph)
        loop.run(5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.run(n)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        row["graph" if graph else "launches"] = {"ms_per_control_step": 1e3 * dt / n, "control_steps_per_s": B * n / dt}
        assert torch.isfinite(loop.q).all()
    out.append(row)
    print(json.dumps(row), flush=True)
