#!/usr/bin/env python3
"""Row-per-lane vs cooperative rollout kernel over the batch size (sets the auto-selection threshold)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
N, H = 3, 30
base = config.panda_config(n_robots=N, horizon=H); base.goal_estimate_mask = 6
b = scenarios.panda_batch(base, 32768, seed=3)
hs = {}
for k in (1, 2):
    c = base.copy(); c.kernel_select = k; hs[k] = FabricHandle(c, 0)
for B in (1, 64, 256, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 16384, 21504, 32768):
    out = []
    for k in (1, 2):
        h = hs[k]
        q, qd, prm = (h.tensor(b[x][:, :B * N]) for x in ("q", "qdot", "params"))
        h.rollout(q, qd, prm); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            h.rollout(q, qd, prm)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 5 * 1e3)
    print(f"B={B:6d}  row-per-lane {out[0]:8.3f} ms   cooperative {out[1]:8.3f} ms")
