#!/usr/bin/env python3
"""Launches only the rollout kernel a few times (for rocprofv3 --pmc / --kernel-trace passes).
usage: python3 tools/prof_rollout.py [B] [dtype] [iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64512
dtype = sys.argv[2] if len(sys.argv) > 2 else "f64"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N, H = 3, 30
cfg = config.panda_config(n_robots=N, horizon=H, scalar=abi.F64 if dtype == "f64" else abi.F32)
cfg.goal_estimate_mask = 0b110
if os.environ.get("MRF_GENERIC") == "1":   # leaf strings outside the reference's set -> runtime-family (generic) kernels
    config.set_strings(cfg, collision_geometry="-0.5 / (x ** 3) * (xdot ** 2)")
batch = scenarios.panda_batch(cfg, B, seed=1000)
h = FabricHandle(cfg, 0)
q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
h.rollout(q, qd, prm)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    avg = h.rollout(q, qd, prm)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"B={B} {dtype}: {dt*1e3:.3f} ms per rollout, {B*N*H/dt:.4g} rollout-steps/s")
