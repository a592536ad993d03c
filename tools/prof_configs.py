#!/usr/bin/env python3
"""Rollout-kernel timing for the BASELINE.json configurations other than the bench workload (kernel only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

cus = torch.cuda.get_device_properties(0).multi_processor_count
only = [a for a in sys.argv[1:] if a in ("C3", "C4", "C5")]       # e.g. `prof_configs.py f64only C5` for a counter pass
for name, N, H, S20, rounds in (("C3 2-Panda RF H=20", 2, 20, False, 6), ("C4 3-Panda RF-CV H=30", 3, 30, False, 6),
                                ("C5 8-Panda RF-CV H=50 S=20", 8, 50, True, 2)):
    if only and name[:2] not in only:
        continue
    for dtype in (("f64",) if "f64only" in sys.argv else ("f64", "f32")):
        cfg = config.panda_config(n_robots=N, horizon=H, scalar=abi.F64 if dtype == "f64" else abi.F32)
        if S20:
            links, offs = config.c5_sphere_table()          # 20 spheres per robot, two on every link incl. the hand
            config.set_spheres(cfg, links, offs)
        cfg.goal_estimate_mask = ((1 << N) - 2) if "CV" in name else 0
        B = rounds * cus * 4 * (64 // N)
        b = scenarios.panda_batch(cfg, B, seed=5, x_min=0.3 if N == 8 else 0.05, q_spread=0.15 if N == 8 else 0.3)
        h = FabricHandle(cfg, 0)
        q, qd, prm = (h.tensor(b[k]) for k in ("q", "qdot", "params"))
        for _ in range(3):
            h.rollout(q, qd, prm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            h.rollout(q, qd, prm)
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / 10
        sb = 8 if dtype == "f64" else 4
        S = cfg.n_spheres
        bytes_unit = sb * (28 + 9 * S * N) + sb * 23 / H
        print(f"{name:28s} {dtype} B={B:7d}: {dt*1e3:8.2f} ms  {B*N*H/dt:.3e} rollout-steps/s  {B/dt:.3e} rollouts/s  "
              f"algorithmic {B*N*H*bytes_unit/dt/1e9:7.0f} GB/s (frac {B*N*H*bytes_unit/dt/8e12:.3f})")
