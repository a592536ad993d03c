#!/usr/bin/env python3
"""mrf_rollout_cartesian_coupled at throughput batch sizes (3 Pandas, H=30): HIP events over the whole call, for the
link-origin sphere table (the joint-space rollout planner's) and for the one-sphere-per-link offset table of the evaluation
scripts (16 obstacle spheres per robot each).  MRF_CART_TILE=0 takes the obstacle-array path (k_publish_obstacles +
k_rollout_cart_panda), default the LDS-tile kernel (k_rollout_cartc_panda).  usage: python3 tools/prof_cart_coupled.py [B ...]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

N, H = 3, 30
cus = torch.cuda.get_device_properties(0).multi_processor_count
sizes = [int(a) for a in sys.argv[1:]] or [2 * cus * 4 * 21, 6 * cus * 4 * 21]
for table in ("link origins", "one offset sphere per link"):
    cfg = config.panda_config(n_robots=N, horizon=H, scalar=abi.F64)
    cfg.kernel_select = 1
    if table != "link origins":
        links, offs = config.sphere_offsets_per_link(1)
        config.set_spheres(cfg, links, offs)
    h = FabricHandle(cfg, 0)
    for B in sizes:
        batch = scenarios.tiled_batch(cfg, B, seed=3, x_min=0.2)
        q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
        for _ in range(2):
            avg = h.rollout_cartesian_coupled(q, qd, prm)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            avg = h.rollout_cartesian_coupled(q, qd, prm)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 6
        print(json.dumps({"table": table, "tile": os.environ.get("MRF_CART_TILE") != "0", "scenarios": B, "rows": B * N,
                          "ms": round(ms, 4), "rollout_steps_per_s": B * N * H / (ms * 1e-3), "checksum": float(avg.double().sum())}))
