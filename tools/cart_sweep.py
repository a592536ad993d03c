#!/usr/bin/env python3
"""k_rollout_cart_panda duration against the number of obstacles M (H=30, no accelerations): the first 10 obstacles of a row
are resident in LDS, the rest stream from HBM / the Infinity Cache in every step -- is a streamed obstacle dearer than a
resident one?  usage: python3 tools/cart_sweep.py"""
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
B = 6 * cus * 4 * (64 // N)
cfg = config.panda_config(n_robots=N, horizon=H, scalar=abi.F64)
batch = scenarios.panda_batch(cfg, B, seed=3)
h = FabricHandle(cfg, 0)
q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
sx, sv, sa = h.fk_spheres(q, qd)
ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, sa)
ox, ov, orad = (torch.cat([t, t], 0).contiguous() for t in (ox, ov, orad))     # 32 obstacles


def timed(fn, iters=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


res = {}
for M in (2, 6, 10, 11, 12, 14, 16, 20, 24):
    res[f"M{M}"] = round(timed(lambda: h.rollout_cartesian(q, qd, prm, ox[:M], ov[:M], None, orad[:M])), 4)
print(json.dumps({"rows": B * N, "H": H, "ms": res}))
