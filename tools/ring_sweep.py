#!/usr/bin/env python3
"""k_action_panda duration against the number of obstacles (fixed cost vs cost per obstacle); MRF_NO_RING=1 for the
register-pipelined kernel.  usage: python3 tools/ring_sweep.py [f64|f32]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
scalar = abi.F64 if dtype == "f64" else abi.F32
N = 3
cus = torch.cuda.get_device_properties(0).multi_processor_count
B = 6 * cus * 4 * (64 // N)
cfg = config.panda_config(n_robots=N, horizon=30, scalar=scalar)
batch = scenarios.panda_batch(cfg, B, seed=3)
h = FabricHandle(cfg, 0)
q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
sx, sv, sa = h.fk_spheres(q, qd)
ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, sa)
ox, ov, oa, orad = (torch.cat([t, t], 0).contiguous() for t in (ox, ov, oa, orad))     # 32 obstacles


def timed(fn, iters=10):
    for _ in range(3):
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
for M in (1, 2, 4, 8, 12, 16, 24, 32):
    for acc in (True, False):
        ms = timed(lambda: h.compute_action(q, qd, prm, ox[:M], ov[:M], oa[:M] if acc else None, orad[:M]))
        res[f"M{M}_{'acc' if acc else 'noacc'}"] = round(ms, 4)
print(json.dumps({"ring": os.environ.get("MRF_NO_RING") != "1", "dtype": dtype, "ms": res}))
