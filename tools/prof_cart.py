#!/usr/bin/env python3
"""Times only the Cartesian rollout kernel (H=30, M=16 obstacles per row, 3 Pandas, chip-filling batch), with and without
obstacle accelerations: HIP events over `iters` launches after 2 warm-ups.  MRF_HIP_LIB selects the build under test.
usage: python3 tools/prof_cart.py [f64|f32] [iters]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N, H = 3, 30
cus = torch.cuda.get_device_properties(0).multi_processor_count
B = 6 * cus * 4 * (64 // N)
cfg = config.panda_config(n_robots=N, horizon=H, scalar=abi.F64 if dtype == "f64" else abi.F32)
batch = scenarios.panda_batch(cfg, B, seed=3)
h = FabricHandle(cfg, 0)
q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
sx, sv, sa = h.fk_spheres(q, qd)
ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, sa)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


ms_a, avg_a = timed(lambda: h.rollout_cartesian(q, qd, prm, ox, ov, oa, orad))
ms_0, avg_0 = timed(lambda: h.rollout_cartesian(q, qd, prm, ox, ov, None, orad))
print(json.dumps({"lib": os.environ.get("MRF_HIP_LIB", "in-tree"), "dtype": dtype, "rows": B * N, "H": H, "M": int(ox.shape[0]),
                  "ms_with_obst_a": ms_a, "ms_obst_a_null": ms_0, "checksum_a": float(avg_a.sum()), "checksum_0": float(avg_0.sum())}))
