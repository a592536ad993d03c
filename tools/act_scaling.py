#!/usr/bin/env python3
"""compute_action (obstacles in HBM) time against the number of obstacle spheres at the bench batch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_robot_fabrics_amd import config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
N=3
cus = torch.cuda.get_device_properties(0).multi_processor_count
B = 6*cus*4*(64//N)
cfg = config.panda_config(n_robots=N, horizon=1)
batch = scenarios.panda_batch(cfg, B, seed=3)
h = FabricHandle(cfg, 0)
q, qd, prm = (h.tensor(batch[k]) for k in ("q","qdot","params"))
sx, sv, sa = h.fk_spheres(q, qd)
ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, sa)
def timed(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/iters
for M in (0, 4, 8, 16, 32):
    if M == 0:
        t = timed(lambda: h.compute_action(q, qd, prm))
    else:
        reps = (M + 15)//16
        X = torch.cat([ox]*reps)[:M].contiguous(); V = torch.cat([ov]*reps)[:M].contiguous(); A = torch.cat([oa]*reps)[:M].contiguous(); Rr = torch.cat([orad]*reps)[:M].contiguous()
        t = timed(lambda: h.compute_action(q, qd, prm, X, V, A, Rr))
    print(f"M={M:3d}: {t:8.1f} us  rows={B*N}")
for rows_frac in (1/6, 2/6, 3/6):
    n = int(B*rows_frac)*N
    t = timed(lambda: h.compute_action(q[:, :n].contiguous(), qd[:, :n].contiguous(), prm[:, :n].contiguous()))
    print(f"M=0 rows={n}: {t:8.1f} us")
