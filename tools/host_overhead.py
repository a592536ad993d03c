#!/usr/bin/env python3
"""Where a single-scenario call spends its time: upload, launch + kernel, download (and pinned-buffer alternatives)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multi_robot_fabrics_amd import config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
cfg = config.panda_config(n_robots=2, horizon=10)
b = scenarios.panda_batch(cfg, 1, seed=0)
h = FabricHandle(cfg, 0)
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
q,qd,prm = b["q"],b["qdot"],b["params"]
print("upload 3 arrays      %.1f us" % t(lambda: h.upload(q,qd,prm)))
dq,dqd,dprm = h.upload(q,qd,prm)
print("rollout launch+sync  %.1f us" % t(lambda: (h.rollout(dq,dqd,dprm), torch.cuda.synchronize())))
avg = h.rollout(dq,dqd,dprm)
print("avg.cpu()            %.1f us" % t(lambda: avg.cpu()))
print("full: upload+rollout+cpu %.1f us" % t(lambda: h.rollout(*h.upload(q,qd,prm)).cpu()))
print("action launch+sync   %.1f us" % t(lambda: (h.compute_action_coupled(dq,dqd,dprm), torch.cuda.synchronize())))
pin = torch.empty(43*2, dtype=torch.float64).pin_memory(); dev = torch.empty(43*2, dtype=torch.float64, device='cuda')
flat = np.concatenate([q.ravel(),qd.ravel(),prm.ravel()])
def up2():
    pin.numpy()[:] = flat; dev.copy_(pin, non_blocking=True)
print("pinned upload        %.1f us" % t(up2))
out_pin = torch.empty(2, dtype=torch.float64).pin_memory()
def down2():
    out_pin.copy_(avg, non_blocking=True); torch.cuda.current_stream().synchronize()
print("pinned download+sync %.1f us" % t(down2))
