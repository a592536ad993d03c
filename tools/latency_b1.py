#!/usr/bin/env python3
"""Single-scenario latency of one control step (rollout + compute_action per robot) through the C ABI:
what a real-time controller of 3 Pandas sees.  usage: python3 tools/latency_b1.py [dtype]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
N, H = 3, 30
sc = abi.F64 if dtype == "f64" else abi.F32
cfg_r = config.panda_config(n_robots=N, horizon=H, scalar=sc)
cfg_r.goal_estimate_mask = 0b110
cfg_a = config.panda_config(n_robots=N, horizon=1, scalar=sc)
b = scenarios.panda_batch(cfg_r, 1, seed=1)
hr, ha = FabricHandle(cfg_r, 0), FabricHandle(cfg_a, 0)
q, qd, prm = (hr.tensor(b[k]) for k in ("q", "qdot", "params"))
S = 8
others = torch.tensor([[j for j in range(N) if j != i] for i in range(N)], device="cuda")
orad = torch.full(((N - 1) * S, N), 0.08, dtype=ha.dtype, device="cuda")


def gather(sph):
    g = sph.view(S, 3, 1, N)[:, :, :, others]
    return g.permute(4, 0, 1, 2, 3).reshape((N - 1) * S, 3, N).contiguous()


def step():
    avg = hr.rollout(q, qd, prm)
    sx, sv, sa = ha.fk_spheres(q, qd)
    act = ha.compute_action(q, qd, prm, gather(sx), gather(sv), gather(sa), orad)
    return avg, act


for _ in range(20):
    step()
torch.cuda.synchronize()
ts = []
for _ in range(200):
    t0 = time.perf_counter()
    avg, act = step()
    a = act.cpu()          # the controller needs the action on the host
    ts.append(time.perf_counter() - t0)
ts = sorted(ts)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(100):
    hr.rollout(q, qd, prm)
ev1.record()
torch.cuda.synchronize()
print(f"{dtype} B=1 control step (rollout H=30 + fk + gather + 3 actions + D2H): median {ts[100]*1e3:.3f} ms "
      f"p90 {ts[180]*1e3:.3f} ms -> {1/ts[100]:.0f} Hz; rollout kernel alone {ev0.elapsed_time(ev1)/100*1e3:.1f} us")
