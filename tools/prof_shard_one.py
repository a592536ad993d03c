#!/usr/bin/env python3
"""ONE ROBOT PER RANK, priced on one GPU without peers ("the added re-walk cost at one robot per rank is
reported from counters"): the per-step kernels of a rank that owns robot 0 of three, fed with a gathered array produced
locally, at the bench batch --
    joint-state payload   k_step_predict_joints + k_step_action_joints<.., XK_JOINTS>: the two remote chains re-walked per lane
    sphere payload        k_step_predict        + k_step_action: the 2 x 6 remote spheres read from memory
Same rows, same result (checked).  Under rocprofv3 (tools/collect_kernel_pmc.sh <tag> tools/prof_shard_one.py shard1) the
counter summary is keyed by kernel name.
usage: python3 tools/prof_shard_one.py [f64]"""
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
B = 6 * cus * 4 * 64            # six rounds of resident waves with ONE owned robot per lane
cfg = config.panda_config(n_robots=N, horizon=H)
cfg.goal_estimate_mask = 0b110
batch = scenarios.tiled_batch(cfg, B, seed=3)
h = FabricHandle(cfg, 0)
q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
S = h.exchange_spheres


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


rows0 = torch.arange(B, device=h.device) * N
q0, qd0, prm0 = q[:, rows0].contiguous(), qd[:, rows0].contiguous(), prm[:, rows0].contiguous()
jst = torch.zeros((N, 21, B), dtype=h.dtype, device=h.device)
sph = torch.zeros((N, S, 9, B), dtype=h.dtype, device=h.device)
qa, qb = q.clone(), q.clone()
h.step_predict_joints(B, 0, N, qa, qd, jst)           # every robot's payload of this step, as the gather would deliver it
h.step_predict(B, 0, N, qb, qd, sph)
q0n = qa[:, rows0].contiguous()
ss = torch.zeros((B,), dtype=h.dtype, device=h.device)
own_j = torch.zeros((1, 21, B), dtype=h.dtype, device=h.device)
own_s = torch.zeros((1, S, 9, B), dtype=h.dtype, device=h.device)
out = {"rows": B, "robots": N, "owned": 1, "remote_robots_per_lane": N - 1, "exchange_spheres": S}
out["predict_joints_ms"] = timed(lambda: h.step_predict_joints(B, 0, 1, q0.clone(), qd0, own_j))
out["predict_spheres_ms"] = timed(lambda: h.step_predict(B, 0, 1, q0.clone(), qd0, own_s))
out["action_joints_ms"] = timed(lambda: h.step_action_joints(B, 0, 1, q0n, qd0.clone(), prm0, jst, ss))
out["action_spheres_ms"] = timed(lambda: h.step_action(B, 0, 1, q0n, qd0.clone(), prm0, sph, ss))
out["clone_ms"] = timed(lambda: qd0.clone())
a, b = qd0.clone(), qd0.clone()
h.step_action_joints(B, 0, 1, q0n, a, prm0, jst, ss)
h.step_action(B, 0, 1, q0n, b, prm0, sph, ss)
out["rel_diff_joints_vs_spheres"] = float((a - b).abs().max() / b.abs().max())
out["step_joints_vs_spheres"] = (out["predict_joints_ms"] + out["action_joints_ms"]) / (out["predict_spheres_ms"] + out["action_spheres_ms"])
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "prof_shard_one.json"), "w"), indent=1)
