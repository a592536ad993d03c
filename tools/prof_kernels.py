#!/usr/bin/env python3
"""Every kernel behind include/mrf.h once, at a chip-filling batch: duration (HIP events, 10 launches after 3 warm-ups),
rows/s and the algorithmic-bytes rate of DESIGN.md section 5 against the 8 TB/s HBM peak.
usage: python3 tools/prof_kernels.py [f64|f32]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
sb = 8 if dtype == "f64" else 4
scalar = abi.F64 if dtype == "f64" else abi.F32
N, H = 3, 30
cus = torch.cuda.get_device_properties(0).multi_processor_count
B = 6 * cus * 4 * (64 // N)
rows = B * N
cfg = config.panda_config(n_robots=N, horizon=H, scalar=scalar)
cfg.goal_estimate_mask = 0b110
batch = scenarios.panda_batch(cfg, B, seed=3)
h = FabricHandle(cfg, 0)
q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
S = cfg.n_spheres
M = S * (N - 1)


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
    return e0.elapsed_time(e1) * 1e-3 / iters


out = []


def report(name, dt, units, bytes_unit, unit_name):
    r = {"kernel": name, "ms": 1e3 * dt, unit_name + "_per_s": units / dt, "algorithmic_GBps": units * bytes_unit / dt / 1e9,
         "frac_of_8TBps": units * bytes_unit / dt / 8e12, "bytes_per_unit": bytes_unit}
    out.append(r)
    print(json.dumps(r), flush=True)


sx, sv, sa = h.fk_spheres(q, qd)
ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, sa)
report("k_fk_spheres_panda", timed(lambda: h.fk_spheres(q, qd)), rows, sb * (14 + 9 * S), "rows")
report("k_action_panda (M=16 from HBM)", timed(lambda: h.compute_action(q, qd, prm, ox, ov, oa, orad)), rows,
       sb * (14 + 29 + 10 * M + 7), "rows")
report("k_action_panda (M=16 from HBM, obst_a = NULL)", timed(lambda: h.compute_action(q, qd, prm, ox, ov, None, orad)), rows,
       sb * (14 + 29 + 7 * M + 7), "rows")
report("k_action_coupled", timed(lambda: h.compute_action_coupled(q, qd, prm)), rows, sb * (14 + 29 + 7), "rows")
report("k_rollout_panda (H=30)", timed(lambda: h.rollout(q, qd, prm)), rows * H, sb * (28 + 9 * S * N) + sb * 23 / H,
       "rollout_steps")
report("k_rollout_cart_panda (H=30, M=16)", timed(lambda: h.rollout_cartesian(q, qd, prm, ox, ov, oa, orad), iters=4), rows * H,
       sb * (43 + 10 * M * H + 14) / H, "rollout_steps")
report("k_rollout_cart_panda (H=30, M=16, obst_a = NULL)", timed(lambda: h.rollout_cartesian(q, qd, prm, ox, ov, None, orad), iters=4),
       rows * H, sb * (43 + 7 * M * H + 14) / H, "rollout_steps")
# the two halves of a robot-sharded rollout step, all robots on this GPU
SX = h.exchange_spheres
sph = torch.empty((N, SX, 9, B), dtype=h.dtype, device="cuda")
ssq = torch.zeros((rows,), dtype=h.dtype, device="cuda")
qq, qqd = q.clone(), qd.clone()
report("k_step_predict", timed(lambda: h.step_predict(B, 0, N, qq, qqd, sph)), rows, sb * (21 + 9 * SX), "rows")
qq, qqd = q.clone(), qd.clone()
h.step_predict(B, 0, N, qq, qqd, sph)
report("k_step_action", timed(lambda: h.step_action(B, 0, N, qq, qd.clone(), prm, sph, ssq)), rows,
       sb * (14 + 29 + 9 * SX * (N - 1) + 8), "rows")
# ... and with the joint-state payload (round 6): all robots on this GPU exchange on chip, nothing is re-walked
jst = torch.empty((N, 21, B), dtype=h.dtype, device="cuda")
qq = q.clone()
report("k_step_predict_joints", timed(lambda: h.step_predict_joints(B, 0, N, qq, qqd, jst)), rows, sb * (21 + 21), "rows")
qq = q.clone()
h.step_predict_joints(B, 0, N, qq, qqd, jst)
# the launch the RCCL transport's loop runs for steps 0 .. H-2: the action AND the following step's position update + joint state
# (into the rank's send block: a group of one gathers in place)
qd_run = qd.clone()
report("k_step_action_joints + next step's predict (all robots local)",
       timed(lambda: h.step_action_predict_joints(B, 0, N, qq, qd_run, prm, jst, ssq, jst)), rows, sb * (7 + 21 + 29 + 8 + 7 + 21), "rows")
# the peer transport's persistent kernel with a group of one (round 6: all robots on this GPU exchange on chip -- the fused
# kernel's step inside the persistent block loop, staging + commit passes around it)
from multi_robot_fabrics_amd.sharded import ShardedRollout
sr = ShardedRollout(cfg, 0, 1, device_index=0, transport="peer", max_scenarios=B)
report("k_rollout_peer (group of one, H=30)", timed(lambda: sr.rollout(q.clone(), qd.clone(), prm), iters=4), rows * H,
       sb * (28 + 9 * SX * N) + sb * 23 / H, "rollout_steps")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"dtype": dtype, "scenarios": B, "robots": N, "kernels": out},
          open(os.path.join(ROOT, "gpurun_out", f"prof_kernels_{dtype}.json"), "w"), indent=1)
