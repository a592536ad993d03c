"""World 1 on one GPU: the robot-sharded rollout (mrf_rollout_sharded) of both transports and both payloads against the
fused kernel (mrf_rollout) on the same batch -- what a rank that owns ALL robots pays for running inside the sharded
machinery (within 1.15x of the fused kernel).  Prints one JSON line.
usage: python tools/shard_ab.py [scenarios] [robots] [horizon] [iters]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
from multi_robot_fabrics_amd.sharded import ShardedRollout


def timed(fn, iters):
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 129024
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    cfg = config.panda_config(n_robots=N, horizon=H)
    cfg.goal_estimate_mask = ((1 << N) - 1) & ~1
    batch = scenarios.tiled_batch(cfg, B, seed=77)
    h = FabricHandle(cfg, 0)
    q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
    out = {"scenarios": B, "robots": N, "horizon": H}
    ms, want = timed(lambda: h.rollout(q, qd, prm), iters)
    out["fused_ms"] = ms
    for transport in ("peer", "rccl"):
        for xname, xk in (("joints", abi.EXCHANGE_JOINTS), ("spheres", abi.EXCHANGE_SPHERES)):
            c = cfg.copy()
            c.exchange = xk
            sr = ShardedRollout(c, 0, 1, device_index=0, transport=transport, max_scenarios=B)
            ms, avg = timed(lambda: sr.rollout(q.clone(), qd.clone(), prm), iters)
            sr.backend.h.comm_status()
            err = float((avg - want).abs().max() / want.abs().max())
            out[f"{transport}_{xname}"] = {"ms": ms, "vs_fused": ms / out["fused_ms"], "rel_err_vs_fused": err}
            sr.backend.h.comm_destroy()
    # ONE ROBOT PER RANK, priced on one GPU without peers: the action kernel of a rank that owns robot 0 only, fed with a
    # gathered array that was produced locally -- joint payload (two remote chains re-walked per lane) against the sphere
    # payload (spheres read from memory).  Same rows, same arithmetic result.
    rows0 = torch.arange(B, device=h.device) * N
    q0, qd0, prm0 = q[:, rows0].contiguous(), qd[:, rows0].contiguous(), prm[:, rows0].contiguous()
    S = h.exchange_spheres
    jst = torch.zeros((N, 21, B), dtype=h.dtype, device=h.device)
    sph = torch.zeros((N, S, 9, B), dtype=h.dtype, device=h.device)
    qq = q.clone()
    h.step_predict_joints(B, 0, N, qq, qd, jst)
    qq2 = q.clone()
    h.step_predict(B, 0, N, qq2, qd, sph)
    q0n = qq[:, rows0].contiguous()
    ss = torch.zeros((B,), dtype=h.dtype, device=h.device)
    res = {}
    for name, fn in (("joints", lambda: h.step_action_joints(B, 0, 1, q0n, qd0.clone(), prm0, jst, ss)),
                     ("spheres", lambda: h.step_action(B, 0, 1, q0n, qd0.clone(), prm0, sph, ss))):
        ms, _ = timed(fn, 4 * iters)
        res[name] = ms
    a, b = qd0.clone(), qd0.clone()
    h.step_action_joints(B, 0, 1, q0n, a, prm0, jst, ss)
    h.step_action(B, 0, 1, q0n, b, prm0, sph, ss)
    out["one_robot_per_rank_action_step"] = {"rows": B, "joints_ms": res["joints"], "spheres_ms": res["spheres"],
                                             "joints_vs_spheres": res["joints"] / res["spheres"],
                                             "rel_diff": float((a - b).abs().max() / b.abs().max()),
                                             "remote_robots_per_lane": N - 1}
    clone_ms, _ = timed(lambda: (q.clone(), qd.clone()), iters)
    out["clone_ms"] = clone_ms
    print(json.dumps(out))


if __name__ == "__main__":
    main()
