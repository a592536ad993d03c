"""Worker of tests/test_gpu_sharded_abi.py: one rank of a robot group whose ranks SHARE the single GPU of the test box.
Launched by torch.distributed.run with the gloo backend (it only carries the bootstrap blobs -- IPC handles -- and the
final verdict); the exchange itself is the library's PEER transport: kernels of the two processes run concurrently on
the one device, store into each other's IPC-mapped buffers and poll each other's flags.

argv: n_robots horizon n_scen table(lo|offsets|offsets20) dtype(f64|f32) [exchange(joints|spheres)]
Every rank also runs the fused single-GPU rollout (mrf_rollout) on the full batch and compares its owned rows."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
from multi_robot_fabrics_amd.sharded import ShardedRollout


def main():
    N, H, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    table, dtype = sys.argv[4], sys.argv[5]
    exchange = sys.argv[6] if len(sys.argv) > 6 else "joints"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    cfg = config.panda_config(n_robots=N, horizon=H, scalar=abi.F64 if dtype == "f64" else abi.F32)
    cfg.goal_estimate_mask = ((1 << N) - 1) & ~1
    cfg.exchange = {"joints": abi.EXCHANGE_JOINTS, "spheres": abi.EXCHANGE_SPHERES}[exchange]
    if table == "offsets":
        links, offs = config.sphere_offsets_per_link(2)
        config.set_spheres(cfg, links, offs, [0.06] * len(links))
    elif table == "offsets20":              # BASELINE config 5's table: 20 spheres with link-local offsets
        links, offs = config.sphere_offsets_per_link(3)
        config.set_spheres(cfg, links[:20], offs[:20], [0.05] * 20)
    batch = scenarios.panda_batch(cfg, B, seed=77, **({"x_min": 0.3, "q_spread": 0.15} if N > 3 else {}))
    sr = ShardedRollout(cfg, rank, world, device_index=0, transport="peer", max_scenarios=B + 5)
    h = sr.backend.h
    rows = sr.own_rows(B).numpy()
    q, qd, prm = (h.tensor(np.ascontiguousarray(batch[k][:, rows])) for k in ("q", "qdot", "params"))
    ref = FabricHandle(cfg, 0)
    fq, fqd, fprm = (ref.tensor(batch[k]) for k in ("q", "qdot", "params"))
    want_avg, tq, tqd = ref.rollout(fq, fqd, fprm, want_traj=True)
    errs = []
    for rep in range(3):        # repeated calls: the flag sequence numbers keep counting across rollouts
        qq, qqd = q.clone(), qd.clone()
        torch.cuda.synchronize()
        dist.barrier()          # ranks sharing ONE device: nothing is queued in front of a peer kernel while the others' already wait
        avg = sr.rollout(qq, qqd, prm)
        h.comm_status()
        e = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))
        errs.append(max(e(avg, want_avg[rows]), e(qq, tq[-1][:, rows]), e(qqd, tqd[-1][:, rows])))
    out = [None] * world
    info = h.comm_info()
    dist.all_gather_object(out, {"rank": rank, "first": sr.first, "count": sr.count, "err": max(errs), "exchange": info["exchange"],
                                 "scalars": info["exchange_scalars_per_robot"], "peers": h.comm_peer_info(),
                                 "peers_one_hop": info["peers_one_hop"], "coresident": info["coresident_workgroups"],
                                 "paired": info["paired_blocks"], "tagged": info["tagged_payload"]})
    if rank == 0:
        print(json.dumps({"ranks": out}))
    dist.barrier()
    h.comm_destroy()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
