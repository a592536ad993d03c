"""Where the waves of the persistent peer kernels spend their time (development build: tools/build_variant.sh timing -
-DMRF_PEER_TIMING, then MRF_HIP_LIB=ab/libtiming.so python tools/peer_timing.py [scenarios] [robots] [horizon] [ranks]).
All ranks of one robot group in this process on one GPU (sharded.InProcessGroup); per rank and per wave: microseconds in
publish (payload stores + release + flags), in the wait for the peers' flags, in the remote fold (payload loads + re-walked
chains) and in the kernel as a whole."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
from multi_robot_fabrics_amd.sharded import InProcessGroup


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 43008
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    G = int(sys.argv[4]) if len(sys.argv) > 4 else N
    cfg = config.panda_config(n_robots=N, horizon=H)
    cfg.goal_estimate_mask = ((1 << N) - 1) & ~1
    batch = scenarios.tiled_batch(cfg, B, seed=77)
    ref = FabricHandle(cfg, 0)
    out = {"scenarios": B, "robots": N, "horizon": H, "ranks": G, "modes": {}}
    for mode, xk, paired in (("joints_paired", abi.EXCHANGE_JOINTS, "1"), ("joints_single", abi.EXCHANGE_JOINTS, "0"),
                             ("joints_tagged", abi.EXCHANGE_JOINTS, "0"), ("spheres", abi.EXCHANGE_SPHERES, "0")):
        os.environ["MRF_PEER_PAIRED"] = paired
        os.environ["MRF_PEER_TAGGED"] = "1" if mode == "joints_tagged" else "0"
        c = cfg.copy()
        c.exchange = xk
        grp = InProcessGroup(c, G, B)
        rows = [grp.own_rows(g, B) for g in range(G)]
        base = [tuple(ref.tensor(np.ascontiguousarray(batch[k][:, r.numpy()])) for k in ("q", "qdot", "params")) for r in rows]
        fn = grp.handles[0].lib.mrf_debug_peer_timing
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
        buf = (C.c_ulonglong * 6)()
        for rep in range(3):
            grp.rollout([(q.clone(), qd.clone(), prm) for q, qd, prm in base])
            torch.cuda.synchronize()
            per_rank = []
            for h in grp.handles:
                assert fn(h._h, buf) == 0
                pub, wait, remote, whole, waves, stage = (int(v) for v in buf)
                us = lambda t: t / max(waves, 1) / 100.0          # wall_clock64: 100 MHz
                per_rank.append({"waves": waves, "publish_us": us(pub), "wait_us": us(wait), "remote_fold_us": us(remote), "of_which_payload_loads_us": us(stage),
                                 "kernel_us": us(whole)})
        out["modes"][mode] = {"paired_blocks": grp.handles[0].comm_info()["paired_blocks"], "per_rank_last_rollout": per_rank}
        grp.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
