"""One parent rank of tests/test_bench_children.py: joins a gloo group and runs bench.robot_sharded_in_children with a
stand-in child command; prints what it got as one JSON line.  argv: rank world port guard_s mode"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, port, guard_s, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], float(sys.argv[4]), sys.argv[5]
os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
import torch.distributed as dist

import bench

dist.init_process_group("gloo")
# the stand-in child: reads RANK from its environment like the real one; what it does depends on the mode
child = r"""
import json, os, sys, time
rank = int(os.environ["RANK"]); mode = sys.argv[1]
assert os.environ["MASTER_PORT"] != sys.argv[2], "the children rendezvous on a port of their own"
assert not any(k.startswith("TORCHELASTIC_") for k in os.environ)
if mode == "fault" and rank == 1:
    os.abort()
if mode in ("fault", "hang"):
    time.sleep(120)
if rank == 0:
    print("a banner line that is not the result")
    print(json.dumps({"peer": {"value": 1.0}, "from_rank": rank}))
"""
os.environ["TORCHELASTIC_RUN_ID"] = "parents-only"      # must not reach the children
t0 = time.monotonic()
block, clean = bench.robot_sharded_in_children(None, rank, world, guard_s, child_cmd=[sys.executable, "-c", child, mode, port])
print(json.dumps({"rank": rank, "block": block, "clean": clean, "took": time.monotonic() - t0}), flush=True)
dist.destroy_process_group()
