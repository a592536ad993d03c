"""bench.robot_sharded_in_children on CPU, world 2 over gloo, with stand-in children: the protocol that keeps the
robot-sharded block of a multi-GPU bench run from taking the headline with it (a result is relayed; a child that dies ends
the other rank's child early and is named; a hang ends at the guard) -- no GPU involved."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def run_parents(mode, guard_s):
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    port = bench.free_port()        # below the ephemeral range: no outgoing connection can take it before the parents bind it
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "bench_children_worker.py"), str(r), "2", str(port), str(guard_s), mode],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=240)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith("{")][-1]))
    return sorted(outs, key=lambda d: d["rank"])


def test_result_is_relayed_from_rank_zeros_child():
    r0, r1 = run_parents("ok", 60)
    assert r0["clean"] and r1["clean"] and r1["block"] is None
    assert r0["block"]["peer"] == {"value": 1.0} and "child process per rank" in r0["block"]["isolation"]


def test_a_dying_child_is_named_and_ends_the_other_child_early():
    r0, r1 = run_parents("fault", 100)
    assert not r0["clean"] and not r1["clean"]
    assert r0["block"]["children"][1] == "killed by signal 6"
    assert "another rank's child had failed" in r0["block"]["children"][0]
    assert "rank 1's child: killed by signal 6" in r0["block"]["error"]
    assert r0["took"] < 60 and r1["took"] < 60          # not the 100 s guard, not the child's 120 s sleep


def test_a_hang_ends_at_the_guard():
    r0, r1 = run_parents("hang", 2)
    assert not r0["clean"] and not r1["clean"]
    assert all("timeout" in c or "failed" in c for c in r0["block"]["children"])
    assert "timeout" in r0["block"]["error"] and r0["took"] < 60
