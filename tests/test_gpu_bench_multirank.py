"""bench.py's world > 1 protocol (per-rank batches, barrier + synchronize around the timed region, MAX over ranks,
one JSON line from rank 0) exercised with two ranks sharing the single GPU of the test box (MRF_BENCH_SHARE_GPU=1:
gloo instead of RCCL for the barrier and the all-reduce of the elapsed time)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_one_gpu_scenario_sharding():
    env = dict(os.environ, MRF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--scenarios", "4032"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                      # rank 0 only
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak"
    assert r["config"]["scenarios_per_gpu"] == 4032
    # whole-job aggregate: both ranks' scenarios over the slowest rank's time
    assert abs(r["value"] - 2 * 4032 * 3 / (r["ms_per_step"] * 3e-3)) / r["value"] < 1e-9
    assert "cpu_baseline" not in r and "single_scenario" not in r
    # the rollout kernel is bound by the f64 vector ALU; the SURVEY 8d algorithmic-HBM figure is a secondary block
    assert r["roofline"]["bound"] in ("valu_f64", "hbm") and r["roofline"]["frac"] > 0
    assert r["roofline"]["hbm_algorithmic"]["bound"] == "hbm"
    assert r["parity_spot_check"]["ok"], r["parity_spot_check"]
    # what the MAX over ranks was taken of: every rank's elapsed time, kernel time, clock and device
    pr = r["per_rank"]
    assert len(pr["elapsed_s"]["all"]) == 2 and pr["elapsed_s"]["max"] >= pr["elapsed_s"]["min"] > 0
    assert abs(pr["elapsed_s"]["max"] - r["ms_per_step"] * 3e-3) / pr["elapsed_s"]["max"] < 0.05
    assert len(pr["rollout_kernel_ms"]["all"]) == 2 and len(pr["devices"]) == 2 and pr["distinct_devices"] == 1   # shared GPU
    assert all(g is None or 0.5 < g < 3.5 for g in pr["shader_ghz"])
    assert r["roofline"]["effective_clock_ghz"] == r["effective_clock_ghz"]


def _audit_fields(roof, link_bytes):
    """What the first run on real links is audited with: the measured HBM traffic of the transport's own
    kernels, and the link model's prediction next to the measured step."""
    link = roof["link"]
    assert abs(link["predicted_ms_per_step"] - link_bytes / 153e9 * 1e3) < 1e-12
    assert abs(link["predicted_ms_per_rollout"] - link["predicted_ms_per_step"] * link["steps"]) < 1e-9
    assert link["measured_ms_per_step"] > 0
    assert set(link["model"]["link_bound_above_scenarios"]) == {"peer", "rccl"} and "peer_vs_rccl" in link["model"]
    assert roof["traffic"] is not None and roof["traffic"] > 0 and roof["traffic_key"].startswith("sharded_")


def _topology_fields(topo, n_devices=1):
    """the node as HIP shows it, recorded before the sharded block."""
    assert topo["n_devices"] == n_devices and topo["status"] == "MRF_OK"
    assert topo["can_access_peer"] == [[1]] and topo["hops"] == [[0]] and topo["link_type"] == [["same device"]]


def test_world_one_robot_sharded_block_carries_the_audit_fields():
    """The default single-GPU run's secondary block: both transports x both payloads with a group of one (no link), the
    measured HBM traffic of their kernels, the link model's prediction for one robot per GPU at this batch -- 21 joint-state
    scalars against 54 sphere scalars per robot -- and the node topology."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--scenarios", "2016",
           "--no-configs", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["exit_code"] == 0
    assert r["build"] == {"has_f32": False, "has_wp": False, "abi_version": 6} or r["build"]["abi_version"] == 6
    _topology_fields(r["robot_sharded"]["topology"])
    pred = {}
    for exchange, scalars, suffix in (("joints", 21, ""), ("spheres", 54, "_spheres")):
        for transport in ("rccl", "peer"):
            blk = r["robot_sharded"][transport + suffix]
            assert "error" not in blk, blk
            assert blk["exchange"] == exchange and blk["config"]["exchange_scalars_per_robot"] == scalars
            assert blk["allgather_bytes_per_rank_per_step"] == 3 * scalars * 8 * 2016     # the 3 robots of the one rank
            roof = blk["roofline"]
            assert roof["bound"] == "hbm" and roof["link"]["bytes_per_link_per_step"] == 0
            assert abs(roof["link"]["predicted_ms_per_step"] - scalars * 8 * 2016 / 153e9 * 1e3) < 1e-12    # one robot per GPU
            assert roof["link"]["measured_ms_per_step"] > 0 and roof["traffic"] > 0
            assert roof["traffic_key"] == f"sharded_{transport}_{exchange}_f64"
            assert blk["parity_vs_fused_kernel"]["ok"], blk["parity_vs_fused_kernel"]
            comm = blk["ranks"][0]["comm"]
            assert comm["exchange"] == exchange and comm["exchange_scalars_per_robot"] == scalars
            if transport == "peer":
                assert blk["ranks"][0]["peers"] == [{"device": 0, "can_access_peer": 1, "link_type": 0, "hops": 0, "rank": 0,
                                                    "link": "same device"}]
            pred[exchange] = roof["link"]["predicted_ms_per_step"]
    assert pred["spheres"] / pred["joints"] >= 2.5            # the link model's prediction, BASELINE config 4
    # ... and the one multi-rank measurement a single GPU can make: three ranks x one robot inside the bench process
    one = r["robot_sharded"]["one_die_group"]
    assert "error" not in one, one
    assert one["scenarios"] == 2016 and one["fused_kernel_ms_same_scenarios"] > 0
    for exchange, scalars in (("joints", 21), ("spheres", 54)):
        leg = one[exchange]
        assert leg["parity_vs_fused_kernel"]["ok"] and leg["ms_device_slowest_rank"] > 0, leg
        assert leg["bytes_per_rank_pair_per_step"] == scalars * 2016 * 8
        assert (leg["comm_rank0"]["world"], leg["comm_rank0"]["robot_count"], leg["comm_rank0"]["peer_buffers_mapped"]) == (3, 1, 2)


def test_bare_launch_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher: the parent starts torch.distributed.run as a child before touching
    the GPU and relays exactly one JSON line."""
    env = dict(os.environ, MRF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scenarios", "2016"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["config"]["scenarios_per_gpu"] == 2016
    # the secondary block of a default run at world > 1: robots of a scenario spread over the two ranks (2 + 1), one
    # scenario batch per replica, the exchange inside the library (peer transport here: the ranks share the GPU)
    peer = r["robot_sharded"]["peer"]
    assert "error" not in peer, peer
    assert peer["config"]["robot_group_ranks"] == 2 and peer["config"]["robots_per_rank"] == [2, 1]
    assert peer["parity_vs_fused_kernel"]["ok"] and peer["roofline"]["bound"] == "xgmi_link"
    # the block proves what ran: every rank's communicator as the library reports it, its device, its own time per rollout
    assert [x["comm"]["transport"] for x in peer["ranks"]] == ["peer", "peer"]
    assert [x["comm"]["world"] for x in peer["ranks"]] == [2, 2] and [x["comm"]["rank"] for x in peer["ranks"]] == [0, 1]
    assert [x["comm"]["peer_buffers_mapped"] for x in peer["ranks"]] == [1, 1]
    assert [(x["comm"]["robot_first"], x["comm"]["robot_count"]) for x in peer["ranks"]] == [(0, 2), (2, 1)]
    assert peer["rollout_ms_per_rank"]["max"] >= peer["rollout_ms_per_rank"]["min"] > 0 and len(peer["devices"]) == 2
    assert peer["rccl_ranks_seen"] is None


@pytest.mark.parametrize("exchange,scalars", [("joints", 21), ("spheres", 54)])
def test_robot_sharded_bench_two_ranks_one_gpu_peer_transport(exchange, scalars):
    """bench.py --shard robots over the PEER transport with the two ranks of the robot group sharing the one GPU, either
    payload: the JSON line carries the link / algorithmic-HBM roofline views, the payload on the wire, where each rank's
    mapped peer buffer lives, and the sharded result agrees with the fused kernel."""
    env = dict(os.environ, MRF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MRF_PEER_TIMEOUT_MS="4000")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scenarios", "2016",
           "--shard", "robots", "--transport", "peer", "--exchange", exchange]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and r["transport"] == "peer" and r["exchange"] == exchange
    assert r["config"]["exchange_scalars_per_robot"] == scalars
    _topology_fields(r["topology"])
    for x in r["ranks"]:            # both ranks sit on device 0: the peer's mapped buffer is reported there, zero hops
        assert [p["device"] for p in x["peers"]] == [0, 0] and x["comm"]["peers_one_hop"] == 0
        assert x["comm"]["exchange"] == exchange and x["comm"]["exchange_scalars_per_robot"] == scalars
    assert r["config"]["robot_group_ranks"] == 2 and r["config"]["robots_per_rank"] == [2, 1]
    assert r["parity_vs_fused_kernel"]["ok"], r["parity_vs_fused_kernel"]
    # 3 robots on 2 ranks: one group [2] whose ranks carry up to 2 robots -> half the scenarios of a 1-robot-per-rank group
    assert r["config"]["robot_groups"] == [2] and r["config"]["scenarios_per_group"] == [1008]
    assert r["config"]["robots_per_rank_all"] == [2, 1]
    assert abs(r["value"] - 1008 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 1e-9
    # both roofline views at world > 1: the link the exchange crosses and the algorithmic HBM bytes of the exchanged formulation
    assert r["roofline"]["bound"] == "xgmi_link" and r["roofline"]["link"]["bytes_per_link_per_step"] == 2 * scalars * 1008 * 8
    assert r["roofline"]["link"]["frac"] > 0 and r["roofline"]["hbm_algorithmic"]["frac"] > 0
    assert r["roofline"]["hbm_algorithmic"]["bytes_per_unit"] > 0
    assert r["allgather_bytes_per_rank_per_step"] == 2 * scalars * 1008 * 8
    # rank 0 owns robots 0 and 1: one on-chip partner, one remote robot read from the local buffer, one peer stored into
    assert r["roofline"]["exchange_bytes_per_row_step"] == {"read_from_the_local_buffer": scalars * 8, "stored_into_peer_buffers": scalars * 8,
                                                            "on_chip_partners": 1}
    _audit_fields(r["roofline"], link_bytes=2 * scalars * 1008 * 8)


def test_stuck_secondary_block_cannot_take_the_headline_with_it():
    """The wall-clock guard around the robot-sharded block (bench.robot_sharded_in_children: one child process per rank,
    killed by its parent when the guard expires): with a guard far shorter than the block, rank 0 still emits the
    scenario-sharded headline -- with an error marker in place of the block -- and the run ends with the distinct exit code
    bench.SHARD_TIMEOUT_RC (3): a hung exchange is not a clean run, but the line is there."""
    env = dict(os.environ, MRF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MRF_BENCH_SHARD_TIMEOUT_S="0.01")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scenarios", "2016"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["parity_spot_check"]["ok"]
    assert r["exit_code"] == 3      # what the launcher's code is taken from (spawn_ranks), not a guess from error texts
    err = r["robot_sharded"]["error"]
    assert "timeout" in err and "rank 0's child" in err, err


def test_faulting_secondary_block_cannot_take_the_headline_with_it():
    """A GPU fault aborts the process it happens in.  The robot-sharded block's first contact with real links must not be
    able to do that to the ranks holding the headline: rank 1's CHILD dies by SIGABRT here (test hook), the parents
    report whose child went how, end the other child early, emit the complete headline and leave with exit code 3."""
    env = dict(os.environ, MRF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MRF_BENCH_CHILD_FAULT="1",
               MRF_PEER_TIMEOUT_MS="60000")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--scenarios", "2016"]
    import time
    t0 = time.monotonic()
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    took = time.monotonic() - t0
    assert out.returncode == 3, (out.returncode, out.stderr[-2000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["parity_spot_check"]["ok"] and "roofline" in r
    assert r["exit_code"] == 3
    assert "rank 1's child: killed by signal 6" in r["robot_sharded"]["error"], r["robot_sharded"]
    assert r["robot_sharded"]["children"][1] == "killed by signal 6"
    assert took < 200, took         # the surviving child was ended by the flag, not by the 420 s guard


def test_four_ranks_one_gpu_one_robot_per_rank_plus_a_replica():
    """BASELINE config 4's layout on the 1/2/4/8 ladder: 3 Pandas on 4 ranks = a group of three ranks with ONE ROBOT EACH
    (peer exchange every rollout step) + a group of one rank carrying all three robots on a third of the scenarios
    (sharded.group_layout -> [3, 1]).  Four processes share the one GPU of the test box; every rank checks its rows against
    the fused kernel."""
    env = dict(os.environ, MRF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MRF_PEER_TIMEOUT_MS="8000")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--scenarios", "504",
           "--shard", "robots", "--transport", "peer"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 4 and r["transport"] == "peer"
    assert r["config"]["robot_groups"] == [3, 1] and r["config"]["scenarios_per_group"] == [504, 168]
    assert r["config"]["robots_per_rank_all"] == [1, 1, 1, 3] and r["config"]["robots_per_rank"] == [1, 1, 1]
    assert r["parity_vs_fused_kernel"]["ok"], r["parity_vs_fused_kernel"]            # MAX over all four ranks
    assert abs(r["value"] - (504 + 168) * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 1e-9
    assert r["exchange"] == "joints"                                                 # the default payload: 21 scalars per robot
    assert r["roofline"]["bound"] == "xgmi_link" and r["roofline"]["link"]["bytes_per_link_per_step"] == 1 * 21 * 504 * 8
    assert r["roofline"]["link"]["model"]["spheres_payload_bytes_per_scenario_link_step"] == 432
    _audit_fields(r["roofline"], link_bytes=1 * 21 * 504 * 8)
    assert [x["comm"]["peers_one_hop"] for x in r["ranks"]] == [0, 0, 0, 0]           # four ranks, one device


def test_eight_ranks_one_gpu_default_run_groups_3_3_2():
    """What `python bench.py --gpus 8` does on an 8-GPU node, with the eight ranks sharing the one GPU of the test box: the
    scenario-sharded headline over 8 ranks, then the robot-sharded block with 3 Pandas on 8 ranks = groups [3, 3, 2]
    (two groups with one robot per rank, one group of two ranks carrying 2 + 1 robots on half the scenarios)."""
    env = dict(os.environ, MRF_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MRF_PEER_TIMEOUT_MS="20000")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--scenarios", "504"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 8 and r["config"]["scenarios_per_gpu"] == 504
    assert abs(r["value"] - 8 * 504 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 1e-9
    peer = r["robot_sharded"]["peer"]
    assert "error" not in peer, peer
    assert "child process per rank" in r["robot_sharded"]["isolation"]
    assert peer["config"]["robot_groups"] == [3, 3, 2] and peer["config"]["scenarios_per_group"] == [504, 504, 252]
    assert peer["config"]["robots_per_rank_all"] == [1, 1, 1, 1, 1, 1, 2, 1]
    assert peer["parity_vs_fused_kernel"]["ok"], peer["parity_vs_fused_kernel"]
    sph = r["robot_sharded"]["peer_spheres"]                 # the sphere payload over the same groups
    assert "error" not in sph and sph["parity_vs_fused_kernel"]["ok"], sph
    assert (peer["exchange"], sph["exchange"]) == ("joints", "spheres")
    assert sph["allgather_bytes_per_rank_per_step"] * 21 == peer["allgather_bytes_per_rank_per_step"] * 54
    _topology_fields(r["robot_sharded"]["topology"])
