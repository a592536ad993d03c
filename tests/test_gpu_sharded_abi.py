"""Robot-sharded rollout through the C ABI (include/mrf.h: mrf_comm_* / mrf_rollout_sharded).

 * world 1: the RCCL transport with a real one-rank communicator (ncclCommInitRank / ncclAllGather issued from C++ on
   the caller's stream) and the PEER transport on its own buffers, both against the fused kernel (mrf_rollout);
 * world 2 on ONE GPU: two processes share the device, the PEER transport exchanges through IPC-mapped memory with
   device-side flags -- even (1+1 of 2 robots) and uneven (2+1 of 3) robot blocks, link-origin and offset sphere tables.
The multi-GPU RCCL path (world > 1) cannot run on the single-GPU test box; its partition / padding logic is the same
code as world 1 plus the slot map covered by the uneven case of tests/test_sharded_gloo.py.
Tolerance: identical arithmetic on identical inputs, different kernels -> f64 <= 1e-9 relative (as every parity test)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle, MrfError
from multi_robot_fabrics_amd.sharded import ShardedRollout

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


@pytest.mark.parametrize("exchange", ["joints", "spheres"])
@pytest.mark.parametrize("n_robots,horizon,n_scen,table", [(3, 6, 37, "lo"), (3, 4, 21, "offsets")])
def test_world1_torch_transport_matches_fused_rollout(n_robots, horizon, n_scen, table, exchange):
    """The Python fallback transport (the step kernels driven from Python; joint payload: one launch per step from the second
    step on, the group of one gathering in place) against the fused kernel."""
    cfg = config.panda_config(n_robots=n_robots, horizon=horizon)
    cfg.exchange = {"joints": abi.EXCHANGE_JOINTS, "spheres": abi.EXCHANGE_SPHERES}[exchange]
    cfg.goal_estimate_mask = 0b110 & ((1 << n_robots) - 1)
    if table == "offsets":
        links, offs = config.sphere_offsets_per_link(2)
        config.set_spheres(cfg, links, offs, [0.06] * len(links))
    batch = scenarios.panda_batch(cfg, n_scen, seed=5)
    sr = ShardedRollout(cfg, 0, 1, device_index=0, transport="torch")
    h = sr.backend.h
    q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
    want_avg, tq, tqd = FabricHandle(cfg, 0).rollout(q, qd, prm, want_traj=True)
    qq, qqd = q.clone(), qd.clone()
    avg = sr.rollout(qq, qqd, prm)
    assert rel(avg, want_avg) < 1e-9 and rel(qq, tq[-1]) < 1e-9 and rel(qqd, tqd[-1]) < 1e-9


@pytest.mark.parametrize("exchange", ["joints", "spheres"])
@pytest.mark.parametrize("transport", ["rccl", "peer"])
@pytest.mark.parametrize("n_robots,horizon,n_scen,table", [(3, 6, 37, "lo"), (2, 5, 130, "lo"), (3, 4, 21, "offsets")])
def test_world1_matches_fused_rollout(transport, n_robots, horizon, n_scen, table, exchange):
    cfg = config.panda_config(n_robots=n_robots, horizon=horizon)
    cfg.exchange = {"joints": abi.EXCHANGE_JOINTS, "spheres": abi.EXCHANGE_SPHERES}[exchange]
    cfg.goal_estimate_mask = 0b110 & ((1 << n_robots) - 1)
    if table == "offsets":
        links, offs = config.sphere_offsets_per_link(2)
        config.set_spheres(cfg, links, offs, [0.06] * len(links))
    batch = scenarios.panda_batch(cfg, n_scen, seed=5)
    sr = ShardedRollout(cfg, 0, 1, device_index=0, transport=transport, max_scenarios=n_scen)
    h = sr.backend.h
    assert h.comm_partition() == (0, n_robots)
    # the communicator as it reports itself (mrf_comm_info): for RCCL what ncclCommCount / UserRank / CuDevice return
    info = h.comm_info()
    assert (info["transport"], info["rank"], info["world"], info["robot_first"], info["robot_count"]) == (transport, 0, 1, 0, n_robots)
    assert info["exchange"] == exchange
    assert info["exchange_scalars_per_robot"] == (21 if exchange == "joints" else 9 * h.exchange_spheres) == h.exchange_scalars
    if transport == "rccl":
        assert (info["rccl_comm_count"], info["rccl_user_rank"]) == (1, 0) and info["rccl_device"] == info["hip_device"] == 0
    else:
        assert (info["rccl_comm_count"], info["rccl_user_rank"], info["peer_buffers_mapped"]) == (0, -1, 0)
        assert info["peers_one_hop"] == 0 and h.comm_peer_info() == [
            {"device": 0, "can_access_peer": 1, "link_type": 0, "hops": 0, "rank": 0, "link": "same device"}]
    q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
    want_avg, tq, tqd = FabricHandle(cfg, 0).rollout(q, qd, prm, want_traj=True)
    for _ in range(2):
        qq, qqd = q.clone(), qd.clone()
        avg = sr.rollout(qq, qqd, prm)
        h.comm_status()
        assert rel(avg, want_avg) < 1e-9 and rel(qq, tq[-1]) < 1e-9 and rel(qqd, tqd[-1]) < 1e-9


def test_comm_argument_errors():
    cfg = config.panda_config(n_robots=2, horizon=3)
    h = FabricHandle(cfg, 0)
    q = torch.zeros((7, 2), dtype=torch.float64, device="cuda")
    with pytest.raises(MrfError):
        h.comm_partition()                                   # no communicator yet
    with pytest.raises(MrfError):
        h.comm_init_rccl(0, 3)                               # more ranks than robots
    with pytest.raises(MrfError):
        h.comm_init_rccl(0, 2, None)                         # world > 1 needs an id
    h.comm_peer_open(0, 1, 4)
    with pytest.raises(MrfError):
        h.comm_peer_open(0, 1, 4)                            # already has one
    prm = torch.zeros((abi.NPARAM, 2 * 9), dtype=torch.float64, device="cuda")
    with pytest.raises(MrfError):
        h.rollout_sharded(torch.zeros((7, 18), dtype=torch.float64, device="cuda"),
                          torch.zeros((7, 18), dtype=torch.float64, device="cuda"), prm)   # 9 scenarios > capacity 4
    h.comm_destroy()
    h.comm_init_rccl(0, 1)                                   # group of one without RCCL
    assert h.comm_partition() == (0, 2)


def _free_port():
    """A port nobody listens on right now and no outgoing connection can take (below the ephemeral range): bench.free_port.
    Back-to-back launches on ONE fixed rendezvous port met the previous launch's socket still closing, a port from bind(0)
    met an outgoing connection (EADDRINUSE in 1 of ~25 launches each)."""
    sys.path.insert(0, ROOT)
    import bench
    return bench.free_port()


def _run_group_on_one_gpu(world, n_robots, horizon, n_scen, table, dtype, exchange, port, roll_call=False, extra_env=None):
    port = _free_port()         # the callers' fixed numbers are kept for reading the logs only
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MRF_PEER_TIMEOUT_MS="8000", MRF_PEER_DEVICE_SHARE=str(world))
    env.update(extra_env or {})
    if roll_call:       # the residency roll call of mrf_comm_peer_connect runs although the ranks share the device
        env["MRF_PEER_ROLL_CALL"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "sharded_worker.py"), str(n_robots),
           str(horizon), str(n_scen), table, dtype, exchange]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    if out.returncode != 0 and "EADDRINUSE" in out.stderr:      # the rendezvous port was taken after all: once more, another port
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
        out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)["ranks"]


@pytest.mark.parametrize("exchange", ["joints", "spheres"])
@pytest.mark.parametrize("n_robots,horizon,n_scen,table,dtype", [(2, 8, 50, "lo", "f64"), (3, 6, 45, "lo", "f64"),
                                                                  (3, 4, 30, "offsets", "f64"), (2, 6, 40, "lo", "f32")])
def test_two_processes_one_gpu_peer_exchange(n_robots, horizon, n_scen, table, dtype, exchange):
    ranks = _run_group_on_one_gpu(2, n_robots, horizon, n_scen, table, dtype, exchange, 29547, roll_call=(table == "lo"))
    assert sorted(r["count"] for r in ranks) == sorted([n_robots // 2, n_robots - n_robots // 2])
    for r in ranks:     # the roll call places between one and 4 x CUs workgroups of the peer kernel's footprint (two processes may
        assert (0 < r["coresident"] <= 4 * 256) if table == "lo" else r["coresident"] == 0, ranks   # run it at the same moment)
    tol = 1e-9 if dtype == "f64" else 2e-3
    for r in ranks:
        assert r["err"] < tol, ranks
        assert r["exchange"] == exchange and r["scalars"] == (21 if exchange == "joints" else 9 * (6 if table == "lo" else 16))
        # both ranks sit on device 0: the peer's mapped buffer is reported on the own device, zero hops
        assert [p["device"] for p in r["peers"]] == [0, 0] and r["peers_one_hop"] == 0


@pytest.mark.parametrize("world,n_robots,horizon,n_scen,table,exchange", [
    (3, 3, 6, 40, "lo", "joints"),            # BASELINE config 4's placement: one robot per rank, two remote robots per lane
    (3, 3, 5, 40, "lo", "spheres"),
    (4, 4, 4, 33, "offsets", "joints"),       # four ranks, generic table: three remote chains re-walked one at a time
    (4, 8, 3, 12, "offsets20", "joints"),     # BASELINE config 5's table on 4 ranks: 2 robots per rank on chip + 6 remote
    (4, 8, 3, 12, "offsets20", "spheres"),
])
def test_three_and_four_processes_one_gpu_peer_exchange(world, n_robots, horizon, n_scen, table, exchange):
    """More than one remote robot per lane (the remote chunk loop of mrf_shard.hpp), local AND remote robots in one wave,
    BASELINE config 5's 20-sphere table: 21 scalars on the wire against 180."""
    ranks = _run_group_on_one_gpu(world, n_robots, horizon, n_scen, table, "f64", exchange, 29551)
    assert sorted(r["count"] for r in ranks) == [n_robots // world] * world
    for r in ranks:
        assert r["err"] < 1e-9, ranks
        assert r["scalars"] == (21 if exchange == "joints" else 9 * {"lo": 6, "offsets": 16, "offsets20": 20}[table])


@pytest.mark.parametrize("world,n_robots,horizon,n_scen,table,dtype,max_grid", [
    (2, 2, 6, 300, "lo", "f64", 2),           # 5 blocks: two pairs and a last unit of ONE block, on two workgroups
    (3, 3, 5, 200, "lo", "f64", 1),           # one robot per rank, two remote chains; one workgroup walks both pairs
    (2, 3, 4, 100, "offsets", "f64", 2),      # 2 + 1 robots: local exchange on chip AND a remote chain, generic table
    (4, 8, 3, 70, "offsets20", "f64", 2),     # BASELINE config 5's table on 4 ranks
    (2, 2, 5, 260, "lo", "f32", 3),
])
def test_paired_blocks_processes_one_gpu(world, n_robots, horizon, n_scen, table, dtype, max_grid):
    """k_rollout_peer_paired (round 6): a workgroup works on two adjacent blocks in turns and publishes the joint state of step
    k + 1 at the end of step k, so that one block's exchange runs under the other block's step.  Opt-in (MRF_PEER_PAIRED=1),
    here at test sizes (MRF_PEER_PAIRED_MIN_BLOCKS=2, a grid of one to three workgroups); separate processes, IPC-mapped
    buffers."""
    ranks = _run_group_on_one_gpu(world, n_robots, horizon, n_scen, table, dtype, "joints", 29555,
                                  extra_env={"MRF_PEER_PAIRED": "1", "MRF_PEER_PAIRED_MIN_BLOCKS": "2", "MRF_PEER_MAX_GRID": str(max_grid)})
    for r in ranks:
        assert r["paired"] == 1, ranks
        assert r["err"] < (1e-9 if dtype == "f64" else 2e-3), ranks


@pytest.mark.parametrize("world,n_robots,horizon,n_scen,table,dtype,max_grid", [
    (2, 2, 6, 150, "lo", "f64", 2),           # three blocks on two workgroups: the tags keep counting across blocks and rollouts
    (3, 3, 5, 100, "lo", "f64", 0),           # one robot per rank: two tagged joint states polled per lane and step
    (2, 3, 4, 60, "offsets", "f64", 0),       # 2 + 1 robots, generic table: the staged (rolled-walk) receiver
    (4, 8, 3, 12, "offsets20", "f64", 0),     # BASELINE config 5's table on 4 ranks: six remote robots per lane
    (2, 2, 5, 70, "lo", "f32", 0),            # float32: one tagged word per scalar
])
def test_tagged_payload_processes_one_gpu(world, n_robots, horizon, n_scen, table, dtype, max_grid):
    """MRF_PEER_TAGGED=1 (opt-in): the joint state travels as 8-byte words that carry the tag of their step and are polled by the
    reader directly -- no store drain, no flags.  Separate processes, IPC-mapped buffers, three rollouts back to back."""
    env = {"MRF_PEER_TAGGED": "1"}
    if max_grid:
        env["MRF_PEER_MAX_GRID"] = str(max_grid)
    ranks = _run_group_on_one_gpu(world, n_robots, horizon, n_scen, table, dtype, "joints", 29557, extra_env=env)
    for r in ranks:
        assert r["tagged"] == 1 and r["paired"] == 0, ranks
        assert r["err"] < (1e-9 if dtype == "f64" else 2e-3), ranks


def test_plain_c_consumer_two_processes_one_gpu():
    """The same two-process peer exchange driven from plain C++ (examples/sharded_rollout_c.cpp: fork, pipes for the IPC
    handles, mrf_comm_peer_open / _connect, mrf_rollout_sharded) -- a non-Python consumer of the ABI runs the north-star
    partitioning; each process checks its rows against the fused kernel."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "examples")])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MRF_PEER_TIMEOUT_MS="4000", MRF_PEER_DEVICE_SHARE="2")
    out = subprocess.run([os.path.join(ROOT, "examples", "sharded_rollout_c"), "150", "10"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr[-2000:]
    assert out.stdout.count("rel err vs fused kernel") == 2, out.stdout


@pytest.mark.parametrize("tagged", [0, 1])
def test_peer_timeout_is_loud_and_recoverable(tagged):
    """A peer that skips a rollout: the waiting rank's exchange times out (bounded), its rollout leaves q / qdot alone and
    returns NaN, mrf_comm_status reports it ON BOTH RANKS; mrf_comm_reset on every rank between two barriers makes the
    group usable again although the ranks had issued different numbers of rollouts.  Then a peer that arrives late:
    neither rank may return a finite result (tests/timeout_worker.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MRF_PEER_TIMEOUT_MS="400", MRF_PEER_DEVICE_SHARE="2",
               MRF_PEER_TAGGED=str(tagged))      # 1: the bounded wait is the poll of the tagged payload words themselves
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "timeout_worker.py")]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    ranks = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["ranks"]
    r0 = next(r for r in ranks if r["rank"] == 0)
    assert r0["status_raised"] is True and r0["avg_all_nan"] and r0["state_untouched"], r0
    assert next(r for r in ranks if r["rank"] == 1)["peer_error_seen"] is True           # the error is the group's
    assert all(r["err_after_reset"] < 1e-9 for r in ranks), ranks
    # a peer that arrives after the other rank gave up cannot return a finite result either; all rows or none are committed
    for r in ranks:
        assert r["late_peer"] == {"status_raised": True, "avg_all_nan": True, "state_untouched": True}, r
    assert all(r["err_after_second_reset"] < 1e-9 for r in ranks), ranks


@pytest.mark.parametrize("seed", range(12))
def test_virtual_ranks_step_kernels_random_configurations(seed):
    """The per-step kernels of both payloads under randomised planners, driven for EVERY rank of a group from one process
    (the gather is a concatenation): random robot counts 2..6 and group sizes, contiguous robot blocks of every shape (all
    robots local / some local + some remote / one robot per rank), link-origin and offset tables with unequal radii,
    runtime leaf families, collision-link subsets, static fabrics, RF-CV masks -- against the fused kernel (mrf_rollout) to
    1e-9.  Covers remote_obstacles_joints (chunks of three staged chains for the link-origin table, one for generic tables),
    the on-chip exchange between the robots of a rank, and the legacy sphere kernels beside them."""
    from multi_robot_fabrics_amd import sharded
    rng = np.random.default_rng(500 + seed)
    N = int(rng.integers(2, 7))
    H = int(rng.integers(2, 5))
    cfg = config.panda_config(n_robots=N, horizon=H, dynamic=int(rng.random() < 0.8))
    if rng.random() < 0.5:              # runtime leaf families (the generic instantiations)
        config.set_strings(cfg, collision_geometry=f"-{rng.uniform(0.2, 0.8):.3f} / (x ** {int(rng.integers(2, 5))}) * xdot ** 2",
                           collision_finsler=f"{rng.uniform(0.005, 0.05):.4f} / (x ** {int(rng.integers(2, 5))}) * (1 - ca.heaviside(xdot)) * xdot ** 2")
    table = rng.choice(["lo", "lo_unequal", "offsets1", "offsets2", "offsets3"])
    if table == "lo_unequal":           # link-origin table whose coincident spheres cannot be merged
        config.set_spheres(cfg, list(range(1, 9)), None, radii=rng.uniform(0.05, 0.08, 8))
    elif table.startswith("offsets"):
        links, offs = config.sphere_offsets_per_link(int(table[-1]))
        config.set_spheres(cfg, links, offs, radii=rng.uniform(0.04, 0.07, len(links)))
    if rng.random() < 0.3:
        cfg.ego_link_mask = int(rng.integers(1, 0x40))
    cfg.goal_estimate_mask = int(rng.integers(0, 1 << N))
    cfg.n_goals = int(rng.integers(1, 4))
    B = int(rng.integers(5, 40))
    batch = scenarios.panda_batch(cfg, B, seed=seed, x_min=0.3 if N > 3 else 0.15, q_spread=0.15 if N > 3 else 0.3)
    ref = FabricHandle(cfg, 0)
    t = ref.tensor
    want_avg, tq, tqd = ref.rollout(t(batch["q"]), t(batch["qdot"]), t(batch["params"]), want_traj=True)
    G = int(rng.integers(1, N + 1))
    parts = sharded.robot_partition(N, G)
    # fused: mrf_step_action_predict_joints -- the action of step k and the position update + joint state of step k + 1 in one
    # launch, every rank into a send block of its own, gathered afterwards (what the RCCL transport's loop runs)
    for exchange, fused in ((abi.EXCHANGE_JOINTS, False), (abi.EXCHANGE_JOINTS, True), (abi.EXCHANGE_SPHERES, False)):
        h = ref
        S = h.exchange_spheres
        shape = (21,) if exchange == abi.EXCHANGE_JOINTS else (S, 9)
        state = []
        for first, count in parts:
            rows = np.array([s * N + first + l for s in range(B) for l in range(count)])
            q, qd, prm = (t(np.ascontiguousarray(batch[k][:, rows])) for k in ("q", "qdot", "params"))
            if (cfg.goal_estimate_mask >> first) & ((1 << count) - 1):
                prm = h.step_prepare(B, first, count, q, qd, prm)
            state.append(dict(rows=rows, q=q, qd=qd, prm=prm, ss=torch.zeros((B * count,), dtype=torch.float64, device="cuda")))
        everybody = torch.zeros((N,) + shape + (B,), dtype=torch.float64, device="cuda")
        send = [torch.zeros((count, 21, B), dtype=torch.float64, device="cuda") for _, count in parts] if fused else None
        for k in range(H):
            for g, ((first, count), st) in enumerate(zip(parts, state)):   # every rank predicts, the "gather" is the shared array
                if fused and k > 0:
                    everybody[first:first + count] = send[g]               # ... or the copy of what its last action launch sent
                elif exchange == abi.EXCHANGE_JOINTS:
                    h.step_predict_joints(B, first, count, st["q"], st["qd"], everybody[first:first + count])
                else:
                    h.step_predict(B, first, count, st["q"], st["qd"], everybody[first:first + count])
            for g, ((first, count), st) in enumerate(zip(parts, state)):
                if fused and k + 1 < H:
                    h.step_action_predict_joints(B, first, count, st["q"], st["qd"], st["prm"], everybody, st["ss"], send[g])
                elif exchange == abi.EXCHANGE_JOINTS:
                    h.step_action_joints(B, first, count, st["q"], st["qd"], st["prm"], everybody, st["ss"])
                else:
                    h.step_action(B, first, count, st["q"], st["qd"], st["prm"], everybody, st["ss"])
        torch.cuda.synchronize()
        for st in state:
            rows = torch.from_numpy(st["rows"]).cuda()
            assert rel(st["ss"] / (H * 7), want_avg[rows]) < 1e-9, (seed, N, G, table, exchange, fused)
            assert rel(st["q"], tq[-1][:, rows]) < 1e-9 and rel(st["qd"], tqd[-1][:, rows]) < 1e-9, (seed, N, G, table, exchange, fused)


@pytest.mark.parametrize("exchange", ["joints", "spheres"])
@pytest.mark.parametrize("n_robots,G,n_scen,horizon", [(3, 3, 21504, 6), (3, 2, 10752, 5), (4, 4, 8192, 4)])
def test_in_process_group_at_production_grid_sizes(n_robots, G, n_scen, horizon, exchange):
    """Every rank of a robot group in THIS process (sharded.InProcessGroup: one handle and one stream per rank, exchange buffers
    connected by device pointers, mrf_comm_peer_connect_local): the persistent peer kernels of all ranks run side by side
    with hundreds of workgroups each and SEVERAL blocks per workgroup -- the grid sizes a real run has, which the
    multi-process tests on one GPU cannot reach reliably (DESIGN.md section 6 "Residency") -- and reproduce the fused kernel."""
    from multi_robot_fabrics_amd.sharded import InProcessGroup
    cfg = config.panda_config(n_robots=n_robots, horizon=horizon)
    cfg.goal_estimate_mask = ((1 << n_robots) - 1) & ~1
    cfg.exchange = {"joints": abi.EXCHANGE_JOINTS, "spheres": abi.EXCHANGE_SPHERES}[exchange]
    batch = scenarios.tiled_batch(cfg, n_scen, seed=5, **({"x_min": 0.3, "q_spread": 0.15} if n_robots > 3 else {}))
    ref = FabricHandle(cfg, 0)
    want, tq, tqd = ref.rollout(*(ref.tensor(batch[k]) for k in ("q", "qdot", "params")), want_traj=True)
    grp = InProcessGroup(cfg, G, n_scen)
    rows = [grp.own_rows(g, n_scen) for g in range(G)]
    for rep in range(2):        # back to back: the sequence numbers keep counting
        states = [tuple(ref.tensor(np.ascontiguousarray(batch[k][:, r.numpy()])) for k in ("q", "qdot", "params")) for r in rows]
        avgs = grp.rollout(states)
        for g, (avg, r) in enumerate(zip(avgs, rows)):
            r = r.cuda()
            assert rel(avg, want[r]) < 1e-9 and rel(states[g][0], tq[-1][:, r]) < 1e-9 and rel(states[g][1], tqd[-1][:, r]) < 1e-9
    info = grp.handles[0].comm_info()
    assert info["world"] == G and info["peer_buffers_mapped"] == G - 1 and info["exchange"] == exchange
    grp.close()


def test_in_process_group_paired_blocks_reproduce_single_blocks(monkeypatch):
    """MRF_PEER_PAIRED=1 above one block per workgroup slot: the joint payload's rollout walks its blocks in pairs
    (k_rollout_peer_paired) at production grid sizes, reproduces the block-at-a-time kernel to rounding (two instantiations:
    the compiler contracts their multiply-adds differently) and the fused kernel to the parity tolerance; switching back and
    forth between the two on one communicator keeps the flag sequence intact."""
    from multi_robot_fabrics_amd.sharded import InProcessGroup
    n_robots, G, n_scen, horizon = 3, 3, 43008 + 64, 6           # 673 blocks per rank (odd: the last unit is one block)
    cfg = config.panda_config(n_robots=n_robots, horizon=horizon)
    cfg.goal_estimate_mask = 0b110
    batch = scenarios.tiled_batch(cfg, n_scen, seed=9)
    ref = FabricHandle(cfg, 0)
    want, tq, tqd = ref.rollout(*(ref.tensor(batch[k]) for k in ("q", "qdot", "params")), want_traj=True)
    grp = InProcessGroup(cfg, G, n_scen)
    rows = [grp.own_rows(g, n_scen) for g in range(G)]
    first = None
    for mode in ("paired", "single", "paired"):
        monkeypatch.setenv("MRF_PEER_PAIRED", "1" if mode == "paired" else "0")
        states = [tuple(ref.tensor(np.ascontiguousarray(batch[k][:, r.numpy()])) for k in ("q", "qdot", "params")) for r in rows]
        avgs = grp.rollout(states)
        assert [h.comm_info()["paired_blocks"] for h in grp.handles] == [int(mode == "paired")] * G
        for g, (avg, r) in enumerate(zip(avgs, rows)):
            r = r.cuda()
            assert rel(avg, want[r]) < 1e-9 and rel(states[g][0], tq[-1][:, r]) < 1e-9 and rel(states[g][1], tqd[-1][:, r]) < 1e-9
        res = [(a.clone(), st[0].clone(), st[1].clone()) for a, st in zip(avgs, states)]
        first = first or res
        for (a, q, qd), (a0, q0, qd0) in zip(res, first):
            if mode == "paired":
                assert torch.equal(a, a0) and torch.equal(q, q0) and torch.equal(qd, qd0)      # the same kernel twice: bit-equal
            else:
                assert rel(a, a0) < 1e-12 and rel(q, q0) < 1e-12 and rel(qd, qd0) < 1e-12
    grp.close()


@pytest.mark.parametrize("n_robots,G,n_scen,horizon", [(3, 3, 21504, 6), (4, 4, 8192, 4), (3, 2, 10752 + 13, 5)])
def test_in_process_group_tagged_payload(n_robots, G, n_scen, horizon, monkeypatch):
    """The tagged payload at production grid sizes (several blocks per workgroup, a ragged last block), all ranks in this process."""
    from multi_robot_fabrics_amd.sharded import InProcessGroup
    monkeypatch.setenv("MRF_PEER_TAGGED", "1")
    cfg = config.panda_config(n_robots=n_robots, horizon=horizon)
    cfg.goal_estimate_mask = ((1 << n_robots) - 1) & ~1
    batch = scenarios.tiled_batch(cfg, n_scen, seed=6, **({"x_min": 0.3, "q_spread": 0.15} if n_robots > 3 else {}))
    ref = FabricHandle(cfg, 0)
    want, tq, tqd = ref.rollout(*(ref.tensor(batch[k]) for k in ("q", "qdot", "params")), want_traj=True)
    grp = InProcessGroup(cfg, G, n_scen)
    rows = [grp.own_rows(g, n_scen) for g in range(G)]
    for rep in range(3):
        states = [tuple(ref.tensor(np.ascontiguousarray(batch[k][:, r.numpy()])) for k in ("q", "qdot", "params")) for r in rows]
        avgs = grp.rollout(states)
        for g, (avg, r) in enumerate(zip(avgs, rows)):
            r = r.cuda()
            assert rel(avg, want[r]) < 1e-9 and rel(states[g][0], tq[-1][:, r]) < 1e-9 and rel(states[g][1], tqd[-1][:, r]) < 1e-9
    assert all(h.comm_info()["tagged_payload"] == 1 for h in grp.handles)
    grp.close()
