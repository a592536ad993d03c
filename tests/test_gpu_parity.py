"""GPU parity: HIP kernels through the C ABI vs the float64 CPU oracle on the same seeded inputs.

Tolerances: f64 kernels <= 1e-9 relative on qddot / actions / trajectories;
f32 kernels <= 1e-4 relative + 1e-5 absolute on the well-conditioned set (all barrier coordinates >= 0.05).
"""
import math

import numpy as np
import pytest
import torch

from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

pytestmark = pytest.mark.gpu

F64_RTOL = 1e-9
F32_RTOL, F32_ATOL = 1e-4, 1e-5


def relerr(got, want):
    return float(np.abs(got - want).max() / max(1e-300, np.abs(want).max()))


def _with_scalar(cfg, scalar):
    c = cfg.copy()
    c.scalar = scalar
    return c


def _obstacles_from_other_robots(cfg, batch, oracle, extra=0, seed=0):
    sx, sv, sa = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, sa)
    if extra:
        rng = np.random.default_rng(seed)
        rows = ox.shape[2]
        ex = np.stack([rng.uniform(-0.5, 1.5, (extra, rows)), rng.uniform(-1.0, 1.0, (extra, rows)),
                       rng.uniform(2.2, 2.8, (extra, rows))], axis=1)       # far above the arms
        ox = np.concatenate([ox, ex]); ov = np.concatenate([ov, rng.uniform(-0.2, 0.2, ex.shape)])
        oa = np.concatenate([oa, rng.uniform(-0.2, 0.2, ex.shape)]); orad = np.concatenate([orad, np.full((extra, rows), 0.1)])
    return ox, ov, oa, orad


@pytest.mark.parametrize("scalar", [abi.F64, abi.F32])
@pytest.mark.parametrize("n_robots,n_scen", [(2, 129), (3, 43)])
def test_compute_action_panda(oracle, scalar, n_robots, n_scen):
    cfg = _with_scalar(config.panda_config(n_robots=n_robots, horizon=1), scalar)
    batch = scenarios.panda_batch(cfg, n_scen, seed=11)
    ox, ov, oa, orad = _obstacles_from_other_robots(cfg, batch, oracle, extra=2)
    want_qdd, want_act = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    h = FabricHandle(cfg, 0)
    act, qdd = h.compute_action(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]), h.tensor(ox),
                                h.tensor(ov), h.tensor(oa), h.tensor(orad), want_qddot=True)
    torch.cuda.synchronize()
    act, qdd = act.cpu().numpy().astype(np.float64), qdd.cpu().numpy().astype(np.float64)
    if scalar == abi.F64:
        assert relerr(qdd, want_qdd) < F64_RTOL
        assert relerr(act, want_act) < F64_RTOL
    else:
        assert np.allclose(qdd, want_qdd, rtol=F32_RTOL, atol=F32_ATOL * max(1.0, np.abs(want_qdd).max()))
        assert np.allclose(act, want_act, rtol=F32_RTOL, atol=F32_ATOL)


def test_compute_action_static_and_grasp(oracle):
    cfg = config.panda_config(n_robots=2, horizon=1)
    batch = scenarios.panda_batch(cfg, 64, seed=5)
    ox, ov, oa, orad = _obstacles_from_other_robots(cfg, batch, oracle)
    h = FabricHandle(cfg, 0)
    q, qd, prm = h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"])
    # static obstacles: v, a omitted (NULL) == explicit zeros == oracle with zeros
    a0 = h.compute_action(q, qd, prm, h.tensor(ox), None, None, h.tensor(orad))
    a1 = h.compute_action(q, qd, prm, h.tensor(ox), h.tensor(np.zeros_like(ov)), h.tensor(np.zeros_like(oa)), h.tensor(orad))
    _, want = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, None, None, orad)
    assert torch.equal(a0, a1)
    assert relerr(a0.cpu().numpy(), want) < F64_RTOL
    # grasp planner: no collision links -> no obstacle and no plane leaves (EXJ:160-166)
    cg = config.panda_config(n_robots=2, horizon=1, n_ego=0)
    hg = FabricHandle(cg, 0)
    ag = hg.compute_action(q, qd, prm, h.tensor(ox), h.tensor(ov), h.tensor(oa), h.tensor(orad))
    _, want_g = oracle.compute_action(cg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    _, want_g0 = oracle.compute_action(cg, batch["q"], batch["qdot"], batch["params"])
    assert relerr(ag.cpu().numpy(), want_g) < F64_RTOL
    assert np.array_equal(want_g, want_g0)
    # no obstacles at all
    an = h.compute_action(q, qd, prm)
    _, want_n = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"])
    assert relerr(an.cpu().numpy(), want_n) < F64_RTOL


def test_compute_action_edge_cases(oracle):
    """robot at rest (all xdot = 0 -> gates at 0.5), q[6] exactly at its goal (|x| = 0 attractor), empty batch."""
    cfg = config.panda_config(n_robots=2, horizon=1)
    batch = scenarios.panda_batch(cfg, 8, seed=2)
    batch["qdot"][:] = 0.0
    batch["q"][6, :] = math.pi / 4
    ox, ov, oa, orad = _obstacles_from_other_robots(cfg, batch, oracle)
    h = FabricHandle(cfg, 0)
    act = h.compute_action(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]), h.tensor(ox),
                           h.tensor(ov), h.tensor(oa), h.tensor(orad))
    _, want = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    assert np.isfinite(want).all()
    assert relerr(act.cpu().numpy(), want) < F64_RTOL
    empty = h.compute_action(h.tensor(np.zeros((7, 0))), h.tensor(np.zeros((7, 0))), h.tensor(np.zeros((abi.NPARAM, 0))))
    assert empty.shape == (7, 0)


@pytest.mark.parametrize("obst_dim", [3, 2])
def test_compute_action_planar(oracle, obst_dim):
    """4 point robots, 6 scene obstacles + 3 other robots (example_pointmasses_static.py:142-199)."""
    cfg = config.planar3_config(n_robots=4, obst_dim=obst_dim)
    rng = np.random.default_rng(4)
    rows = 4 * 33
    q = np.stack([rng.uniform(-2.5, 2.5, rows), rng.uniform(-2.5, 3.7, rows), rng.uniform(-1, 1, rows)])
    qd = rng.uniform(-0.5, 0.5, (3, rows))
    prm = np.zeros((abi.NPARAM, rows))
    prm[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + 2] = rng.uniform(-2.5, 2.5, (2, rows))
    prm[abi.P_WEIGHT_GOAL_0] = 1.0
    prm[abi.P_RADIUS_BODY] = 0.2
    M = 9
    ang = rng.uniform(0, 2 * math.pi, (M, rows))
    dist = rng.uniform(1.5, 4.0, (M, rows))
    ox = np.stack([q[0] + dist * np.cos(ang), q[1] + dist * np.sin(ang), np.zeros((M, rows))], axis=1)
    ov = rng.uniform(-0.3, 0.3, (M, 3, rows)); oa = rng.uniform(-0.3, 0.3, (M, 3, rows))
    orad = np.concatenate([np.ones((6, rows)), np.full((3, rows), 0.2)])
    want_qdd, want_act = oracle.compute_action(cfg, q, qd, prm, ox, ov, oa, orad)
    h = FabricHandle(cfg, 0)
    act = h.compute_action(h.tensor(q), h.tensor(qd), h.tensor(prm), h.tensor(ox), h.tensor(ov), h.tensor(oa), h.tensor(orad))
    assert relerr(act.cpu().numpy(), want_act) < F64_RTOL


@pytest.mark.parametrize("kernel", [1, 2, 3])     # 1 = row-per-lane (throughput), 2 = one wave per scenario (latency),
                                                   # 3 = wave pair per row (float64; float32 falls back to 1)
@pytest.mark.parametrize("scalar", [abi.F64, abi.F32])
@pytest.mark.parametrize("n_robots,horizon,n_scen,mask,dynamic", [
    (2, 20, 37, 0, 1),        # BASELINE config 3: 2-Panda RF H=20
    (3, 30, 25, 0b110, 1),    # BASELINE config 4: 3-Panda RF-CV H=30
    (3, 6, 22, 0, 0),         # static fabrics: exchanged v, a zeroed (FPJ:215-217); 22 scenarios -> ragged last wave
    (1, 5, 70, 0, 1),         # a single robot: no obstacles at all
])
def test_rollout_jointspace(oracle, scalar, n_robots, horizon, n_scen, mask, dynamic, kernel):
    cfg = _with_scalar(config.panda_config(n_robots=n_robots, horizon=horizon, dynamic=dynamic), scalar)
    cfg.goal_estimate_mask = mask
    cfg.kernel_select = kernel
    batch = scenarios.panda_batch(cfg, n_scen, seed=21, x_min=0.08)
    want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
    h = FabricHandle(cfg, 0)
    avg, tq, tqd = h.rollout(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]), want_traj=True)
    avg_only = h.rollout(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]))
    torch.cuda.synchronize()
    assert torch.equal(avg, avg_only)
    avg, tq, tqd = (t.cpu().numpy().astype(np.float64) for t in (avg, tq, tqd))
    if scalar == abi.F64:
        assert relerr(tq, want_q) < F64_RTOL
        assert relerr(tqd, want_qd) < F64_RTOL
        assert relerr(avg, want_avg) < F64_RTOL
    else:
        assert np.allclose(tqd, want_qd, rtol=1e-3, atol=1e-4)
        assert np.allclose(tq, want_q, rtol=1e-4, atol=1e-5)
        assert np.allclose(avg, want_avg, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("n_goals,n_planes,use_limits,dynamic,n_robots", [(0, 1, 1, 1, 3), (1, 1, 1, 1, 2), (2, 0, 1, 1, 3),
                                                                          (3, 1, 0, 0, 4), (3, 0, 0, 1, 2), (3, 1, 1, 1, 5)])
def test_rollout_wave_pair_planner_switches(oracle, n_goals, n_planes, use_limits, dynamic, n_robots):
    """k_rollout_panda_wp (kernel_select = 3) splits the solve over two waves: attractors, plane leaves, joint limits and the
    no-goal composition (h_g handed to the other wave) each live in one of them -- every planner switch, robot counts with
    idle tail lanes (3, 5) and a ragged last workgroup, against the oracle and bit-for-bit repeatable."""
    if not abi.has_wp():
        pytest.skip("library built without the wave-pair kernel (MRF_WITH_WP=1 python __graft_entry__.py)")
    cfg = config.panda_config(n_robots=n_robots, horizon=6, dynamic=dynamic)
    cfg.n_goals, cfg.n_planes, cfg.use_limits = n_goals, n_planes, use_limits
    cfg.goal_estimate_mask = ((1 << n_robots) - 1) & ~1 if n_goals else 0
    cfg.kernel_select = 3
    n_scen = 2 * (64 // n_robots) + 5
    batch = scenarios.panda_batch(cfg, n_scen, seed=31 + n_goals, x_min=0.1 if n_robots < 4 else 0.2,
                                  q_spread=0.3 if n_robots < 4 else 0.15)
    want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
    h = FabricHandle(cfg, 0)
    t = h.tensor
    avg, tq, tqd = h.rollout(t(batch["q"]), t(batch["qdot"]), t(batch["params"]), want_traj=True)
    assert relerr(tqd.cpu().numpy(), want_qd) < F64_RTOL and relerr(tq.cpu().numpy(), want_q) < F64_RTOL
    assert relerr(avg.cpu().numpy(), want_avg) < F64_RTOL
    assert torch.equal(h.rollout(t(batch["q"]), t(batch["qdot"]), t(batch["params"])), avg)
    # the row kernel on the same inputs: the same numbers to round-off (another summation order)
    c1 = cfg.copy()
    c1.kernel_select = 1
    avg1 = FabricHandle(c1, 0).rollout(t(batch["q"]), t(batch["qdot"]), t(batch["params"]))
    assert relerr(avg.cpu().numpy(), avg1.cpu().numpy()) < 1e-11


@pytest.mark.parametrize("kernel", [1, 2])
def test_rollout_eight_pandas(oracle, kernel):
    """BASELINE config 5 shape: 8 Pandas on the build-defined ring, 20 spheres per robot with link-local offsets."""
    cfg = config.panda_config(n_robots=8, horizon=4)
    cfg.kernel_select = kernel
    links, offs = config.sphere_offsets_per_link(3)
    keep = [i for i in range(len(links))][:20]
    config.set_spheres(cfg, [links[i] for i in keep], [offs[i] for i in keep])
    cfg.goal_estimate_mask = 0xFE
    batch = scenarios.panda_batch(cfg, 9, seed=8, x_min=0.3, q_spread=0.15)
    want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
    h = FabricHandle(cfg, 0)
    avg, tq, tqd = h.rollout(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]), want_traj=True)
    assert relerr(tqd.cpu().numpy(), want_qd) < F64_RTOL
    assert relerr(avg.cpu().numpy(), want_avg) < F64_RTOL


@pytest.mark.parametrize("scalar", [abi.F64, abi.F32])
def test_rollout_cartesian(oracle, scalar):
    cfg = _with_scalar(config.panda_config(n_robots=2, horizon=10), scalar)
    batch = scenarios.panda_batch(cfg, 150, seed=31, x_min=0.1)
    ox, ov, oa, orad = _obstacles_from_other_robots(cfg, batch, oracle)
    oa[:] = 0.0   # forward_planner_Cartesian.py:33 a_obsts_dyn = 0
    want_avg, want_q, want_qd = oracle.rollout_cartesian(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad, traj=True)
    h = FabricHandle(cfg, 0)
    avg, tq, tqd = h.rollout_cartesian(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]),
                                       h.tensor(ox), h.tensor(ov), h.tensor(oa), h.tensor(orad), want_traj=True)
    avg, tq, tqd = (t.cpu().numpy().astype(np.float64) for t in (avg, tq, tqd))
    if scalar == abi.F64:
        assert relerr(tqd, want_qd) < F64_RTOL
        assert relerr(tq, want_q) < F64_RTOL
        assert relerr(avg, want_avg) < F64_RTOL
    else:
        assert np.allclose(tqd, want_qd, rtol=1e-3, atol=1e-4)
        assert np.allclose(avg, want_avg, rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("accel", ["zeros", "none", "given"])
def test_rollout_cartesian_more_obstacles_than_resident(oracle, accel):
    """16 obstacle spheres per row: more than the LDS-resident prefix (7 with accelerations, 10 without), so both the
    resident and the streamed part of the obstacle loop fold.  obst_a = None (the reference's use, FPC:33) takes the
    instantiation without acceleration loads and must equal the oracle with zero accelerations; nonzero accelerations
    take the other one."""
    cfg = config.panda_config(n_robots=3, horizon=6)
    batch = scenarios.panda_batch(cfg, 70, seed=37, x_min=0.1)
    ox, ov, oa, orad = _obstacles_from_other_robots(cfg, batch, oracle)
    assert ox.shape[0] == 16
    if accel != "given":
        oa[:] = 0.0
    want_avg, want_q, want_qd = oracle.rollout_cartesian(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad, traj=True)
    h = FabricHandle(cfg, 0)
    t = h.tensor
    avg, tq, tqd = h.rollout_cartesian(t(batch["q"]), t(batch["qdot"]), t(batch["params"]), t(ox), t(ov),
                                       None if accel == "none" else t(oa), t(orad), want_traj=True)
    assert relerr(tqd.cpu().numpy(), want_qd) < F64_RTOL and relerr(tq.cpu().numpy(), want_q) < F64_RTOL
    assert relerr(avg.cpu().numpy(), want_avg) < F64_RTOL
    # compute_action takes the same two instantiations
    _, want_act = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    act = h.compute_action(t(batch["q"]), t(batch["qdot"]), t(batch["params"]), t(ox), t(ov),
                           None if accel == "none" else t(oa), t(orad))
    assert relerr(act.cpu().numpy(), want_act) < F64_RTOL


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("n_robots,per_link,dynamic", [(2, 1, 1), (3, 2, 1), (2, 4, 1), (3, 1, 0), (1, 1, 1), (3, 1, 1), (4, 1, 1)])
def test_rollout_cartesian_coupled(oracle, n_robots, per_link, dynamic, kernel):
    """mrf_rollout_cartesian_coupled (EXC:330-399 on the device): every robot's Cartesian rollout against the configured
    spheres of the other robots of its scenario -- the simulator's table with per_link spheres per link, positions and
    velocities at the start state, zero accelerations -- equals the oracle's rollout_cartesian fed with the host-side
    assembly of those obstacles (static fabrics: zero obstacle velocities; one robot: no obstacles).  kernel 1, one sphere per
    link (the link-origin table): the other robots' start states in the LDS tile for the whole horizon
    (k_rollout_cartc_panda); kernel 1, other tables: obstacle arrays assembled on the device + the row-per-lane rollout
    kernel; kernel 2: one wave per scenario, the start states staged in LDS (k_coop_panda in its Cartesian mode)."""
    cfg = config.panda_config(n_robots=n_robots, horizon=5, dynamic=dynamic)
    cfg.kernel_select = kernel
    links, offs = config.sphere_offsets_per_link(per_link)
    config.set_spheres(cfg, links, offs, [0.07 + 0.001 * (s % 5) for s in range(len(links))])
    batch = scenarios.panda_batch(cfg, 23, seed=43, x_min=0.1)
    sx, sv, _ = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv if dynamic else None, None)
    assert ox.shape[0] == 8 * per_link * (n_robots - 1)
    want_avg, want_q, want_qd = oracle.rollout_cartesian(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad, traj=True)
    h = FabricHandle(cfg, 0)
    t = h.tensor
    avg, tq, tqd = h.rollout_cartesian_coupled(t(batch["q"]), t(batch["qdot"]), t(batch["params"]), want_traj=True)
    assert relerr(tqd.cpu().numpy(), want_qd) < F64_RTOL and relerr(tq.cpu().numpy(), want_q) < F64_RTOL
    assert relerr(avg.cpu().numpy(), want_avg) < F64_RTOL
    # a larger batch afterwards grows the handle's obstacle work buffer; results of the first rows do not change
    big = scenarios.panda_batch(cfg, 200, seed=43, x_min=0.1)
    avg2 = h.rollout_cartesian_coupled(t(big["q"]), t(big["qdot"]), t(big["params"]))
    assert avg2.shape == (200 * n_robots,) and torch.isfinite(avg2).all()
    assert torch.equal(h.rollout_cartesian_coupled(t(batch["q"]), t(batch["qdot"]), t(batch["params"])), avg)


@pytest.mark.parametrize("per_link", [0, 1])
def test_rollout_cartesian_coupled_float32_tile(oracle, per_link):
    """The LDS-tile form of the coupled Cartesian rollout in float32 (link-origin table and one offset sphere per link)
    against the float64 oracle, and bit-identical between two calls."""
    if not abi.has_f32():       # asked inside the test: an unloadable library must fail tests, not the collection
        pytest.skip("library built without float32 kernels (MRF_WITH_F32=1 python __graft_entry__.py)")
    cfg = config.panda_config(n_robots=3, horizon=8, scalar=abi.F32)
    cfg.kernel_select = 1
    if per_link:
        links, offs = config.sphere_offsets_per_link(per_link)
        config.set_spheres(cfg, links, offs)
    batch = scenarios.panda_batch(cfg, 50, seed=47, x_min=0.25)
    c64 = cfg.copy()
    c64.scalar = abi.F64
    sx, sv, _ = oracle.fk_spheres(c64, batch["q"], batch["qdot"])
    o = scenarios.other_robot_obstacles(c64, batch, sx, sv, None)
    want_avg, _, want_qd = oracle.rollout_cartesian(c64, batch["q"], batch["qdot"], batch["params"], *o, traj=True)
    h = FabricHandle(cfg, 0)
    t = h.tensor
    avg, _, tqd = h.rollout_cartesian_coupled(t(batch["q"]), t(batch["qdot"]), t(batch["params"]), want_traj=True)
    assert avg.dtype == torch.float32
    assert relerr(avg.double().cpu().numpy(), want_avg) < 2e-3 and relerr(tqd.double().cpu().numpy(), want_qd) < 2e-3
    assert torch.equal(h.rollout_cartesian_coupled(t(batch["q"]), t(batch["qdot"]), t(batch["params"])), avg)


def test_fk_spheres_with_offsets(oracle):
    cfg = config.panda_config(n_robots=3, horizon=1)
    links, offs = config.sphere_offsets_per_link(4)
    config.set_spheres(cfg, links, offs)
    batch = scenarios.panda_batch(cfg, 40, seed=41)
    wx, wv, wa = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
    h = FabricHandle(cfg, 0)
    x, v, a = h.fk_spheres(h.tensor(batch["q"]), h.tensor(batch["qdot"]))
    assert relerr(x.cpu().numpy(), wx) < 1e-12
    assert relerr(v.cpu().numpy(), wv) < 1e-11
    assert relerr(a.cpu().numpy(), wa) < 1e-11
    x_only, v_none, _ = h.fk_spheres(h.tensor(batch["q"]))
    assert v_none is None and torch.equal(x_only, x)


@pytest.mark.parametrize("robot_first,robot_count,r1", [(0, 3, 0.08), (1, 1, 0.08), (1, 2, 0.08), (1, 2, 0.11)])
def test_sharded_step_matches_fused_rollout(oracle, robot_first, robot_count, r1):
    """predict -> (gather) -> action for a subset of robots reproduces the fused rollout's trajectory."""
    N, H, B = 3, 7, 19
    cfg = config.panda_config(n_robots=N, horizon=H)
    cfg.sphere_radius[1] = r1        # != sphere_radius[0]: the link-1/2 pair is no longer merged
    batch = scenarios.panda_batch(cfg, B, seed=51, x_min=0.08)
    want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
    h = FabricHandle(cfg, 0)
    S = cfg.n_spheres
    SX = h.exchange_spheres          # the coincident origins of links 1/2 and 5/6 travel once
    keep = [0, 2, 3, 4, 6, 7] if r1 == 0.08 else [0, 1, 2, 3, 4, 6, 7]
    assert SX == len(keep)
    sel = np.array([s * N + robot_first + l for s in range(B) for l in range(robot_count)])
    q = h.tensor(batch["q"][:, sel]); qd = h.tensor(batch["qdot"][:, sel]); prm = h.tensor(batch["params"][:, sel])
    sumsq = torch.zeros(B * robot_count, dtype=torch.float64, device="cuda")
    for k in range(H):
        sph_own = torch.empty((robot_count, SX, 9, B), dtype=torch.float64, device="cuda")
        h.step_predict(B, robot_first, robot_count, q, qd, sph_own)
        # spheres of the robots this "rank" does not own come from the oracle trajectory (stand-in for the all-gather)
        qk, qdk = want_q[k], (want_qd[k - 1] if k else batch["qdot"])
        sx, sv, sa = oracle.fk_spheres(cfg, qk, qdk)
        sph_all = np.concatenate([sx, sv, sa], axis=1).reshape(S, 9, B, N).transpose(3, 0, 1, 2)[:, keep].copy()
        sph_all_t = h.tensor(sph_all)
        assert relerr(sph_own.cpu().numpy(), sph_all[robot_first:robot_first + robot_count]) < 1e-9
        sph_all_t[robot_first:robot_first + robot_count] = sph_own
        h.step_action(B, robot_first, robot_count, q, qd, prm, sph_all_t, sumsq)
        assert relerr(q.cpu().numpy(), want_q[k][:, sel]) < F64_RTOL
        assert relerr(qd.cpu().numpy(), want_qd[k][:, sel]) < F64_RTOL
    assert relerr((sumsq / (H * 7)).cpu().numpy(), want_avg[sel]) < F64_RTOL


# ---------------------------------------------------------------------------------- properties at full size
def test_properties_at_bench_size():
    """Size-independent properties on the full bench batch (no oracle: too slow at this size):
    obstacle-permutation invariance, static == dynamic with zero v/a, H=1 rollout == Euler step + compute_action."""
    N = 3
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    B = 6 * cus * 4 * (64 // N)                            # bench.py's batch: 129 024 scenarios on a 256-CU MI355X
    cfg = config.panda_config(n_robots=N, horizon=1)
    batch = scenarios.panda_batch(cfg, B, seed=61)
    h = FabricHandle(cfg, 0)
    q, qd, prm = h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"])
    # H=1 rollout: q1 = q + dt*qd, obstacles = other robots' link origins at (q1, qd)
    avg, tq, tqd = h.rollout(q, qd, prm, want_traj=True)
    q1 = q + cfg.dt * qd
    assert torch.allclose(tq[0], q1, rtol=0, atol=1e-15)
    sx, sv, sa = h.fk_spheres(q1, qd)
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, sa)
    act = h.compute_action(q1, qd, prm, ox, ov, oa, orad)
    assert float((act - tqd[0]).abs().max() / tqd[0].abs().max()) < 1e-10
    assert torch.allclose(avg, (act ** 2).sum(0) / 7, rtol=1e-10, atol=0)
    # permutation of the obstacle list leaves the action unchanged up to summation order
    perm = torch.randperm(ox.shape[0], device="cuda")
    act_p = h.compute_action(q1, qd, prm, ox[perm].contiguous(), ov[perm].contiguous(), oa[perm].contiguous(),
                             orad[perm].contiguous())
    assert float((act - act_p).abs().max() / act.abs().max()) < 1e-10
    # the static planner (dynamic=0 rollout) equals compute_action with v = a = 0
    cs = config.panda_config(n_robots=N, horizon=1, dynamic=0)
    hs = FabricHandle(cs, 0)
    _, _, tqd_s = hs.rollout(q, qd, prm, want_traj=True)
    act_s = h.compute_action(q1, qd, prm, ox, None, None, orad)
    assert float((act_s - tqd_s[0]).abs().max() / tqd_s[0].abs().max()) < 1e-10
    assert np.isfinite(act.cpu().numpy()).all()
    # the cooperative (latency) and row-per-lane (throughput) kernels agree on the whole batch
    c1, c2 = config.panda_config(n_robots=N, horizon=5), config.panda_config(n_robots=N, horizon=5)
    c1.kernel_select, c2.kernel_select = 1, 2
    r1, r2 = FabricHandle(c1, 0).rollout(q, qd, prm), FabricHandle(c2, 0).rollout(q, qd, prm)
    assert float((r1 - r2).abs().max() / r1.abs().max()) < 1e-10


def test_create_rejects_bad_config():
    from multi_robot_fabrics_amd.runtime import MrfError
    cfg = config.panda_config(n_robots=2, horizon=1)
    cfg.n_ego = 4
    with pytest.raises(MrfError):
        FabricHandle(cfg, 0)
    cfg = config.panda_config(n_robots=2, horizon=1)
    cfg.mode = abi.MODE_ACC
    h = FabricHandle(cfg, 0)
    z = h.tensor(np.zeros((7, 2)))
    with pytest.raises(MrfError):
        h.rollout(z, z, h.tensor(np.zeros((abi.NPARAM, 2))))


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("n_per_link,use_accel,dynamic", [(1, False, 1), (1, True, 1), (2, False, 1), (1, False, 0)])
def test_compute_action_coupled(oracle, n_per_link, use_accel, dynamic, kernel):
    """Device-side obstacle assembly (EXJ:394-412) == fk_spheres + host gather + compute_action, and == the oracle."""
    N, B = 3, 45
    cfg = config.panda_config(n_robots=N, horizon=1, dynamic=dynamic)
    cfg.kernel_select = kernel
    if n_per_link > 1:
        links, offs = config.sphere_offsets_per_link(n_per_link)
        config.set_spheres(cfg, links, offs)
    batch = scenarios.panda_batch(cfg, B, seed=71, x_min=0.15 if n_per_link > 1 else 0.05)
    sx, sv, sa = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv if dynamic else None,
                                                       sa if (dynamic and use_accel) else None)
    _, want = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    h = FabricHandle(cfg, 0)
    q, qd, prm = h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"])
    act, qdd = h.compute_action_coupled(q, qd, prm, use_accel=use_accel, want_qddot=True)
    assert relerr(act.cpu().numpy(), want) < F64_RTOL
    two_step = h.compute_action(q, qd, prm, h.tensor(ox), h.tensor(ov), h.tensor(oa), h.tensor(orad))
    assert float((two_step - act).abs().max() / act.abs().max()) < 1e-11


def test_baseline_config_2_mrdf_two_pandas_ten_spheres(oracle):
    """BASELINE configs[1]: 2-Panda MRDF joint-space (no rollout), ~10 collision spheres per robot."""
    cfg = scenarios.baseline_config("C2")["cfg"]             # the configuration bench.py's `configs` block times
    batch = scenarios.panda_batch(cfg, 300, seed=81, x_min=0.15)
    sx, sv, sa = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, None)
    assert ox.shape[0] == 10
    want_qdd, want = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    h = FabricHandle(cfg, 0)
    act, qdd = h.compute_action_coupled(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]),
                                        want_qddot=True)
    assert relerr(act.cpu().numpy(), want) < F64_RTOL and relerr(qdd.cpu().numpy(), want_qdd) < F64_RTOL


@pytest.mark.parametrize("kernel", [1, 2])
def test_baseline_config_5_eight_pandas_h50(oracle, kernel):
    """BASELINE configs[4] at its full shape: 8 Pandas, RF-CV, H=50, 20 spheres per robot (140 obstacle spheres each)."""
    cfg = scenarios.baseline_config("C5")["cfg"]            # the configuration bench.py's `configs` block times
    cfg.kernel_select = kernel
    links = list(cfg.sphere_link[:cfg.n_spheres])           # 20 spheres, two of them on link 8 / the hand
    assert len(links) == 20 and links.count(8) == 2 and all(links.count(l) >= 2 for l in range(1, 9))
    assert cfg.goal_estimate_mask == 0xFE
    batch = scenarios.panda_batch(cfg, 10, seed=91, x_min=0.3, q_spread=0.15)
    want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
    h = FabricHandle(cfg, 0)
    avg, tq, tqd = h.rollout(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]), want_traj=True)
    assert relerr(tq.cpu().numpy(), want_q) < F64_RTOL
    # 50 coupled steps of 8 robots: the measured round-off growth (tools/c5_error_growth.py, profiles/r04_c5_error_growth.json)
    # is 2e-15 at step 1 -> 1.6e-14 at step 50, five orders below the stated tolerance; no loosening is needed
    assert relerr(tqd.cpu().numpy(), want_qd) < F64_RTOL
    assert relerr(avg.cpu().numpy(), want_avg) < F64_RTOL


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("radii", [
    (0.08,) * 8,                                           # reference table: links 1/2 and 5/6 merged (weight 2)
    (0.08, 0.11, 0.08, 0.08, 0.08, 0.08, 0.08, 0.08),      # links 1/2 differ -> only 5/6 merged
    (0.08, 0.08, 0.08, 0.08, 0.06, 0.10, 0.08, 0.08),      # links 5/6 differ -> only 1/2 merged
    (0.05, 0.06, 0.07, 0.08, 0.09, 0.10, 0.07, 0.06),      # nothing merged
])
def test_link_origin_table_coincident_spheres(oracle, radii, kernel):
    """The origins of links 1/2 and 5/6 coincide (zero joint offsets): with equal radii the kernels evaluate the leaf
    once with weight 2.  Every merge pattern must equal the oracle, which evaluates all 8 spheres."""
    N, B = 3, 23
    cfg = config.panda_config(n_robots=N, horizon=7)
    cfg.kernel_select = kernel
    cfg.goal_estimate_mask = 0b010
    for s, r in enumerate(radii):
        cfg.sphere_radius[s] = r
    batch = scenarios.panda_batch(cfg, B, seed=33, x_min=0.1)
    want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
    h = FabricHandle(cfg, 0)
    q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
    avg, tq, tqd = h.rollout(q, qd, prm, want_traj=True)
    assert relerr(tqd.cpu().numpy(), want_qd) < F64_RTOL
    assert relerr(avg.cpu().numpy(), want_avg) < F64_RTOL
    sx, sv, sa = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, sa)
    _, want_act = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    act = h.compute_action_coupled(q, qd, prm, use_accel=True)
    assert relerr(act.cpu().numpy(), want_act) < F64_RTOL


@pytest.mark.parametrize("kernel", [1, 2])
@pytest.mark.parametrize("n_robots,n_scen", [(5, 27), (16, 5)])
def test_rollout_odd_robot_counts(oracle, n_robots, n_scen, kernel):
    """Robot counts that do not divide the wave (5 -> 12 scenarios + 4 shadow lanes) and the ABI maximum (16)."""
    cfg = config.panda_config(n_robots=n_robots, horizon=3)
    if kernel == 2 and 5 * n_robots > 64:
        pytest.skip("the cooperative kernel needs 5 lanes per robot")
    cfg.kernel_select = kernel
    cfg.goal_estimate_mask = 0b1010
    batch = scenarios.panda_batch(cfg, n_scen, seed=14, x_min=0.3, q_spread=0.15)
    want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
    h = FabricHandle(cfg, 0)
    q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
    avg, tq, tqd = h.rollout(q, qd, prm, want_traj=True)
    assert relerr(tqd.cpu().numpy(), want_qd) < F64_RTOL
    assert relerr(avg.cpu().numpy(), want_avg) < F64_RTOL
    act = h.compute_action_coupled(q, qd, prm, use_accel=True)
    sx, sv, sa = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, sa)
    _, want_act = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    assert relerr(act.cpu().numpy(), want_act) < F64_RTOL


@pytest.mark.parametrize("seed", range(10))
def test_random_planner_configurations(oracle, seed):
    """Randomised planner definitions (leaf families / exponents / gates through the string front-end, every named
    constant of mrf_config, mode, number of goals, plane/limit switches): the generic kernel instantiations against the
    oracle -- compute_action with explicit obstacle arrays, the coupled action and the rollout."""
    rng = np.random.default_rng(100 + seed)
    N = int(rng.integers(2, 4))
    cfg = config.panda_config(n_robots=N, horizon=4, dynamic=int(rng.integers(0, 2)))
    gate = lambda: rng.choice(["", " * (1 - ca.heaviside(xdot))", " * (-0.5 * (ca.sign(xdot) - 1))"])
    def pow_leaf(sign, kmax, pmax):
        return f"{sign}{rng.uniform(0.05, kmax):.4f} / (x ** {int(rng.integers(1, pmax + 1))}){gate()} * xdot ** 2"
    def logistic(sign):
        return f"{sign}{rng.uniform(1, 12):.3f}*(1/(1+{rng.uniform(0.5, 2):.3f}*ca.exp(-{rng.uniform(2, 12):.3f}*x))-1) * (xdot**2)"
    config.set_strings(cfg,
                       collision_geometry=pow_leaf("-", 1.0, 5) if rng.random() < 0.7 else logistic(""),
                       collision_finsler=pow_leaf("", 0.1, 5),
                       geometry_plane_constraint=logistic("") if rng.random() < 0.5 else pow_leaf("-", 1.0, 3),
                       finsler_plane_constraint=pow_leaf("", 0.5, 3),
                       limit_geometry=pow_leaf("-", 0.5, 3), limit_finsler=pow_leaf("", 0.5, 3))
    cfg.base_mass = rng.uniform(0.05, 1.0)
    cfg.attr_k, cfg.attr_alpha = rng.uniform(1, 10), rng.uniform(2, 20)
    cfg.attr_mu, cfg.attr_ml, cfg.attr_a = rng.uniform(1, 3), rng.uniform(0.1, 0.9), rng.uniform(0.2, 2)
    cfg.beta_a, cfg.beta_r, cfg.beta_b, cfg.beta_s = rng.uniform(0.1, 2), rng.uniform(0.01, 0.2), rng.uniform(1, 10), rng.uniform(0, 0.1)
    cfg.eta_a, cfg.eta_s = rng.uniform(0.1, 1), rng.uniform(0.1, 1)
    cfg.eps = 10.0 ** rng.uniform(-8, -4)
    cfg.goal_estimate_T = rng.uniform(0.05, 0.5)
    cfg.goal_estimate_mask = int(rng.integers(0, 1 << N))
    cfg.n_goals = int(rng.integers(0, 4))
    cfg.n_planes = int(rng.integers(0, 2))
    cfg.use_limits = int(rng.integers(0, 2))
    cfg.plane_abs = int(rng.integers(0, 2))
    cfg.zero_small_action = int(rng.integers(0, 2))
    cfg.dt = rng.uniform(0.002, 0.02)
    if rng.random() < 0.5:
        links, offs = config.sphere_offsets_per_link(int(rng.integers(1, 3)))
        config.set_spheres(cfg, links, offs, radii=rng.uniform(0.05, 0.1, len(links)))
    batch = scenarios.panda_batch(cfg, 19, seed=seed, x_min=0.25)
    h = FabricHandle(cfg, 0)
    q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
    # rollout (mode 'vel' only) and coupled action
    want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
    avg, tq, tqd = h.rollout(q, qd, prm, want_traj=True)
    assert relerr(tqd.cpu().numpy(), want_qd) < F64_RTOL, "rollout"
    sx, sv, sa = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv if cfg.dynamic else None, sa if cfg.dynamic else None)
    for mode in (abi.MODE_VEL, abi.MODE_ACC):
        c2 = cfg.copy()
        c2.mode = mode
        h2 = FabricHandle(c2, 0)
        want_qdd, want_act = oracle.compute_action(c2, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
        act, qdd = h2.compute_action(q, qd, prm, h2.tensor(ox), h2.tensor(ov), h2.tensor(oa), h2.tensor(orad), want_qddot=True)
        assert relerr(qdd.cpu().numpy(), want_qdd) < F64_RTOL, "compute_action qddot"
        assert relerr(act.cpu().numpy(), want_act) < F64_RTOL, "compute_action action"
        act_c = h2.compute_action_coupled(q, qd, prm, use_accel=True)
        assert relerr(act_c.cpu().numpy(), want_act) < F64_RTOL, "coupled action"


@pytest.mark.parametrize("kernel", [1, 2])
def test_rollout_large_time_step_takes_the_full_sincos_path(oracle, kernel):
    """dt = 0.1 s makes |dq| exceed the 0.125 rad window of the incremental cos/sin update in some lanes, so the
    wave falls back to full sincos evaluations (both paths must agree with the oracle)."""
    cfg = config.panda_config(n_robots=3, horizon=4)
    cfg.dt = 0.1
    cfg.kernel_select = kernel
    batch = scenarios.panda_batch(cfg, 21, seed=77, x_min=0.3, qd_spread=2.0)
    assert np.abs(batch["qdot"]).max() * cfg.dt > 0.125
    want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
    h = FabricHandle(cfg, 0)
    avg, tq, tqd = h.rollout(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]), want_traj=True)
    assert relerr(tq.cpu().numpy(), want_q) < F64_RTOL
    assert relerr(tqd.cpu().numpy(), want_qd) < F64_RTOL


def test_joint_sincos_multi_revolution_angles(oracle):
    """Joint angles far outside the joint limits (several revolutions, both signs, and beyond the fast path's 1e4 rad
    window): the kernels' own range reduction must agree with the oracle's libm sin/cos."""
    cfg = config.panda_config(n_robots=2, horizon=1)
    rng = np.random.default_rng(2)
    rows = 400
    q = rng.uniform(-40.0, 40.0, (7, rows))
    q[:, :20] = rng.uniform(-3e4, 3e4, (7, 20))               # library path
    q[:, 20:27] = np.array([k * np.pi / 4 for k in range(-3, 4)])[None, :] * np.ones((7, 1))   # quadrant boundaries
    qd = rng.uniform(-1, 1, (7, rows))
    h = FabricHandle(cfg, 0)
    x, v, a = h.fk_spheres(h.tensor(q), h.tensor(qd))
    wx, wv, wa = oracle.fk_spheres(cfg, q, qd)
    assert relerr(x.cpu().numpy(), wx) < 1e-11
    assert relerr(v.cpu().numpy(), wv) < 1e-11
    assert relerr(a.cpu().numpy(), wa) < 1e-11


@pytest.mark.parametrize("mask", [0b000100, 0b010000, 0b101001, 0b001100, 0b111110])
def test_collision_link_subsets(oracle, mask):
    """mrf_config.ego_link_mask: collision + plane leaves on a subset of links 3..8 (set_components' collision_links,
    EXJ:91-96; the Cartesian rollout class defaults to link 7 alone, FPC:20-21; set_planner_panda's own default is link
    5).  0b001100 = links 5 and 6, the two that share one point; 0b000100 / 0b001000-type masks split that pair.
    Every kernel family against the oracle: explicit-obstacle action, coupled action, both coupled rollouts
    (row-per-lane and cooperative), Cartesian rollout, robot-sharded step kernels."""
    N, H, B = 3, 5, 23
    cfg = config.panda_config(n_robots=N, horizon=H)
    cfg.ego_link_mask = mask
    cfg.goal_estimate_mask = 0b110
    batch = scenarios.panda_batch(cfg, B, seed=mask, x_min=0.1)
    rng = np.random.default_rng(mask)
    batch["params"][abi.P_RADIUS_BODY:abi.P_RADIUS_BODY + 6] = rng.uniform(0.05, 0.09, (6, B * N))   # distinct radii per link
    want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
    sx, sv, sa = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv, sa)
    want_qdd, want_act = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    full = cfg.copy()
    full.ego_link_mask = 0x3F
    _, act_full = oracle.compute_action(full, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    assert relerr(want_act, act_full) > 1e-6          # the mask matters on this batch
    wc_avg, _, wc_qd = oracle.rollout_cartesian(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad, traj=True)
    for kernel in (1, 2):
        c = cfg.copy()
        c.kernel_select = kernel
        h = FabricHandle(c, 0)
        q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
        avg, tq, tqd = h.rollout(q, qd, prm, want_traj=True)
        assert relerr(tqd.cpu().numpy(), want_qd) < F64_RTOL and relerr(avg.cpu().numpy(), want_avg) < F64_RTOL, kernel
        act_c = h.compute_action_coupled(q, qd, prm, use_accel=True)
        assert relerr(act_c.cpu().numpy(), want_act) < F64_RTOL, kernel
    act, qdd = h.compute_action(q, qd, prm, h.tensor(ox), h.tensor(ov), h.tensor(oa), h.tensor(orad), want_qddot=True)
    assert relerr(act.cpu().numpy(), want_act) < F64_RTOL and relerr(qdd.cpu().numpy(), want_qdd) < F64_RTOL
    cavg, _, ctqd = h.rollout_cartesian(q, qd, prm, h.tensor(ox), h.tensor(ov), h.tensor(oa), h.tensor(orad), want_traj=True)
    assert relerr(ctqd.cpu().numpy(), wc_qd) < F64_RTOL and relerr(cavg.cpu().numpy(), wc_avg) < F64_RTOL
    from multi_robot_fabrics_amd.sharded import ShardedRollout
    for transport in ("rccl", "peer"):
        sr = ShardedRollout(cfg, 0, 1, device_index=0, transport=transport, max_scenarios=B)
        qq, qqd = q.clone(), qd.clone()
        savg = sr.rollout(qq, qqd, prm)
        sr.backend.h.comm_status()
        assert relerr(savg.cpu().numpy(), want_avg) < F64_RTOL and relerr(qqd.cpu().numpy(), want_qd[-1]) < F64_RTOL, transport


def test_validate_rejects_an_empty_link_mask():
    from multi_robot_fabrics_amd.runtime import MrfError
    cfg = config.panda_config(n_robots=2, horizon=2)
    cfg.ego_link_mask = 0
    with pytest.raises(MrfError):
        FabricHandle(cfg, 0)
    cfg.n_ego = 0               # "grasp" planner: no collision links at all, the mask is not looked at
    FabricHandle(cfg, 0)


@pytest.mark.parametrize("seed", range(12))
def test_coupled_action_packed_tile_random_tables(oracle, seed):
    """k_action_coupled<TAB_PACKED> (round 6): tables of up to 12 spheres without obstacle accelerations are emitted by the
    solve's own unrolled chain walk and folded from 6-row tile entries.  Random tables (1..12 spheres drawn from the
    simulator's 1-3-per-link tables, incl. the hand's x/y-shifted spheres and several spheres on one link, unequal radii),
    2-5 robots, runtime leaf families and collision-link subsets (the two-walk form), static fabrics, 'acc' and 'vel' --
    against the oracle fed with the host-side assembly (EXJ:394-412, a = 0 as EXJ:411)."""
    rng = np.random.default_rng(900 + seed)
    N = int(rng.integers(2, 6))
    cfg = config.panda_config(n_robots=N, horizon=1, dynamic=int(rng.random() < 0.8))
    cfg.kernel_select = 1
    cfg.mode = int(rng.integers(0, 2))
    links, offs = config.sphere_offsets_per_link(int(rng.integers(1, 4)))
    S = int(rng.integers(1, 13))
    keep = sorted(rng.choice(len(links), size=min(S, len(links)), replace=False).tolist())
    config.set_spheres(cfg, [links[i] for i in keep], [offs[i] for i in keep], radii=rng.uniform(0.04, 0.08, len(keep)))
    if rng.random() < 0.5:
        config.set_strings(cfg, collision_geometry=f"-{rng.uniform(0.2, 0.8):.3f} / (x ** {int(rng.integers(2, 5))}) * xdot ** 2",
                           collision_finsler=f"{rng.uniform(0.005, 0.05):.4f} / (x ** {int(rng.integers(2, 5))}) * (1 - ca.heaviside(xdot)) * xdot ** 2")
    if rng.random() < 0.4:
        cfg.ego_link_mask = int(rng.integers(1, 0x40))
    cfg.n_goals = int(rng.integers(0, 4))
    B = int(rng.integers(3, 70))
    batch = scenarios.panda_batch(cfg, B, seed=seed, x_min=0.3 if N > 3 else 0.15, q_spread=0.15 if N > 3 else 0.3)
    sx, sv, _ = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
    ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv if cfg.dynamic else None, None)
    want_qdd, want = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
    h = FabricHandle(cfg, 0)
    act, qdd = h.compute_action_coupled(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]), want_qddot=True)
    assert relerr(act.cpu().numpy(), want) < F64_RTOL and relerr(qdd.cpu().numpy(), want_qdd) < F64_RTOL, (seed, N, len(keep))
    again = h.compute_action_coupled(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]))
    assert torch.equal(act, again)
