"""CPU: host logic and the C-ABI surface (no compute calls -- there is no GPU here)."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from multi_robot_fabrics_amd import abi, config, leafspec, sharded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------ leaf strings
REFERENCE_STRINGS = {
    # example_pandas_Jointspace.py:87-89
    "10*(1/(1+1*ca.exp(-10*x))-1) * (xdot**2)": (abi.FAMILY_LOGISTIC, abi.GATE_NONE, 0, 10.0, 1.0, 10.0),
    "-0.5 / (x ** 4) * (xdot ** 2)": (abi.FAMILY_POW, abi.GATE_NONE, 4, -0.5, 0.0, 0.0),
    "0.01/(x**4) * xdot**2": (abi.FAMILY_POW, abi.GATE_NONE, 4, 0.01, 0.0, 0.0),
    # example_pointmasses_static.py:106-107
    "-2.0 / (x ** 1) * xdot ** 2": (abi.FAMILY_POW, abi.GATE_NONE, 1, -2.0, 0.0, 0.0),
    "1.0/(x**2) * (1 - ca.heaviside(xdot))* xdot**2": (abi.FAMILY_POW, abi.GATE_NEG, 2, 1.0, 0.0, 0.0),
    # library defaults (recalled)
    "-0.5 / (x ** 5) * (-0.5 * (ca.sign(xdot) - 1)) * xdot ** 2": (abi.FAMILY_POW, abi.GATE_NEG, 5, -0.5, 0.0, 0.0),
    "0.1/(x**1) * (-0.5 * (ca.sign(xdot) - 1)) * xdot**2": (abi.FAMILY_POW, abi.GATE_NEG, 1, 0.1, 0.0, 0.0),
}


@pytest.mark.parametrize("expr,want", list(REFERENCE_STRINGS.items()))
def test_reference_leaf_strings_parse(expr, want):
    fn = leafspec.parse_leaf(expr)
    got = fn.as_tuple()
    assert got[:3] == want[:3]
    assert np.allclose(got[3:], want[3:], rtol=1e-15, atol=0)


def test_gate_value_at_zero_velocity():
    fn = leafspec.parse_leaf("1.0/(x**2) * (1 - ca.heaviside(xdot))* xdot**2")
    assert leafspec.metric(fn, 0.5, -1.0) == pytest.approx(8.0)
    assert leafspec.metric(fn, 0.5, 1.0) == 0.0
    assert leafspec.metric(fn, 0.5, 0.0) == pytest.approx(4.0)      # casadi heaviside(0) = 0.5


@pytest.mark.parametrize("expr", ["x*xdot", "ca.exp(x)*xdot**2", "1/(x**0.5)*xdot**2", "xdot**2 + x", "import os",
                                  "1/(x**2) * ca.tanh(xdot) * xdot**2"])
def test_unsupported_leaf_strings_raise(expr):
    with pytest.raises(leafspec.LeafSpecError):
        leafspec.parse_leaf(expr)


# ------------------------------------------------------------------------------------------ config / ABI
def test_library_loads_and_exports_every_declared_symbol():
    lib = abi.load_library()
    header = open(os.path.join(ROOT, "include", "mrf.h")).read()
    declared = set(re.findall(r"\b(mrf_[a-z0-9_]+)\s*\(", header)) - {"mrf_handle"}
    assert declared == set(abi.EXPORTS), declared ^ set(abi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mrf_abi_version() == abi.MRF_ABI_VERSION
    assert lib.mrf_config_sizeof() == C.sizeof(abi.Config)


def _struct_equal(a, b, skip=()):
    for name, _ in abi.Config._fields_:
        if name in skip:
            continue
        va, vb = getattr(a, name), getattr(b, name)
        if isinstance(va, abi.LeafFn):
            assert va.as_tuple() == pytest.approx(vb.as_tuple()), name
        elif hasattr(va, "__len__"):
            assert np.allclose(np.array(va[:]).astype(float) if not hasattr(va[0], "__len__") else np.array([list(r) for r in va]),
                               np.array(vb[:]).astype(float) if not hasattr(vb[0], "__len__") else np.array([list(r) for r in vb]),
                               atol=1e-15), name
        else:
            assert va == pytest.approx(vb), name


@pytest.mark.parametrize("n_robots", [2, 3, 8])
def test_c_defaults_match_python_config(n_robots):
    """mrf_default_config_panda (C, host-only) and config.panda_config (Python, from the leaf strings) agree."""
    lib = abi.load_library()
    c = abi.Config()
    lib.mrf_default_config_panda(C.byref(c), n_robots, 30)
    p = config.panda_config(n_robots=n_robots, horizon=30)
    _struct_equal(c, p)


def test_c_defaults_planar():
    lib = abi.load_library()
    c = abi.Config()
    lib.mrf_default_config_planar3(C.byref(c), 4)
    p = config.planar3_config(n_robots=4)
    _struct_equal(c, p, skip=("plane_geometry",))


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(abi.MrfLibraryError):
        abi.load_library(str(tmp_path / "libmrf_hip.so"))


def test_no_gpu_means_no_handle():
    """The product path has no CPU fallback: without a HIP device a planner cannot be concretized."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from multi_robot_fabrics_amd.runtime import FabricHandle, MrfError
    with pytest.raises(MrfError):
        FabricHandle(config.panda_config(2, 5), 0)
    lib = abi.load_library()
    h = C.c_void_p()
    cfg = config.panda_config(2, 5)
    rc = lib.mrf_create(C.byref(cfg), 0, C.byref(h))
    assert rc == -3 and b"no HIP device" in lib.mrf_last_error(h)
    lib.mrf_destroy(h)
    bad = config.panda_config(2, 5)
    bad.n_ego = 4
    rc = lib.mrf_create(C.byref(bad), 0, C.byref(h))
    assert rc == -2 and b"n_ego" in lib.mrf_last_error(h)
    lib.mrf_destroy(h)


def test_mounts_and_sphere_table():
    p = config.panda_config(n_robots=3, horizon=1)
    m1 = np.array(p.mount[1][:]).reshape(3, 4)
    assert np.allclose(m1, [[-1, 0, 0, 1.0], [0, -1, 0, 0.0], [0, 0, 1, 0.65]], atol=1e-15)    # PM:101-105,139-150
    assert list(p.sphere_link[:8]) == list(range(1, 9)) and p.sphere_radius[0] == 0.08
    links, offs = config.sphere_offsets_per_link(4)
    assert len(links) == 32 and links == sorted(links)
    assert offs[0] == [0.0, 0.0, -0.333] and offs[1][2] == pytest.approx(-0.333 + 0.333 / 4)   # SIM:220-226
    with pytest.raises(ValueError):
        config.set_spheres(p, [1] * 33)


# ------------------------------------------------------------------------------------------ planner front-end (no GPU)
def _panda_planner():
    from multi_robot_fabrics_amd.planner import ParameterizedFabricPlanner, GenericURDFFk
    from multi_robot_fabrics_amd.goals import GoalComposition
    goal = GoalComposition("goal", {
        "subgoal0": dict(weight=2.0, is_primary_goal=True, indices=[0, 1, 2], parent_link="world", child_link="panda_hand",
                         desired_position=[0.1, 0.6, 0.8], epsilon=0.05, type="staticSubGoal"),
        "subgoal1": dict(weight=10.0, is_primary_goal=False, indices=[0, 1, 2], parent_link="panda_link7",
                         child_link="panda_hand", desired_position=[0.107, 0.0, 0.0], angle=[-0.366, 0.0, 0.0, 0.3305],
                         epsilon=0.05, type="staticSubGoal"),
        "subgoal2": dict(weight=1.0, is_primary_goal=False, indices=[6], desired_position=[math.pi / 4], epsilon=0.05,
                         type="staticJointSpaceSubGoal")})
    fk = GenericURDFFk(None, "panda_link0", "panda_leftfinger")
    planner = ParameterizedFabricPlanner(7, fk, collision_geometry="-0.5 / (x ** 4) * (xdot ** 2)",
                                         collision_finsler="0.01/(x**4) * xdot**2",
                                         geometry_plane_constraint="10*(1/(1+1*ca.exp(-10*x))-1) * (xdot**2)")
    planner.set_components(collision_links=["panda_link%d" % i for i in range(1, 9)], goal=goal, number_obstacles=0,
                           number_dynamic_obstacles=2, dynamic_obstacle_dimension=3, number_plane_constraints=1,
                           limits=config.PANDA_LIMITS)
    return planner, goal


def test_planner_input_keys_follow_reference_order():
    planner, goal = _panda_planner()
    keys = planner.input_keys()
    # forward_planner_Jointspace.py:227-232
    assert keys[:4] == ["angle_goal_1", "constraint_0", "q", "qdot"]
    assert keys[4:10] == ["radius_body_panda_link%d" % l for l in range(3, 9)]
    assert keys[10:12] == ["radius_obst_dynamic_0", "radius_obst_dynamic_1"]
    assert keys[12:18] == ["weight_goal_0", "weight_goal_1", "weight_goal_2", "x_goal_0", "x_goal_1", "x_goal_2"]
    assert keys[18:] == ["x_obst_dynamic_0", "x_obst_dynamic_1", "xddot_obst_dynamic_0", "xddot_obst_dynamic_1",
                         "xdot_obst_dynamic_0", "xdot_obst_dynamic_1"]
    assert len(goal._config) == 3 and goal._config.subgoal1.desired_position == [0.107, 0.0, 0.0]
    assert len(planner.leaves) == 6 * 2 + 6 + 14 + 3


def test_planner_argument_marshalling():
    planner, _ = _panda_planner()
    kw = dict(q=np.zeros(7), qdot=np.zeros(7), x_goal_0=[0.1, 0.2, 0.3], weight_goal_0=2.0, angle_goal_1=np.eye(3),
              x_goal_1=[0.107, 0, 0], weight_goal_1=20.0, x_goal_2=np.array([0.7]), weight_goal_2=1.0,
              constraint_0=np.array([0, 0, 1, -0.65]), radius_body_panda_links={str(l): np.array(0.08) for l in range(3, 9)},
              radius_body_panda_hand=np.array([0.08]),        # ignored, as in the reference (EXJ:434)
              x_obsts_dynamic=[[1, 2, 3], [4, 5, 6]], xdot_obsts_dynamic=[[0, 0, 1]] * 2,
              xddot_obsts_dynamic=[np.zeros(3)] * 2, radius_obsts_dynamic=[0.08, 0.09],
              x_obsts=[[9, 9, 9]], radius_obsts=[1.0])        # static lists ignored: built with number_obstacles=0
    p = planner.params_row(kw)
    assert p[abi.P_X_GOAL_0:abi.P_X_GOAL_0 + 3].tolist() == [0.1, 0.2, 0.3] and p[abi.P_WEIGHT_GOAL_1] == 20.0
    assert p[abi.P_CONSTRAINT_0 + 3] == -0.65 and p[abi.P_RADIUS_BODY + 5] == 0.08
    ox, ov, oa, orad, ns = planner.obstacle_arrays(kw)
    assert ns == 0 and ox.tolist() == [[1, 2, 3], [4, 5, 6]] and orad.tolist() == [0.08, 0.09] and ov[1, 2] == 1
    del kw["xdot_obsts_dynamic"]
    with pytest.raises(KeyError):
        planner.obstacle_arrays(kw)
    # a subset of the collision links (the reference's set_planner_panda default is link 5 alone, EXJ:64) is a mask
    planner.set_components(collision_links=["panda_link1", "panda_link5", "panda_hand"], goal=None, number_obstacles=1,
                           number_plane_constraints=1)
    assert planner._ego_links == [5, 8] and config.ego_link_mask(planner._ego_links) == 0b100100
    assert sorted(k for k in planner.leaves if "obst" in k) == ["panda_link5_obst_0_leaf", "panda_link8_obst_0_leaf"]
    assert "radius_body_panda_link5" in planner.input_keys() and "radius_body_panda_link3" not in planner.input_keys()
    with pytest.raises(KeyError):
        planner.set_components(collision_links=["panda_link9"], goal=None)


def test_urdf_constants_are_checked():
    from multi_robot_fabrics_amd.kinematics import GenericURDFFk
    ok = "<robot>" + "".join(
        f'<joint name="panda_joint{i+1}"><origin rpy="{r[0]} {r[1]} {r[2]}" xyz="{x[0]} {x[1]} {x[2]}"/></joint>'
        for i, (x, r) in enumerate([((0, 0, 0.333), (0, 0, 0)), ((0, 0, 0), (-math.pi / 2, 0, 0)),
                                    ((0, -0.316, 0), (math.pi / 2, 0, 0)), ((0.0825, 0, 0), (math.pi / 2, 0, 0)),
                                    ((-0.0825, 0.384, 0), (-math.pi / 2, 0, 0)), ((0, 0, 0), (math.pi / 2, 0, 0)),
                                    ((0.088, 0, 0), (math.pi / 2, 0, 0)), ((0, 0, 0.107), (0, 0, 0))])) + "</robot>"
    GenericURDFFk(ok, "panda_link0", "panda_leftfinger")
    with pytest.raises(ValueError):
        GenericURDFFk(ok.replace("0.333", "0.4"), "panda_link0", "panda_leftfinger")
    with pytest.raises(NotImplementedError):
        GenericURDFFk(None, "base", "tool0")


# ------------------------------------------------------------------------------------------ sharding plan
def test_robot_groups_and_partition():
    # groups of N ranks (one robot per GPU: BASELINE config 4 "robots sharded 1/GPU"), then one smaller group
    assert sharded.group_layout(3, 1) == [1]
    assert sharded.group_layout(3, 2) == [2]
    assert sharded.group_layout(3, 3) == [3]
    assert sharded.group_layout(3, 4) == [3, 1]
    assert sharded.group_layout(3, 8) == [3, 3, 2]
    assert sharded.group_layout(8, 8) == [8]
    assert sharded.group_layout(8, 4) == [4]
    assert sharded.group_layout(2, 8) == [2, 2, 2, 2]
    assert [sharded.rank_placement(3, 8, r) for r in range(8)] == [
        (0, 0, 0, 3), (0, 1, 0, 3), (0, 2, 0, 3), (1, 0, 3, 3), (1, 1, 3, 3), (1, 2, 3, 3), (2, 0, 6, 2), (2, 1, 6, 2)]
    assert sharded.rank_placement(3, 4, 3) == (1, 0, 3, 1)
    # a group's scenario share is inversely proportional to its robots per rank: the ranks finish together
    assert sharded.group_scenarios(3, 8, 1200) == [1200, 1200, 600]
    assert sharded.group_scenarios(3, 4, 1200) == [1200, 400]
    assert sharded.group_scenarios(8, 8, 1200) == [1200] and sharded.group_scenarios(3, 1, 1200) == [400]
    for n in range(1, 9):
        for w in range(1, 17):
            sizes = sharded.group_layout(n, w)
            assert sum(sizes) == w and all(1 <= g <= n for g in sizes) and sizes == sorted(sizes, reverse=True)
    assert sharded.robot_partition(3, 2) == [(0, 2), (2, 1)]
    assert sharded.robot_partition(8, 4) == [(0, 2), (2, 2), (4, 2), (6, 2)]
    for n in range(1, 17):
        for g in range(1, n + 1):
            parts = sharded.robot_partition(n, g)
            assert sum(c for _, c in parts) == n and all(c >= 1 for _, c in parts)
            assert [f for f, _ in parts] == list(np.cumsum([0] + [c for _, c in parts[:-1]]))


# ------------------------------------------------------------------------------------------ deadlock logic (DP:50-118)
def test_deadlock_checking_known_answers():
    from multi_robot_fabrics_amd.deadlock import deadlockprevention
    dp = deadlockprevention([7, 7], 2, 10)
    x = [np.array([0.5, 0.0, 1.0]), np.array([0.6, 0.1, 1.0])]
    goals = [np.array([0.2, 0.6, 1.15]), np.array([0.8, -0.6, 1.15])]
    weights = [2.0, 2.0]
    # moving robots: nothing happens, the wait counter keeps its "idle" value
    g, w, t = dp.deadlock_checking(x, list(goals), list(weights), time_step=50, time_deadlock_out=1000, avg_sum=0.5,
                                   state_machine_robots=[0, 0])
    assert w == [2.0, 2.0] and t == 1000 and np.array_equal(g[1], goals[1])
    # stalled, close end-effectors, both approaching: robot 0 is closer to its goal -> leader (weight 3), robot 1 backs off
    g, w, t = dp.deadlock_checking(x, list(goals), list(weights), time_step=50, time_deadlock_out=1000, avg_sum=0.01,
                                   state_machine_robots=[0, 1])
    d0, d1 = np.linalg.norm(x[0] - goals[0]), np.linalg.norm(x[1] - goals[1])
    lead, foll = (0, 1) if d0 <= d1 else (1, 0)
    assert t == 0 and w[lead] == 3 and w[foll] == 2
    diff = (x[lead] - x[foll]) * 2
    assert np.allclose(g[foll], x[foll] - 0.3 / np.linalg.norm(diff) * diff)
    # afterwards the rewritten goal is held for time_wait steps
    g2, w2, t2 = dp.deadlock_checking(x, list(goals), list(weights), time_step=51, time_deadlock_out=t, avg_sum=0.5,
                                      state_machine_robots=[0, 1])
    assert t2 == 1 and w2[lead] == 3 and np.allclose(g2[foll], g[foll])
    # too early in the episode (time_step <= 10): never a deadlock
    dp2 = deadlockprevention([7, 7, 7], 3, 10)
    g3, w3, t3 = dp2.deadlock_checking(x + [np.array([2.0, 2.0, 1.0])], list(goals) + [np.zeros(3)], [2.0] * 3, time_step=5,
                                       time_deadlock_out=1000, avg_sum=0.0, state_machine_robots=[0, 0, 0])
    assert w3 == [2.0] * 3 and t3 == 1000


def test_deadlock_config_defaults_and_null_handles():
    """Host-only parts of the control-step ABI: the thresholds of deadlock_prevention.py:12-27 (+ literals of :50-118)
    and argument checking without a device."""
    lib = abi.load_library()
    assert lib.mrf_deadlock_config_sizeof() == C.sizeof(abi.DeadlockConfig)
    c = abi.DeadlockConfig()
    lib.mrf_default_deadlock_config(C.byref(c), 0)
    assert (c.avg_vel_constant, c.dist_constant, c.goal_weight_follower, c.goal_weight_leader, c.time_wait,
            c.nr_goal_scale) == (0.16, 0.0, 2.0, 3.0, 300, 2.0)
    assert (c.ee_distance, c.follower_offset, c.min_goal_norm, c.z_floor, c.min_time_step, c.grasp_state,
            c.grasp_timeout) == (0.35, 0.3, 0.05, 0.1, 10, 2, 400)
    lib.mrf_default_deadlock_config(C.byref(c), 1)
    assert (c.avg_vel_constant, c.dist_constant, c.goal_weight_follower, c.goal_weight_leader, c.time_wait,
            c.nr_goal_scale) == (0.03, 1.0, 10.0, 1.0, 50, 100.0)
    vl = (C.c_double * 7)(*[1.0] * 7)
    assert lib.mrf_deadlock_init(None, 4, None, None, None) == -1
    assert lib.mrf_control_prepare(None, 4, None, None, None, None, 1, None, None) == -1
    assert lib.mrf_deadlock_step(None, 4, C.byref(c), -1, None, None, None, None, None, None, None) == -1
    assert lib.mrf_apply_action(None, 4, None, None, None, vl, -1.0, None) == -1
    assert lib.mrf_episode_run(None, None, 4, 1, None, 0, vl, -1.0, *([None] * 10), 0, None) == -1


def test_plain_c_consumer_of_the_sharded_abi_builds():
    """examples/sharded_rollout_c.cpp drives mrf_comm_peer_* / mrf_rollout_sharded from C++ without Python or torch;
    it must compile and link against the library as built (it RUNS in tests/test_gpu_sharded_abi.py)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "examples")])
    assert os.path.exists(os.path.join(root, "examples", "sharded_rollout_c"))


def test_bench_watchdog_fires_only_when_the_block_overruns():
    """bench.run_guarded: the wall-clock guard around the robot-sharded block (no GPU needed: the module's GPU work is
    under main())."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fired = []
    assert bench.run_guarded(lambda: 7, None, lambda: fired.append("x")) == 7 and not fired      # no guard requested
    assert bench.run_guarded(lambda: 8, 5.0, lambda: fired.append("x")) == 8
    time.sleep(0.05)
    assert not fired                                                                              # returned in time
    assert bench.run_guarded(lambda: (time.sleep(0.3), 9)[1], 0.05, lambda: fired.append("late")) == 9
    assert fired == ["late"]                                                                      # overran: guard fired once


def test_topology_and_exchange_schema_without_a_gpu():
    """the fields a first multi-GPU line audits itself with exist whatever the machine -- here, without a
    device, the topology block reports zero devices and the library's status instead of raising, and the payload arithmetic
    of the two exchange kinds is what include/mrf.h states (21 joint-state scalars; 9 per exchanged sphere)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t = bench.node_topology()
    assert set(t) >= {"n_devices", "status", "can_access_peer", "link_type", "hops", "visible_devices_env"}, t
    assert len(t["can_access_peer"]) == len(t["link_type"]) == len(t["hops"]) == min(t["n_devices"], 16)
    assert set(bench.SHARDED_KEYS) >= {"exchange", "allgather_bytes_per_rank_per_step", "roofline", "ranks", "parity_vs_fused_kernel"}
    assert abi.COMM_INFO_KEYS[10:] == ("exchange", "exchange_scalars_per_robot", "peers_one_hop", "coresident_workgroups", "paired_blocks", "tagged_payload")
    assert len(abi.COMM_INFO_KEYS) == 16
    assert abi.PEER_INFO_KEYS == ("device", "can_access_peer", "link_type", "hops")
    header = open(os.path.join(ROOT, "include", "mrf.h")).read()
    assert "#define MRF_COMM_INFO_N 16" in header and "#define MRF_PEER_INFO_N 4" in header
    assert "#define MRF_JOINT_STATE_SCALARS 21" in header and abi.JOINT_STATE_SCALARS == 21
    cfg = config.panda_config(n_robots=3, horizon=2)
    assert cfg.exchange == abi.EXCHANGE_JOINTS                       # the default payload
    c2 = cfg.copy()
    c2.exchange = 7
    out = C.c_void_p()
    rc = abi.load_library().mrf_create(C.byref(c2), 0, C.byref(out))
    assert rc == -2 and b"exchange" in abi.load_library().mrf_last_error(out)     # MRF_E_CONFIG, before any device is touched
    abi.load_library().mrf_destroy(out)


def test_free_port_is_bindable_and_below_the_ephemeral_range():
    """bench.free_port: the rendezvous ports handed to OTHER processes (self-spawned ranks, the robot-sharded block's children,
    the multi-process tests) are free now and cannot be taken by an outgoing connection before the listener binds them."""
    import socket
    import bench
    lo = 32768
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            lo = int(f.read().split()[0])
    except (OSError, ValueError):
        pass
    seen = set()
    for _ in range(8):
        port = bench.free_port()
        assert 10000 <= port < max(min(lo, 32768), 12000)
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", port))
        seen.add(port)
    assert len(seen) > 1          # drawn at random, not one fixed number
