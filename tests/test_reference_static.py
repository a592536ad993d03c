"""What can be pinned against the reference's own files today, from tests/golden/reference_static.npz
(made by tests/golden/make_reference_static_golden.py out of the URDF, parameters_manipulators.py and the YAML):

 * the chain constants compiled into csrc/mrf_device.hpp (and used by the oracle) are the URDF's joint origins / rolls,
   the planner's limit table is the URDF's, and a generic URDF forward kinematics agrees with the oracle's chain;
 * the joint-index assumptions of config.sphere_offsets_per_link (simulator index 16 = panda_joint8, 11 = panda_joint5,
   create_simulation_manipulators.py:232,240);
 * parameters.manipulator_parameters agrees attribute by attribute with the reference's class for 2 and 3 robots;
 * parameters.load_yaml_settings reads the reference's eight-key YAML."""
import math
import os
import re

import numpy as np
import pytest

from multi_robot_fabrics_amd import config, parameters

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "reference_static.npz"))


def header_array(name):
    src = open(os.path.join(ROOT, "multi-robot-fabrics_amd", "csrc", "mrf_device.hpp")).read()
    m = re.search(r"%s\[8\]\s*=\s*\{([^}]*)\}" % name, src)
    return [float(v) for v in m.group(1).split(",")]


def test_chain_constants_in_the_kernels_are_the_urdfs():
    names = [str(n) for n in G["chain_name"]]
    assert names[:9] == ["panda_joint%d" % j for j in range(1, 9)] + ["panda_hand_joint"]
    assert [str(t) for t in G["chain_type"][:9]] == ["revolute"] * 7 + ["fixed", "fixed"]
    xyz, rpy = G["chain_xyz"], G["chain_rpy"]
    np.testing.assert_allclose(header_array("kPX"), xyz[:8, 0], atol=0)
    np.testing.assert_allclose(header_array("kPY"), xyz[:8, 1], atol=0)
    np.testing.assert_allclose(header_array("kPZ"), xyz[:8, 2], atol=0)
    roll = np.array(header_array("kROLL")) * math.pi / 2
    np.testing.assert_allclose(roll, rpy[:8, 0], atol=1e-6)          # the URDF prints pi/2 as 1.57079632679
    assert np.abs(rpy[:8, 1:]).max() == 0.0                           # pitch = yaw = 0 on the arm joints
    assert np.abs(G["chain_axis"][:7] - np.array([0, 0, 1.0])).max() == 0.0   # every revolute axis is local z
    assert np.abs(xyz[8]).max() == 0.0                                # panda_hand origin == panda_link8 origin
    np.testing.assert_allclose(np.array(config.PANDA_LIMITS), np.stack([G["chain_lower"][:7], G["chain_upper"][:7]], 1))
    np.testing.assert_allclose(config.PANDA_VEL_LIMITS, G["chain_velocity"][:7])


def urdf_fk(q, mount):
    """Generic URDF forward kinematics from the fixture: origins of panda_link1..8 (world)."""
    T = np.array(mount, dtype=float)
    out = []
    for j in range(8):
        x, y, z = G["chain_xyz"][j]
        r, p, yw = G["chain_rpy"][j]
        Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
        Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
        Rz = np.array([[math.cos(yw), -math.sin(yw), 0], [math.sin(yw), math.cos(yw), 0], [0, 0, 1]])
        O = np.identity(4)
        O[:3, :3] = Rz @ Ry @ Rx
        O[:3, 3] = [x, y, z]
        T = T @ O
        if str(G["chain_type"][j]) == "revolute":
            c, s = math.cos(q[j]), math.sin(q[j])
            Q = np.identity(4)
            Q[:2, :2] = [[c, -s], [s, c]]
            T = T @ Q
        out.append(T[:3, 3].copy())
    return np.array(out)


def test_urdf_forward_kinematics_agrees_with_the_oracle_chain(oracle):
    rng = np.random.default_rng(0)
    for n in (2, 3):
        cfg = config.panda_config(n_robots=n, horizon=1)
        q = rng.uniform(-2.5, 2.5, (7, n))
        x, _, _ = oracle.fk_spheres(cfg, q, np.zeros_like(q))
        for i in range(n):
            want = urdf_fk(q[:, i], G[f"pm{n}_mount_transform"][i])      # the reference's own mount transforms
            assert np.abs(x[:, :, i] - want).max() < 1e-10           # 1.57079632679 vs pi/2 in the URDF: ~5e-12


def test_simulator_joint_indices_behind_the_sphere_offsets():
    order = [str(n) for n in G["urdf_joint_order"]]
    assert order.index("panda_joint8") == 16 and order.index("panda_joint5") == 11      # SIM:232, SIM:240
    links, offs = config.sphere_offsets_per_link(4)
    offs = np.array(offs).reshape(8, 4, 3)
    length = [0.333, 0.2, 0.3164, 0.2, 0.3840, 0.2, 0.088, 0.2]
    for li in range(8):
        z_start = length[li] if li % 2 == 0 else length[li] / 2
        z = -z_start + np.arange(4) * length[li] / 4
        if li == 7:
            z[1] = -z_start + 2 * length[li] / 4                                      # SIM:235
        np.testing.assert_allclose(offs[li, :, 2], z, atol=1e-15)
    assert offs[7, 1, :2].tolist() == [0.03, 0.03] and offs[7, 2, :2].tolist() == [-0.03, -0.03]   # hand, SIM:232-239
    assert offs[4, 2, :2].tolist() == [0.0, 0.02] and offs[4, 3, :2].tolist() == [0.0, 0.06]       # link 5, SIM:240-246
    untouched = [(l, i) for l in range(8) for i in range(4) if (l, i) not in ((7, 1), (7, 2), (4, 2), (4, 3))]
    assert all(np.abs(offs[l, i, :2]).max() == 0.0 for l, i in untouched)


@pytest.mark.parametrize("n", [2, 3])
def test_manipulator_parameters_match_the_reference_class(n):
    p = parameters.manipulator_parameters(nr_robots=n, n_obst_per_link=4)
    g = lambda k: G[f"pm{n}_{k}"]
    for k in ("dt", "n_cubes", "nr_robots", "radius_sphere", "z_table", "N_HORIZON", "STATIC_OR_DYN_FABRICS", "n_obst_per_link"):
        assert float(getattr(p, k)) == float(g(k)), k
    for k in ("dof", "nr_obsts", "nr_constraints", "nr_obsts_dyn", "nr_obsts_dyn_all", "collision_links_nrs", "r_robots",
              "mount_positions", "mount_orientations", "mount_transform", "rotation_matrix_pandas", "start_goals",
              "constraints", "r_dyns_obsts"):
        np.testing.assert_allclose(np.array(getattr(p, k), dtype=float), g(k), atol=1e-15, err_msg=k)
    np.testing.assert_allclose(np.array([x[:7] for x in p.pos0]), g("pos0_7"), atol=0)
    assert p.fabrics_mode == str(g("fabrics_mode"))
    assert [list(x) for x in p.collision_links] == [list(map(str, x)) for x in g("collision_links")]
    assert p.robot_types == [str(x) for x in g("robot_types")]
    assert sorted(p.radius_body_panda_links) == [str(x) for x in g("radius_body_keys")]
    assert [float(p.radius_body_panda_links[k]) for k in sorted(p.radius_body_panda_links)] == list(g("radius_body_values"))
    assert [float(x) for x in p.get_settings()] == list(g("get_settings"))
    assert [float(x) for x in p.define_settings(True, False, 1, 1, False, 10, False, 4)] == list(g("define_settings"))


def test_load_yaml_settings_reads_the_reference_yaml(tmp_path):
    keys, vals = [str(k) for k in G["yaml_keys"]], G["yaml_values"]
    assert keys == sorted(["n_robots", "ROLLOUT_FABRICS", "ROLLOUTS_PLOTTING", "STATIC_OR_DYN_FABRICS", "RESOLVE_DEADLOCKS",
                           "ESTIMATE_GOAL", "N_HORIZON", "n_obst_per_link"])
    ref = dict(zip(keys, vals))
    # the committed example configuration carries the reference's values ...
    p, setup = parameters.load_yaml_settings(os.path.join(ROOT, "examples", "configs", "panda_config.yaml"))
    assert {k: float(v) for k, v in setup.items()} == ref
    # ... and they land where the reference's driver puts them (EXJ:517-531)
    assert p.nr_robots == ref["n_robots"] and p.n_obst_per_link == ref["n_obst_per_link"] and p.N_HORIZON == ref["N_HORIZON"]
    assert p.ROLLOUT_FABRICS is True and p.ROLLOUTS_PLOTTING is False and p.ESTIMATE_GOAL is False
    assert p.STATIC_OR_DYN_FABRICS == 1 and p.RESOLVE_DEADLOCKS == 1
    assert p.nr_obsts_dyn_all == [8 * 4] * 2 and p.nr_obsts_dyn == [8] * 2


def test_reference_requirements_are_the_lock_file():
    """tests/golden/reference_requirements.txt (the environment of the pin recipe) is generated, not typed: it equals what
    make_reference_requirements.py renders from the reference's poetry.lock, and names the three packages that hold the
    arithmetic at the versions SURVEY 8c cites."""
    import os
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(here, "reference_requirements.txt")) as f:
        text = f.read()
    for pin in ("casadi==3.5.5", "fabrics==0.9.5", "forwardkinematics==1.2.3"):
        assert pin + " \\\n" in text, pin
    assert text.count("--hash=sha256:") > 300
    lock = "/root/reference/poetry.lock"
    if not os.path.exists(lock):
        pytest.skip("the reference is not present here")
    sys.path.insert(0, here)
    import make_reference_requirements as gen
    assert gen.render(lock) == text
