"""The example / evaluation entry points keep the reference's call contract (north_star "examples/configs
surface unchanged"): every function the reference's drivers define is importable under the reference's module path with
the same positional parameters and defaults (this build's extras are keyword-only), and `manipulator_parameters` carries
the simulator-facing attributes too.  Expected values: tests/golden/reference_static.npz, read from the reference's own
files with `ast` by tests/golden/make_reference_static_golden.py.  CPU only: nothing is launched."""
import importlib
import inspect
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "reference_static.npz"))
CONTRACT = list(zip(G["sig_names"].tolist(), G["sig_params"].tolist()))


def _positional(sig):
    out = []
    for p in sig.parameters.values():
        if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD):
            out.append(p.name if p.default is p.empty else f"{p.name}={p.default!r}")
    return ",".join(out)


@pytest.mark.parametrize("name,params", CONTRACT, ids=[n.split("/", 1)[1] for n, _ in CONTRACT])
def test_entry_point_signature(name, params):
    rel, fn = name.split(":")
    mod = importlib.import_module(rel[:-3].replace("/", "."))
    f = getattr(mod, fn)
    sig = inspect.signature(f)
    assert _positional(sig) == params, f"{name}{sig}"
    # anything beyond the reference's parameters must be keyword-only, so that no reference call changes meaning
    extra = [p for p in sig.parameters.values() if p.kind not in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    assert all(p.kind == p.KEYWORD_ONLY and p.default is not p.empty for p in extra)


def test_reference_smoke_test_call_is_accepted():
    """examples/test_examples.py:8-36 calls test_main(n_steps=100, render=False); the two evaluation scripts' __main__
    add n_runs (evaluate_horizon.py:121-124)."""
    for mod, fn in (("examples.example_pointmasses_static", "run_point_example"),
                    ("examples.example_pointmasses_dynamic", "run_point_example"),
                    ("examples.example_pandas_Jointspace", "define_run_panda_example"),
                    ("examples.example_pandas_cartesian", "define_run_panda_example")):
        inspect.signature(getattr(importlib.import_module(mod), fn)).bind(n_steps=100, render=False)
    for mod in ("examples.evaluation.evaluate_horizon", "examples.evaluation.evaluate_random_dynamic_scenarios"):
        inspect.signature(importlib.import_module(mod).define_run_evaluations).bind(n_steps=7000, render=False, n_runs=5)


@pytest.mark.parametrize("n", [2, 3])
def test_manipulator_parameters_simulator_attributes(n):
    import examples.parameters_manipulators as pm
    p = pm.manipulator_parameters(nr_robots=n)
    loc = p.get_urdf_locations()
    assert loc is p.urdf_links and sorted(loc) == G[f"pm{n}_urdf_link_keys"].tolist()
    assert [os.path.basename(loc[k]) for k in sorted(loc)] == G[f"pm{n}_urdf_basenames"].tolist()
    assert os.path.exists(loc["URDF_file_panda"])          # the one file the planner-side code opens (EXJ:80)
    assert np.array_equal(np.array(p.tray_positions, dtype=float), G[f"pm{n}_tray_positions"])
    assert np.array_equal(np.array(p.tray_orientations, dtype=float), G[f"pm{n}_tray_orientations"])
    assert np.array_equal(np.array(p.table_position, dtype=float), G[f"pm{n}_table_position"])


def test_shipped_urdf_is_the_compiled_chain():
    """The kinematics-only URDF the drivers read is generated from the compiled constants and equals the reference's
    chain (tests/golden/reference_static.npz chain_*) joint by joint."""
    import xml.etree.ElementTree as ET
    import examples.parameters_manipulators as pm
    root = ET.parse(pm.manipulator_parameters(nr_robots=2).URDF_file_panda).getroot()
    joints = {j.get("name"): j for j in root.findall("joint")}
    for i, name in enumerate(G["chain_name"].tolist()):
        j = joints[name]
        assert j.get("type") == G["chain_type"][i]
        assert j.find("parent").get("link") == G["chain_parent"][i] and j.find("child").get("link") == G["chain_child"][i]
        org = j.find("origin")
        assert np.allclose([float(v) for v in org.get("xyz").split()], G["chain_xyz"][i], atol=1e-12)
        assert np.allclose([float(v) for v in org.get("rpy").split()], G["chain_rpy"][i], atol=1e-9)
        if j.get("type") != "fixed":
            assert np.allclose([float(v) for v in j.find("axis").get("xyz").split()], G["chain_axis"][i])
            lim = j.find("limit")
            assert float(lim.get("lower")) == G["chain_lower"][i] and float(lim.get("upper")) == G["chain_upper"][i]


def test_cube_layouts():
    """cell.cube_layout without a GPU: the fixed places (two columns between the mounts, three cubes per robot for two
    robots, two for three; SIM:37-60,113-137) and the random scene's spacing rule (SIM:62-76)."""
    import examples.parameters_manipulators as pm
    from multi_robot_fabrics_amd.cell import cube_layout
    for n in (2, 3):
        p = pm.manipulator_parameters(nr_robots=n)
        fixed = cube_layout(p)
        assert fixed.shape == (1, 6, 3)
        assert np.allclose(fixed[0, :, 2], 0.65 + 0.025) and set(np.round(fixed[0, :, 0], 3)) == {0.4, 0.6}
        assert np.allclose(sorted(set(np.round(fixed[0, :, 1] - (0.2 if n == 3 else 0.0), 3))), [-0.15, 0.0, 0.15])
        assert len({tuple(np.round(c, 6)) for c in fixed[0]}) == 6
        rnd = cube_layout(p, random_scene=True, rng=np.random.default_rng(5), scenes=3)
        assert rnd.shape == (3, 6, 3)
        for P in rnd:
            d = np.linalg.norm(P[:, None, :2] - P[None, :, :2], axis=2) + 10 * np.identity(6)
            assert d.min() > 0.05 + 0.06 and P[:, 0].min() >= 0.4 and P[:, 0].max() <= 0.6
        assert not np.allclose(rnd[0], rnd[1])
    two = cube_layout(pm.manipulator_parameters(nr_robots=2), n_cubes=2)[0]
    assert np.allclose(two[:, :2], [[0.4, 0.0], [0.6, -0.15]])      # one cube each: the first place of either column
