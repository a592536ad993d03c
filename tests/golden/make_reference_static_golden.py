#!/usr/bin/env python3
"""Pins what CAN be pinned against /root/reference today: everything here needs numpy / xml only.

    python tests/golden/make_reference_static_golden.py [--reference /root/reference]

Writes tests/golden/reference_static.npz (data only):
  urdf_joint_order      names of all <joint> elements of panda_with_finger.urdf in file order -- the index space of the
                        simulator's joint map that create_simulation_manipulators.py:202-207,232,240 indexes into
  chain_*               the kinematic chain panda_link0 -> panda_leftfinger: names, types, xyz, rpy, axis, lower/upper/velocity
  pm{2,3}_*             attributes of examples/parameters_manipulators.manipulator_parameters(nr_robots) (numpy-only
                        module, imported from the reference) that the planner-facing code reads
  yaml_*                the eight keys of examples/configs/panda_config.yaml
  sig_*                 the call contract of the example / evaluation entry points (examples/test_examples.py:8-36 calls
                        them as f(n_steps=100, render=False)): "module:function" names with their parameter names and the
                        repr of their defaults, read from the reference's files with `ast` (nothing is imported)
  result_keys_*         keys of the dictionary run_panda_example returns (example_pandas_Jointspace.py:509-515,
                        example_pandas_cartesian.py:518-523)
"""
import argparse
import importlib.util
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def urdf_chain(path):
    root = ET.parse(path).getroot()
    joints = root.findall("joint")
    order = [j.get("name") for j in joints]
    by_child = {j.find("child").get("link"): j for j in joints}
    chain, link = [], "panda_leftfinger"
    while link != "panda_link0":
        j = by_child[link]
        chain.append(j)
        link = j.find("parent").get("link")
    chain.reverse()

    def vec(el, attr, default):
        return [float(v) for v in (el.get(attr) if el is not None and el.get(attr) else default).split()]

    rec = dict(name=[], type=[], parent=[], child=[], xyz=[], rpy=[], axis=[], lower=[], upper=[], velocity=[])
    for j in chain:
        org, ax, lim = j.find("origin"), j.find("axis"), j.find("limit")
        rec["name"].append(j.get("name"))
        rec["type"].append(j.get("type"))
        rec["parent"].append(j.find("parent").get("link"))
        rec["child"].append(j.find("child").get("link"))
        rec["xyz"].append(vec(org, "xyz", "0 0 0"))
        rec["rpy"].append(vec(org, "rpy", "0 0 0"))
        rec["axis"].append(vec(ax, "xyz", "0 0 0"))
        for k in ("lower", "upper", "velocity"):
            rec[k].append(float(lim.get(k)) if lim is not None and lim.get(k) is not None else np.nan)
    return order, rec


def parameters(ref_root, n):
    spec = importlib.util.spec_from_file_location("ref_parameters_manipulators",
                                                  os.path.join(ref_root, "examples", "parameters_manipulators.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class _Np:      # PM:111-115 builds a ragged pos0 (7, 9, 7 entries): an object array on the numpy the reference pins,
        def __getattr__(self, k):                                      # an error on numpy >= 1.24 -- same fallback here
            return getattr(np, k)

        @staticmethod
        def array(x, *a, **kw):
            try:
                return np.array(x, *a, **kw)
            except ValueError:
                return np.array(x, dtype=object)
    mod.np = _Np()
    p = mod.manipulator_parameters(nr_robots=n, n_obst_per_link=4)
    out = {}
    for k in ("dt", "n_cubes", "nr_robots", "radius_sphere", "z_table", "N_HORIZON", "STATIC_OR_DYN_FABRICS",
              "n_obst_per_link"):
        out[k] = np.array(getattr(p, k))
    for k in ("dof", "nr_obsts", "nr_constraints", "nr_obsts_dyn", "nr_obsts_dyn_all", "collision_links_nrs", "r_robots",
              "mount_positions", "mount_orientations", "mount_transform", "rotation_matrix_pandas", "start_goals",
              "constraints", "r_dyns_obsts"):
        out[k] = np.array(getattr(p, k), dtype=float)
    out["pos0_7"] = np.array([np.asarray(x, dtype=float)[:7] for x in p.pos0])
    out["fabrics_mode"] = np.array(p.fabrics_mode)
    out["collision_links"] = np.array(p.collision_links)
    out["robot_types"] = np.array(p.robot_types)
    out["radius_body_keys"] = np.array(sorted(p.radius_body_panda_links))
    out["radius_body_values"] = np.array([float(p.radius_body_panda_links[k]) for k in sorted(p.radius_body_panda_links)])
    out["get_settings"] = np.array(p.get_settings(), dtype=float)
    out["define_settings"] = np.array(p.define_settings(True, False, 1, 1, False, 10, False, 4), dtype=float)
    return out


ENTRY_POINTS = {
    "examples/example_pandas_Jointspace.py": ["create_dummy_goal_panda", "set_planner_panda", "define_planners",
                                              "define_rollout_planners", "run_panda_example", "define_run_panda_example"],
    "examples/example_pandas_cartesian.py": ["create_dummy_goal_panda", "set_planner_panda", "define_planners",
                                             "define_rollout_planners", "run_panda_example", "define_run_panda_example"],
    "examples/example_pointmasses_static.py": ["set_planner_point", "run_point_example"],
    "examples/example_pointmasses_dynamic.py": ["set_planner_point", "run_point_example"],
    "examples/evaluation/evaluate_horizon.py": ["get_std", "define_run_evaluations"],
    "examples/evaluation/evaluate_random_dynamic_scenarios.py": ["get_std", "define_run_evaluations"],
}


def entry_point_contract(ref_root):
    """[(module:function, "name=default,name,...")] and the result keys of the two run_panda_example functions."""
    import ast
    names, params, result_keys = [], [], {}
    for rel, funcs in ENTRY_POINTS.items():
        with open(os.path.join(ref_root, rel)) as f:
            tree = ast.parse(f.read())
        defs = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
        for fn in funcs:
            a = defs[fn].args
            pos = a.posonlyargs + a.args
            defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
            items = [arg.arg if d is None else f"{arg.arg}={ast.unparse(d)}" for arg, d in zip(pos, defaults)]
            names.append(f"{rel}:{fn}")
            params.append(",".join(items))
        if "run_panda_example" in defs:
            ret = [n for n in ast.walk(defs["run_panda_example"]) if isinstance(n, ast.Return) and isinstance(n.value, ast.Dict)]
            result_keys[rel] = [k.value for k in ret[-1].value.keys]
    return names, params, result_keys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("MRF_REFERENCE", "/root/reference"))
    ref = os.path.abspath(ap.parse_args().reference)
    order, chain = urdf_chain(os.path.join(ref, "examples", "simulation_environments", "urdfs", "panda_with_finger.urdf"))
    out = {"urdf_joint_order": np.array(order)}
    for k, v in chain.items():
        out["chain_" + k] = np.array(v)
    for n in (2, 3):
        for k, v in parameters(ref, n).items():
            out[f"pm{n}_{k}"] = v
    import yaml
    with open(os.path.join(ref, "examples", "configs", "panda_config.yaml")) as f:
        y = yaml.safe_load(f)
    out["yaml_keys"] = np.array(sorted(y))
    out["yaml_values"] = np.array([float(y[k]) for k in sorted(y)])
    # format (and summary statistics) of the reference's recorded benchmark pickle, evaluation/results_horizon: what
    # examples/evaluation/evaluate_horizon.py has to reproduce in shape; the numbers are the ones BASELINE.md quotes
    import pickle
    with open(os.path.join(ref, "evaluation", "results_horizon"), "rb") as f:
        rec = pickle.load(f)
    out["results_horizon_container"] = np.array(type(rec).__name__)
    out["results_horizon_shapes"] = np.array([np.asarray(r).shape for r in rec])
    out["results_horizon_dtype"] = np.array(str(np.asarray(rec[0]).dtype))
    out["results_horizon_mean_s"] = np.array([float(np.asarray(r).mean()) for r in rec])
    out["results_horizon_median_s"] = np.array([float(np.median(np.asarray(r))) for r in rec])
    names, params, result_keys = entry_point_contract(ref)
    out["sig_names"], out["sig_params"] = np.array(names), np.array(params)
    out["result_keys_jointspace"] = np.array(result_keys["examples/example_pandas_Jointspace.py"])
    out["result_keys_cartesian"] = np.array(result_keys["examples/example_pandas_cartesian.py"])
    for n in (2, 3):      # simulator-facing attributes of manipulator_parameters that the scene stand-in reads
        spec = importlib.util.spec_from_file_location("ref_pm", os.path.join(ref, "examples", "parameters_manipulators.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _np = mod.np

        class _Np:
            def __getattr__(self, k):
                return getattr(_np, k)

            @staticmethod
            def array(x, *a, **kw):
                try:
                    return _np.array(x, *a, **kw)
                except ValueError:
                    return _np.array(x, dtype=object)
        mod.np = _Np()
        p = mod.manipulator_parameters(nr_robots=n)
        out[f"pm{n}_urdf_link_keys"] = np.array(sorted(p.get_urdf_locations()))
        out[f"pm{n}_urdf_basenames"] = np.array([os.path.basename(p.get_urdf_locations()[k]) for k in sorted(p.get_urdf_locations())])
        out[f"pm{n}_tray_positions"] = np.array(p.tray_positions, dtype=float)
        out[f"pm{n}_tray_orientations"] = np.array(p.tray_orientations, dtype=float)
        out[f"pm{n}_table_position"] = np.array(p.table_position, dtype=float)
    np.savez(os.path.join(HERE, "reference_static.npz"), **out)
    print("wrote reference_static.npz:", len(out), "arrays;", "joint order", order[:18])


if __name__ == "__main__":
    sys.exit(main())
