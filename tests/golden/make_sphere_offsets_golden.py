#!/usr/bin/env python3
"""Pins SURVEY row f2 (the simulator's collision-sphere offsets) by RUNNING the reference's own generator
(VERDICT r5 item 5), in the build container:

    python tests/golden/make_sphere_offsets_golden.py [--reference /root/reference]

`create_manipulators_simulation.add_collision_spheres` (examples/simulation_environments/create_simulation_manipulators.py:176-257)
is plain Python + numpy arithmetic around two simulator calls.  This script takes the method's AST out of the reference's
file, compiles THAT (nothing is copied, the file is read where it lies), and calls it with
  * `self`  = a namespace carrying the attributes the method reads, taken from the reference's own
              parameters_manipulators.manipulator_parameters(nr_robots, n_obst_per_link) (a numpy-only module, imported);
  * `env`   = a stub whose joint map is the <joint> order of the reference's panda_with_finger.urdf (what
              env.env.env._robots[0]._urdf_robot._joint_map is in the simulator: yourdfpy keeps file order) and whose
              add_collision_link(...) records its keyword arguments.
Writes tests/golden/sphere_offsets.npz -- arrays only:
  n{K}_r{N}_robot / _link_index / _sphere_on_link / _offset[.,3] / _rotation_is_identity / _size     every recorded call, in call order,
                                                                        for n_obst_per_link K in 1..4 and N in {2, 3} robots
  n{K}_r{N}_link_transform_list      the method's own return structure [robot][link][sphere][4][4]
"""
import argparse
import ast
import importlib.util
import os
import types
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REL = os.path.join("examples", "simulation_environments", "create_simulation_manipulators.py")


def reference_method(ref_root, cls="create_manipulators_simulation", name="add_collision_spheres"):
    """The reference's method as a plain function, compiled from its own file's AST with numpy as its only global."""
    path = os.path.join(ref_root, REL)
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    klass = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    fn = next(n for n in klass.body if isinstance(n, ast.FunctionDef) and n.name == name)
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, path, "exec"), ns)
    return ns[name], (fn.lineno, fn.end_lineno)


def reference_parameters(ref_root, nr_robots, n_obst_per_link):
    spec = importlib.util.spec_from_file_location("ref_parameters_manipulators",
                                                  os.path.join(ref_root, "examples", "parameters_manipulators.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class _Np:      # PM:111-115 builds a ragged pos0: an object array on the numpy the reference pins, an error on numpy >= 1.24
        def __getattr__(self, k):
            return getattr(np, k)

        @staticmethod
        def array(x, *a, **kw):
            try:
                return np.array(x, *a, **kw)
            except ValueError:
                return np.array(x, dtype=object)
    mod.np = _Np()
    return mod.manipulator_parameters(nr_robots=nr_robots, n_obst_per_link=n_obst_per_link)


class StubEnv:
    """What add_collision_spheres touches of the simulator: the URDF joint map and add_collision_link."""

    def __init__(self, joint_order):
        jm = {n: None for n in joint_order}              # dicts keep insertion order: the URDF's file order
        robot = types.SimpleNamespace(_urdf_robot=types.SimpleNamespace(_joint_map=jm))
        self.env = types.SimpleNamespace(env=types.SimpleNamespace(_robots=[robot]))
        self.calls = []

    def add_collision_link(self, **kw):
        self.calls.append(kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("MRF_REFERENCE", "/root/reference"))
    ref = os.path.abspath(ap.parse_args().reference)
    urdf = os.path.join(ref, "examples", "simulation_environments", "urdfs", "panda_with_finger.urdf")
    order = [j.get("name") for j in ET.parse(urdf).getroot().findall("joint")]
    method, lines = reference_method(ref)
    out = {"source_lines": np.array(lines), "urdf_joint_order": np.array(order)}
    for K in (1, 2, 3, 4):
        for N in (2, 3):
            p = reference_parameters(ref, N, K)
            me = types.SimpleNamespace(collision_links_nrs=p.collision_links_nrs, radius_sphere=p.radius_sphere,
                                       robot_types=p.robot_types, n_obst_per_link=p.n_obst_per_link,
                                       link_transform_list=[[] for _ in range(N)])
            env = StubEnv(order)
            assert method(me, env) is env
            T = np.array([c["link_transformation"] for c in env.calls])
            assert all(c["shape_type"] == "sphere" for c in env.calls)
            key = f"n{K}_r{N}_"
            out[key + "robot"] = np.array([c["robot_index"] for c in env.calls])
            out[key + "link_index"] = np.array([c["link_index"] for c in env.calls])
            out[key + "sphere_on_link"] = np.array([c["sphere_on_link_index"] for c in env.calls])
            out[key + "offset"] = T[:, 0:3, 3].copy()
            out[key + "rotation_is_identity"] = np.array([bool(np.array_equal(t[:3, :3], np.identity(3))) for t in T])
            out[key + "size"] = np.array([c["size"][0] for c in env.calls], dtype=float)
            out[key + "link_transform_list"] = np.array(me.link_transform_list)
    path = os.path.join(HERE, "sphere_offsets.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: add_collision_spheres = lines {lines[0]}-{lines[1]} of {REL}; "
          f"{sum(v.size for v in out.values())} values")


if __name__ == "__main__":
    main()
