#!/usr/bin/env python3
"""Reference-side pin of the fabric solve: evaluates the REAL reference stack on the seeded inputs of the committed
golden files and writes the outputs next to them.

    python tests/golden/make_reference_golden.py [--reference /path/to/multi-robot-fabrics]

Needs an environment in which the reference itself runs (its pyproject.toml pins: python >=3.8,<3.10,
fabrics==0.9.5, forwardkinematics==1.2.3, casadi==3.5.5, mpscenes, quaternionic, matplotlib).  None of those is
importable in the build container, so this script has never run there: it is the one command that turns
"parity unpinned" into "pinned" for anyone who has the wheels.  It reads INPUTS from
    tests/golden/panda_actions.npz, planar_actions.npz, panda_rollout.npz        (made by make_golden.py)
and writes data only (no reference source) to
    tests/golden/reference_panda_actions.npz     action [cases,7]        planner.compute_action(**kwargs)
    tests/golden/reference_planar_actions.npz    action [cases,3]
    tests/golden/reference_panda_rollout.npz     {dyn,stat}_{q,qd}[2,H,7], {dyn,stat}_avg[2]
    tests/golden/reference_panda_rollout_c4.npz  avg[3] (+ q_last, qd_last [3,7])   BASELINE config 4: 3-Panda RF-CV H=30
    tests/golden/reference_panda_cartesian.npz   r{0,1}_{q,qd}[H,7], r{0,1}_avg      FabricsRollouts (Cartesian variant)
As soon as those files exist, tests/test_reference_pin.py (CPU: the oracle; -m gpu: the HIP kernels) compares against
them instead of skipping, and tests/reconcile_constants.py fits the recalled constants of mrf_config to them.

The planners are built call by call as the reference's drivers build them:
    Panda        examples/example_pandas_Jointspace.py:64-134  (set_planner_panda), kwargs of :421-439
    rollouts     examples/example_pandas_Jointspace.py:172-193 (define_rollout_planners), :354-375 (inputs_action),
                 multi_robot_fabrics/fabrics_planner/forward_planner_Jointspace.py:298-423
    point robot  examples/example_pointmasses_static.py:102-129 (set_planner_point), kwargs of :191-199
    config 4     as "rollouts" with parameters_manipulators(nr_robots=3), N_HORIZON=30 and robot 1's goal replaced by the
                 RF-CV estimate x_ee + 20*0.01*v_ee (EXJ:346-348) with v_ee = J qdot from
                 UtilsKinematics.define_symbolic_endeffector / compute_endeffector (utils.py:120-136, utils_apply_fk.py:35-43)
    Cartesian    examples/example_pandas_cartesian.py:160-192 (define_rollout_planners): FabricsRollouts +
                 symbolic_forward_fabrics + preset_radii_obsts_dyn + define_arguments_numerical + rollouts_numerical /
                 get_velocity_rollouts (forward_planner_Cartesian.py:347-563)
The only liberty: the mount transform T_0 and the obstacle counts come from the golden case instead of from
parameters_manipulators (set_mount_transformation / set_components take them as plain arguments anyway).
"""
import argparse
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = HERE            # --out: where reference_*.npz are written (the inputs are always read from HERE)


def need(mod):
    try:
        return __import__(mod)
    except Exception as e:  # noqa: BLE001
        sys.exit("cannot import %r (%s).\nThis script needs the reference's own environment: python<3.10 with "
                 "fabrics==0.9.5 forwardkinematics==1.2.3 casadi==3.5.5 mpscenes quaternionic matplotlib." % (mod, e))


def panda_goal(GoalComposition, n_goals=3):
    """create_dummy_goal_panda (EXJ:25-62); the values are placeholders, the runtime kwargs carry the real ones."""
    goal_dict = {
        "subgoal0": {"weight": 2.0, "is_primary_goal": True, "indices": [0, 1, 2], "parent_link": "world",
                     "child_link": "panda_hand", "desired_position": [0.1, 0.6, 0.8], "epsilon": 0.05,
                     "type": "staticSubGoal"},
        "subgoal1": {"weight": 10.0, "is_primary_goal": False, "indices": [0, 1, 2], "parent_link": "panda_link7",
                     "child_link": "panda_hand", "desired_position": [0.107, 0.0, 0.0],
                     "angle": [-0.366, 0.0, 0.0, 0.3305], "epsilon": 0.05, "type": "staticSubGoal"},
        "subgoal2": {"weight": 1.0, "is_primary_goal": False, "indices": [6], "desired_position": [math.pi / 4],
                     "epsilon": 0.05, "type": "staticJointSpaceSubGoal"},
    }
    keep = ["subgoal0", "subgoal1", "subgoal2"][:n_goals]
    return GoalComposition(name="goal", content_dict={k: goal_dict[k] for k in keep})


def set_planner_panda(mods, urdf, T_0, nr_obst, nr_obst_dyn, collision_links_nr, with_goal=True):
    """EXJ:64-134 with the mount transform passed in."""
    fk = mods["GenericURDFFk"](urdf, "panda_link0", "panda_leftfinger")
    planner = mods["ParameterizedFabricPlanner"](
        7, fk,
        geometry_plane_constraint="10*(1/(1+1*ca.exp(-10*x))-1) * (xdot**2)",
        collision_geometry="-0.5 / (x ** 4) * (xdot ** 2)",
        collision_finsler="0.01/(x**4) * xdot**2",
    )
    collision_links = ["panda_link%d" % l if l < 9 else "panda_hand" for l in collision_links_nr]
    panda_limits = [[-2.8973, 2.8973], [-1.7628, 1.7628], [-2.8973, 2.8973], [-3.0718, -0.0698], [-2.8973, 2.8973],
                    [-0.0175, 3.7525], [-2.8973, 2.8973]]
    planner._forward_kinematics.set_mount_transformation(np.array(T_0))
    goal = panda_goal(mods["GoalComposition"]) if with_goal else None
    planner.set_components(collision_links=collision_links, goal=goal, number_obstacles=nr_obst,
                           number_dynamic_obstacles=nr_obst_dyn, dynamic_obstacle_dimension=3,
                           number_plane_constraints=1, limits=panda_limits)
    planner.concretize(mode="vel", time_step=0.01)
    return planner, goal


R1 = np.array([[0.0, 0.0, -1.0], [0.0, 1.0, 0.0], [1.0, 0.0, 0.0]])


def panda_actions(mods, urdf):
    g = np.load(os.path.join(HERE, "panda_actions.npz"))
    out = []
    for ci, kind in enumerate(g["kinds"]):
        kind = str(kind)
        M = g["ox"].shape[1]
        static = kind == "static"
        grasp = kind == "grasp"
        planner, _ = set_planner_panda(mods, urdf, g["mount"][ci], nr_obst=M if static else 0,
                                       nr_obst_dyn=0 if (static or grasp) else M,
                                       collision_links_nr=[] if grasp else [1, 2, 3, 4, 5, 6, 7, 8],
                                       with_goal=kind != "nogoal")
        rb = g["rb"][ci]
        kw = dict(q=g["q"][ci], qdot=g["qd"][ci], constraint_0=np.array([0.0, 0.0, 1.0, -0.65]),
                  radius_body_panda_links={str(l): np.array(rb[l - 3]) for l in range(3, 9)},
                  radius_body_panda_hand=np.array([rb[5]]))
        if kind != "nogoal":
            kw.update(x_goal_0=g["g0"][ci], weight_goal_0=2.0, angle_goal_1=R1, x_goal_1=np.array([0.107, 0.0, 0.0]),
                      weight_goal_1=20.0, x_goal_2=np.array([math.pi / 4]), weight_goal_2=1.0)
        if static:
            kw.update(x_obsts=list(g["ox"][ci]), radius_obsts=list(g["orad"][ci]))
        elif not grasp:
            kw.update(x_obsts_dynamic=list(g["ox"][ci]), xdot_obsts_dynamic=list(g["ov"][ci]),
                      xddot_obsts_dynamic=list(g["oa"][ci]), radius_obsts_dynamic=list(g["orad"][ci]))
        action = np.asarray(planner.compute_action(**kw), dtype=float).reshape(-1)
        print("panda action", ci, kind, action)
        out.append(action)
    np.savez(os.path.join(OUT, "reference_panda_actions.npz"), action=np.stack(out), kinds=g["kinds"])


def planar_actions(mods, ref_root):
    g = np.load(os.path.join(HERE, "planar_actions.npz"))
    with open(os.path.join(ref_root, "examples", "simulation_environments", "urdfs", "pointRobot1.urdf")) as f:
        urdf = f.read()
    goal = mods["GoalComposition"](name="goal", content_dict={
        "subgoal0": {"weight": 1, "is_primary_goal": True, "indices": [0, 1], "parent_link": "world",
                     "child_link": "base_link", "desired_position": [1.5, 0.99], "epsilon": 0.1, "type": "staticSubGoal"}})
    out = []
    for ci in range(len(g["q"])):
        n_static, dyn = int(g["n_static"][ci]), int(g["dyn"][ci])
        M = g["ox"].shape[1]
        fk = mods["GenericURDFFk"](urdf, "world", "base_link")
        planner = mods["ParameterizedFabricPlanner"](
            3, fk, collision_geometry="-2.0 / (x ** 1) * xdot ** 2",
            collision_finsler="1.0/(x**2) * (1 - ca.heaviside(xdot))* xdot**2")
        kw = dict(q=g["q"][ci], qdot=g["qd"][ci], x_goal_0=g["g0"][ci], weight_goal_0=1.0, radius_body_base_link=0.2)
        if not dyn:          # example_pointmasses_static.py:122-128
            planner.set_components(["base_link"], {}, goal=goal, number_obstacles=M)
            kw.update(x_obsts=list(g["ox"][ci]), radius_obsts=list(g["orad"][ci]))
        else:                # example_pointmasses_dynamic.py: static scene spheres + 2-D dynamic spheres
            ns = n_static
            planner.set_components(["base_link"], {}, goal=goal, number_obstacles=ns, number_dynamic_obstacles=M - ns,
                                   dynamic_obstacle_dimension=2)
            if ns:
                kw.update(x_obsts=list(g["ox"][ci][:ns]), radius_obsts=list(g["orad"][ci][:ns]))
            kw.update(x_obsts_dynamic=[x[:2] for x in g["ox"][ci][ns:]], xdot_obsts_dynamic=[v[:2] for v in g["ov"][ci][ns:]],
                      xddot_obsts_dynamic=[a[:2] for a in g["oa"][ci][ns:]], radius_obsts_dynamic=list(g["orad"][ci][ns:]))
        planner.concretize()
        action = np.asarray(planner.compute_action(**kw), dtype=float).reshape(-1)
        print("planar action", ci, action)
        out.append(action)
    np.savez(os.path.join(OUT, "reference_planar_actions.npz"), action=np.stack(out))


def rollouts(mods, urdf):
    """2-Panda coupled rollout H=3 through the reference's ForwardFabricsPlanner (EXJ:172-193, 354-375)."""
    import examples.parameters_manipulators as parameters_manipulators
    from multi_robot_fabrics.fabrics_planner.forward_planner_Jointspace import ForwardFabricsPlanner
    from multi_robot_fabrics.utils.utils import UtilsKinematics
    g = np.load(os.path.join(HERE, "panda_rollout.npz"))
    H = g["dyn_q"].shape[1]
    out = {}
    for name, dynamic in (("dyn", 1), ("stat", 0)):
        params = parameters_manipulators.manipulator_parameters(nr_robots=2)
        params.define_settings(ROLLOUT_FABRICS=True, ROLLOUTS_PLOTTING=True, STATIC_OR_DYN_FABRICS=dynamic,
                               RESOLVE_DEADLOCKS=True, ESTIMATE_GOAL=False, N_HORIZON=H)
        planners, goal_structs = [], []
        for i in range(2):
            p, gs = set_planner_panda(mods, urdf, params.mount_transform[i], nr_obst=params.nr_obsts[i],
                                      nr_obst_dyn=params.nr_obsts_dyn[i], collision_links_nr=params.collision_links_nrs[i])
            planners.append(p)
            goal_structs.append(gs)
        fk_dict = UtilsKinematics().define_forward_kinematics(planners=planners, collision_links_nrs=params.collision_links_nrs,
                                                              collision_links=params.collision_links)
        fwd = ForwardFabricsPlanner(params=params, planners=planners, N_steps=10, fk_dict=fk_dict,
                                    goal_struct_robots=goal_structs)
        fwd.forward_multi_fabrics_symbolic()
        inputs_action = {"q_robots": [g[name + "_q0"][i] for i in range(2)], "q_dot_robots": [g[name + "_qd0"][i] for i in range(2)],
                         "x_obsts": [[] * 2], "x_goals0": [g[name + "_g0"][i] for i in range(2)],
                         "x_goals1": [np.array([0.107, 0.0, 0.0])] * 2, "x_goals2": [np.array([math.pi / 4])] * 2,
                         "weight_goals0": [2.0, 2.0], "weight_goals1": [20.0, 20.0], "weight_goals2": [1.0, 1.0],
                         "constraints": [np.array([0, 0, 1, -params.z_table])] * 2}
        avg = fwd.get_velocity_rollouts(inputs_action=inputs_action)
        qN, qdN, _ = fwd.rollouts_numerical(inputs_action=inputs_action)
        # rollouts_numerical returns, per robot, a one-element list holding an array [7, H] (FPJ:381-391)
        out[name + "_q"] = np.stack([np.asarray(qN["robot_%d" % i][0]).reshape(7, H).T for i in range(2)])
        out[name + "_qd"] = np.stack([np.asarray(qdN["robot_%d" % i][0]).reshape(7, H).T for i in range(2)])
        out[name + "_avg"] = np.array([float(np.asarray(a).reshape(-1)[0]) for a in avg])
        print("rollout", name, out[name + "_avg"])
    np.savez(os.path.join(OUT, "reference_panda_rollout.npz"), **out)


def rollout_c4(mods, urdf):
    """BASELINE config 4 through the reference's ForwardFabricsPlanner with three robots."""
    import examples.parameters_manipulators as parameters_manipulators
    from multi_robot_fabrics.fabrics_planner.forward_planner_Jointspace import ForwardFabricsPlanner
    from multi_robot_fabrics.utils.utils import UtilsKinematics
    from multi_robot_fabrics.utils.utils_apply_fk import compute_endeffector
    g = np.load(os.path.join(HERE, "panda_rollout_c4.npz"))
    N, H = 3, int(g["horizon"])
    params = parameters_manipulators.manipulator_parameters(nr_robots=N)
    params.define_settings(ROLLOUT_FABRICS=True, ROLLOUTS_PLOTTING=False, STATIC_OR_DYN_FABRICS=1, RESOLVE_DEADLOCKS=True,
                           ESTIMATE_GOAL=True, N_HORIZON=H)
    assert np.allclose(np.array(params.mount_transform, dtype=float), g["mounts"]), "fixture mounts != parameters_manipulators"
    planners, goal_structs = [], []
    for i in range(N):
        p, gs = set_planner_panda(mods, urdf, params.mount_transform[i], nr_obst=params.nr_obsts[i],
                                  nr_obst_dyn=params.nr_obsts_dyn[i], collision_links_nr=params.collision_links_nrs[i])
        planners.append(p)
        goal_structs.append(gs)
    kin = UtilsKinematics()
    fk_dict = kin.define_forward_kinematics(planners=planners, collision_links_nrs=params.collision_links_nrs,
                                            collision_links=params.collision_links)
    fk_endeff = kin.define_symbolic_endeffector(planners)
    q0, qd0 = [g["q0"][i] for i in range(N)], [g["qd0"][i] for i in range(N)]
    x_ee, v_ee = compute_endeffector(q0, qd0, fk_endeff, nr_robots=N)
    goals = [g["g0"][i] for i in range(N)]
    goals[1] = x_ee[1] + 20 * 0.01 * v_ee[1]                                     # EXJ:346-348 / EXC:355-357
    print("estimated goal of robot 1:", goals[1], "(fixture:", g["estimated_goal_1"], ")")
    fwd = ForwardFabricsPlanner(params=params, planners=planners, N_steps=10, fk_dict=fk_dict, goal_struct_robots=goal_structs)
    fwd.forward_multi_fabrics_symbolic()
    inputs_action = {"q_robots": q0, "q_dot_robots": qd0, "x_obsts": [[] * N], "x_goals0": goals,
                     "x_goals1": [np.array([0.107, 0.0, 0.0])] * N, "x_goals2": [np.array([math.pi / 4])] * N,
                     "weight_goals0": [2.0] * N, "weight_goals1": [20.0] * N, "weight_goals2": [1.0] * N,
                     "constraints": [np.array([0, 0, 1, -params.z_table])] * N}
    avg = fwd.get_velocity_rollouts(inputs_action=inputs_action)
    out = {"avg": np.array([float(np.asarray(a).reshape(-1)[0]) for a in avg]), "estimated_goal_1": np.asarray(goals[1], dtype=float)}
    print("rollout C4 avg", out["avg"])
    try:
        # rollouts_numerical is written for two robots (its result dictionaries hold robot_0 / robot_1 only, FPJ:346-348):
        # the last state of all three comes from the per-robot functions it wraps, fed the same positional list
        args = []
        for i in range(N):
            args.append(fwd.rotation_matrices_pandas[i])
            args += [inputs_action["constraints"][i]] * fwd.nr_constraints[i]
            args += [q0[i], qd0[i]]
            args += [inputs_action["weight_goals%d" % s][i] for s in range(fwd.nr_subgoals[i])]
            args += [inputs_action["x_goals%d" % s][i] for s in range(fwd.nr_subgoals[i])]
            args += list(fwd.r_robots_args[i])
            args += list(fwd.r_dyns_obsts[0 if i else 1])
        out["q_last"] = np.stack([np.asarray(list(fwd.q_N_fun[i](*args))[-1].full()).reshape(-1) for i in range(N)])
        out["qd_last"] = np.stack([np.asarray(list(fwd.q_dot_N_fun[i](*args))[-1].full()).reshape(-1) for i in range(N)])
    except Exception as e:  # noqa: BLE001
        print("last state not recorded (%s); the average velocities pin the rollout" % e)
    np.savez(os.path.join(OUT, "reference_panda_rollout_c4.npz"), **out)


def cartesian(mods, urdf):
    """FabricsRollouts exactly as example_pandas_cartesian.py:160-192 builds it, on the inputs of panda_cartesian.npz."""
    import examples.parameters_manipulators as parameters_manipulators
    from multi_robot_fabrics.fabrics_planner.forward_planner_Cartesian import FabricsRollouts
    g = np.load(os.path.join(HERE, "panda_cartesian.npz"))
    H = int(g["horizon"])
    params = parameters_manipulators.manipulator_parameters(nr_robots=2, n_obst_per_link=1)
    params.define_settings(ROLLOUT_FABRICS=True, ROLLOUTS_PLOTTING=True, STATIC_OR_DYN_FABRICS=1, RESOLVE_DEADLOCKS=True,
                           ESTIMATE_GOAL=True, N_HORIZON=H, n_obst_per_link=1)
    assert np.allclose(np.array(params.mount_transform, dtype=float), g["mounts"])
    out = {}
    for i in range(2):
        planner, goal_struct = set_planner_panda(mods, urdf, params.mount_transform[i], nr_obst=0,
                                                 nr_obst_dyn=params.nr_obsts_dyn_all[i],
                                                 collision_links_nr=params.collision_links_nrs[i])
        v_obsts_dyn = [np.zeros((3,))] * params.nr_obsts_dyn_all[0]
        fp = FabricsRollouts(N=H, dt=params.dt, nx=params.dof[i] * 2, nu=params.dof[i], dof=params.dof[i],
                             nr_obsts=params.nr_obsts[i], bool_ring=False, nr_obsts_dyn=params.nr_obsts_dyn_all[i],
                             v_obsts_dyn=v_obsts_dyn, fabrics_mode=params.fabrics_mode,
                             collision_links_nrs=params.collision_links_nrs[i], nr_constraints=params.nr_constraints[i],
                             radius_sphere=params.radius_sphere, constraints=params.constraints[i],
                             nr_goals=len(goal_struct._config))
        fp.symbolic_forward_fabrics(planner=planner, goal_struct=goal_struct)
        fp.preset_radii_obsts_dyn(radii_obst_dyn=params.r_dyns_obsts[i])
        weight_goals = {"subgoal0": 2.0, "subgoal1": 20.0, "subgoal2": 1.0}
        x_goals = {"subgoal0": g["r%d_goal" % i], "subgoal1": np.array([0.107, 0.0, 0.0]), "subgoal2": np.array([math.pi / 4])}
        arguments = fp.define_arguments_numerical(q_robot=g["q0"][i], q_dot_robot=g["qd0"][i], constraints=params.constraints[i],
                                                  weight_goals=weight_goals, x_goals=x_goals, x_obsts=[],
                                                  x_obsts_dyn=list(g["r%d_ox" % i]), v_obsts_dyn=list(g["r%d_ov" % i]))
        q_N, q_dot_N, _ = fp.rollouts_numerical(arguments)                       # each [7, H] (FPC:538-559)
        avg = fp.get_velocity_rollouts(arguments)
        out["r%d_q" % i] = np.asarray(q_N, dtype=float).reshape(7, H).T
        out["r%d_qd" % i] = np.asarray(q_dot_N, dtype=float).reshape(7, H).T
        out["r%d_avg" % i] = np.array(float(np.asarray(avg.full() if hasattr(avg, "full") else avg).reshape(-1)[0]))
        print("cartesian rollout robot", i, out["r%d_avg" % i])
    np.savez(os.path.join(OUT, "reference_panda_cartesian.npz"), **out)


def mirrors_as_reference_modules(repo_root):
    """--dry-run-with-mirrors: this build's mirror classes under the reference's import names, so that the recipe below runs
    end to end on the GPU WITHOUT the reference stack.  It proves the recipe's plumbing (call sequence, keyword names,
    result shapes, file keys) -- not parity: the files it writes come from this build and must never be committed as
    reference_*.npz (main() refuses to write them next to the fixtures)."""
    import types
    sys.path.insert(0, repo_root)
    from multi_robot_fabrics_amd import goals, kinematics, planner, rollouts

    def module(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    module("casadi")
    for pkg in ("fabrics", "fabrics.planner", "forwardkinematics", "forwardkinematics.urdfFks", "mpscenes", "mpscenes.goals",
                "multi_robot_fabrics", "multi_robot_fabrics.fabrics_planner", "multi_robot_fabrics.utils"):
        module(pkg, __path__=[])
    module("fabrics.planner.parameterized_planner", ParameterizedFabricPlanner=planner.ParameterizedFabricPlanner)
    module("forwardkinematics.urdfFks.generic_urdf_fk", GenericURDFFk=kinematics.GenericURDFFk)
    module("mpscenes.goals.goal_composition", GoalComposition=goals.GoalComposition)
    module("multi_robot_fabrics.fabrics_planner.forward_planner_Jointspace", ForwardFabricsPlanner=rollouts.ForwardFabricsPlanner)
    module("multi_robot_fabrics.fabrics_planner.forward_planner_Cartesian", FabricsRollouts=rollouts.FabricsRollouts)
    module("multi_robot_fabrics.utils.utils", UtilsKinematics=kinematics.UtilsKinematics)
    module("multi_robot_fabrics.utils.utils_apply_fk", compute_endeffector=kinematics.compute_endeffector,
           compute_x_obsts_dyn_0=kinematics.compute_x_obsts_dyn_0)


PATH_FILES = ("multi_robot_fabrics/fabrics_planner/forward_planner_Jointspace.py",
              "multi_robot_fabrics/fabrics_planner/forward_planner_Cartesian.py", "multi_robot_fabrics/utils/utils.py",
              "multi_robot_fabrics/utils/utils_apply_fk.py", "examples/example_pandas_Jointspace.py",
              "examples/example_pandas_cartesian.py", "examples/parameters_manipulators.py",
              "examples/simulation_environments/urdfs/panda_with_finger.urdf", "poetry.lock")
# sha256 of those files in the checkout SURVEY.md describes (/root/reference of the build container): the vectors are only
# a pin of THAT reference if the checkout they were made from holds the same files
EXPECTED_SHA256 = {
    "multi_robot_fabrics/fabrics_planner/forward_planner_Jointspace.py":
        "11ac8abe21043bd566a436428fe55db62c85b1847e0f0912fca9e9eee94210ac",
    "multi_robot_fabrics/fabrics_planner/forward_planner_Cartesian.py":
        "73f1dc3fabc3c97a496e84da4e861c67f6eaae87d91099592cf2df533878a3d1",
    "multi_robot_fabrics/utils/utils.py":
        "133b282b0cca057add7aa7c84afdda6a514c03af1df4b04bdd174a44598ff733",
    "multi_robot_fabrics/utils/utils_apply_fk.py":
        "3cef85a7adc8d3bae221443552f23e161ceea78106f33cc923ade277e4ed950c",
    "examples/example_pandas_Jointspace.py":
        "3f3d80a6c00e876d997a2e4e6aff34657b9a46b2fedca018ae14cc28633e87a6",
    "examples/example_pandas_cartesian.py":
        "d5d28244832c349bd23fa5b7ef439bfe5287387b0f0c5fe7609dd4db71bd32ad",
    "examples/parameters_manipulators.py":
        "db00c45434450fadc74bf1304fdba6fc73dfc4ece969b832572ce9ab3efd5277",
    "examples/simulation_environments/urdfs/panda_with_finger.urdf":
        "8c76433a527f530e6d4bed507d8c12d75272a1096f5b434825cd511ce99d6376",
    "poetry.lock":
        "19c578057dfe5b04eabd3819b20efad84e4a49e3878168fc0466594b00d17b45",
}


def provenance(ref_root, dry_run):
    """Which reference the vectors came from: git HEAD of the checkout (if it is one), the lock's content-hash next to the
    one reference_requirements.txt was generated from, and the sha256 of every reference file on the path.  A checkout whose
    files differ from the surveyed ones is refused (the hash-pinned environment and SURVEY's file:line citations belong to
    one revision)."""
    import hashlib
    import subprocess
    out = {"git_head": "unknown", "files": [], "sha256": []}
    try:
        out["git_head"] = subprocess.run(["git", "-C", ref_root, "rev-parse", "HEAD"], capture_output=True, text=True,
                                         timeout=20).stdout.strip() or "unknown"
    except Exception:       # noqa: BLE001
        pass
    bad = []
    for rel in PATH_FILES:
        path = os.path.join(ref_root, rel)
        h = hashlib.sha256(open(path, "rb").read()).hexdigest() if os.path.exists(path) else "absent"
        out["files"].append(rel)
        out["sha256"].append(h)
        if not dry_run and EXPECTED_SHA256.get(rel) != h:
            bad.append(rel)
    lock_hash = req_hash = "unknown"
    try:
        import tomli
        with open(os.path.join(ref_root, "poetry.lock"), "rb") as f:
            lock_hash = tomli.load(f)["metadata"]["content-hash"]
    except Exception:       # noqa: BLE001
        pass
    with open(os.path.join(HERE, "reference_requirements.txt")) as f:
        for line in f:
            if "lock content-hash" in line:
                req_hash = line.split("lock content-hash")[1].split(";")[0].strip()
    out["lock_content_hash"], out["requirements_lock_content_hash"] = lock_hash, req_hash
    if not dry_run and (bad or lock_hash != req_hash):
        sys.exit("the reference checkout at %s is not the surveyed revision: %s differ%s -- check out the commit whose files "
                 "match tests/golden/make_reference_golden.py:EXPECTED_SHA256" %
                 (ref_root, bad or "no file", "" if lock_hash == req_hash else "; poetry.lock content-hash %s != %s" % (lock_hash, req_hash)))
    return {k: np.array(v) for k, v in out.items()}


def main():
    global OUT
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("MRF_REFERENCE", "/root/reference"),
                    help="checkout of tud-amr/multi-robot-fabrics (for its URDFs and its rollout classes)")
    ap.add_argument("--out", default=HERE, help="directory for reference_*.npz (default: next to the fixtures)")
    ap.add_argument("--dry-run-with-mirrors", action="store_true",
                    help="run the recipe on this build's mirror classes (needs a GPU and --out elsewhere); plumbing check only")
    args = ap.parse_args()
    OUT = os.path.abspath(args.out)
    os.makedirs(OUT, exist_ok=True)
    if args.dry_run_with_mirrors:
        if os.path.samefile(OUT, HERE):
            sys.exit("--dry-run-with-mirrors writes this build's OWN outputs: give --out a scratch directory")
        repo_root = os.path.dirname(os.path.dirname(HERE))
        args.reference = repo_root               # examples/simulation_environments/urdfs/*.urdf, examples.parameters_manipulators
        mirrors_as_reference_modules(repo_root)
    ref_root = os.path.abspath(args.reference)
    urdf_path = os.path.join(ref_root, "examples", "simulation_environments", "urdfs", "panda_with_finger.urdf")
    if not os.path.exists(urdf_path):
        sys.exit("reference checkout not found at %s (--reference)" % ref_root)
    need("casadi")
    need("fabrics")
    need("forwardkinematics")
    need("mpscenes")
    sys.path.insert(0, ref_root)
    from fabrics.planner.parameterized_planner import ParameterizedFabricPlanner
    from forwardkinematics.urdfFks.generic_urdf_fk import GenericURDFFk
    from mpscenes.goals.goal_composition import GoalComposition
    mods = dict(ParameterizedFabricPlanner=ParameterizedFabricPlanner, GenericURDFFk=GenericURDFFk,
                GoalComposition=GoalComposition)
    with open(urdf_path) as f:
        urdf = f.read()
    np.savez(os.path.join(OUT, "reference_provenance.npz"), **provenance(ref_root, args.dry_run_with_mirrors))
    panda_actions(mods, urdf)
    planar_actions(mods, ref_root)
    rollouts(mods, urdf)
    rollout_c4(mods, urdf)
    cartesian(mods, urdf)
    print("wrote tests/golden/reference_{panda_actions,planar_actions,panda_rollout,panda_rollout_c4,panda_cartesian}.npz -- "
          "now run: python -m pytest tests/test_reference_pin.py   and   python tests/reconcile_constants.py")


if __name__ == "__main__":
    main()
