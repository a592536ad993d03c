"""Shared by tests/test_reference_pin.py and tests/reconcile_constants.py: evaluates a backend (the CPU oracle, or the
HIP kernels through the C ABI) on the inputs of the committed golden files, in the shape of
tests/golden/reference_*.npz (written by tests/golden/make_reference_golden.py from the REAL reference stack)."""
import os

import numpy as np

from multi_robot_fabrics_amd import abi, config
from test_oracle_golden import params_row

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KINDS = ("panda_actions", "planar_actions", "panda_rollout", "panda_rollout_c4", "panda_cartesian")
FILES = {k: os.path.join(GOLD, f"reference_{k}.npz") for k in KINDS}
HOW = ("parity with the CasADi path is UNPINNED: tests/golden/reference_*.npz are absent.  Generate them where the "
       "reference runs (python 3.9 + tests/golden/reference_requirements.txt): python tests/golden/make_reference_golden.py")


def have(kind):
    return os.path.exists(FILES[kind])


def apply_constants(cfg, constants):
    for k, v in (constants or {}).items():
        if k == "strings":
            config.set_strings(cfg, **v)
        else:
            setattr(cfg, k, v)
    return cfg


def panda_action_cases(constants=None):
    """-> list of (kind, cfg, q[7,1], qd, prm[29,1], ox[M,3,1], ov, oa, orad[M,1], n_static)."""
    g = np.load(os.path.join(GOLD, "panda_actions.npz"))
    out = []
    for i, kind in enumerate(g["kinds"]):
        kind = str(kind)
        cfg = config.panda_config(n_robots=1, horizon=1, mounts=[g["mount"][i]], n_ego=0 if kind == "grasp" else 6)
        if kind == "nogoal":
            cfg.n_goals = 0
        apply_constants(cfg, constants)
        M = g["ox"].shape[1]
        prm = params_row(g["g0"][i], g["rb"][i])
        out.append((kind, cfg, g["q"][i][:, None], g["qd"][i][:, None], prm[:, None], g["ox"][i][:, :, None],
                    g["ov"][i][:, :, None], g["oa"][i][:, :, None], g["orad"][i][:, None], M if kind == "static" else 0))
    return out


def planar_action_cases(constants=None):
    g = np.load(os.path.join(GOLD, "planar_actions.npz"))
    out = []
    for i in range(len(g["dyn"])):
        cfg = config.planar3_config(n_robots=1, obst_dim=2 if g["dyn"][i] else 3)
        apply_constants(cfg, constants)
        prm = np.zeros(abi.NPARAM)
        prm[0:2] = g["g0"][i]
        prm[abi.P_WEIGHT_GOAL_0] = 1.0
        prm[abi.P_RADIUS_BODY] = 0.2
        ns = int(g["n_static"][i]) if g["dyn"][i] else g["ox"].shape[1]
        out.append(("planar", cfg, g["q"][i][:, None], g["qd"][i][:, None], prm[:, None], g["ox"][i][:, :, None],
                    g["ov"][i][:, :, None], g["oa"][i][:, :, None], g["orad"][i][:, None], ns))
    return out


def rollout_cases(constants=None):
    """-> list of (name, cfg, q0[7,2], qd0, prm[29,2])."""
    g = np.load(os.path.join(GOLD, "panda_rollout.npz"))
    out = []
    for name, dynamic in (("dyn", 1), ("stat", 0)):
        cfg = config.panda_config(n_robots=2, horizon=g[f"{name}_q"].shape[1], dynamic=dynamic, mounts=list(g["mounts"]))
        apply_constants(cfg, constants)
        prm = np.stack([params_row(g[f"{name}_g0"][i], [0.08] * 6) for i in range(2)], axis=1)
        out.append((name, cfg, g[f"{name}_q0"].T.copy(), g[f"{name}_qd0"].T.copy(), prm))
    return out


def c4_case(constants=None):
    """BASELINE config 4 (3-Panda RF-CV H=30, panda_rollout_c4.npz) -> (cfg, q0[7,3], qd0, prm[29,3]).  Robot 1's goal
    is estimated on the device / in the oracle (goal_estimate_mask = 0b010); the parameter row carries the start goal."""
    g = np.load(os.path.join(GOLD, "panda_rollout_c4.npz"))
    cfg = config.panda_config(n_robots=3, horizon=int(g["horizon"]), dynamic=1, mounts=list(g["mounts"]))
    cfg.goal_estimate_mask = int(g["estimate_mask"])
    apply_constants(cfg, constants)
    prm = np.stack([params_row(g["g0"][i], [0.08] * 6) for i in range(3)], axis=1)
    return cfg, g["q0"].T.copy(), g["qd0"].T.copy(), prm


def cartesian_cases(constants=None):
    """FabricsRollouts cases of panda_cartesian.npz -> list of (name, cfg, q0[7,1], qd0, prm[29,1], ox[8,3,1], ov, orad[8,1]).
    The obstacle accelerations are zero (FPC:33): the mirrored class passes obst_a = NULL."""
    g = np.load(os.path.join(GOLD, "panda_cartesian.npz"))
    out = []
    for i in range(2):
        cfg = config.panda_config(n_robots=1, horizon=int(g["horizon"]), dynamic=1, mounts=[g["mounts"][i]])
        apply_constants(cfg, constants)
        prm = params_row(g[f"r{i}_goal"], [0.08] * 6)
        out.append((f"r{i}", cfg, g["q0"][i][:, None].copy(), g["qd0"][i][:, None].copy(), prm[:, None],
                    g[f"r{i}_ox"][:, :, None].copy(), g[f"r{i}_ov"][:, :, None].copy(), np.full((8, 1), 0.08)))
    return out


def oracle_c4(oracle, case):
    cfg, q0, qd0, prm = case
    avg, tq, tqd = oracle.rollout(cfg, q0, qd0, prm, traj=True)
    return {"avg": avg, "q_last": tq[-1].T, "qd_last": tqd[-1].T}


def oracle_cartesian(oracle, cases):
    out = {}
    for name, cfg, q0, qd0, prm, ox, ov, orad in cases:
        avg, tq, tqd = oracle.rollout_cartesian(cfg, q0, qd0, prm, ox, ov, np.zeros_like(ox), orad, traj=True)
        out[name + "_q"], out[name + "_qd"], out[name + "_avg"] = tq[:, :, 0], tqd[:, :, 0], avg[0]
    return out


def oracle_actions(oracle, cases):
    return np.stack([oracle.compute_action(c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], n_static=c[9])[1][:, 0]
                     for c in cases])


def oracle_rollouts(oracle, cases):
    out = {}
    for name, cfg, q0, qd0, prm in cases:
        avg, tq, tqd = oracle.rollout(cfg, q0, qd0, prm, traj=True)
        out[name + "_q"], out[name + "_qd"], out[name + "_avg"] = tq.transpose(2, 0, 1), tqd.transpose(2, 0, 1), avg
    return out
