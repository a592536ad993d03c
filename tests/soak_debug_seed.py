"""Takes one seed of tests/soak_random_configs.py apart (script):  python3 tests/soak_debug_seed.py <seed>
Prints the drawn planner definition, the worst row / step of the rollout against the oracle, and the oracle's own
sensitivity to a 1e-13 relative perturbation of the inputs (an ill-conditioned sample shows a self-sensitivity above
the GPU-vs-oracle difference)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as oracle
import test_gpu_parity as t
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
seed = int(sys.argv[1])
# replicate the config construction by monkeypatching FabricHandle.rollout to capture
cap = {}
orig = FabricHandle.rollout
def roll(self, q, qd, prm, want_traj=False, **kw):
    out = orig(self, q, qd, prm, want_traj=want_traj, **kw)
    cap["h"] = self; cap["q"] = q; cap["qd"] = qd; cap["prm"] = prm; cap["out"] = out
    return out
FabricHandle.rollout = roll
try:
    t.test_random_planner_configurations(oracle, seed)
    print("passed")
except AssertionError as e:
    print("failed:", e)
h = cap["h"]; cfg = h.cfg
q, qd, prm = (x.cpu().numpy() for x in (cap["q"], cap["qd"], cap["prm"]))
want_avg, want_q, want_qd = oracle.rollout(cfg, q, qd, prm, traj=True)
avg, tq, tqd = cap["out"]
tq, tqd = tq.cpu().numpy(), tqd.cpu().numpy()
print("N", cfg.n_robots, "H", cfg.horizon, "dt", cfg.dt, "mode", cfg.mode, "dynamic", cfg.dynamic, "n_goals", cfg.n_goals, "planes", cfg.n_planes, "limits", cfg.use_limits, "n_spheres", cfg.n_spheres, "eps", cfg.eps, "kernel_select", cfg.kernel_select)
for name in ("collision_geometry", "collision_finsler", "plane_geometry", "plane_finsler", "limit_geometry", "limit_finsler"):
    f = getattr(cfg, name); print(name, f.family, f.gate, f.p, f.k, f.c, f.s)
err = np.abs(tqd - want_qd); scale = np.abs(want_qd).max()
print("scale", scale, "max err", err.max(), "rel", err.max() / scale)
k, j, r = np.unravel_index(err.argmax(), err.shape)
print("worst at step", k, "joint", j, "row", r, "got", tqd[k, j, r], "want", want_qd[k, j, r])
for kk in range(cfg.horizon):
    print(" step", kk, "row err", np.abs(tqd[kk, :, r] - want_qd[kk, :, r]).max(), "|qd|", np.abs(want_qd[kk, :, r]).max())
# per-row max err
rowerr = err.max(axis=(0, 1)); print("rows with err>1e-9*scale:", np.nonzero(rowerr > 1e-9 * scale)[0], rowerr.max())
print("q of row", q[:, r], "limits lo", [cfg.limits[j][0] for j in range(7)], "hi", [cfg.limits[j][1] for j in range(7)])
# conditioning: the oracle against itself on inputs perturbed by 1e-13 (relative)
rng = np.random.default_rng(0)
q2 = q * (1 + 1e-13 * rng.standard_normal(q.shape)); qd2 = qd * (1 + 1e-13 * rng.standard_normal(qd.shape))
_, _, w2 = oracle.rollout(cfg, q2, qd2, prm, traj=True)
for kk in range(cfg.horizon):
    print(" oracle self-sensitivity step", kk, np.abs(w2[kk][:, r] - want_qd[kk][:, r]).max() / np.abs(want_qd[kk][:, r]).max(),
          " gpu vs oracle", np.abs(tqd[kk, :, r] - want_qd[kk, :, r]).max() / np.abs(want_qd[kk][:, r]).max())
sc = r // cfg.n_robots
print("scenario rows", [sc * cfg.n_robots + i for i in range(cfg.n_robots)])
for i in range(cfg.n_robots):
    rr = sc * cfg.n_robots + i
    print(" robot", i, "|qd0|", np.abs(qd[:, rr]).max(), "|qd step0|", np.abs(want_qd[0][:, rr]).max())
