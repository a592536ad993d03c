#!/usr/bin/env python3
"""Round-off growth of the coupled rollout along the horizon, kernel vs float64 oracle: BASELINE config 5 (8 Pandas, 20
spheres each, H = 50) and config 4 (3 Pandas, H = 30), 10 scenarios.  Prints, per 10 steps, the relative error of the
joint velocities / positions of that step (max over rows / max magnitude) -- the measurement behind the tolerance of
tests/test_gpu_parity.py::test_baseline_config_5_eight_pandas_h50.
usage: python3 tools/c5_error_growth.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from multi_robot_fabrics_amd import config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

out = {}
for name, N, H, kw in (("C5", 8, 50, dict(x_min=0.3, q_spread=0.15)), ("C4", 3, 30, {})):
    for kernel in (1, 2):
        cfg = config.panda_config(n_robots=N, horizon=H)
        cfg.kernel_select = kernel
        if name == "C5":
            links, offs = config.c5_sphere_table()
            config.set_spheres(cfg, links, offs)
        cfg.goal_estimate_mask = ((1 << N) - 1) & ~1
        batch = scenarios.panda_batch(cfg, 10, seed=91, **kw)
        _, wq, wqd = oracle_lib.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
        h = FabricHandle(cfg, 0)
        _, tq, tqd = h.rollout(h.tensor(batch["q"]), h.tensor(batch["qdot"]), h.tensor(batch["params"]), want_traj=True)
        tq, tqd = tq.cpu().numpy(), tqd.cpu().numpy()
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
        steps = [k for k in range(H) if k % 10 == 9 or k == 0]
        out[f"{name}_kernel{kernel}"] = {"qdot_rel_err_at_step": {str(k + 1): rel(tqd[k], wqd[k]) for k in steps},
                                         "q_rel_err_at_step": {str(k + 1): rel(tq[k], wq[k]) for k in steps},
                                         "qdot_whole_trajectory": rel(tqd, wqd), "q_whole_trajectory": rel(tq, wq)}
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c5_error_growth.json"), "w"), indent=1)
