#!/usr/bin/env python3
"""Soak of mrf_rollout_cartesian_coupled beyond the collected tests: random robot counts, sphere tables, horizons, static /
dynamic fabrics and modes; the cooperative (one wave per scenario) and the row-per-lane forms against each other and,
every fifth draw, against the float64 oracle.
usage: python3 tests/soak_cartesian_coupled.py [n_draws] [first_seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

n, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 60), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"coop_vs_row": 0.0, "vs_oracle": 0.0}
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    N, per_link, H = int(rng.integers(2, 5)), int(rng.choice([0, 1, 2, 4])), int(rng.integers(2, 13))
    cfg = config.panda_config(n_robots=N, horizon=H, dynamic=int(rng.integers(0, 2)))
    cfg.mode = abi.MODE_VEL if rng.random() < 0.7 else abi.MODE_ACC
    if per_link:
        links, offs = config.sphere_offsets_per_link(per_link)
        config.set_spheres(cfg, links, offs, [0.06 + 0.002 * (s % 7) for s in range(len(links))])
    B = int(rng.integers(1, 40))
    batch = scenarios.panda_batch(cfg, B, seed=seed, x_min=0.1)
    out = []
    for kernel in (1, 2):
        c = cfg.copy()
        c.kernel_select = kernel
        h = FabricHandle(c, 0)
        avg, tq, tqd = h.rollout_cartesian_coupled(*(h.tensor(batch[k]) for k in ("q", "qdot", "params")), want_traj=True)
        out.append((avg.cpu().numpy(), tq.cpu().numpy(), tqd.cpu().numpy()))
    rel = lambda a, b: float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
    e = max(rel(a, b) for a, b in zip(out[1], out[0]))
    worst["coop_vs_row"] = max(worst["coop_vs_row"], e)
    line = f"seed {seed}: N={N} S={cfg.n_spheres} H={H} dyn={cfg.dynamic} mode={'vel' if cfg.mode == abi.MODE_VEL else 'acc'} B={B} coop-vs-row {e:.2e}"
    if seed % 5 == 0:
        sx, sv, _ = oracle_lib.fk_spheres(cfg, batch["q"], batch["qdot"])
        o = scenarios.other_robot_obstacles(cfg, batch, sx, sv if cfg.dynamic else None, None)
        want = oracle_lib.rollout_cartesian(cfg, batch["q"], batch["qdot"], batch["params"], *o, traj=True)
        eo = max(rel(a, b) for a, b in zip(out[0], want))
        worst["vs_oracle"] = max(worst["vs_oracle"], eo)
        line += f"  row-vs-oracle {eo:.2e}"
    print(line, flush=True)
print("worst:", worst)
assert worst["coop_vs_row"] < 1e-9 and worst["vs_oracle"] < 1e-9
