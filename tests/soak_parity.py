"""Randomised parity soak (script, not collected by pytest): python3 tests/soak_parity.py [n_seeds]
Many seeds of wide-spread scenarios -- including near-contact and penetrating sphere pairs (x < 0) -- through the
rollout and the coupled action, f64, against the oracle.  Prints the worst relative errors."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import oracle_lib as oracle
    from multi_robot_fabrics_amd import abi, config, scenarios
    from multi_robot_fabrics_amd.runtime import FabricHandle
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    worst = {"rollout": 0.0, "action": 0.0, "rollout_penetrating": 0.0}
    nonfinite = 0
    for seed in range(n_seeds):
        rng = np.random.default_rng(1000 + seed)
        N = int(rng.integers(2, 5))
        cfg = config.panda_config(n_robots=N, horizon=int(rng.integers(2, 9)), dynamic=int(rng.integers(0, 2)))
        cfg.goal_estimate_mask = int(rng.integers(0, 1 << N))
        cfg.kernel_select = int(rng.integers(0, 3))
        B = int(rng.integers(1, 60))
        lim = np.array(config.PANDA_LIMITS)
        p0 = scenarios.pos0(N)
        batch = scenarios.panda_batch(cfg, B, seed=seed, x_min=0.2)
        spread = rng.choice([0.3, 0.8, 1.5])
        q = np.clip(p0[None] + rng.uniform(-spread, spread, (B, N, 7)), lim[:, 0] + 0.05, lim[:, 1] - 0.05)
        xm = scenarios.min_barrier_coordinate(cfg, q)
        batch["q"] = np.ascontiguousarray(q.reshape(-1, 7).T)
        h = FabricHandle(cfg, 0)
        qt, qdt, pt = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
        want_avg, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
        avg, tq, tqd = h.rollout(qt, qdt, pt, want_traj=True)
        got = tqd.cpu().numpy()
        ok = np.isfinite(want_qd).all(axis=(0, 1)) & np.isfinite(got).all(axis=(0, 1))
        nonfinite += int((~ok).sum())
        rowx = np.repeat(xm, N)
        for name, sel in (("rollout", ok & (rowx >= 0.02)), ("rollout_penetrating", ok & (rowx < 0.02))):
            if sel.any():
                e = np.abs(got[:, :, sel] - want_qd[:, :, sel]).max() / max(1e-300, np.abs(want_qd[:, :, sel]).max())
                worst[name] = max(worst[name], float(e))
        sx, sv, sa = oracle.fk_spheres(cfg, batch["q"], batch["qdot"])
        ox, ov, oa, orad = scenarios.other_robot_obstacles(cfg, batch, sx, sv if cfg.dynamic else None, None)
        _, want_act = oracle.compute_action(cfg, batch["q"], batch["qdot"], batch["params"], ox, ov, oa, orad)
        act = h.compute_action_coupled(qt, qdt, pt).cpu().numpy()
        sel = np.isfinite(want_act).all(0) & np.isfinite(act).all(0) & (rowx >= 0.02)
        if sel.any():
            worst["action"] = max(worst["action"], float(np.abs(act[:, sel] - want_act[:, sel]).max() / np.abs(want_act[:, sel]).max()))
    print({"seeds": n_seeds, "worst_rel_err": worst, "rows_with_nonfinite_results": nonfinite})


if __name__ == "__main__":
    main()
