"""Post-mortem of tests/soak_parity.py (script):  python3 tests/soak_parity_debug.py <first_seed> <n_seeds> [threshold]
Re-draws the seeds, finds those whose "clear" rows (sphere / plane barrier coordinates >= 0.05 over the whole oracle
trajectory) differ from the oracle by more than the threshold (default 1e-9, the suite's tolerance), and prints for the
worst row of each: the error per rollout step, the oracle's OWN sensitivity to a 1e-13 relative perturbation of its inputs
per step, the smallest sphere / plane barrier coordinate and the smallest joint-limit distance over the trajectory, and
the size of the commanded velocities.  A row whose oracle self-sensitivity exceeds the kernel-vs-oracle difference is an
ill-conditioned sample, not a disagreement."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle_lib as oracle
    import soak_parity as sp
    from multi_robot_fabrics_amd import config
    from multi_robot_fabrics_amd.runtime import FabricHandle
    first, n = int(sys.argv[1]), int(sys.argv[2])
    thr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-9
    lim = np.array(config.PANDA_LIMITS)
    found = 0
    for seed in range(first, first + n):
        cfg, batch, B, N = sp.draw_seed(seed)
        h = FabricHandle(cfg, 0)
        qt, qdt, pt = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
        _, want_q, want_qd = oracle.rollout(cfg, batch["q"], batch["qdot"], batch["params"], traj=True)
        xm = sp.trajectory_min_barrier(oracle, config, cfg, [batch["q"]] + [want_q[k] for k in range(cfg.horizon)], B, N)
        _, tq, tqd = h.rollout(qt, qdt, pt, want_traj=True)
        got = tqd.cpu().numpy()
        ok = np.isfinite(want_qd).all(axis=(0, 1)) & np.isfinite(got).all(axis=(0, 1))
        sel = ok & (np.repeat(xm, N) >= 0.05)
        if not sel.any():
            continue
        scale = np.abs(want_qd[:, :, sel]).max()
        err = np.abs(got - want_qd)
        err[:, :, ~sel] = 0.0
        if err.max() / scale <= thr:
            continue
        found += 1
        k, j, r = np.unravel_index(err.argmax(), err.shape)
        rng = np.random.default_rng(0)
        q2 = batch["q"] * (1 + 1e-13 * rng.standard_normal(batch["q"].shape))
        qd2 = batch["qdot"] * (1 + 1e-13 * rng.standard_normal(batch["qdot"].shape))
        _, _, w2 = oracle.rollout(cfg, q2, qd2, batch["params"], traj=True)
        traj_q = np.stack([batch["q"][:, r]] + [want_q[kk][:, r] for kk in range(cfg.horizon)])      # [H+1, 7]
        lim_dist = np.minimum(traj_q - lim[:, 0], lim[:, 1] - traj_q)
        print(f"seed {seed}: N={N} H={cfg.horizon} B={B} dynamic={cfg.dynamic} kernel_select={cfg.kernel_select} "
              f"ego_mask={cfg.ego_link_mask:#x} spheres={cfg.n_spheres} est_mask={cfg.goal_estimate_mask:#x}  "
              f"seed-wide rel err {err.max() / scale:.2e} (scale {scale:.3g}), worst row {r} (robot {r % N}), joint {j}, step {k}")
        print(f"   row's min sphere/plane barrier x over the trajectory {xm[r // N]:.4f}; min joint-limit distance "
              f"{lim_dist.min():.4f} rad (joint {int(np.unravel_index(lim_dist.argmin(), lim_dist.shape)[1])}, "
              f"at trajectory point {int(np.unravel_index(lim_dist.argmin(), lim_dist.shape)[0])})")
        for kk in range(cfg.horizon):
            den = max(1e-300, np.abs(want_qd[kk][:, r]).max())
            print(f"   step {kk}: |qd| {den:9.3g}   kernel vs oracle {np.abs(got[kk, :, r] - want_qd[kk][:, r]).max() / den:.2e}"
                  f"   oracle self-sensitivity(1e-13) {np.abs(w2[kk][:, r] - want_qd[kk][:, r]).max() / den:.2e}"
                  f"   min limit distance {lim_dist[kk].min():.4f}", flush=True)
    print(f"seeds {first}..{first + n - 1}: {found} above {thr:g}")


if __name__ == "__main__":
    main()
