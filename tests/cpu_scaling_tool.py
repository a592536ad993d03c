"""Thread scaling of the CPU oracle (context for bench.py's cpu_baseline): python3 tests/cpu_scaling_tool.py.
Lives under tests/ because only tests may run the oracle; it is a script, not a pytest module."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import oracle_lib as ol
    from multi_robot_fabrics_amd import config, scenarios
    cfg = config.panda_config(3, 30); cfg.goal_estimate_mask = 6
    b = scenarios.panda_batch(cfg, 8192, seed=1)
    print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
    for th in (1, 8, 32, 64, 128, 256):
        ol.set_threads(th)
        n = min(8192, 64 * th)
        sel = slice(0, n * 3)
        t = time.time(); ol.rollout(cfg, b["q"][:, sel], b["qdot"][:, sel], b["params"][:, sel]); dt = time.time() - t
        print(th, "threads", n, "scen: %.3fs  %.0f rollout-steps/s  (%.1fx)" % (dt, n * 90 / dt, 0))


if __name__ == "__main__":
    main()
