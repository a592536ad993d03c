"""Soak of tests/test_gpu_parity.py::test_random_planner_configurations over many more seeds than the suite runs
(script, not collected by pytest):  python3 tests/soak_random_configs.py [first_seed] [n_seeds]
Every seed draws a planner definition (leaf strings, constants, goal / plane / limit switches, sphere table, mode) and
compares rollout, compute_action (both modes) and the coupled action with the oracle at the suite's tolerance."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle_lib
    import test_gpu_parity as t
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    bad = []
    for seed in range(first, first + n):
        try:
            t.test_random_planner_configurations(oracle_lib, seed)
        except AssertionError as e:
            bad.append((seed, str(e)))
            print("seed", seed, "FAILED:", e, flush=True)
        except Exception:
            bad.append((seed, traceback.format_exc(limit=2)))
            print("seed", seed, "ERROR", flush=True)
    print(f"seeds {first}..{first + n - 1}: {n - len(bad)} passed, {len(bad)} failed {[b[0] for b in bad]}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
