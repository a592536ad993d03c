"""bench.py's `configs` block: every BASELINE configuration beside the headline is launched at its
bench size, timed by HIP events, given a roofline entry from the stamped counter file and checked against the oracle on
64 scenarios."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("name", ["C2", "C3", "C5", "CART", "CARTC", "CART32", "CARTC32"])
def test_config_entry(name):
    import bench
    from multi_robot_fabrics_amd import scenarios
    assert name in scenarios.BASELINE_CONFIGS
    r = bench.run_config(name, "f64", 0, iters=2, warmup=1)
    assert r["kernel_ms"] > 0 and r["units_per_s"] > 0
    # six whole rounds of resident single-wave workgroups (4 per CU), as the headline batch
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert r["scenarios"] == bench.CONFIG_ROUNDS * cus * 4 * (64 // r["robots"])
    assert r["parity_spot_check"]["ok"] and r["parity_spot_check"]["tol"] == 1e-9, r["parity_spot_check"]
    roof = r["roofline"]
    assert roof["hbm_algorithmic"]["frac"] > 0
    if roof["bound"] == "valu_f64":                       # a counter pass is on record for this configuration
        assert 0.05 < roof["frac"] < 1.0 and roof["flops_per_unit"] > 1000 and roof["traffic_key"].startswith(f"config_{name}_f64_B")
        assert roof["roofline_inputs_stale"] in (True, False)


def test_baseline_configs_are_the_parity_tests_configs():
    """scenarios.baseline_config is the single place the configurations are built: C5's table is the 20-sphere one, C2's the
    10-sphere one, the Cartesian entry has M = 16."""
    from multi_robot_fabrics_amd import scenarios
    c2, c3, c5, ca = (scenarios.baseline_config(n)["cfg"] for n in ("C2", "C3", "C5", "CART"))
    assert (c2.n_robots, c2.n_spheres, c2.horizon) == (2, 10, 1)
    assert (c3.n_robots, c3.n_spheres, c3.horizon) == (2, 8, 20)
    assert (c5.n_robots, c5.n_spheres, c5.horizon, c5.goal_estimate_mask) == (8, 20, 50, 0xFE)
    assert (ca.n_robots, ca.n_spheres * (ca.n_robots - 1), ca.horizon) == (3, 16, 30)
    c32 = scenarios.baseline_config("CARTC32")["cfg"]       # the reference's default example: panda_config.yaml n_obst_per_link: 4
    assert (c32.n_robots, c32.n_spheres, c32.horizon) == (2, 32, 30)
