#!/usr/bin/env python3
"""The BASELINE.json configurations beside the bench headline -- C2, C3, C5 and the Cartesian rollout, built by
scenarios.baseline_config and launched by bench.run_config exactly as bench.py's `configs` block launches them -- for
rocprofv3 passes (tools/collect_config_pmc.sh) and for a quick look at their kernel times.
usage: python3 tools/prof_configs.py [f64|f32] [C2 C3 C5 CART CARTC ...] [--check]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from multi_robot_fabrics_amd import scenarios

dtype = next((a for a in sys.argv[1:] if a in ("f64", "f32")), "f64")
names = [a for a in sys.argv[1:] if a in scenarios.BASELINE_CONFIGS] or list(scenarios.BASELINE_CONFIGS)
for name in names:
    r = bench.run_config(name, dtype, 0, check="--check" in sys.argv)
    print(json.dumps({name: r}), flush=True)
