#!/usr/bin/env python3
"""300 control steps of the default two-Panda cell (one scene) for tools/step_timeline.py."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_robot_fabrics_amd.cell import PandaCell
from multi_robot_fabrics_amd.parameters import load_yaml_settings
params, _ = load_yaml_settings()
params.n_obst_per_link = 1
params.nr_obsts_dyn_all = [8 * (params.nr_robots - 1)] * params.nr_robots
cell = PandaCell.from_parameters(params, rollouts="jointspace", dynamic=True)
log = cell.run(300, stop_when_done=False)
print("device ms per control step:", 1e3 * float(log.solver_s[50:].mean()))
