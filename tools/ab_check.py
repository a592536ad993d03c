#!/usr/bin/env python3
"""A/B correctness of a kernel build (MRF_HIP_LIB): the coupled 3-Panda RF-CV H=30 rollout of 2016 seeded scenarios,
trajectory end state and velocity signal saved to gpurun_out/abcheck_<tag>.npz; with two tags, compares them.
usage: MRF_HIP_LIB=ab/libX.so python3 tools/ab_check.py save X   |   python3 tools/ab_check.py diff X Y"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out")
if sys.argv[1] == "diff":
    a, b = (np.load(os.path.join(out, f"abcheck_{t}.npz")) for t in sys.argv[2:4])
    for k in a.files:
        print(sys.argv[2], sys.argv[3], k, "max rel diff", float(np.abs(a[k] - b[k]).max() / np.abs(a[k]).max()))
    sys.exit(0)
import torch
from multi_robot_fabrics_amd import config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
cfg = config.panda_config(n_robots=3, horizon=30)
cfg.goal_estimate_mask = 0b110
cfg.kernel_select = 1
batch = scenarios.panda_batch(cfg, 2016, seed=1000)
h = FabricHandle(cfg, 0)
avg, tq, tqd = h.rollout(*(h.tensor(batch[k]) for k in ("q", "qdot", "params")), want_traj=True)
os.makedirs(out, exist_ok=True)
np.savez(os.path.join(out, f"abcheck_{sys.argv[2]}.npz"), avg=avg.cpu().numpy(), q_end=tq[-1].cpu().numpy(), qd_end=tqd[-1].cpu().numpy())
print("saved", sys.argv[2])
