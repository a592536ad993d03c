#!/usr/bin/env python3
"""Cycle stamps of the cooperative rollout kernel's phases (needs a library built with -DMRF_COOP_CLOCKS:
tools/build_variant.sh clk - -DMRF_COOP_CLOCKS ; MRF_HIP_LIB=ab/libclk.so python3 tools/coop_phases.py)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
cfg = config.panda_config(n_robots=3, horizon=30)
cfg.goal_estimate_mask = 0b110
cfg.kernel_select = 2
b = scenarios.panda_batch(cfg, 1, seed=1)
h = FabricHandle(cfg, 0)
q, qd, prm = (h.tensor(b[k]) for k in ("q", "qdot", "params"))
for _ in range(5):
    h.rollout(q, qd, prm)
torch.cuda.synchronize()
out = (C.c_longlong * 8)()
h.lib.mrf_debug_clocks.argtypes = [C.c_void_p, C.c_int]
assert h.lib.mrf_debug_clocks(out, 8) == 0
names = ["integrate", "walk", "barrier+publish", "fold my chunk (+plane)", "chunk reduce", "finish"]
t = list(out)
print("cycles per phase at horizon step 5 (s_memtime / shader clock):")
for i, n in enumerate(names):
    print(f"  {n:26s} {t[i + 1] - t[i]:7d}")
print(f"  {'step total':26s} {t[6] - t[0]:7d}")
