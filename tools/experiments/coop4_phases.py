#!/usr/bin/env python3
"""Cycle stamps of the four-wave rollout kernel's phases, per wave (needs a library built with -DMRF_COOP_CLOCKS:
tools/build_variant.sh c4clk - -DMRF_COOP4 -DMRF_COOP_CLOCKS ; MRF_HIP_LIB=ab/libc4clk.so MRF_ABI_ANY=1 python3 tools/coop4_phases.py;
the kernel is an experiment, csrc/experiments/coop4.hpp)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
cfg = config.panda_config(n_robots=3, horizon=30)
cfg.goal_estimate_mask = 0b110
cfg.kernel_select = 3
b = scenarios.panda_batch(cfg, 1, seed=1)
h = FabricHandle(cfg, 0)
q, qd, prm = (h.tensor(b[k]) for k in ("q", "qdot", "params"))
for _ in range(5):
    h.rollout(q, qd, prm)
torch.cuda.synchronize()
out = (C.c_longlong * 32)()
h.lib.mrf_debug_clocks.argtypes = [C.c_void_p, C.c_int]
assert h.lib.mrf_debug_clocks(out, 32) == 0
names = ["integrate + walk + publish", "barrier 1", "fold + chunk sum", "barrier 2", "wave job", "barrier 3 + solve", "barrier 4 + finish"]
t = list(out)
t0 = min(t[w * 8] for w in range(4))
print("cycles per phase at horizon step 5 (shader clock), waves 0..3:")
for i, n in enumerate(names):
    print(f"  {n:28s}" + "".join(f"{t[w * 8 + i + 1] - t[w * 8 + i]:8d}" for w in range(4)))
print(f"  {'step total':28s}" + "".join(f"{t[w * 8 + 7] - t[w * 8]:8d}" for w in range(4)))
print(f"  {'start skew':28s}" + "".join(f"{t[w * 8] - t0:8d}" for w in range(4)))
