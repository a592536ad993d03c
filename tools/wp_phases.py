#!/usr/bin/env python3
"""Phase timeline of the wave-pair rollout kernel (needs a library built with -DMRF_WP_CLOCKS:
tools/build_variant.sh wpclk - -DMRF_WP_CLOCKS -DMRF_DEV_F64_PANDA_ONLY ; MRF_HIP_LIB=ab/libwpclk.so python3 tools/wp_phases.py [B]).
Prints, for both waves of the middle workgroup at horizon step 5, the shader-cycle stamps relative to wave A's step start."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multi_robot_fabrics_amd import config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 129024
cfg = config.panda_config(n_robots=3, horizon=30)
cfg.goal_estimate_mask = 0b110
b = scenarios.panda_batch(cfg, B, seed=1)
h = FabricHandle(cfg, 0)
q, qd, prm = (h.tensor(b[k]) for k in ("q", "qdot", "params"))
for _ in range(3):
    h.rollout(q, qd, prm)
torch.cuda.synchronize()
out = (C.c_longlong * 32)()
h.lib.mrf_debug_wp_clocks.argtypes = [C.c_void_p, C.c_int]
assert h.lib.mrf_debug_wp_clocks(out, 32) == 0
t = list(out)
names = ["step start", "B2 arrive", "B2 leave", "fold done", "Bfree arrive", "Bfree leave", "Bx arrive", "Bx leave",
         "Bc arrive", "Bc leave", "Bd arrive", "Bd leave"]
t0 = t[0]
print(f"B={B}: shader cycles since wave A's step start (horizon step 5, middle workgroup)")
print(f"{'':14s} {'wave A':>9s} {'wave B':>9s}")
for i, n in enumerate(names):
    a, bb = t[i] - t0, t[16 + i] - t0
    print(f"{n:14s} {a if t[i] else -1:9d} {bb if t[16 + i] else -1:9d}")
