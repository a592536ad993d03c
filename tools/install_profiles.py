#!/usr/bin/env python3
"""After `gpurun -- bash tools/collect_all.sh <tag>`: copies the judged summaries from gpurun_out/ into profiles/ and
regenerates profiles/traffic.json from them (headline, `configs` block at bench.config_sizes, robot-sharded transports), so
that every entry's `source` is the committed file and carries the hash of the kernel sources the counters were taken with.
usage: python3 tools/install_profiles.py r05 [--cus 256]"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
cus = int(sys.argv[sys.argv.index("--cus") + 1]) if "--cus" in sys.argv else 256
os.chdir(ROOT)
for name in ("kernel_stats.csv", "pmc.json", "kernels_pmc.json", "per_kernel_f64.json", "configs_pmc.json",
             "c32_configs_pmc.json", "shard1_pmc.json"):
    shutil.copy(f"gpurun_out/profiles/{tag}_{name}", f"profiles/{tag}_{name}")
if os.path.exists("gpurun_out/ex/summary.json"):
    shutil.copy("gpurun_out/ex/summary.json", f"profiles/{tag}_examples.json")
with open("profiles/traffic.json") as f:
    tj = json.load(f)
for k in [k for k in tj if k.startswith(("config_", "sharded_"))]:
    tj.pop(k)
with open("profiles/traffic.json", "w") as f:
    json.dump(tj, f, indent=1)


def mt(*a):
    subprocess.check_call([sys.executable, "tools/make_traffic.py"] + list(a), stdout=subprocess.DEVNULL)


def size(n_robots, rounds=6):
    return rounds * cus * 4 * (64 // n_robots)


rows = json.load(open("gpurun_out/prof_kernels_f64.json"))["scenarios"] * 3
mt(f"profiles/{tag}_pmc.json", f"rollout_f64_N3_H30_B{size(3)}")
c = f"profiles/{tag}_configs_pmc.json"
# k_action_coupled is persistent (round 6): its waves walk several blocks each, the work is B / 32 wave-sized blocks
mt(c, f"config_C2_f64_B{size(2)}", "--horizon", "1", "--kernel-substring", "k_action_coupled<", "--work-waves", str(size(2) // 32))
mt(c, f"config_C3_f64_B{size(2)}", "--horizon", "20", "--kernel-substring", "k_rollout_panda<double, LS_reference, true>")
mt(c, f"config_C5_f64_B{size(8)}", "--horizon", "50", "--kernel-substring", "k_rollout_panda<double, LS_reference, false>")
mt(c, f"config_CART_f64_B{size(3)}", "--horizon", "30", "--kernel-substring", "k_rollout_cart_panda<")
mt(c, f"config_CARTC_f64_B{size(3)}", "--horizon", "30", "--kernel-substring", "k_rollout_cartc_panda<double, LS_reference, 0>")
c32 = f"profiles/{tag}_c32_configs_pmc.json"      # the 32-sphere Cartesian shapes: their own pass set (same kernel name as CART)
mt(c32, f"config_CART32_f64_B{size(2)}", "--horizon", "30", "--kernel-substring", "k_rollout_cart_panda<")
mt(c32, f"config_CARTC32_f64_B{size(2)}", "--horizon", "30", "--kernel-substring", "k_rollout_carts_panda<")
k = f"profiles/{tag}_kernels_pmc.json"
mt(k, "sharded_rccl_spheres_f64", "--rows", str(rows), "--sum-kernels", "k_step_predict<", "k_step_action<", "--steps-per-launch", "1")
mt(k, "sharded_rccl_joints_f64", "--rows", str(rows), "--sum-kernels", "k_step_action_joints<",
   "--steps-per-launch", "1")   # the loop's one launch per step (action + next predict); the first step's predict is 1 of 30
# a group of one exchanges nothing: the same persistent kernel (XK_NONE) whatever the configured payload
mt(k, "sharded_peer_joints_f64", "--rows", str(rows), "--sum-kernels", "k_rollout_peer<", "--steps-per-launch", "30")
mt(k, "sharded_peer_spheres_f64", "--rows", str(rows), "--sum-kernels", "k_rollout_peer<", "--steps-per-launch", "30")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_traffic  # noqa: E402
tj = json.load(open("profiles/traffic.json"))
print("kernel sources now:", make_traffic.kernel_source_sha256()[:12])
for key, v in tj.items():
    if isinstance(v, dict) and "kernel_source_sha256" in v:
        print(f"  {key:34s} {str(v['kernel_source_sha256'])[:12]}  {v['source']}")
