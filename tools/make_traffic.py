#!/usr/bin/env python3
"""Derives bench.py's roofline inputs from a PMC summary -- no hand copy.

usage: make_traffic.py <profiles/TAG_pmc.json> <key> [--horizon H] [--kernel-substring k_rollout_panda] [--out profiles/traffic.json]
  key: the traffic.json entry to (re)write: "rollout_<dtype>_N<N>_H<H>_B<B>" (the headline, what bench.py looks up) or
       "config_<NAME>_<dtype>_B<B>" (bench.py's `configs` block; --horizon = lane-steps per launch: H, or 1 for compute_action)

From the kernel's entry in the PMC file (tools/summarize_prof.py: means over the full-batch dispatches of separate
--pmc passes):
    bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024        FETCH_SIZE / WRITE_SIZE are KiB; FETCH doubled per
                                                                  MI355X_MICROARCH.md "HBM" (gfx950 correction)
    flops_per_unit   = (2 * SQ_INSTS_VALU_FMA_F64 + SQ_INSTS_VALU_MUL_F64 + SQ_INSTS_VALU_ADD_F64) / SQ_WAVES / H
                       (f32 kernels: the _F32 counters) = executed flops per lane and rollout step
and stamps the entry with the sha256 of the kernel sources (csrc/mrf_kernels.hip + csrc/mrf_device.hpp) that the PMC
file itself was taken with (its "_meta" block), so that bench.py can say "roofline_inputs_stale" when the kernel changed
afterwards.  tests/test_roofline_inputs.py recomputes every entry from its `source` file."""
import argparse
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("multi-robot-fabrics_amd/csrc/mrf_kernels.hip", "multi-robot-fabrics_amd/csrc/mrf_device.hpp",
                  "multi-robot-fabrics_amd/csrc/mrf_tile.hpp", "multi-robot-fabrics_amd/csrc/mrf_shard.hpp",
                  "multi-robot-fabrics_amd/csrc/mrf_comm.hip", "multi-robot-fabrics_amd/csrc/mrf_shard_step.hip")


def kernel_source_sha256(root=ROOT):
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def derive(entry, horizon, dtype, work_waves=None):
    """work_waves: wave-sized blocks of rows one launch works on, for PERSISTENT kernels whose waves walk several blocks each
    (k_action_coupled since round 6: SQ_WAVES is the resident grid, not the amount of work); None: SQ_WAVES."""
    sfx = "F64" if dtype == "f64" else "F32"
    out = {"bytes_per_launch": (2.0 * entry["FETCH_SIZE"] + entry["WRITE_SIZE"]) * 1024.0}
    need = [f"SQ_INSTS_VALU_FMA_{sfx}", f"SQ_INSTS_VALU_MUL_{sfx}", f"SQ_INSTS_VALU_ADD_{sfx}", "SQ_WAVES"]
    if all(k in entry for k in need):
        out["flops_per_unit"] = (2.0 * entry[need[0]] + entry[need[1]] + entry[need[2]]) / (work_waves or entry["SQ_WAVES"]) / horizon
    if work_waves:
        out["work_waves"] = work_waves
    return out


def sharded_entry(pmc, substrings, rows, steps_per_launch):
    """bytes_per_row_step = sum over the named kernels of (2*FETCH_SIZE + WRITE_SIZE)*1024 / rows / steps_per_launch."""
    kernels = [pick(pmc, sub) for sub in substrings]
    total = sum((2.0 * pmc[k]["FETCH_SIZE"] + pmc[k]["WRITE_SIZE"]) * 1024.0 for k in kernels)
    return {"bytes_per_row_step": total / rows / steps_per_launch, "kernels": [k[:60] for k in kernels], "rows": rows,
            "steps_per_launch": steps_per_launch}


def pick(pmc, substring):
    hits = [k for k in pmc if substring in k and not k.startswith("_")]
    if len(hits) != 1:
        raise SystemExit(f"{len(hits)} kernels match {substring!r}: {hits}")
    return hits[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc")
    ap.add_argument("key")
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--kernel-substring", default="k_rollout_panda<")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "traffic.json"))
    ap.add_argument("--rows", type=int, default=None, help="sharded_* keys: rows the counted launches worked on")
    ap.add_argument("--sum-kernels", nargs="+", default=None, help="sharded_* keys: kernel-name substrings whose bytes add up")
    ap.add_argument("--steps-per-launch", type=int, default=1, help="sharded_* keys: rollout steps one launch covers")
    ap.add_argument("--work-waves", type=int, default=None,
                    help="persistent kernels: wave-sized blocks of rows per launch (replaces SQ_WAVES in flops_per_unit)")
    args = ap.parse_args()
    ms = re.fullmatch(r"sharded_(rccl|peer|torch)_(joints|spheres)_(f64|f32)", args.key)
    if ms:
        # measured HBM bytes per owned row and rollout step of a robot-sharded transport's kernels (sharded.roofline)
        if not (args.rows and args.sum_kernels):
            raise SystemExit("sharded_* keys need --rows and --sum-kernels")
        with open(args.pmc) as f:
            pmc = json.load(f)
        rel = os.path.relpath(os.path.abspath(args.pmc), ROOT)
        rec = sharded_entry(pmc, args.sum_kernels, args.rows, args.steps_per_launch)
        rec.update(source=rel, kernel_source_sha256=pmc.get("_meta", {}).get("kernel_source_sha256"))
        tj = {}
        if os.path.exists(args.out):
            with open(args.out) as f:
                tj = json.load(f)
        tj[args.key] = rec
        with open(args.out, "w") as f:
            json.dump(tj, f, indent=1)
            f.write("\n")
        print(json.dumps({args.key: rec}, indent=1))
        return
    m = re.fullmatch(r"rollout_(f64|f32)_N(\d+)_H(\d+)_B(\d+)", args.key)
    mc = re.fullmatch(r"config_([A-Za-z0-9]+)_(f64|f32)_B(\d+)", args.key)
    if m:
        dtype, H = m.group(1), int(m.group(3))
        H = args.horizon or H
    elif mc and args.horizon:
        dtype, H = mc.group(2), args.horizon
    else:
        raise SystemExit("key must look like rollout_f64_N3_H30_B129024, or config_C3_f64_B65536 with --horizon")
    with open(args.pmc) as f:
        pmc = json.load(f)
    kernel = pick(pmc, args.kernel_substring)
    rec = derive(pmc[kernel], H, dtype, args.work_waves)
    rel = os.path.relpath(os.path.abspath(args.pmc), ROOT)
    rec.update(source=rel, kernel=kernel[:60], horizon=H,
               flops_source=f"(2*SQ_INSTS_VALU_FMA_{dtype.upper()} + MUL + ADD) / {'work_waves' if args.work_waves else 'SQ_WAVES'} / H of {rel}",
               kernel_source_sha256=pmc.get("_meta", {}).get("kernel_source_sha256"))
    tj = {}
    if os.path.exists(args.out):
        with open(args.out) as f:
            tj = json.load(f)
    tj[args.key] = rec
    with open(args.out, "w") as f:
        json.dump(tj, f, indent=1)
        f.write("\n")
    print(json.dumps({args.key: rec}, indent=1))


if __name__ == "__main__":
    main()
