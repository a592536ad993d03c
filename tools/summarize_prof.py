#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small files committed under profiles/.
usage: summarize_prof.py <tag> <stats_dir> <out_dir> [name=pmc_dir ...]
  stats_dir : output of  rocprofv3 --kernel-trace --stats
  pmc_dir   : output of  rocprofv3 --kernel-trace --pmc <counters>   (one directory per pass)
Writes <out_dir>/<tag>_kernel_stats.csv (copy) and <out_dir>/<tag>_pmc.json with, per kernel, the mean of each counter
over the launches of the LARGEST grid (the B=1 latency launches of bench.py would otherwise pollute the means)
and the mean duration of those same launches."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def short_name(k):
    """Demangled kernel name without its argument list and with the leaf-policy set abbreviated, so that instantiations
    that differ only in their trailing flags (LO, RES, ACC ...) keep distinct keys."""
    k = k.split("(")[0]
    k = k.replace("mrf::vgpr_literals::", "mrf::").replace("mrf::sgpr_literals::", "mrf::")  # inline namespaces of mrf_device.hpp
    k = k.replace("mrf::LeafSet<mrf::LeafPow<4, 4, 0, 0>, mrf::SLeaf<1, 0, 0, 1, 1>, mrf::SLeaf<0, 1, 0, 1, 1> >", "LS_reference")
    k = k.replace("mrf::LeafSet<mrf::LeafGeneric, mrf::SLeafGeneric, mrf::SLeafGeneric>", "LS_generic")
    return k.replace("void ", "")[:120]


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main():
    tag, stats_dir, out_dir = sys.argv[1:4]
    os.makedirs(out_dir, exist_ok=True)
    ks = find(stats_dir, "_kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    summary = collections.defaultdict(dict)
    for spec in sys.argv[4:]:
        _, d = spec.split("=", 1)
        cc = find(d, "_counter_collection.csv")
        if not cc:
            continue
        rows = list(csv.DictReader(open(cc)))
        by_kernel = collections.defaultdict(list)
        for r in rows:
            by_kernel[r["Kernel_Name"]].append(r)
        for k, rs in by_kernel.items():
            if "mrf::" not in k:
                continue
            gmax = max(int(r["Grid_Size"]) for r in rs)
            rs = [r for r in rs if int(r["Grid_Size"]) == gmax]
            key = short_name(k)
            s = summary[key]
            s["grid_size"] = gmax
            s["vgpr"], s["agpr"], s["lds_bytes"], s["scratch_bytes"] = (
                int(rs[0]["VGPR_Count"]), int(rs[0]["Accum_VGPR_Count"]), int(rs[0]["LDS_Block_Size"]), int(rs[0]["Scratch_Size"]))
            per = collections.defaultdict(list)
            disp = {}
            for r in rs:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
                disp[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            for c, v in per.items():
                s[c] = sum(v) / len(v)
            s.setdefault("launches", {})[spec.split("=")[0]] = len(disp)
            s.setdefault("duration_us_under_pmc", {})[spec.split("=")[0]] = sum(disp.values()) / len(disp)
    # which kernel sources these counters belong to (tools/make_traffic.py, bench.py "roofline_inputs_stale")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from make_traffic import kernel_source_sha256
    summary["_meta"] = {"kernel_source_sha256": kernel_source_sha256(), "tag": tag}
    with open(os.path.join(out_dir, f"{tag}_pmc.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    print(json.dumps(summary, indent=1, sort_keys=True)[:6000])


if __name__ == "__main__":
    main()
