#!/usr/bin/env python3
"""Kernel timeline of the device-resident control step at B = 1 from a rocprofv3 --kernel-trace run of
tools/step_timeline_run.py: per kernel of one replayed control step its mean duration and the mean gap to its predecessor.
usage: python3 tools/step_timeline.py <dir with *_kernel_trace.csv>"""
import collections
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "mrf::" in r["Kernel_Name"]]
rows = rows[len(rows) // 2:]                       # the warmed-up half
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev_end = None
for r in rows:
    name = r["Kernel_Name"].split("<")[0].replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[name].append(e - s)
    if prev_end is not None:
        gap[name].append(s - prev_end)
    prev_end = e
tot = 0
for name in dur:
    d, g = sum(dur[name]) / len(dur[name]) / 1e3, (sum(gap[name]) / len(gap[name]) / 1e3 if gap[name] else 0)
    print(f"{name:38s} n={len(dur[name]):5d}  kernel {d:8.2f} us   gap before {g:7.2f} us")
