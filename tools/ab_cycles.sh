#!/bin/bash
# Runs ON THE GPU BOX: clock-independent A/B of kernel builds (ab/lib<name>.so from tools/build_variant.sh).
#   bash tools/ab_cycles.sh "<names>" [B] [dtype] ["<counters>"]      (KERNEL=substring, PROG=tools/prof_kernels.py to look at another kernel)
# Prints, per build, the rollout kernel's mean GRBM_GUI_ACTIVE (GPU cycles, summed over the 8 XCDs) and wait counters.
names=$1; B=${2:-129024}; dt=${3:-f64}; ctr=${4:-"GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES"}
root=$(pwd); cd /tmp; export TMPDIR=/tmp
for n in $names; do
  out=$root/gpurun_out/ab_$n; rm -rf $out
  MRF_HIP_LIB=$root/ab/lib$n.so rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python3 $root/${PROG:-tools/prof_rollout.py} ${PROGARGS:-$B $dt 4} > $out.log 2>&1
  python3 - "$n" $out "${KERNEL:-k_rollout_panda}" <<'PY'
import csv, glob, sys, collections
n, d, kern = sys.argv[1:4]
f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"]]
g = max(int(r["Grid_Size"]) for r in rows)
per = collections.defaultdict(list); dur = {}
for r in rows:
    if int(r["Grid_Size"]) == g:
        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(n, {k: f"{sum(v)/len(v):.4g}" for k, v in per.items()}, f"dur_us={sum(dur.values())/len(dur):.1f}")
PY
done
