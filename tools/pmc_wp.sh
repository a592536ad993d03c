#!/bin/bash
# Runs ON THE GPU BOX: SQ counter passes over tools/prof_rollout.py for the kernel variant selected by MRF_ROLLOUT_WP
# usage: tools/pmc_wp.sh <tag> [B]     -> gpurun_out/pmc_<tag>.txt  (one line per counter, per kernel)
#        PROG="tools/prof_configs.py f64 CART32" tools/pmc_wp.sh cart32      any other launcher of rollout kernels
tag=${1:-wp}; B=${2:-129024}
root=$(pwd); out=$root/gpurun_out/pmcwp_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
# every pass under its own timeout (a pass with FETCH_SIZE and WRITE_SIZE together aborted and hung for ten minutes: they
# go in separate passes, as the guide says); PASSES overrides the list (semicolon-separated)
passes=${PASSES:-"SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES;SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS;SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY;SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64;SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_VMEM;SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY;SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS;FETCH_SIZE;WRITE_SIZE"}
IFS=';' read -ra plist <<< "$passes"
for pass in "${plist[@]}"; do
  name=$(echo $pass | tr ' ' '+')
  timeout 180 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/$name -- python3 $root/${PROG:-tools/prof_rollout.py $B f64 2} > $out/$name.log 2>&1 || echo "pass $name: rc=$?"
done
python3 - $out > $root/gpurun_out/pmc_$tag.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rollout" not in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        v = d[c]
        print(f"  {c:34s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
cat $root/gpurun_out/pmc_$tag.txt
