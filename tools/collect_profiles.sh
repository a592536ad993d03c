#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh <tag> [B] [dtype]').
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command (kernel time shares and average durations)
# 2. separate --pmc passes over tools/prof_rollout.py (the rollout kernel alone, same B): HBM traffic counters and
#    the SQ issue counters.  --pmc is never combined with any other trace domain.
# Summaries land in gpurun_out/profiles/<tag>_*; copy the ones to be judged into profiles/.
tag=${1:-rXX}; B=${2:-129024}; dt=${3:-f64}
root=$(pwd); out=$root/gpurun_out/prof_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --dtype $dt > $out/bench_under_stats.json 2> $out/stats.err
specs=""
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  name=$(echo $pass | tr ' ' '+')
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/pmc_$name -- python3 $root/tools/prof_rollout.py $B $dt 3 > $out/pmc_$name.log 2>&1
  specs="$specs $name=$out/pmc_$name"
done
python3 $root/tools/summarize_prof.py $tag $out/stats $root/gpurun_out/profiles $specs > $out/summary.log 2>&1
# bench.py's roofline inputs, derived from the summary just written (never copied by hand)
python3 $root/tools/make_traffic.py $root/gpurun_out/profiles/${tag}_pmc.json rollout_${dt}_N3_H30_B$B --out $root/gpurun_out/profiles/${tag}_traffic.json >> $out/summary.log 2>&1
tail -c 3000 $out/summary.log
