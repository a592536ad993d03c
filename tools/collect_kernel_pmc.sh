#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_kernel_pmc.sh <tag>'): counter evidence for every kernel behind
# include/mrf.h EXCEPT the bench's rollout kernel (tools/collect_profiles.sh does that one): separate rocprofv3 --pmc passes
# (never combined with another trace domain) over tools/prof_kernels.py (all kernels at the bench batch, f64).  The BASELINE
# configurations C2 / C3 / C5 / Cartesian have their own script, tools/collect_config_pmc.sh.
# Writes gpurun_out/profiles/<tag>_kernels_pmc.json (per kernel: FETCH_SIZE / WRITE_SIZE in KiB, instruction and wait
# counters, registers, LDS, scratch) and <tag>_per_kernel_f64.json with measured-vs-algorithmic bytes per kernel.
# usage: collect_kernel_pmc.sh <tag> [program] [suffix]     default: tools/prof_kernels.py, suffix "kernels"
#   round 6: collect_kernel_pmc.sh <tag> tools/prof_shard_one.py shard1  -> <tag>_shard1_pmc.json: the action kernels of a rank
#   that owns ONE robot of three, joint-state payload (two remote chains re-walked per lane) against the sphere payload
tag=${1:-rXX}; prog=${2:-tools/prof_kernels.py}; suffix=${3:-kernels}
root=$(pwd); out=$root/gpurun_out/kpmc_${tag}_$suffix
mkdir -p $out $root/gpurun_out/profiles; cd /tmp; export TMPDIR=/tmp
python3 $root/$prog f64 > $out/prof_kernels.txt 2>&1
[ "$suffix" = "kernels" ] && cp $root/gpurun_out/prof_kernels_f64.json $out/prof_kernels_f64.json
specs=""
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
  name=$(echo $pass | tr ' ' '+')
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/k_$name -- python3 $root/$prog f64 > $out/k_$name.log 2>&1
  specs="$specs $name=$out/k_$name"
done
python3 $root/tools/summarize_prof.py ${tag}_$suffix $out/none $root/gpurun_out/profiles $specs > $out/summary_k.log 2>&1
[ "$suffix" != "kernels" ] && { tail -c 600 $out/prof_kernels.txt; exit 0; }
python3 $root/tools/kernel_byte_ratios.py $root/gpurun_out/profiles/${tag}_kernels_pmc.json $out/prof_kernels_f64.json > $root/gpurun_out/profiles/${tag}_per_kernel_f64.json 2> $out/ratios.err
tail -c 1500 $root/gpurun_out/profiles/${tag}_per_kernel_f64.json
