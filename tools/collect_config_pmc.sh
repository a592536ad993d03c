#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/collect_config_pmc.sh <tag>'): counter passes for bench.py's `configs` block
# (C2, C3, C5, Cartesian rollout; tools/prof_configs.py launches them as bench.py does) -- separate rocprofv3 --pmc passes,
# never combined with another trace domain -- and the traffic.json entries derived from them (tools/make_traffic.py).
# Writes gpurun_out/profiles/<tag>_configs_pmc.json and <tag>_configs_traffic.json; copy what is to be judged into profiles/.
# usage: collect_config_pmc.sh <tag> [f64|f32] ["C2 C3 ..."]      default set: C2 C3 C5 CART CARTC
# The summary is keyed by kernel name, and CART32 launches the same k_rollout_cart_panda instantiation as CART on another
# obstacle count: the 32-sphere Cartesian shapes (the reference's default, panda_config.yaml:8) get their own pass set,
#   collect_config_pmc.sh <tag>_c32 f64 "CART32 CARTC32"      -> <tag>_c32_configs_pmc.json   (round 6)
tag=${1:-rXX}; dt=${2:-f64}
root=$(pwd); out=$root/gpurun_out/cpmc_$tag
mkdir -p $out $root/gpurun_out/profiles; cd /tmp; export TMPDIR=/tmp
cfgs=${3:-"C2 C3 C5 CART CARTC"}
python3 $root/tools/prof_configs.py $dt $cfgs > $out/prof_configs.txt 2>&1
specs=""
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM SQ_INSTS_VMEM" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  name=$(echo $pass | tr ' ' '+')
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $out/c_$name -- python3 $root/tools/prof_configs.py $dt $cfgs > $out/c_$name.log 2>&1
  specs="$specs $name=$out/c_$name"
done
python3 $root/tools/summarize_prof.py ${tag}_configs $out/none $root/gpurun_out/profiles $specs > $out/summary.log 2>&1
pmc=$root/gpurun_out/profiles/${tag}_configs_pmc.json
tj=$root/gpurun_out/profiles/${tag}_configs_traffic.json
# keys carry the batch bench.config_sizes gives on THIS device (six whole rounds of resident waves)
size() { python3 -c "import sys; sys.path.insert(0, '$root'); import torch, bench; print(bench.config_sizes('$1', torch.cuda.get_device_properties(0).multi_processor_count))" 2>/dev/null | tail -1; }
has() { case " $cfgs " in *" $1 "*) return 0;; esac; return 1; }
mt() { python3 $root/tools/make_traffic.py $pmc config_$1_${dt}_B$(size $1) --horizon $2 --kernel-substring "$3" $4 $5 --out $tj >> $out/summary.log 2>&1; }
has C2 && mt C2 1 "k_action_coupled<" --work-waves $(( $(size C2) / 32 ))   # persistent kernel: blocks of work, not resident waves
has C3 && mt C3 20 "k_rollout_panda<double, LS_reference, true>"
has C5 && mt C5 50 "k_rollout_panda<double, LS_reference, false>"
has CART && mt CART 30 "k_rollout_cart_panda<"
has CARTC && mt CARTC 30 "k_rollout_cartc_panda<double, LS_reference, 0>"
has CART32 && mt CART32 30 "k_rollout_cart_panda<"
has CARTC32 && mt CARTC32 30 "k_rollout_carts_panda<"
tail -c 2500 $out/summary.log; cat $out/prof_configs.txt | cut -c1-400
