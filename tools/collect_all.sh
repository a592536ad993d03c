#!/bin/bash
# Runs ON THE GPU BOX: every record a round keeps under profiles/ in one call, every profiler invocation under its own timeout.
#   gpurun --timeout 2400 -- 'bash tools/collect_all.sh r05'
# -> gpurun_out/profiles/<tag>_{kernel_stats.csv,pmc.json,traffic.json,kernels_pmc.json,per_kernel_f64.json,configs_pmc.json,
#    configs_traffic.json}, gpurun_out/bench_default.json, gpurun_out/ex/summary.json
tag=${1:-rXX}
root=$(pwd)
mkdir -p gpurun_out/profiles
timeout 900 bash tools/collect_profiles.sh $tag > gpurun_out/collect_profiles.log 2>&1; echo "collect_profiles rc=$?"
timeout 900 bash tools/collect_kernel_pmc.sh $tag > gpurun_out/collect_kernel_pmc.log 2>&1; echo "collect_kernel_pmc rc=$?"
timeout 900 bash tools/collect_config_pmc.sh $tag > gpurun_out/collect_config_pmc.log 2>&1; echo "collect_config_pmc rc=$?"
timeout 900 bash tools/collect_config_pmc.sh ${tag}_c32 f64 "CART32 CARTC32" > gpurun_out/collect_config32_pmc.log 2>&1; echo "collect_config_pmc c32 rc=$?"
timeout 600 bash tools/collect_kernel_pmc.sh $tag tools/prof_shard_one.py shard1 > gpurun_out/collect_shard1_pmc.log 2>&1; echo "collect_kernel_pmc shard1 rc=$?"
# robot-sharded transports: measured HBM bytes per owned row and step (sharded.roofline's traffic)
rows=$(python3 -c "import json; print(json.load(open('$root/gpurun_out/prof_kernels_f64.json'))['scenarios'] * 3)")
k=$root/gpurun_out/profiles/${tag}_kernels_pmc.json
t=$root/gpurun_out/profiles/${tag}_sharded_traffic.json
python3 tools/make_traffic.py $k sharded_rccl_spheres_f64 --rows $rows --sum-kernels "k_step_predict<" "k_step_action<" --steps-per-launch 1 --out $t > /dev/null
python3 tools/make_traffic.py $k sharded_rccl_joints_f64 --rows $rows --sum-kernels "k_step_action_joints<" --steps-per-launch 1 --out $t > /dev/null
# a group of one exchanges nothing: the same persistent kernel (XK_NONE) whatever the configured payload
python3 tools/make_traffic.py $k sharded_peer_joints_f64 --rows $rows --sum-kernels "k_rollout_peer<" --steps-per-launch 30 --out $t > /dev/null
python3 tools/make_traffic.py $k sharded_peer_spheres_f64 --rows $rows --sum-kernels "k_rollout_peer<" --steps-per-launch 30 --out $t > /dev/null
cat $t | head -30
timeout 900 bash tools/run_examples.sh > gpurun_out/run_examples.log 2>&1; echo "run_examples rc=$?"
