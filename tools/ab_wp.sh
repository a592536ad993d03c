#!/bin/bash
# A/B of the wave-pair rollout kernel against the row-per-lane kernel on one box (MRF_ROLLOUT_WP=0 / 1), alternating runs
mkdir -p gpurun_out
for i in 1 2 3; do
  for wp in 0 1; do
    echo -n "WP=$wp  "; MRF_ROLLOUT_WP=$wp python3 tools/prof_rollout.py ${1:-129024} f64 10
  done
done 2>&1 | tee gpurun_out/ab_wp.txt
