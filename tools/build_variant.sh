#!/bin/bash
# Build a kernel variant into ab/lib<name>.so for same-box A/B timing:
#   tools/build_variant.sh <name> [git-rev|-] [extra hipcc flags for the solve kernels...]   (KERNEL_FLAGS= overrides -ffast-math)
#   MRF_HIP_LIB=ab/lib<name>.so python3 tools/prof_rollout.py ...
# With a git revision the csrc/ and include/ trees of that revision are compiled (from a temp copy).
set -e
cd "$(dirname "$0")/.."
name=$1; rev=$2; shift; [ $# -gt 0 ] && shift
mkdir -p ab
src=.
if [ -n "$rev" ] && [ "$rev" != "-" ]; then
  src=$(mktemp -d)
  git archive "$rev" multi-robot-fabrics_amd/csrc include | tar -x -C "$src"
fi
# same per-file flags as __graft_entry__.build(): the solve kernels with -ffast-math, the control-step unit without
tmp=$(mktemp -d)
hip="/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC"
[ "$MRF_WITH_F32" = "1" ] && hip="$hip -DMRF_WITH_F32"
$hip ${KERNEL_FLAGS--ffast-math} "$@" -c -o $tmp/k.o "$src/multi-robot-fabrics_amd/csrc/mrf_kernels.hip" &
objs="$tmp/k.o"
if [ -f "$src/multi-robot-fabrics_amd/csrc/mrf_rollout_wp.hip" ]; then
  $hip ${KERNEL_FLAGS--ffast-math} "$@" -c -o $tmp/w.o "$src/multi-robot-fabrics_amd/csrc/mrf_rollout_wp.hip" &
  objs="$objs $tmp/w.o"
fi
if [ -f "$src/multi-robot-fabrics_amd/csrc/mrf_control.hip" ]; then
  $hip -c -o $tmp/c.o "$src/multi-robot-fabrics_amd/csrc/mrf_control.hip" &
  objs="$objs $tmp/c.o"
fi
if [ -f "$src/multi-robot-fabrics_amd/csrc/mrf_comm.hip" ]; then
  $hip ${KERNEL_FLAGS--ffast-math} "$@" -c -o $tmp/m.o "$src/multi-robot-fabrics_amd/csrc/mrf_comm.hip" &
  objs="$objs $tmp/m.o"
fi
if [ -f "$src/multi-robot-fabrics_amd/csrc/mrf_shard_step.hip" ]; then
  $hip ${KERNEL_FLAGS--ffast-math} "$@" -c -o $tmp/s.o "$src/multi-robot-fabrics_amd/csrc/mrf_shard_step.hip" &
  objs="$objs $tmp/s.o"
fi
if [ -f "$src/multi-robot-fabrics_amd/csrc/mrf_hostpath.hip" ]; then
  $hip -c -o $tmp/h.o "$src/multi-robot-fabrics_amd/csrc/mrf_hostpath.hip" &
  objs="$objs $tmp/h.o"
fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/lib$name.so $objs -ldl
