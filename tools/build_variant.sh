#!/bin/bash
# Build a kernel variant into ab/lib<name>.so for same-box A/B timing:
#   tools/build_variant.sh <name> [git-rev] [extra hipcc flags...]
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
srcs="$src/multi-robot-fabrics_amd/csrc/mrf_kernels.hip"
[ -f "$src/multi-robot-fabrics_amd/csrc/mrf_control.hip" ] && srcs="$srcs $src/multi-robot-fabrics_amd/csrc/mrf_control.hip"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared "$@" -o ab/lib$name.so $srcs
